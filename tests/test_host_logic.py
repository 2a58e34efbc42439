"""CPU tests of host-side logic that needs no kernel launch."""
import torch

import bevfusion_amd  # noqa: F401
from bevfusion_amd.spconv import SparseConv3d, SubMConv3d
from bevfusion_amd.sparse_encoder import BEVFusionSparseEncoder


def test_spconv1_checkpoint_is_rotated_on_load():
    """write_spconv2.py:43-74: a state dict without module version 2 stores kernels as (k0,k1,k2,in,out)."""
    conv = SubMConv3d(5, 16, 3, bias=True, indice_key="subm1")
    w_new = torch.randn(16, 3, 3, 3, 5)
    old = {"weight": w_new.permute(1, 2, 3, 4, 0).contiguous(), "bias": torch.randn(16)}  # spconv 1.x layout
    conv.load_state_dict(old)  # no _metadata -> version None
    assert torch.equal(conv.weight, w_new) and torch.equal(conv.bias, old["bias"])


def test_spconv2_checkpoint_round_trips_unchanged():
    a = SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key="spconv2")
    b = SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key="spconv2")
    sd = a.state_dict()
    assert sd._metadata[""]["version"] == 2
    b.load_state_dict(sd)
    assert torch.equal(a.weight, b.weight)


def test_encoder_state_dict_keys_follow_the_reference_names():
    """BF/sparse_encoder.py:45-131: conv_input.0.weight, encoder_layers.encoder_layer1.0.conv1.weight, conv_out.0.weight."""
    enc = BEVFusionSparseEncoder(in_channels=5, sparse_shape=[1440, 1440, 41],
                                 encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                 encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)), block_type="basicblock")
    keys = set(enc.state_dict().keys())
    for k in ("conv_input.0.weight", "conv_input.1.running_mean", "encoder_layers.encoder_layer1.0.conv1.weight",
              "encoder_layers.encoder_layer1.0.norm2.weight", "encoder_layers.encoder_layer1.2.0.weight",
              "encoder_layers.encoder_layer4.1.conv2.weight", "conv_out.0.weight", "conv_out.1.bias"):
        assert k in keys, k
    assert enc.state_dict()["conv_out.0.weight"].shape == (128, 1, 1, 3, 128)
    # mixed-version import: an old-layout encoder checkpoint loads into the new layout
    old = {k: (v.permute(1, 2, 3, 4, 0).contiguous() if v.dim() == 5 else v.clone()) for k, v in enc.state_dict().items()}
    enc2 = BEVFusionSparseEncoder(in_channels=5, sparse_shape=[1440, 1440, 41],
                                  encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                                  encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, (1, 1, 0)), (0, 0)), block_type="basicblock")
    enc2.load_state_dict(old)
    for k, v in enc.state_dict().items():
        assert torch.equal(v, enc2.state_dict()[k]), k


def test_head_and_neck_state_dict_keys_follow_the_reference_names():
    """The reference builds these layers as mmcv ConvModule / FFN (children `conv`, `bn`, `layers`): heat-map head
    (BF/bevfusion_head.py:104-126), SeparateHead (centerpoint_head.py:54-82), GeneralizedLSSFPN (BF/bevfusion_necks.py:50-72),
    decoder FFN / attention (BF/transformer.py:26-31 via mmdet's DetrTransformerDecoderLayer).  Reference checkpoints must load."""
    from bevfusion_amd.dense_modules import BEVFusionHead, GeneralizedLSSFPN
    head = BEVFusionHead(num_proposals=20, in_channels=64, hidden_channel=32, num_classes=10,
                         decoder_layer=dict(self_attn_cfg=dict(embed_dims=32, num_heads=4, dropout=0.1),
                                            cross_attn_cfg=dict(embed_dims=32, num_heads=4, dropout=0.1),
                                            ffn_cfg=dict(embed_dims=32, feedforward_channels=64, num_fcs=2, ffn_drop=0.1),
                                            pos_encoding_cfg=dict(input_channel=2, num_pos_feats=32)),
                         grid_size=(64, 64, 8), out_size_factor=8)
    keys = set(head.state_dict().keys())
    want = ["shared_conv.weight", "shared_conv.bias",
            "heatmap_head.0.conv.weight", "heatmap_head.0.bn.weight", "heatmap_head.0.bn.bias", "heatmap_head.0.bn.running_mean",
            "heatmap_head.0.bn.running_var", "heatmap_head.0.bn.num_batches_tracked", "heatmap_head.1.weight", "heatmap_head.1.bias",
            "class_encoding.weight", "class_encoding.bias",
            "decoder.0.self_attn.attn.in_proj_weight", "decoder.0.self_attn.attn.in_proj_bias",
            "decoder.0.self_attn.attn.out_proj.weight", "decoder.0.cross_attn.attn.out_proj.bias",
            "decoder.0.ffn.layers.0.0.weight", "decoder.0.ffn.layers.0.0.bias", "decoder.0.ffn.layers.1.weight",
            "decoder.0.ffn.layers.1.bias", "decoder.0.norms.0.weight", "decoder.0.norms.2.bias",
            "decoder.0.self_posembed.position_embedding_head.0.weight", "decoder.0.self_posembed.position_embedding_head.1.running_mean",
            "decoder.0.cross_posembed.position_embedding_head.3.bias"]
    for h in ("center", "height", "dim", "rot", "vel", "heatmap"):
        want += [f"prediction_heads.0.{h}.0.conv.weight", f"prediction_heads.0.{h}.0.bn.weight", f"prediction_heads.0.{h}.0.bn.running_var",
                 f"prediction_heads.0.{h}.1.weight", f"prediction_heads.0.{h}.1.bias"]
    for k in want:
        assert k in keys, k
    assert not any(".conv.bias" in k for k in keys)          # bias='auto' + norm -> no conv bias
    import re
    pats = [r"shared_conv\.(weight|bias)", r"heatmap_head\.0\.(conv\.weight|bn\.\w+)", r"heatmap_head\.1\.(weight|bias)",
            r"class_encoding\.(weight|bias)", r"decoder\.0\.(self|cross)_attn\.attn\.(in_proj_weight|in_proj_bias|out_proj\.weight|out_proj\.bias)",
            r"decoder\.0\.ffn\.layers\.(0\.0|1)\.(weight|bias)", r"decoder\.0\.norms\.[012]\.(weight|bias)",
            r"decoder\.0\.(self|cross)_posembed\.position_embedding_head\.(0|3)\.(weight|bias)",
            r"decoder\.0\.(self|cross)_posembed\.position_embedding_head\.1\.\w+",
            r"prediction_heads\.0\.\w+\.0\.(conv\.weight|bn\.\w+)", r"prediction_heads\.0\.\w+\.1\.(weight|bias)"]
    for k in keys:
        assert any(re.fullmatch(p, k) for p in pats), k
    fpn = GeneralizedLSSFPN(in_channels=[8, 16, 32], out_channels=8, num_outs=3, start_level=0)
    fk = set(fpn.state_dict().keys())
    for i in (0, 1):
        for sub in ("lateral_convs", "fpn_convs"):
            for leaf in ("conv.weight", "bn.weight", "bn.bias", "bn.running_mean", "bn.running_var", "bn.num_batches_tracked"):
                assert f"{sub}.{i}.{leaf}" in fk
    assert len(fk) == 24


def test_grouped_weight_gradient_plan_is_a_partition():
    """bfhip_conv2d_wgrad_group_plan (host only, csrc/conv2d.hip): the table it writes for the 77 dense layers of the `full` workload's
    backward pass (shapes of profiles/r03_conv_wgrad_layers.txt) cuts every layer's pixel range into splits that cover it exactly
    once, gives every layer its own slab range, lists the workgroups of each tile shape once, and hands the 8 XCDs consecutive
    chunks of (nearly) equal total steps.  The item layout below mirrors `struct WgradItem` (internal; test in lock-step)."""
    import ctypes

    import numpy as np
    from bevfusion_amd import _lib, conv2d as c2
    lib = _lib.load()
    #          x: N  Cin   H    W     w: Cout k  s   layers
    layers = [(4, 128, 180, 180, 128, 3, 1, 5), (4, 256, 90, 90, 256, 3, 1, 5), (4, 336, 180, 180, 256, 3, 1, 1),
              (24, 256, 16, 44, 256, 3, 1, 6), (24, 64, 64, 176, 64, 3, 1, 3), (24, 256, 32, 88, 256, 3, 1, 2),
              (4, 512, 180, 180, 128, 3, 1, 1), (24, 64, 64, 176, 256, 1, 1, 4), (24, 256, 16, 44, 1024, 1, 1, 6),
              (24, 1024, 16, 44, 256, 1, 1, 5), (24, 320, 32, 88, 256, 3, 1, 1), (24, 128, 32, 88, 512, 1, 1, 4),
              (24, 128, 32, 88, 128, 3, 1, 3), (4, 80, 360, 360, 80, 3, 1, 1), (24, 512, 8, 22, 2048, 1, 1, 3),
              (24, 2048, 8, 22, 512, 1, 1, 2), (4, 80, 360, 360, 80, 3, 2, 1), (24, 32, 64, 176, 64, 5, 2, 1),
              (24, 64, 64, 176, 64, 1, 1, 1), (4, 128, 180, 180, 256, 1, 1, 1)]
    rows = []
    for N, Cin, H, W, Cout, k, s, cnt in layers:
        assert lib.bfhip_conv2d_wgrad_groupable(N, H, W, Cin, Cout, k, k, s, k // 2, 1)
        rows += [(0x100000 + 0x1000 * len(rows), 0x200000, 0x300000 + 0x100 * len(rows), Cin, Cout, N, H, W, Cin, Cout, k, k, s, k // 2, 1,
                  len(rows) % 2, 0)] * 1
        rows += [rows[-1]] * (cnt - 1)
    n = len(rows)
    L = np.array(rows, dtype=c2._layer_dtype())
    item_dt = np.dtype([("x", "<u8"), ("dy", "<u8"), ("dw", "<u8"), ("slab_off", "<u8"), ("M", "<i8"), ("rps", "<i8"), ("total", "<i8")] +
                       [(k, "<i4") for k in ("N", "H", "W", "C", "ldx", "OH", "OW", "KH", "KW", "stride", "pad", "dil", "nq", "Cout", "ldg",
                                             "splits", "tiles_co", "tiles_k", "dw_bf16", "shape", "first_block", "n_blocks",
                                             "first_rblock", "n_rblocks")])
    nb = lib.bfhip_conv2d_wgrad_group_table_bytes(n)
    assert nb == 256 + n * item_dt.itemsize
    for target in (0, 24, 200):
        img = np.zeros(nb, np.uint8)
        slab = ctypes.c_size_t(0)
        _lib.call("bfhip_conv2d_wgrad_group_plan", L.ctypes.data, n, target, img.ctypes.data, nb, ctypes.byref(slab))
        hd = img[:256].view(np.int32)
        n_items, n_shape, first, blocks, grid, rblocks = hd[0], hd[1:4], hd[4:7], hd[7:10], hd[10:13], hd[16]
        chunks = hd[18:45].reshape(3, 9)
        items = img[256:].view(item_dt)
        assert n_items == n and n_shape[0] == 0 and n_shape[1] + n_shape[2] == n and hd[17] == (target or 96)
        assert sorted(items["dw"].tolist()) == sorted(L["dw"].tolist())            # every layer exactly once
        end = 0
        for it in items:                                                          # table order = slab order
            steps = -(-int(it["M"]) // 64)
            per = int(it["rps"]) // 64
            assert it["rps"] % 64 == 0 and per >= 6 and it["splits"] * per >= steps > (it["splits"] - 1) * per
            assert it["M"] == it["N"] * it["OH"] * it["OW"] and it["nq"] * 8 == it["KH"] * it["KW"] * it["C"] and it["total"] == it["Cout"] * it["nq"] * 8
            assert it["n_blocks"] == it["tiles_co"] * it["tiles_k"] * it["splits"]
            assert it["tiles_co"] * (256 if it["shape"] == 2 else 128) >= it["Cout"] and it["tiles_k"] * (256 if it["shape"] == 1 else 128) >= it["nq"] * 8
            assert it["slab_off"] == end and end % 256 == 0
            end += -(-int(it["splits"]) * int(it["total"]) * 4 // 256) * 256
        assert end == slab.value
        assert (np.cumsum(items["n_rblocks"]) - items["n_rblocks"] == items["first_rblock"]).all() and items["n_rblocks"].sum() == rblocks
        for sh in (1, 2):
            its = items[first[sh]:first[sh] + n_shape[sh]]
            assert (its["shape"] == sh).all() and (np.diff(its["rps"]) <= 0).all()  # longest workgroups first
            assert (np.cumsum(its["n_blocks"]) - its["n_blocks"] == its["first_block"]).all() and its["n_blocks"].sum() == blocks[sh]
            c = chunks[sh]
            assert c[0] == 0 and c[8] == blocks[sh] and (np.diff(c) >= 0).all() and grid[sh] == 8 * np.diff(c).max()
            per_block = np.repeat(its["rps"] // 64, its["n_blocks"])
            load = np.array([per_block[c[i]:c[i + 1]].sum() for i in range(8)])
            assert load.sum() == per_block.sum() and load.max() - load.min() <= 2 * per_block.max()
        if target == 200:   # deeper splits for the launch with many workgroups, not for the one that would fall under ~4 rounds
            assert items[first[1]]["rps"] // 64 > 150 and items[first[2]]["rps"] // 64 < 100
    # a layer the wide kernels do not take (fewer than six 64-pixel steps) is refused, with a message
    assert not lib.bfhip_conv2d_wgrad_groupable(1, 12, 12, 64, 64, 3, 3, 1, 1, 1)
    bad = np.array([(0x1000, 0x2000, 0x3000, 64, 64, 1, 12, 12, 64, 64, 3, 3, 1, 1, 1, 0, 0)], dtype=c2._layer_dtype())
    img = np.zeros(lib.bfhip_conv2d_wgrad_group_table_bytes(1), np.uint8)
    assert lib.bfhip_conv2d_wgrad_group_plan(bad.ctypes.data, 1, 0, img.ctypes.data, img.size, ctypes.byref(slab)) != 0
    assert b"not groupable" in lib.bfhip_last_error()
