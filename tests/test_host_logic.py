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
