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
