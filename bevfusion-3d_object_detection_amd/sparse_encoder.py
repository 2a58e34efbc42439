"""Host mirror of the LiDAR middle encoder: `BEVFusionSparseEncoder` (BF/sparse_encoder.py:12-156),
its layer factory `make_encoder_layers` (mmdet3d/models/middle_encoders/sparse_encoder.py:165-241),
`SparseBasicBlock` and `make_sparse_convmodule` (mmdet3d/models/layers/sparse_block.py:94-224).
Conv arithmetic: csrc/spconv.hip via spconv.py; BatchNorm1d / ReLU on the [N, C] feature matrix are torch.
"""
import os
from typing import Optional, Tuple, Union

import torch
from torch import nn

from .registry import MODELS
from .spconv import (BatchNorm1dAct, CapacityMonitor, SparseConv3d, SparseConvTensor, SparseModule,  # noqa: F401
                     SparseSequential, SubMConv3d, prepare_strided_rulebooks, replace_feature, round_capacity)

_CONV_TYPES = {"SubMConv3d": SubMConv3d, "SparseConv3d": SparseConv3d}


def build_norm_1d(norm_cfg, channels):
    cfg = dict(norm_cfg or dict(type="BN1d"))
    typ = cfg.pop("type")
    assert typ in ("BN1d", "BN"), typ
    cfg.pop("requires_grad", None)
    return BatchNorm1dAct(channels, **cfg)  # nn.BatchNorm1d with fused (+residual)(+ReLU) HIP kernels in training


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0,
                           conv_type="SubMConv3d", norm_cfg=None, order=("conv", "norm", "act"), **kwargs):
    """conv (bias=False) / BN1d / ReLU in the requested order, as a SparseSequential."""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {"conv", "norm", "act"} == {"conv", "norm", "act"}
    layers = []
    for layer in order:
        if layer == "conv":
            layers.append(_CONV_TYPES[conv_type](in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                                 bias=False, indice_key=indice_key))
        elif layer == "norm":
            layers.append(build_norm_1d(norm_cfg, out_channels))
        elif layer == "act":
            layers.append(nn.ReLU(inplace=True))
    return SparseSequential(*layers)


class SparseBasicBlock(SparseModule):
    """Two 3x3x3 SubM convs + BN + ReLU with an identity shortcut
    (mmdet3d/models/layers/sparse_block.py:94-154; attribute names conv1/norm1/conv2/norm2 as mmdet's BasicBlock)."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, indice_key=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        conv_cfg = dict(conv_cfg or dict(type="SubMConv3d"))
        conv_cfg.setdefault("indice_key", indice_key)
        conv_cls = _CONV_TYPES[conv_cfg.pop("type")]
        self.conv1 = conv_cls(inplanes, planes, 3, stride=stride, padding=1, bias=False, **conv_cfg)
        self.norm1 = build_norm_1d(norm_cfg, planes)
        self.conv2 = conv_cls(planes, planes, 3, padding=1, bias=False, **conv_cfg)
        self.norm2 = build_norm_1d(norm_cfg, planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x.features
        assert x.features.dim() == 2
        out = self.conv1(x)
        out = replace_feature(out, self.norm1(out.features, relu=True, rows_dev=out.n_valid))
        out = self.conv2(out)
        if self.downsample is not None:
            identity = self.downsample(x).features
        # norm2 + identity + ReLU in one pass
        return replace_feature(out, self.norm2(out.features, residual=identity, relu=True, rows_dev=out.n_valid))


@MODELS.register_module()
class BEVFusionSparseEncoder(nn.Module):
    """Sparse encoder of BEVFusion; spatial order (X, Y, Z) as produced by this fork's voxelization.

    forward(voxel_features [N,C], coors int[N,4]=(b,x,y,z), batch_size) -> [B, C_out*Z_out, X_out, Y_out]
    """

    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type="conv_module",
                 return_middle_feats=False):
        super().__init__()
        assert block_type in ["conv_module", "basicblock"]
        assert isinstance(order, tuple) and len(order) == 3 and set(order) == {"conv", "norm", "act"}
        self.sparse_shape = sparse_shape
        self.in_channels = in_channels
        self.order = order
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(encoder_channels)
        self.fp16_enabled = False
        self.return_middle_feats = return_middle_feats
        # False: NCHW-contiguous fp32 output like the reference; True: the same tensor with channels-last strides (in the
        # autocast dtype when autocast is on) so the NHWC convs of the fuser / BEV backbone take it without a copy
        self.bev_channels_last = False
        # count the outputs of all strided layers up front: one host read per forward (BFHIP_PRESIZE_RULEBOOKS=0: one per layer)
        self.presize_rulebooks = os.environ.get("BFHIP_PRESIZE_RULEBOOKS", "1") == "1"
        self._nout_hints = []  # N_out of the strided layers in the previous forward: sizes the capped buffers of the next one
        # static capacity mode (forward(..., n_valid=...)): row capacities of the strided layers (grow-only) and the
        # stall-free watcher of the true counts
        self.static_caps = None
        self._monitor = CapacityMonitor()
        self.overflowed = False
        # static capacity mode: device bool, True when THIS forward's strided layers produced more rows than their capacities
        # (rows beyond a capacity are dropped from the rulebooks on both sides, so the BEV map of that step is the truncated
        # problem's: BEVFusion.loss poisons the step's losses with it instead of training on that map)
        self.capacity_status = None
        self._caps_dev = (None, None)
        first_order = ("conv",) if order[0] != "conv" else order  # pre-activation variant keeps a bare first conv
        self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                 indice_key="subm1", conv_type="SubMConv3d", order=first_order)
        encoder_out_channels = self.make_encoder_layers(make_sparse_convmodule, norm_cfg, base_channels,
                                                        block_type=block_type)
        self.conv_out = make_sparse_convmodule(encoder_out_channels, output_channels, kernel_size=(1, 1, 3),
                                               stride=(1, 1, 2), norm_cfg=norm_cfg, padding=0,
                                               indice_key="spconv_down2", conv_type="SparseConv3d")

    def make_encoder_layers(self, make_block, norm_cfg, in_channels, block_type="conv_module",
                            conv_cfg=dict(type="SubMConv3d")):
        self.encoder_layers = SparseSequential()
        out_channels = in_channels
        for i, blocks in enumerate(self.encoder_channels):
            blocks_list = []
            for j, out_channels in enumerate(tuple(blocks)):
                padding = tuple(self.encoder_paddings[i])[j]
                last_of_stage = j == len(blocks) - 1 and i != len(self.encoder_channels) - 1
                if i != 0 and j == 0 and block_type == "conv_module":
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                  padding=padding, indice_key=f"spconv{i + 1}",
                                                  conv_type="SparseConv3d"))
                elif block_type == "basicblock":
                    if last_of_stage:
                        blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                      padding=padding, indice_key=f"spconv{i + 1}",
                                                      conv_type="SparseConv3d"))
                    else:
                        blocks_list.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg,
                                                            conv_cfg=conv_cfg))
                else:
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, padding=padding,
                                                  indice_key=f"subm{i + 1}", conv_type="SubMConv3d"))
                in_channels = out_channels
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", SparseSequential(*blocks_list))
        return out_channels

    def update_static_caps(self, n_outs):
        """Grow-only capacities from observed strided-layer output counts."""
        caps = [round_capacity(n) for n in n_outs]
        if self.static_caps is None or len(self.static_caps) != len(caps):
            self.static_caps = caps
        else:
            self.static_caps = [max(a, b) for a, b in zip(self.static_caps, caps)]

    def forward(self, voxel_features, coors, batch_size, n_valid=None):
        """n_valid (device i32[1]): static capacity mode -- `voxel_features` / `coors` are capacity-sized with the active rows
        as a prefix (BEVFusion.voxelize_static); every row count stays on the device and NO host read happens."""
        coors = coors.int()
        x = SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size)
        if n_valid is not None:
            assert self.static_caps is not None, "static capacity mode needs capacities (run one exact forward first)"
            seen = self._monitor.poll()  # true counts of an EARLIER forward, if they have arrived: no stall
            if seen is not None:
                if any(n > c for n, c in zip(seen, self.static_caps)):
                    self.overflowed = True  # that forward dropped rows beyond its capacity: grown below for the next ones
                    import warnings
                    warnings.warn("sparse encoder: a strided layer exceeded its row capacity %s < %s; capacities grown"
                                  % (self.static_caps, seen))
                self.update_static_caps(seen)
            x.n_valid = n_valid
            chain = [m for m in self.modules() if isinstance(m, SparseConv3d)]
            plans, true_counts = prepare_strided_rulebooks(
                coors, batch_size, self.sparse_shape, [(m.kernel_size, m.stride, m.padding, m.dilation) for m in chain],
                static_caps=self.static_caps, n_in_dev=n_valid)
            x.indice_dict["_strided_plans"] = plans
            if true_counts is not None:
                self._monitor.submit(true_counts)
                key = tuple(self.static_caps)
                if self._caps_dev[0] != key:  # host list -> device, once per capacity change (a plain H2D copy, no read)
                    self._caps_dev = (key, torch.tensor(key[:true_counts.numel()], dtype=torch.int32, device=coors.device))
                self.capacity_status = (true_counts > self._caps_dev[1]).any()
            else:
                self.capacity_status = None
        elif self.presize_rulebooks and coors.is_cuda:
            self.capacity_status = None
            # all strided layers' output counts in ONE host read (instead of one per layer)
            chain = [m for m in self.modules() if isinstance(m, SparseConv3d)]  # registration order = execution order
            x.indice_dict["_strided_plans"] = prepare_strided_rulebooks(
                coors, batch_size, self.sparse_shape, [(m.kernel_size, m.stride, m.padding, m.dilation) for m in chain],
                hints=self._nout_hints)
            if len(self._nout_hints) == len(chain) and all(h > 0 for h in self._nout_hints):
                self.update_static_caps(self._nout_hints)
        x = self.conv_input(x)
        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        out = self.conv_out(encode_features[-1])
        # out.dense() -> [N, C, X, Y, Z] -> permute(0,1,4,2,3) -> view(N, C*Z, X, Y)   (BF/sparse_encoder.py:147-151)
        if self.bev_channels_last:
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
            spatial_features = out.to_bev(channels_last=True, dtype=dt if dt == torch.bfloat16 else torch.float32)
        else:
            spatial_features = out.to_bev()
        if self.return_middle_feats:
            return spatial_features, encode_features
        return spatial_features
