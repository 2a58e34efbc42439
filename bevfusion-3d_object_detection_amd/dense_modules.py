"""Dense (MFMA-bound) modules of the BEVFusion graph.

The BEV / view-transform / head convolutions (SURVEY 8 a-8 ... a-11) are `conv2d.Conv2d`: nn.Conv2d whose bf16 channels-last
calls run the hand-written implicit-GEMM kernels of csrc/conv2d.hip (forward, dgrad, wgrad, BatchNorm statistics in the
epilogue); the image backbone (ResNet-50, a consumer of the hot path, not a row of it) stays on torch.nn (MIOpen / CK).
The modules exist so that the full fwd+bwd step of BASELINE.json configs 3-5 runs without mmcv/mmdet/mmengine (absent here).
Layer shapes follow the reference:
  ResNet50            BASELINE.json's image backbone (the reference config uses mmdet Swin-T; any backbone
                      returning 3 levels works, BF/bevfusion.py:55,161-171)
  GeneralizedLSSFPN   BF/bevfusion_necks.py:11-96
  ConvFuser           BF/bevfusion_head.py:25-38
  SECOND / SECONDFPN  mmdet3d/models/backbones/second.py:27-95, mmdet3d/models/necks/second_fpn.py:30-94
  BEVFusionHead       BF/bevfusion_head.py:41-299 (forward only: shared conv, heat-map head, top-k proposals,
                      one TransformerDecoderLayer BF/transformer.py:26-113, SeparateHead
                      mmdet3d/models/dense_heads/centerpoint_head.py:20-121).  Target assignment, losses and
                      box decoding are "next" rows.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .bn2d import BatchNorm2dAct, BatchNormRows, bn_act
from .conv2d import Conv2d, Conv2dHipWgrad, fp32_island
from .linear_rows import linear_rows
from . import attention as split_attention
from .registry import MODELS


# ----------------------------------------------------------------------------- image backbone
# Which kernels the ResNet-50 trunk's 52 convolutions take (BFHIP_RESNET_CONV).  The weight gradient is csrc/conv2d.hip's in
# every mode but "lib" (the library's brings an fp32 zero-fill and a cast launch per call and is no faster).  Forward and data
# gradient, GPU time per call from `tools/resnet_conv_micro.py` (graph replay, batch 24 x 64 x 176 after the stem; sums over the
# trunk): library forward 1.72 ms, HIP 1.96 ms -- but a HIP forward hands the BatchNorm behind it its statistics (0.34 ms of
# statistics passes over the trunk); library data gradient 2.52 ms (with its zero-fill of dx), HIP 2.49 ms, where the HIP side
# wins every stride-1 layer (1x1: conv_pw_kernel, 20-35 % ahead; 3x3 up to 256 channels: 10-25 % ahead) and loses every stride-2
# one by 1.5-2.3x (the 1x1 stride-2 shortcuts write three zero pixels out of four through the full tile machinery).  Hence
#   "tuned" (default): 1x1 stride 1 -> HIP forward + HIP data gradient; 3x3 up to 256 channels -> HIP forward (ahead of the library
#                      by 2-8 % since the one-stage tile variant, plus the statistics pass it saves) and, at stride 1, HIP data
#                      gradient; the 512-channel 3x3 layers and the stride-2 data gradients -> library; stride-2 1x1 shortcuts run
#                      on the subsampled input as stride-1 pointwise layers (_Bottleneck.forward)
#   "hipwgrad": library forward and data gradient everywhere (round 2's default); "lib": the library for everything;
#   "hip3x3" / "hip" / "hip3x3+hipwgrad": the 3x3 / all / 3x3-only layers entirely on the HIP kernels (round-2 experiments:
#   34.65 / 35.28 / 34.0-34.3 ms per `full` step against 33.7-33.8 for "hipwgrad")
# (The 7x7 stem has 3 input channels and always stays on the library.)
_RESNET_CONV = os.environ.get("BFHIP_RESNET_CONV", "tuned")


def _resnet_conv(cin, cout, k, stride=1, padding=0, bias=False):
    mode = _RESNET_CONV
    if mode == "tuned":
        m = Conv2dHipWgrad(cin, cout, k, stride=stride, padding=padding, bias=bias)
        if stride == 1 and k == 1:
            m.fwd = m.dgrad = "hip"
        elif k == 3 and cin <= 256:
            m.fwd = "hip"
            if stride == 1:
                m.dgrad = "hip"
        elif k == 1:
            m.cache_wt = True   # stride-2 shortcut: run on the subsampled input by _Bottleneck (forward_unstrided)
        return m
    if k == 1:
        cls = Conv2d if mode == "hip" else (Conv2dHipWgrad if mode in ("hipwgrad", "hip3x3+hipwgrad") else nn.Conv2d)
    else:
        cls = Conv2d if mode in ("hip", "hip3x3", "hip3x3+hipwgrad") else (Conv2dHipWgrad if mode == "hipwgrad" else nn.Conv2d)
    return cls(cin, cout, k, stride=stride, padding=padding, bias=bias)


_Conv3x3 = _Conv1x1 = _resnet_conv
_SHORTCUT_FORK = os.environ.get("BFHIP_SHORTCUT_FORK", "1") == "1"
_HEAD_ROWS = os.environ.get("BFHIP_HEAD_ROWS", "1") == "1"   # prediction heads as GEMMs over [B*L, C] rows (SeparateHead.forward)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _Conv1x1(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2dAct(planes)
        self.conv2 = _Conv3x3(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2dAct(planes)
        self.conv3 = _Conv1x1(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2dAct(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        ds = self.downsample
        if ds is None and hasattr(self.conv1, "forward_fork"):
            out, identity = self.conv1.forward_fork(x)   # the identity's gradient is added in conv1's data gradient
            out = self.bn1(out, relu=True)
        elif (ds is not None and hasattr(self.conv1, "forward_fork") and isinstance(ds[0], Conv2dHipWgrad) and ds[0].stride == (2, 2)
              and ds[0].kernel_size == (1, 1) and ds[0].padding == (0, 0) and _SHORTCUT_FORK and ds[0].training):
            # stride-2 1x1 shortcut: conv1 also hands out x at its even pixels (dense); the shortcut runs on that as a stride-1
            # pointwise conv (HIP forward + statistics, compact data gradient) and its gradient is added at the even pixels in
            # conv1's data-gradient epilogue -- instead of a library data gradient that zero-fills and writes the full-size
            # tensor plus autograd's add over it
            out, x_sub = self.conv1.forward_fork(x, subsample=2)
            out = self.bn1(out, relu=True)
            identity = ds(x) if x_sub is None else ds[1](ds[0].forward_unstrided(x_sub))
        else:
            identity = x if self.downsample is None else self.downsample(x)
            out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=identity, relu=True)  # BN + identity + ReLU in one pass


class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        tap = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        _lib.call("bfhip_maxpool3x3s2_fwd", x.data_ptr(), N, H, W, C, y.data_ptr(), tap.data_ptr(), _lib.stream_of(x))
        ctx.save_for_backward(tap)
        ctx.dims = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (tap,) = ctx.saved_tensors
        N, C, H, W = ctx.dims
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        _lib.call("bfhip_maxpool3x3s2_bwd", dy.data_ptr(), tap.data_ptr(), N, H, W, C, dx.data_ptr(), _lib.stream_of(dy))
        return dx


class MaxPool3x3s2(nn.MaxPool2d):
    """nn.MaxPool2d(3, stride=2, padding=1) (the ResNet stem's pooling); a channels-last bf16 CUDA map takes csrc/pool.hip
    (forward with the winning tap per element, backward as a gather: 0.10 + 0.25 ms -> see DESIGN.md), anything else the
    library.  `BFHIP_MAXPOOL=0` switches the HIP path off."""

    ENABLED = os.environ.get("BFHIP_MAXPOOL", "1") == "1"

    def __init__(self):
        super().__init__(3, stride=2, padding=1)

    def forward(self, x):
        if (self.ENABLED and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0
                and x.numel() < (1 << 31) and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0):
            return _MaxPool3x3s2.apply(x)
        return super().forward(x)


@MODELS.register_module()
class ResNet50(nn.Module):
    """Standard ResNet-50; returns the stride-8/16/32 maps (512, 1024, 2048 channels)."""

    def __init__(self, out_indices=(1, 2, 3)):
        super().__init__()
        self.out_indices = out_indices
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2dAct(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = MaxPool3x3s2()
        self.inplanes = 64
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(_Conv1x1(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 BatchNorm2dAct(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled("cuda"):
            # dtype and layout in ONE pass over the images (autocast's cast + the library's channels-last conversion of the stem's
            # input are three)
            x = x.to(dtype=torch.get_autocast_dtype("cuda"), memory_format=torch.channels_last)
        x = self.maxpool(self.bn1(self.conv1(x), relu=True))
        outs = []
        for i, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class _Upsample2x(torch.autograd.Function):
    """Exact-2x bilinear upsampling of a channels-last map (csrc/upsample.hip); the backward is a gather."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        out = torch.empty((B, C, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _lib.call("bfhip_upsample2x_nhwc", x.data_ptr(), out.data_ptr(), B, H, W, C, 1 if x.dtype == torch.bfloat16 else 0, 0,
                  _lib.stream_of(x))
        ctx.dims = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.dims
        g = g.contiguous(memory_format=torch.channels_last)
        gin = torch.empty((B, C, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        _lib.call("bfhip_upsample2x_nhwc", g.data_ptr(), gin.data_ptr(), B, H, W, C, 1 if g.dtype == torch.bfloat16 else 0, 1,
                  _lib.stream_of(g))
        return gin


def upsample_to(x, size, cfg):
    """F.interpolate(x, size=size, **cfg); the exact-2x bilinear case on a channels-last GPU map runs in the input dtype
    with the gather backward (autocast would widen it to fp32, and the library backward uses atomics)."""
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if (x.is_cuda and x.dim() == 4 and cfg.get("mode") == "bilinear" and not cfg.get("align_corners", False)
            and tuple(size) == (2 * x.shape[2], 2 * x.shape[3]) and x.dtype in (torch.float32, torch.bfloat16)
            and x.shape[1] % vec == 0 and x.is_contiguous(memory_format=torch.channels_last)):
        return _Upsample2x.apply(x)
    return F.interpolate(x, size=size, **cfg)


class ConvModule(nn.Module):
    """The subset of mmcv's ConvModule the reference uses: conv -> (BatchNorm) -> (ReLU) with the children named `conv` and
    `bn`, so that state-dict keys are the reference's (`<name>.conv.weight`, `<name>.bn.weight`, ...;
    BF/bevfusion_necks.py:50-72, BF/bevfusion_head.py:104-115, centerpoint_head.py:60-70).  `bias='auto'` = no bias when a
    norm follows.  2-D modules fuse BN + ReLU (csrc/bn2d.hip); `dim=1` builds the Conv1d / BatchNorm1d form."""

    def __init__(self, cin, cout, k, stride=1, padding=0, norm=True, act=True, bias="auto", eps=1e-5, momentum=0.1, dim=2):
        super().__init__()
        bias = (not norm) if bias == "auto" else bool(bias)
        if dim == 2:
            self.conv = Conv2d(cin, cout, k, stride=stride, padding=padding, bias=bias)
            self.bn = BatchNorm2dAct(cout, eps=eps, momentum=momentum, act=act) if norm else None
        else:
            self.conv = nn.Conv1d(cin, cout, k, stride=stride, padding=padding, bias=bias)
            self.bn = BatchNormRows(cout, eps=eps, momentum=momentum) if norm else None  # an nn.BatchNorm1d (+ fused [M, C] path)
        self.dim, self.act = dim, act

    def forward(self, x):
        x = self.conv(x)
        if self.bn is None:
            return F.relu(x) if self.act else x
        if self.dim == 2:
            return self.bn(x)          # BatchNorm2dAct applies the ReLU itself
        x = self.bn(x)
        return F.relu(x) if self.act else x


class FFN(nn.Module):
    """mmcv FFN (num_fcs = 2, add_identity): `layers` = Sequential(Sequential(Linear, ReLU, Dropout), Linear, Dropout);
    keys `layers.0.0.weight`, `layers.1.weight` as in the reference's checkpoints (BF/transformer.py:26 builds it through
    mmdet's DetrTransformerDecoderLayer)."""

    def __init__(self, embed_dims, feedforward_channels, ffn_drop=0.0):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))

    def forward(self, x):
        return x + self.layers(x)


@MODELS.register_module()
class GeneralizedLSSFPN(nn.Module):
    """Top-down: upsample level i+1, concat with level i, 1x1 conv, 3x3 conv; returns levels [0, n-1)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=dict(type="BN2d"), act_cfg=dict(type="ReLU"),
                 upsample_cfg=dict(mode="bilinear", align_corners=True)):
        super().__init__()
        assert isinstance(in_channels, list)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.upsample_cfg = dict(upsample_cfg)
        self.backbone_end_level = self.num_ins - 1 if end_level == -1 else end_level
        self.start_level, self.end_level = start_level, end_level
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            cin = in_channels[i] + (in_channels[i + 1] if i == self.backbone_end_level - 1 else out_channels)
            self.lateral_convs.append(ConvModule(cin, out_channels, 1, norm=not no_norm_on_lateral))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1))

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [inputs[i + self.start_level] for i in range(len(inputs))]
        used = len(laterals) - 1
        for i in range(used - 1, -1, -1):
            x = upsample_to(laterals[i + 1], laterals[i].shape[2:], self.upsample_cfg)
            if x.dtype != laterals[i].dtype:
                x = x.to(laterals[i].dtype)
            laterals[i] = self.fpn_convs[i](self.lateral_convs[i](torch.cat([laterals[i], x], dim=1)))
        return tuple(laterals[i] for i in range(used))


# ----------------------------------------------------------------------------- fusion + BEV backbone
@MODELS.register_module()
class ConvFuser(nn.Sequential):
    def __init__(self, in_channels, out_channels):
        self.in_channels, self.out_channels = in_channels, out_channels
        super().__init__(Conv2d(sum(in_channels), out_channels, 3, padding=1, bias=False),
                         *bn_act(out_channels))

    def forward(self, inputs):
        return super().forward(torch.cat(inputs, dim=1))


@MODELS.register_module()
class SECOND(nn.Module):
    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False)):
        super().__init__()
        assert len(layer_strides) == len(layer_nums) == len(out_channels)
        eps, mom = norm_cfg.get("eps", 1e-5), norm_cfg.get("momentum", 0.1)
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, n in enumerate(layer_nums):
            block = [Conv2d(in_filters[i], out_channels[i], 3, stride=layer_strides[i], padding=1, bias=False),
                     *bn_act(out_channels[i], eps=eps, momentum=mom)]
            for _ in range(n):
                block += [Conv2d(out_channels[i], out_channels[i], 3, padding=1, bias=False),
                          *bn_act(out_channels[i], eps=eps, momentum=mom)]
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        outs = []
        for b in self.blocks:
            x = b(x)
            outs.append(x)
        return tuple(outs)


@MODELS.register_module()
class SECONDFPN(nn.Module):
    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=False):
        super().__init__()
        eps, mom = norm_cfg.get("eps", 1e-5), norm_cfg.get("momentum", 0.1)
        deblocks = []
        for i, oc in enumerate(out_channels):
            s = upsample_strides[i]
            if s > 1 or (s == 1 and not use_conv_for_no_stride):
                up = nn.ConvTranspose2d(in_channels[i], oc, s, stride=s, bias=False)
            else:
                k = int(round(1 / s))
                up = Conv2d(in_channels[i], oc, k, stride=k, bias=False)
            deblocks.append(nn.Sequential(up, *bn_act(oc, eps=eps, momentum=mom)))
        self.deblocks = nn.ModuleList(deblocks)

    def forward(self, x):
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]


# ----------------------------------------------------------------------------- TransFusion head (forward)
class PositionEncodingLearned(nn.Module):
    """Conv1d(k=1) -> BN1d -> ReLU -> Conv1d(k=1) over positions (BF/transformer.py:10-23).  A k=1 Conv1d is a linear map
    over channels, so the stack is evaluated on the row-major [B*N, C] matrix (`forward_nlc`): the BatchNorm then is a
    column reduction of a contiguous matrix (fused kernel) instead of a strided [B, C, N] reduction, and the decoder gets
    the [B, N, C] layout it transposes to anyway.  Parameters and state-dict keys are those of the Conv1d/BN1d stack."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(nn.Conv1d(input_channel, num_pos_feats, 1),
                                                     BatchNormRows(num_pos_feats), nn.ReLU(inplace=True),
                                                     nn.Conv1d(num_pos_feats, num_pos_feats, 1))

    def forward_nlc(self, xyz):
        """xyz [B, N, in] -> [B, N, C]."""
        conv1, bn, _, conv2 = self.position_embedding_head
        B, N, _ = xyz.shape
        h = linear_rows(xyz.reshape(B * N, -1), conv1.weight.squeeze(-1), conv1.bias)
        h = bn(h.contiguous(), relu=True)
        h = linear_rows(h, conv2.weight.squeeze(-1), conv2.bias)
        return h.view(B, N, -1)

    def forward(self, xyz):
        """xyz [B, N, in] -> [B, C, N] (the reference's layout)."""
        return self.forward_nlc(xyz).transpose(1, 2)


class _MHA(nn.Module):
    """mmcv MultiheadAttention semantics: identity + dropout(attn(q + q_pos, k + k_pos, v)), batch_first.
    Parameters live in an nn.MultiheadAttention (same state-dict keys as the reference's mmcv wrapper: attn.in_proj_weight,
    attn.out_proj.weight, ...); the computation is spelled out -- three projections, scaled_dot_product_attention (what
    nn.MultiheadAttention runs when no weights are requested), output projection -- so that the projections of the
    32 400 BEV keys per sample go through `linear_rows` (split-K weight gradient)."""

    def __init__(self, embed_dims, num_heads, dropout=0.0):
        super().__init__()
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, dropout=dropout, batch_first=True)
        self.dropout = nn.Dropout(dropout)
        self.embed_dims, self.num_heads = embed_dims, num_heads

    def forward(self, query, key, value, query_pos=None, key_pos=None):
        q = query if query_pos is None else query + query_pos
        k = key if key_pos is None else key + key_pos
        return self.attend(query, q, k, value)

    def attend(self, query, q, k, value):
        """identity + dropout(attn(q, k, value)) with the position terms already added by the caller (the decoder layer feeds the
        SAME tensor as key + key_pos and as value -- mmcv's convention -- so it adds once instead of once per role)."""
        E, H = self.embed_dims, self.num_heads
        # the three projection blocks as unbind() views of the packed parameter: their gradients come back through ONE stack
        # (slices w[i*E:(i+1)*E] cost a zero-filled [3E, E] tensor, a copy and an add per block in the backward)
        ws = self.attn.in_proj_weight.view(3, E, E).unbind(0)
        bs = self.attn.in_proj_bias.view(3, E).unbind(0) if self.attn.in_proj_bias is not None else (None, None, None)
        B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]

        def proj(t, i):
            return linear_rows(t.reshape(-1, E), ws[i], bs[i]).view(B, -1, E)  # [B, L, H*d]

        qp, kp, vp = proj(q, 0), proj(k, 1), proj(value, 2)
        p_drop = self.attn.dropout if self.training else 0.0
        if split_attention.supported(qp, kp, vp, H):
            # 200 queries x 32 400 keys: split the key axis over the chip (csrc/attn.hip)
            o = split_attention.cross_attention(qp, kp, vp, H, p_drop)
        else:
            heads = lambda t: t.view(B, -1, H, E // H).transpose(1, 2)  # noqa: E731  [B, H, L, d]
            o = F.scaled_dot_product_attention(heads(qp), heads(kp), heads(vp), dropout_p=p_drop)
            o = o.transpose(1, 2).reshape(B, Lq, E)
        return query + self.dropout(self.attn.out_proj(o))


@MODELS.register_module()
class TransformerDecoderLayer(nn.Module):
    def __init__(self, self_attn_cfg=dict(embed_dims=128, num_heads=8, dropout=0.1),
                 cross_attn_cfg=dict(embed_dims=128, num_heads=8, dropout=0.1),
                 ffn_cfg=dict(embed_dims=128, feedforward_channels=256, num_fcs=2, ffn_drop=0.1),
                 norm_cfg=dict(type="LN"), pos_encoding_cfg=dict(input_channel=2, num_pos_feats=128), **kwargs):
        super().__init__()
        d = self_attn_cfg["embed_dims"]
        self.self_attn = _MHA(d, self_attn_cfg["num_heads"], self_attn_cfg.get("dropout", 0.0))
        self.cross_attn = _MHA(d, cross_attn_cfg["num_heads"], cross_attn_cfg.get("dropout", 0.0))
        ff, drop = ffn_cfg["feedforward_channels"], ffn_cfg.get("ffn_drop", 0.0)
        self.ffn = FFN(d, ff, drop)
        self.norms = nn.ModuleList([nn.LayerNorm(d) for _ in range(3)])
        self.self_posembed = PositionEncodingLearned(**pos_encoding_cfg)
        self.cross_posembed = PositionEncodingLearned(**pos_encoding_cfg)

    def forward(self, query, key=None, query_pos=None, key_pos=None):
        """query [B, C, Nq], key [B, C, Nk], *_pos [B, N, 2] -> [B, C, Nq]."""
        qp = self.self_posembed.forward_nlc(query_pos)
        kp = self.cross_posembed.forward_nlc(key_pos)
        q, k = query.transpose(1, 2), key.transpose(1, 2)
        # (reference: self_attn(q, q, q + qp, qp, qp) and cross_attn(q, k, k + kp, qp, kp) -- query, key and value of the self
        # attention are all q + qp, key and value of the cross attention both k + kp: each sum is formed once)
        qq = q + qp
        q = self.norms[0](self.self_attn.attend(q, qq, qq, qq))
        kk = k + kp
        q = self.norms[1](self.cross_attn.attend(q, q + qp, kk, kk))
        q = self.norms[2](self.ffn(q))
        return q.transpose(1, 2)


class SeparateHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c_in, head_conv, final_kernel, padding=final_kernel // 2, dim=1))
                c_in = head_conv
            layers.append(nn.Conv1d(head_conv, classes, final_kernel, padding=final_kernel // 2, bias=True))
            self.add_module(head, nn.Sequential(*layers))
        getattr(self, "heatmap")[-1].bias.data.fill_(init_bias)

    def _pointwise(self):
        return all(isinstance(m, (ConvModule, nn.Conv1d)) and (m.conv if isinstance(m, ConvModule) else m).kernel_size == (1,)
                   and (m.conv if isinstance(m, ConvModule) else m).stride == (1,) for head in self.heads for m in getattr(self, head))

    def forward(self, x):
        """x [B, C, L] -> {head: [B, out, L]}.  With kernel size 1 (the reference's configuration) every layer is a linear map
        over channels: the stacks run on the row-major [B*L, C] matrix -- a view when x is the decoder's [B, L, C] output
        transposed -- as GEMMs + the fused row BatchNorm, instead of library Conv1d calls on [B, C, 200] (each of which brings
        zero-fill / cast helper launches in its backward).  Same parameters, same values."""
        if not (_HEAD_ROWS and x.is_cuda and x.dim() == 3 and self._pointwise()):
            return {head: getattr(self, head)(x) for head in self.heads}
        B, C, L = x.shape
        rows = x.transpose(1, 2).reshape(B * L, C)
        if torch.is_autocast_enabled("cuda") and rows.dtype != torch.get_autocast_dtype("cuda"):
            rows = rows.to(torch.get_autocast_dtype("cuda"))   # once, not once per head inside every F.linear
        out = {}
        for head in self.heads:
            h = rows
            for m in getattr(self, head):
                if isinstance(m, ConvModule):
                    h = F.linear(h, m.conv.weight.squeeze(-1), m.conv.bias)
                    if m.bn is not None:
                        h = m.bn(h, relu=m.act)
                    elif m.act:
                        h = F.relu(h)
                else:
                    h = F.linear(h, m.weight.squeeze(-1), m.bias)
            out[head] = h.view(B, L, -1).transpose(1, 2)
        return out


@MODELS.register_module()
class BEVFusionHead(nn.Module):
    """Forward of the TransFusion head (BF/bevfusion_head.py:198-299)."""

    def __init__(self, num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
                 num_decoder_layers=1, decoder_layer=dict(), num_heads=8, nms_kernel_size=3, bn_momentum=0.1,
                 common_heads=dict(center=[2, 2], height=[1, 2], dim=[3, 2], rot=[2, 2], vel=[2, 2]),
                 num_heatmap_convs=2, grid_size=(1440, 1440, 41), out_size_factor=8, train_cfg=None, test_cfg=None,
                 bbox_coder=None,
                 loss_cls=dict(type="mmdet.FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0),
                 loss_bbox=dict(type="mmdet.L1Loss", reduction="mean", loss_weight=0.25),
                 loss_heatmap=dict(type="mmdet.GaussianFocalLoss", reduction="mean", loss_weight=1.0), **kwargs):
        super().__init__()
        from .head_targets import TransFusionBBoxCoder, _assigner_weights
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.loss_cls_cfg, self.loss_bbox_cfg, self.loss_heatmap_cfg = dict(loss_cls), dict(loss_bbox), dict(loss_heatmap)
        if train_cfg is not None:
            grid_size, out_size_factor = train_cfg["grid_size"], train_cfg["out_size_factor"]
            assert train_cfg["assigner"]["type"] == "HungarianAssigner3D", "only the Hungarian assigner is implemented"
            self.assign_weights = _assigner_weights(train_cfg["assigner"])
        if bbox_coder is not None:
            bc = dict(bbox_coder)
            bc.pop("type", None)
            self.bbox_coder = TransFusionBBoxCoder(**bc)
        else:
            self.bbox_coder = None
        self.num_classes, self.num_proposals = num_classes, num_proposals
        self.num_decoder_layers, self.nms_kernel_size, self.auxiliary = num_decoder_layers, nms_kernel_size, auxiliary
        self.shared_conv = Conv2d(in_channels, hidden_channel, 3, padding=1)
        self.heatmap_head = nn.Sequential(ConvModule(hidden_channel, hidden_channel, 3, padding=1),
                                          Conv2d(hidden_channel, num_classes, 3, padding=1))
        self.class_encoding = nn.Conv1d(num_classes, hidden_channel, 1)
        self.decoder = nn.ModuleList([TransformerDecoderLayer(**decoder_layer) for _ in range(num_decoder_layers)])
        heads = dict(common_heads)
        heads.update(heatmap=(num_classes, num_heatmap_convs))
        self.prediction_heads = nn.ModuleList([SeparateHead(hidden_channel, heads) for _ in range(num_decoder_layers)])
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = bn_momentum
        xs, ys = grid_size[0] // out_size_factor, grid_size[1] // out_size_factor
        bx, by = torch.meshgrid(torch.linspace(0, xs - 1, xs), torch.linspace(0, ys - 1, ys), indexing="ij")
        self.register_buffer("bev_pos", torch.stack([bx + 0.5, by + 0.5], 0).view(1, 2, -1).permute(0, 2, 1),
                             persistent=False)

    def forward(self, feats, metas=None):
        inputs = feats[0] if isinstance(feats, (list, tuple)) else feats
        B = inputs.shape[0]
        fusion_feat = self.shared_conv(inputs)
        flat = fusion_feat.reshape(B, fusion_feat.shape[1], -1)
        bev_pos = self.bev_pos.expand(B, -1, -1)
        mixed = fusion_feat.is_cuda and torch.is_autocast_enabled("cuda")
        with torch.autocast("cuda", enabled=False), fp32_island(mixed):   # the reference's fp32 island (BF/bevfusion_head.py:218)
            dense_heatmap = self.heatmap_head(fusion_feat.float())
        heatmap = dense_heatmap.detach().sigmoid()
        pad = self.nms_kernel_size // 2
        local_max = torch.zeros_like(heatmap)
        inner = F.max_pool2d(heatmap, kernel_size=self.nms_kernel_size, stride=1, padding=0)
        local_max[:, :, pad:-pad, pad:-pad] = inner
        heatmap = (heatmap * (heatmap == local_max)).reshape(B, heatmap.shape[1], -1)
        top = heatmap.reshape(B, -1).topk(self.num_proposals, dim=-1).indices  # = argsort(descending)[:num_proposals]
        top_class, top_index = top // heatmap.shape[-1], top % heatmap.shape[-1]
        query_feat = flat.gather(index=top_index[:, None, :].expand(-1, flat.shape[1], -1), dim=-1)
        one_hot = F.one_hot(top_class, num_classes=self.num_classes)   # [B, P, classes]
        if _HEAD_ROWS and query_feat.is_cuda and self.class_encoding.kernel_size == (1,):
            # the k = 1 Conv1d over [B, classes, P] is a GEMM over the rows [B, P, classes]
            enc = F.linear(one_hot.to(query_feat.dtype), self.class_encoding.weight.squeeze(-1), self.class_encoding.bias)
            query_feat = query_feat + enc.permute(0, 2, 1)
        else:
            query_feat = query_feat + self.class_encoding(one_hot.permute(0, 2, 1).to(query_feat.dtype))
        query_pos = bev_pos.gather(index=top_index[:, :, None].expand(-1, -1, 2), dim=1)
        rets = []
        for i in range(self.num_decoder_layers):
            query_feat = self.decoder[i](query_feat, key=flat, query_pos=query_pos, key_pos=bev_pos)
            res = self.prediction_heads[i](query_feat)
            res["center"] = res["center"] + query_pos.permute(0, 2, 1)
            rets.append(res)
            query_pos = res["center"].detach().clone().permute(0, 2, 1)
        rets[0]["query_heatmap_score"] = heatmap.gather(index=top_index[:, None, :].expand(-1, self.num_classes, -1), dim=-1)
        rets[0]["dense_heatmap"] = dense_heatmap
        rets[0]["query_labels"] = top_class
        self.query_labels = top_class
        if not self.auxiliary:
            return ([rets[-1]],)
        # all decoder layers concatenated along the proposal axis (BF/bevfusion_head.py:292-299)
        keep = ("dense_heatmap", "dense_heatmap_old", "query_heatmap_score", "query_labels")
        new_res = {k: (rets[0][k] if k in keep else (torch.cat([r[k] for r in rets], dim=-1) if len(rets) > 1 else rets[0][k]))
                   for k in rets[0]}
        return ([new_res],)

    # ------------------------------------------------------------------ targets and losses
    def get_targets(self, batch_gt_instances_3d, preds_dict):
        """BF/bevfusion_head.py:450-674 for the whole batch on the device (csrc/head.hip): decode, three matching costs,
        Hungarian assignment per decoder layer, target scatter, dense heat-map.  `preds_dict` is the (single) dict of
        forward().  Returns labels i32[B, L*P], label_weights, bbox_targets, bbox_weights, ious, num_pos (host int),
        matched_ious (0-dim device tensor), heatmap f32[B, C, Y', X'].  No device->host read."""
        from . import head_targets as ht
        tc = self.train_cfg
        dev = preds_dict["center"].device
        packed = batch_gt_instances_3d if isinstance(batch_gt_instances_3d, ht.PackedGT) else None
        if packed is not None:
            gt_boxes, gt_labels, n_gt, counts = packed.boxes, packed.labels, packed.n_gt, packed.counts
        else:
            gt_boxes, gt_labels, n_gt, counts = ht.pack_gt(batch_gt_instances_3d, dev)
        P = self.num_proposals
        L = self.num_decoder_layers if self.auxiliary else 1
        vel = preds_dict.get("vel")
        outs, statuses = [], []
        for layer in range(L):
            boxes = self.bbox_coder.decode_boxes(preds_dict["rot"], preds_dict["dim"], preds_dict["center"],
                                                 preds_dict["height"], vel, p_off=layer * P, num=P)
            assigned, iou, _, status = ht.assign_batch(boxes, preds_dict["heatmap"], gt_boxes, gt_labels, n_gt,
                                                       tc["point_cloud_range"], self.assign_weights, p_off=layer * P)
            statuses.append(status)
            outs.append(ht.build_targets(assigned, iou, gt_boxes, gt_labels, self.num_classes, self.bbox_coder.code_size,
                                         tc["point_cloud_range"], tc["out_size_factor"], tc["voxel_size"],
                                         tc.get("pos_weight", -1)))
        labels, label_weights, bbox_targets, bbox_weights, ious = [torch.cat(x, dim=1) if L > 1 else x[0] for x in zip(*outs)]
        # status != 0: a NaN / inf matching cost (scipy's linear_sum_assignment raises "matrix contains invalid numeric
        # entries" there, BF/utils.py:270).  Kept on the device and folded into the losses as NaN by loss_by_feat, so the
        # failure is visible without a per-step host read; `assignment_status` is there for callers that want to raise.
        self.assignment_status = torch.stack(statuses) if L > 1 else statuses[0]
        pos_per_sample = [min(c, P) * L for c in counts]  # the Hungarian step matches min(#GT, #proposals) pairs
        num_pos = sum(pos_per_sample)
        if packed is not None:
            denom = packed.denom(P, L)
        else:
            denom = torch.tensor([max(n, 1) for n in pos_per_sample], dtype=torch.float32).to(dev, non_blocking=True)
        matched_ious = (ious.sum(1) / denom).mean()
        heatmap = ht.draw_heatmap(gt_boxes, gt_labels, n_gt, self.num_classes, tc["grid_size"], tc["point_cloud_range"],
                                  tc["voxel_size"], tc["out_size_factor"], tc["gaussian_overlap"], tc["min_radius"])
        return labels, label_weights, bbox_targets, bbox_weights, ious, num_pos, matched_ious, heatmap

    def loss_by_feat(self, preds_dicts, batch_gt_instances_3d, *args, **kwargs):
        """BF/bevfusion_head.py:696-796."""
        from . import head_targets as ht
        preds_dict = preds_dicts[0][0]
        (labels, label_weights, bbox_targets, bbox_weights, ious, num_pos, matched_ious,
         heatmap) = self.get_targets(batch_gt_instances_3d, preds_dict)
        loss_dict = dict()
        loss_dict["loss_heatmap"] = ht.gaussian_focal_loss_with_logits(
            preds_dict["dense_heatmap"], heatmap, 1e-4, self.loss_heatmap_cfg.get("loss_weight", 1.0))
        P = self.num_proposals
        L = self.num_decoder_layers if self.auxiliary else 1
        keys = ["center", "height", "dim", "rot"] + (["vel"] if "vel" in preds_dict else [])
        preds = torch.cat([preds_dict[k] for k in keys], dim=1)  # [B, code_size, L*P]
        code_weights = self.train_cfg.get("code_weights", None) or [1.0] * preds.shape[1]
        cw = getattr(self, "_code_weights", None)
        if cw is None or cw.device != preds.device:
            cw = self._code_weights = torch.tensor(code_weights[:preds.shape[1]], dtype=torch.float32).to(preds.device)
        avg = float(max(num_pos, 1))
        for layer in range(L):
            prefix = "layer_-1" if (layer == self.num_decoder_layers - 1 or (layer == 0 and not self.auxiliary)) else f"layer_{layer}"
            sl = slice(layer * P, (layer + 1) * P)
            cls_sum, box_sum = ht.query_losses(preds_dict["heatmap"], preds, labels[:, sl], label_weights[:, sl],
                                               bbox_targets[:, sl], bbox_weights[:, sl], cw, layer * P, P,
                                               self.loss_cls_cfg.get("gamma", 2.0), self.loss_cls_cfg.get("alpha", 0.25))
            loss_dict[f"{prefix}_loss_cls"] = cls_sum / avg * self.loss_cls_cfg.get("loss_weight", 1.0)
            loss_dict[f"{prefix}_loss_bbox"] = box_sum / avg * self.loss_bbox_cfg.get("loss_weight", 1.0)
        loss_dict["matched_ious"] = matched_ious
        # an invalid matching cost (NaN / inf) leaves every query unmatched: poison the losses instead of training on
        # background-only targets with the wrong normaliser (the reference raises from scipy at this point)
        # MULTIPLIED in (x * nan), not added: the gradient of every parameter then is NaN too, so the step is not only visibly
        # invalid but is skipped by the optimizers of this package (amp.skip_nonfinite_step) instead of being applied
        poison = torch.where(self.assignment_status.ne(0).any(), float("nan"), 1.0)
        for k in loss_dict:
            if "loss" in k:
                loss_dict[k] = loss_dict[k] * poison
        return loss_dict

    def check_assignment(self):
        """Host-side check of the last get_targets() (one device read): raises like scipy's linear_sum_assignment does on
        a cost matrix with invalid entries (BF/utils.py:267-270)."""
        if getattr(self, "assignment_status", None) is not None and bool(self.assignment_status.ne(0).any()):
            raise ValueError("matrix contains invalid numeric entries")

    def loss(self, batch_feats, batch_data_samples):
        """BF/bevfusion_head.py:676-694.  `batch_data_samples`: objects with `.gt_instances_3d` (+ `.metainfo`), or the
        ground truth itself as (boxes, labels) pairs."""
        from .head_targets import PackedGT
        if isinstance(batch_data_samples, PackedGT):  # ground truth prepared on the device ahead of the step
            return self.loss_by_feat(self(batch_feats, None), batch_data_samples)
        gts = [getattr(d, "gt_instances_3d", d) for d in batch_data_samples]
        metas = [getattr(d, "metainfo", None) for d in batch_data_samples]
        return self.loss_by_feat(self(batch_feats, metas), gts)

    def predict_by_feat(self, preds_dicts, metas=None, img=None, rescale=False, for_roi=False):
        """BF/bevfusion_head.py:322-448 with nms_type None (the nuScenes configs): score = sigmoid(cls) * heat-map score
        of the query's own class, decode with the score / centre-range filter.  One dict per sample."""
        preds = preds_dicts[0][0]
        P = self.num_proposals
        score = preds["heatmap"][..., -P:].sigmoid()
        one_hot = F.one_hot(self.query_labels, num_classes=self.num_classes).permute(0, 2, 1)
        score = score * preds["query_heatmap_score"] * one_hot
        vel = preds["vel"][..., -P:] if "vel" in preds else None
        nms_type = (self.test_cfg or {}).get("nms_type", None)
        rets = self.bbox_coder.decode(score, preds["rot"][..., -P:], preds["dim"][..., -P:], preds["center"][..., -P:],
                                      preds["height"][..., -P:], vel, filter=True)
        if nms_type is not None:
            from .head_targets import circle_nms, nms_bev, xywhr2xyxyr
            # nuScenes tasks (:358-378): classes 0-7 untouched, pedestrians (8) and traffic cones (9) with radius 0.175
            tasks = [(list(range(8)), -1.0), ([8], 0.175), ([9], 0.175)] if (self.test_cfg or {}).get("dataset") == "nuScenes" \
                else [([0], 0.7), ([1], 0.7), ([2], 0.7)]
            out = []
            for r in rets:
                boxes, scores, labels = r["bboxes"], r["scores"], r["labels"]
                keep_mask = torch.zeros_like(scores, dtype=torch.bool)
                for classes, radius in tasks:
                    task_mask = torch.zeros_like(keep_mask)
                    for c in classes:
                        task_mask |= labels == c
                    if radius > 0:
                        idx = torch.where(task_mask)[0]
                        if nms_type == "circle":
                            kept = circle_nms(torch.cat([boxes[idx, :2], scores[idx, None]], dim=1), radius)
                        else:  # any other value: rotated-IoU NMS with the task radius as the IoU threshold (:414-423)
                            bev = boxes[idx][:, [0, 1, 3, 4, 6]]
                            kept = nms_bev(xywhr2xyxyr(bev), scores[idx], radius, self.test_cfg.get("pre_max_size"),
                                           self.test_cfg.get("post_max_size"))
                        keep_mask[idx[kept]] = True
                    else:
                        keep_mask |= task_mask
                out.append(dict(bboxes=boxes[keep_mask], scores=scores[keep_mask], labels=labels[keep_mask]))
            rets = out
        return [dict(bboxes_3d=r["bboxes"], scores_3d=r["scores"], labels_3d=r["labels"].int()) for r in rets]

    def predict(self, batch_feats, batch_input_metas=None):
        return self.predict_by_feat(self(batch_feats, batch_input_metas), batch_input_metas)
