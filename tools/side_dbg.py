"""Determinism check: which parts of the step are bit-reproducible run to run (same inputs, same weights)."""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bevfusion_amd
from bevfusion_amd.bevfusion import nuscenes_config
from bevfusion_amd.registry import MODELS
from test_model_gpu import _inputs
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = MODELS.build(nuscenes_config()).to(dev).train()
inp = _inputs(dev, 1)
def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-30))
# LiDAR branch alone (hand-written kernels + fused BN1d), fp32 and bf16-autocast
for amp in (False, True):
    outs = []
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            outs.append(model.extract_pts_feat(inp).detach().clone())
    print("lidar branch amp=%s bit-identical:" % amp, torch.equal(outs[0], outs[1]), rel(outs[0], outs[1]))
# camera branch: backbone+neck (MIOpen) then view transform (ours) on FIXED neck features
with torch.autocast("cuda", dtype=torch.bfloat16):
    B, N = 1, 6
    x = model.img_neck(model.img_backbone(inp["imgs"].reshape(6, 3, 256, 704)))[0]
    x2 = model.img_neck(model.img_backbone(inp["imgs"].reshape(6, 3, 256, 704)))[0]
print("ResNet-50 + FPN (MIOpen, bf16) bit-identical:", torch.equal(x, x2), rel(x.float(), x2.float()))
xx = x.detach().reshape(1, 6, *x.shape[1:]).float()
vt = model.view_transform
outs = []
for _ in range(2):
    with torch.autocast("cuda", enabled=False):
        o, _ = vt(xx, inp["points"], inp["lidar2img"], inp["cam2img"], inp["cam2lidar"], inp["img_aug_matrix"], inp["lidar_aug_matrix"], None)
    outs.append(o.detach().float().clone())
print("view transform (plan + raster + depthnet(MIOpen bf16) + lift-splat + downsample) bit-identical:", torch.equal(outs[0], outs[1]), rel(outs[0], outs[1]))
