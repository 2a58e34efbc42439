import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch, torch.nn.functional as F
from torch import nn
import bevfusion_amd
import bevfusion_amd.conv2d as _c2s
_c2s._SPLIT_SCOPE[0] = 1  # as inside an fp32 island of a mixed-precision step
import bevfusion_amd.conv2d as c2
import bevfusion_amd.bn2d as b2
from bevfusion_amd.conv2d import Conv2d
from bevfusion_amd.bn2d import BatchNorm2dAct
dev = torch.device("cuda:0")
def l2(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
torch.manual_seed(0)
N, C, H, W = 2, 80, 45, 52
x = torch.randn(N, C, H, W)
seed = torch.randn(N, C, (H + 1) // 2, (W + 1) // 2)
def build():
    torch.manual_seed(1)
    return nn.Sequential(Conv2d(C, C, 3, padding=1, bias=False), BatchNorm2dAct(C, act=True), Conv2d(C, C, 3, stride=2, padding=1, bias=False), BatchNorm2dAct(C, act=True))
ref = build().double().train()
xr = x.double().requires_grad_(True)
yr = ref(xr); (yr * seed.double()).sum().backward()
for split in (True, False):
    for fused in (True, False):
        c2.FP32_SPLIT, b2.FUSED_BN2D = split, fused
        m = build().to(dev).train()
        xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = m(xg); (y * seed.to(dev)).sum().backward()
        print("split", split, "fusedBN", fused, "y %.2e dx %.2e" % (l2(y.detach(), yr.detach()), l2(xg.grad, xr.grad)),
              " ".join("%s %.2e" % (n, l2(p.grad, dict(ref.named_parameters())[n].grad)) for n, p in m.named_parameters()))
