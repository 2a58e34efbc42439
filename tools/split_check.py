import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch, torch.nn.functional as F
import bevfusion_amd
import bevfusion_amd.conv2d as _c2s
_c2s._SPLIT_SCOPE[0] = 1  # as inside an fp32 island of a mixed-precision step
from bevfusion_amd.conv2d import Conv2d
dev = torch.device("cuda:0")
def l2(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for (N, H, W, Cin, Cout, k, s, p) in [(2, 45, 52, 128, 128, 3, 1, 1), (2, 44, 52, 128, 256, 3, 2, 1), (2, 45, 53, 128, 256, 3, 2, 1), (2, 45, 52, 256, 256, 3, 1, 1), (2, 45, 52, 64, 128, 1, 1, 0), (4, 90, 90, 256, 256, 3, 1, 1), (4, 180, 180, 256, 128, 3, 1, 1)]:
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=s, padding=p)
    gy = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    ref.backward(gy.double())
    conv = Conv2d(Cin, Cout, k, stride=s, padding=p, bias=False).to(dev).train()
    with torch.no_grad():
        conv.weight.copy_(w)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    print((N, H, W, Cin, Cout, k, s, p), "split_eligible", conv.split_eligible(xg))
    y = conv(xg)
    y.backward(gy.to(dev))
    print("   y %.2e  dx %.2e  dw %.2e" % (l2(y.detach().cpu(), ref.detach()), l2(xg.grad.cpu(), xr.grad), l2(conv.weight.grad.cpu(), wr.grad)))
