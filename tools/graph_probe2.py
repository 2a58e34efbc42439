import os, sys
sys.path.insert(0, "/root/repo")
os.environ["BENCH_GRAPH"] = "1"
import torch, bench
dev = torch.device("cuda:0")
if os.environ.get("P2_SETDEV"): torch.cuda.set_device(0)
wl = bench.FullModel(dev, 4, 40000)
mode = os.environ.get("P2_MODE", "step")
if mode == "step":
    for i in range(8):
        r = wl.step(); print(i, "ok", flush=True)
elif mode == "direct":
    for i in range(3):
        wl._eager_step()
    wl._capture()
    print("captured", flush=True)
    r = wl._static_loss
elif mode == "direct_sync":
    for i in range(3):
        wl._eager_step()
    torch.cuda.synchronize()
    wl._capture()
    print("captured", flush=True)
    r = wl._static_loss
else:
    from bevfusion_amd.head_targets import PackedGT
    for i in range(3):
        wl._eager_step()
    torch.cuda.synchronize()
    gts = PackedGT(wl.gts, dev)
    wl.opt.zero_grad()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r = wl._forward_backward(gts)
        wl._update()
    print("captured", flush=True)
    for i in range(3):
        g.replay()
torch.cuda.synchronize()
print("done", float(r))
