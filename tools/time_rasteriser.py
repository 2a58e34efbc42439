import os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bevfusion_amd
from bevfusion_amd import synthetic
from bevfusion_amd.depth_lss import DepthLSSTransform
N = synthetic.NUSC
dev = torch.device("cuda:0")
vt = DepthLSSTransform(in_channels=256, out_channels=80, image_size=N["image_size"], feature_size=N["feature_size"], xbound=N["xbound"], ybound=N["ybound"], zbound=N["zbound"], dbound=N["dbound"], downsample=2).to(dev)
B = 4
rig = synthetic.camera_rig(batch=B, seed=1, train_aug=True)
t = {k: torch.from_numpy(v).to(dev) for k, v in rig.items()}
pts = [torch.from_numpy(synthetic.lidar_sweep(40000, seed=1000 + i)).to(dev) for i in range(B)]
img = torch.zeros(B, 6, 256, 32, 88, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
d = vt.rasterise_depth(img, pts, t["lidar2image"], t["img_aug_matrix"], t["lidar_aug_matrix"])
print("rasterise_depth ms", timeit(lambda: vt.rasterise_depth(img, pts, t["lidar2image"], t["img_aug_matrix"], t["lidar_aug_matrix"])))
print("gt_depth_distribution ms", timeit(lambda: vt.gt_depth_distribution(d.view(B * 6, 1, 256, 704), B, 6)))
print("nonzero depth pixels per frame", int((d > 0).sum()) / B)
