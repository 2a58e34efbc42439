"""Seeded synthetic nuScenes-shaped inputs for tests and bench.py (no dataset, no network).

Shapes follow the reference's nuScenes configs
(projects/BEVFusion/configs/nuscenes/bevfusion_lidar-cam_voxel0075_second_secfpn_8xb4-cyclic-20e_nus-3d.py:45-57
and ..._lidar_voxel0075...py:10-11,49-65): 6 cameras, 256x704 images, 32x88 feature maps,
D=118 depth bins, C=80, BEV 360x360, voxels 0.075/0.075/0.2 m on [-54,54]^2 x [-5,3].

The camera rig is a NOMINAL nuScenes rig built from the publicly documented sensor layout
(front/back cameras, yaw 0, +-55, +-110, 180 degrees, 1600x900 images with f ~ 1266 px; back
camera f ~ 809 px).  The real calibration shipped with the reference is a pickle and is not
loaded.  Matrix conventions are the reference's (BF/loading.py:138-158,
BF/transforms_3d.py:51-61,113-117): cam2img 4x4 with K top-left, camera2lidar = [R | t],
img_aug_matrix = resize 0.48 then crop (32,176)  ->  diag(0.48,0.48,1,1) with t = (-32,-176,0).
"""
import math

import numpy as np

NUSC = dict(
    point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0],
    voxel_size=[0.075, 0.075, 0.2],
    max_num_points=10,
    max_voxels=(120000, 160000),
    sparse_shape=[1440, 1440, 41],
    image_size=(256, 704),
    feature_size=(32, 88),
    xbound=[-54.0, 54.0, 0.3],
    ybound=[-54.0, 54.0, 0.3],
    zbound=[-10.0, 10.0, 20.0],
    dbound=[1.0, 60.0, 0.5],
    num_cams=6,
    C=80,
)

# HDL-32E: 32 beams from -30.67 to +10.67 degrees
_ELEV = np.deg2rad(np.linspace(-30.67, 10.67, 32))


def lidar_sweep(n=40000, seed=1000, features=5, sensor_height=1.84):
    """One synthetic 32-beam spinning-LiDAR sweep of a street-like scene: fixed azimuth firing steps,
    ground returns on the down-looking beams, two building fronts (a street canyon) and a few box-shaped
    obstacles.  Vertical surfaces make neighbouring rings land in the same (x, y) column, as in real sweeps.
    Tuned so that 40 k points give ~18-20 k voxels at 0.075/0.075/0.2 m and sparse-encoder stage sizes
    close to the real nuScenes sample quoted in SURVEY.md 8 (17.7 k voxels from a 34.7 k-point sweep).
    Returns f32[n, features] = (x, y, z, intensity, dt)."""
    rng = np.random.default_rng(seed)
    per_ring = (n + 31) // 32
    ring = np.repeat(np.arange(32), per_ring)[:n]
    step = np.tile(np.arange(per_ring), 32)[:n]
    az = 2.0 * math.pi * (step + rng.uniform(-0.08, 0.08, n)) / per_ring
    el = _ELEV[ring]
    ca, sa = np.cos(az), np.sin(az)
    # horizontal range to the scene along each azimuth
    yaw = rng.uniform(-0.2, 0.2)
    half_w = rng.uniform(9.0, 14.0, 2)                    # lateral distance of the two building fronts
    lat = ca * math.cos(yaw) + sa * math.sin(yaw)         # component across the street
    with np.errstate(divide="ignore"):
        wall = np.where(lat > 1e-3, half_w[0] / lat, np.where(lat < -1e-3, -half_w[1] / lat, np.inf))
    horiz = np.minimum(wall, 68.0)
    for _ in range(14):                                   # boxes: cars / poles, 1.5-5 m wide
        c_az, c_r = rng.uniform(0, 2 * math.pi), rng.uniform(4.0, 40.0)
        w_ang = rng.uniform(0.75, 2.5) / c_r
        d = np.abs(np.angle(np.exp(1j * (az - c_az))))
        horiz = np.where(d < w_ang, np.minimum(horiz, c_r + rng.uniform(0, 0.3)), horiz)
    height_cap = rng.uniform(1.0, 2.2)                    # box tops: higher beams pass over them to the wall
    r_scene = horiz / np.maximum(np.cos(el), 1e-3)
    over = (r_scene * np.sin(el) + 0.0 > height_cap) & (horiz < wall - 1e-3)
    r_scene = np.where(over, wall / np.maximum(np.cos(el), 1e-3), r_scene)
    with np.errstate(divide="ignore"):
        ground = np.where(el < -0.01, sensor_height / np.sin(-el), np.inf)
    r = np.minimum(ground, r_scene) * rng.normal(1.0, 0.0015, n)
    keep_r = np.clip(r, 0.8, 75.0)
    x = keep_r * np.cos(el) * ca
    y = keep_r * np.cos(el) * sa
    z = keep_r * np.sin(el) + rng.normal(0.0, 0.01, n)
    pts = np.zeros((n, features), np.float32)
    pts[:, 0], pts[:, 1], pts[:, 2] = x, y, z
    if features > 3:
        pts[:, 3] = rng.uniform(0.0, 255.0, n)
    perm = rng.permutation(n) if False else np.argsort(step * 32 + ring, kind="stable")  # firing order: azimuth-major
    return pts[perm]


def uniform_points(n=40000, seed=0, features=5, rng_range=(-54.0, -54.0, -5.0, 54.0, 54.0, 3.0), margin=1.0):
    """Stress distribution: uniform in (slightly more than) the box -> M ~ N voxels, some points outside."""
    rng = np.random.default_rng(seed)
    lo = np.array(rng_range[:3]) - margin
    hi = np.array(rng_range[3:]) + margin
    pts = np.zeros((n, features), np.float32)
    pts[:, :3] = rng.uniform(lo, hi, (n, 3))
    if features > 3:
        pts[:, 3:] = rng.uniform(0.0, 1.0, (n, features - 3))
    return pts


def _cam2lidar(yaw_deg, forward=0.0, lateral=0.0, z=-0.3, pitch_deg=0.0, roll_deg=0.0):
    """Camera optical frame (x right, y down, z forward) -> lidar frame (x right, y forward, z up);
    yaw counter-clockwise from +y; small mounting pitch/roll applied in the camera frame (a perfectly
    level rig would map all 32 feature rows of a column to the same BEV cell, which real rigs do not)."""
    psi = math.radians(yaw_deg)
    f = np.array([-math.sin(psi), math.cos(psi), 0.0])
    r = np.array([math.cos(psi), math.sin(psi), 0.0])
    dn = np.array([0.0, 0.0, -1.0])
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2] = r, dn, f
    p, q = math.radians(pitch_deg), math.radians(roll_deg)
    Rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]])
    Rz = np.array([[math.cos(q), -math.sin(q), 0], [math.sin(q), math.cos(q), 0], [0, 0, 1]])
    T[:3, :3] = T[:3, :3] @ Rx @ Rz
    T[:3, 3] = f * forward + r * lateral + np.array([0.0, 0.0, z])
    return T


def camera_rig(batch=1, seed=None, train_aug=False):
    """Returns dict of f32 arrays: camera_intrinsics [B,6,4,4], camera2lidar [B,6,4,4],
    lidar2image [B,6,4,4], img_aug_matrix [B,6,4,4], lidar_aug_matrix [B,4,4].

    train_aug: seeded rotation in +-pi/4, scale in [0.9,1.1], translation sigma 0.5 for the lidar
    augmentation (reference config ...lidar-cam...py:92-96); image aug stays the eval one."""
    yaws = [0.0, -55.0, 55.0, 180.0, 110.0, -110.0]  # F, FR, FL, B, BL, BR
    fwd = [1.70, 1.55, 1.55, 0.05, 1.05, 1.05]
    lat = [0.0, 0.50, -0.50, 0.0, -0.48, 0.48]
    # nominal ~1 degree mounting tolerances; chosen so that the eval-aug frustum statistics match the
    # real sample quoted in SURVEY.md 8 (kept ~1.83 M, ~97 k intervals, mean length ~19, max ~890)
    pitch = [0.92, -0.69, 1.15, -1.04, 0.58, -1.27]
    roll = [-0.35, 1.04, 0.58, -0.81, 1.27, 0.46]
    K = np.zeros((6, 4, 4))
    c2l = np.zeros((6, 4, 4))
    for i in range(6):
        fx = 809.22 if i == 3 else 1266.42
        cx, cy = (829.22, 481.78) if i == 3 else (816.27, 491.51)
        K[i] = np.eye(4)
        K[i, 0, 0] = K[i, 1, 1] = fx
        K[i, 0, 2], K[i, 1, 2] = cx, cy
        c2l[i] = _cam2lidar(yaws[i], fwd[i], lat[i], pitch_deg=pitch[i], roll_deg=roll[i])
    l2c = np.linalg.inv(c2l)
    l2i = K @ l2c
    aug = np.eye(4)
    aug[0, 0] = aug[1, 1] = 0.48
    aug[0, 3], aug[1, 3] = -32.0, -176.0
    out = dict(
        camera_intrinsics=np.broadcast_to(K, (batch, 6, 4, 4)).astype(np.float32).copy(),
        camera2lidar=np.broadcast_to(c2l, (batch, 6, 4, 4)).astype(np.float32).copy(),
        lidar2image=np.broadcast_to(l2i, (batch, 6, 4, 4)).astype(np.float32).copy(),
        img_aug_matrix=np.broadcast_to(aug, (batch, 6, 4, 4)).astype(np.float32).copy(),
    )
    la = np.broadcast_to(np.eye(4), (batch, 4, 4)).copy()
    if train_aug:
        rng = np.random.default_rng(0 if seed is None else seed)
        for b in range(batch):
            th = rng.uniform(-math.pi / 4, math.pi / 4)
            s = rng.uniform(0.9, 1.1)
            R = np.array([[math.cos(th), -math.sin(th), 0], [math.sin(th), math.cos(th), 0], [0, 0, 1]])
            la[b, :3, :3] = s * R
            la[b, :3, 3] = rng.normal(0.0, 0.5, 3)
    out["lidar_aug_matrix"] = la.astype(np.float32)
    return out


def create_frustum(image_size=NUSC["image_size"], feature_size=NUSC["feature_size"], dbound=NUSC["dbound"]):
    """(D, fH, fW, 3) pixel-depth grid (BF/depth_lss.py:53-66): xs/ys = linspace over the image,
    ds = arange(dbound) -- evaluated in fp32 like torch.arange/linspace(dtype=float)."""
    import torch
    iH, iW = image_size
    fH, fW = feature_size
    ds = torch.arange(*dbound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = ds.shape[0]
    xs = torch.linspace(0, iW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
    ys = torch.linspace(0, iH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
    return torch.stack((xs, ys, ds), -1).contiguous()


def gen_dx_bx(xbound=NUSC["xbound"], ybound=NUSC["ybound"], zbound=NUSC["zbound"]):
    """dx, bx, nx of the BEV grid (BF/depth_lss.py:14-18; nx by float division then truncation)."""
    import torch
    rows = [xbound, ybound, zbound]
    dx = torch.tensor([r[2] for r in rows], dtype=torch.float32)
    bx = torch.tensor([r[0] + r[2] / 2.0 for r in rows], dtype=torch.float32)
    nx = torch.tensor([int((r[1] - r[0]) / r[2]) for r in rows], dtype=torch.int64)
    return dx, bx, nx


# nuScenes detection classes (config class_names order) with typical (dx, dy, dz) metres
_CLASS_SIZES = np.array([
    [4.6, 1.95, 1.73],   # car
    [6.9, 2.5, 2.84],    # truck
    [6.4, 2.85, 3.2],    # construction_vehicle
    [11.0, 2.95, 3.5],   # bus
    [12.3, 2.9, 3.87],   # trailer
    [0.5, 2.5, 0.98],    # barrier
    [2.1, 0.77, 1.47],   # motorcycle
    [1.7, 0.6, 1.28],    # bicycle
    [0.73, 0.67, 1.77],  # pedestrian
    [0.41, 0.41, 1.07],  # traffic_cone
], np.float32)
_CLASS_FREQ = np.array([0.43, 0.08, 0.013, 0.014, 0.022, 0.13, 0.011, 0.01, 0.20, 0.09])


def gt_boxes(seed=3000, n=None, point_cloud_range=NUSC["point_cloud_range"]):
    """Synthetic ground truth of one frame: boxes f32[G, 9] = (x, y, z_bottom, dx, dy, dz, yaw, vx, vy) in the LiDAR
    frame (the layout of `LiDARInstance3DBoxes.tensor` with velocities) and labels i64[G].  G ~ U(15, 60) unless given;
    class frequencies and sizes are nuScenes-like, centres stay inside the detection range."""
    rs = np.random.RandomState(seed)
    g = int(rs.randint(15, 61)) if n is None else int(n)
    labels = rs.choice(10, size=g, p=_CLASS_FREQ / _CLASS_FREQ.sum()).astype(np.int64)
    lo, hi = np.array(point_cloud_range[:2], np.float32) + 1.0, np.array(point_cloud_range[3:5], np.float32) - 1.0
    xy = (rs.uniform(0, 1, (g, 2)) * (hi - lo) + lo).astype(np.float32)
    size = _CLASS_SIZES[labels] * rs.uniform(0.85, 1.2, (g, 3)).astype(np.float32)
    z = rs.uniform(-2.2, -1.2, (g, 1)).astype(np.float32)
    yaw = rs.uniform(-np.pi, np.pi, (g, 1)).astype(np.float32)
    vel = (rs.normal(0, 2.0, (g, 2)) * (rs.uniform(0, 1, (g, 1)) < 0.4)).astype(np.float32)
    return np.concatenate([xy, z, size, yaw, vel], 1).astype(np.float32), labels


def bev_pool_case(seed, n, C, B, D, H, W, integer):
    """Seeded sorted bev_pool inputs with a long-tailed interval-length distribution (numpy PCG64: stable across boxes).
    Returns x f32[n, C], geom i32[n, 4] (x, y, z, b), ranks i64[n] (depth_lss.py:165-170 rank formula), sorted by rank."""
    rng = np.random.default_rng(seed)
    cells = B * D * H * W
    lengths = []
    while sum(lengths) < n:
        r = rng.random()
        ln = int(rng.geometric(1 / 14.0)) if r < 0.97 else int(rng.integers(200, 900))
        lengths.append(ln)
    lengths[-1] -= sum(lengths) - n
    m = len(lengths)
    assert m <= cells
    cell_ids = np.sort(rng.choice(cells, m, replace=False))
    rank = np.repeat(cell_ids, lengths).astype(np.int64)
    # rank = x*(W*D*B) + y*(D*B) + z*B + b   with nx = (H, W, D) in the reference's naming (x <-> H axis)
    xs, rem = rank // (W * D * B), rank % (W * D * B)
    ys, rem = rem // (D * B), rem % (D * B)
    zs, bs = rem // B, rem % B
    geom = np.stack([xs, ys, zs, bs], 1).astype(np.int32)
    if integer:
        x = rng.integers(-8, 9, (n, C)).astype(np.float32)
    else:
        x = rng.standard_normal((n, C)).astype(np.float32)
    return x, geom, rank
