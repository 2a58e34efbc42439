"""CPU, world_size 2, gloo: the N>1 path of bench.py (env rendezvous on 127.0.0.1, DDP gradient all-reduce, barrier,
MAX-over-ranks timing, single JSON line from rank 0).  The HIP ops have no CPU path, so the rehearsal workload is the
dense BEV tail only; what is tested is the distributed plumbing the GPU run shares."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nproc, tmp_path, extra=(), grad_sync="flat"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_FINGERPRINT_DIR=str(tmp_path), OMP_NUM_THREADS="2",
               BENCH_GRAD_SYNC=grad_sync)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc),
           "--workload", "dist_selftest", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--batch", "2", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line from rank 0, got %d" % len(lines)
    return json.loads(lines[0])


def test_two_ranks_gloo(tmp_path):
    out = _run(2, tmp_path)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 4 and out["config"]["frames_per_gpu_per_step"] == 2
    # value = frames of ALL ranks / max-over-ranks time
    assert abs(out["value"] - 2 * 2 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-3
    # both ranks hold the same (averaged) gradients although they saw different batches
    fps = [float(open(os.path.join(tmp_path, "rank%d.txt" % r)).read()) for r in range(2)]
    assert fps[0] > 0 and abs(fps[0] - fps[1]) <= 1e-9 * abs(fps[0])
    assert "flat all-reduce" in out["config"]["parallelism"]
    # the flat per-dtype all-reduce (bevfusion_amd/grad_sync.py) and torch DDP produce the same averaged gradients
    out_ddp = _run(2, tmp_path, grad_sync="ddp")
    assert "DDP" in out_ddp["config"]["parallelism"]
    fps_ddp = [float(open(os.path.join(tmp_path, "rank%d.txt" % r)).read()) for r in range(2)]
    assert abs(fps_ddp[0] - fps_ddp[1]) <= 1e-9 * abs(fps_ddp[0])
    assert abs(fps_ddp[0] - fps[0]) <= 1e-5 * abs(fps[0])


def test_single_rank_json_contract(tmp_path):
    env = dict(os.environ, BENCH_FINGERPRINT_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "dist_selftest", "--backend", "gloo",
                        "--steps", "2", "--warmup", "1", "--batch", "2"], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["vs_baseline"] is None


def test_flat_grad_allreduce_handles_missing_grads_single_process(tmp_path):
    """FlatGradAllReduce on a 1-rank gloo group: gradients survive the round trip unchanged, a parameter without a gradient
    gets zeros, mixed dtypes go through separate flat buffers."""
    code = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bevfusion_amd
from bevfusion_amd.grad_sync import FlatGradAllReduce, broadcast_parameters
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
dist.init_process_group("gloo", rank=0, world_size=1)
a = torch.nn.Parameter(torch.randn(3, 4)); b = torch.nn.Parameter(torch.randn(5).to(torch.bfloat16)); c = torch.nn.Parameter(torch.randn(2))
m = torch.nn.ParameterList([a, b, c])
broadcast_parameters(m)
(a.sum() * 2 + b.float().sum() * 3).backward()
ga, gb = a.grad.clone(), b.grad.clone()
gs = FlatGradAllReduce(m.parameters())
assert len(gs.groups) == 2 and gs.bytes_per_step() == (12 + 2) * 4 + 5 * 2
gs.reduce()
assert torch.equal(a.grad, ga) and torch.equal(b.grad, gb) and c.grad is not None and float(c.grad.abs().sum()) == 0.0
print("ok")
""" % (ROOT, _free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


_WS4 = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bevfusion_amd
from bevfusion_amd.grad_sync import FlatGradAllReduce
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
shapes = [((33, 7, 3, 3), torch.bfloat16), ((129,), torch.float32), ((64, 31), torch.bfloat16), ((5, 5), torch.float32)]
def grads(r):
    g = torch.Generator().manual_seed(100 + r)
    return [(torch.randn(s, generator=g) * (1 + 3 * i)).to(dt) for i, (s, dt) in enumerate(shapes)]
exact = [sum(g.double() for g in col) / world for col in zip(*[grads(r) for r in range(world)])]   # fp64 mean of what each rank holds
res = {}
for mode in ("a2a", "allreduce_fp32", "allreduce_bf16"):
    ps = [torch.nn.Parameter(torch.zeros(s, dtype=dt)) for s, dt in shapes]
    if rank == 1:
        ps[1].requires_grad_(True)
    for p, g in zip(ps, grads(rank)):
        p.grad = g.clone()
    if mode == "a2a" and rank == 2:
        ps[3].grad = None                       # an unused parameter on one rank contributes zeros
    gs = FlatGradAllReduce(ps, exchange=mode)
    assert len(gs.groups) == 2
    gs.reduce()
    errs = []
    for i, (p, e) in enumerate(zip(ps, exact)):
        if mode == "a2a" and i == 3:
            e = e - grads(2)[3].double() / world
        errs.append(float((p.grad.double() - e).norm() / e.norm()))
        gathered = [torch.empty_like(p.grad) for _ in range(world)]
        dist.all_gather(gathered, p.grad)
        assert all(torch.equal(gathered[0], t) for t in gathered), "ranks disagree"
    res[mode] = errs
if rank == 0:
    print("ERRS", res)
    for mode in ("a2a", "allreduce_fp32"):
        assert res[mode][0] < 2.5e-3 and res[mode][2] < 2.5e-3, res      # ONE bf16 rounding of the mean (2^-9 = 1.95e-3 max rel)
        assert res[mode][1] < 1e-6 and res[mode][3] < 1e-6, res
    assert max(res["a2a"][0], res["a2a"][2]) <= max(res["allreduce_bf16"][0], res["allreduce_bf16"][2]) + 1e-9, res
    print("ok")
dist.destroy_process_group()
"""


def test_flat_exchange_world4_mixed_dtypes_fp32_accumulation(tmp_path):
    """World size 4 on gloo, mixed bf16 / fp32 parameters, ragged sizes (padding of the all-to-all shards), one parameter
    without a gradient on one rank: every rank ends with identical gradients; the bf16 group's mean is accumulated in fp32
    (error of ONE bf16 rounding vs the fp64 mean) with both the all-to-all exchange and the widened all-reduce, and is no
    worse than the lossy bf16-sum all-reduce."""
    script = tmp_path / "ws4.py"
    script.write_text(_WS4 % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
