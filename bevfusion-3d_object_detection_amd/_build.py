"""Builds csrc/*.hip into csrc/libbevfusion_hip.so with hipcc for gfx950 (in-tree, incremental).

Usage: python _build.py [--force]
hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the source tree.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libbevfusion_hip.so")
ARCH = "gfx950"
FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH,
    "-ffp-contract=off",                       # index math must match the reference's unfused fp32
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
] + os.environ.get("BFHIP_EXTRA_FLAGS", "").split()


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(out, deps):
    return (not os.path.exists(out)) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "bevfusion_hip.h")]
    cc = hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [cc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        return s, r.stderr

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s, err in ex.map(compile_one, jobs):
                if verbose:
                    print("[build] compiled", os.path.basename(s), file=sys.stderr)
                    if err.strip():
                        print(err, file=sys.stderr)
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[build] linked", LIB, file=sys.stderr)
    build_torch_binding(force=force, verbose=verbose)
    return LIB


TORCH_EXT = os.path.join(CSRC, "bfhip_torch_ext.so")


def build_torch_binding(force=False, verbose=True):
    """csrc/torch_binding.cpp -> csrc/bfhip_torch_ext.so: host-only C++ autograd front-ends (pybind11 module) over the C
    ABI, compiled with g++ against the installed torch.  Optional: the Python (ctypes) path is used when it is absent."""
    import sysconfig
    import torch
    src = os.path.join(CSRC, "torch_binding.cpp")
    hdr = os.path.join(HERE, "..", "include", "bevfusion_hip.h")
    if not (force or _stale(TORCH_EXT, [src, hdr, LIB])):
        return TORCH_EXT
    tdir = os.path.dirname(torch.__file__)
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", TORCH_EXT,
           "-I" + os.path.join(tdir, "include"), "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=bfhip_torch_ext",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-L" + os.path.join(tdir, "lib"), "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_python",
           "-L" + CSRC, "-lbevfusion_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(tdir, "lib"), "-Wno-attributes"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch binding failed to build:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("[build] built", TORCH_EXT, file=sys.stderr)
    return TORCH_EXT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
