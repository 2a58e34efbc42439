"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares.
No compute call is made (no GPU here)."""
import ctypes
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(bfhip_\w+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def libpath():
    import bevfusion_amd
    from bevfusion_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        bevfusion_amd.build()
    return _lib.LIB_PATH


def test_header_declares_symbols():
    assert len(declared_symbols()) >= 7


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_python_binding_covers_header(libpath):
    from bevfusion_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.bfhip_abi_version() >= 1


def test_library_is_gfx950_only(libpath):
    """The fat binary embedded in the .so carries gfx950 code objects and no other GPU target."""
    blob = open(libpath, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "bevfusion_hip.h"\nint (*fp)(void) = bfhip_abi_version;\nint main(void){return fp != 0 ? 0 : 1;}\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_product_never_imports_oracle():
    """The product path must not route through the oracle or any CPU fallback."""
    pkg = os.path.join(ROOT, "bevfusion-3d_object_detection_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M) or "libbevfusion_oracle" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bevfusion_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected():
    import torch
    from bevfusion_amd.ops import Voxelization
    vox = Voxelization([1.0, 1.0, 1.0], [0, 0, 0, 4, 4, 4], 5, 20)
    with pytest.raises(RuntimeError):
        vox(torch.zeros(10, 4))


def test_conv2d_supported_rejects_what_the_kernels_cannot_hold():
    """bfhip_conv2d_supported is what routes a layer to the HIP kernels or to the library (conv2d.Conv2d.hip_eligible): it must say
    no to geometries whose tap tables do not fit beside the LDS stages -- for Cin (forward, weight gradient) AND Cout (the data
    gradient gathers over Cout) -- so that such layers fall back to torch instead of failing at launch.  Host-only call."""
    from bevfusion_amd import _lib
    lib = _lib.load()
    ok = lambda *a: lib.bfhip_conv2d_supported(*a)  # noqa: E731  (N, H, W, Cin, Cout, KH, KW, stride, pad, dil)
    assert ok(4, 180, 180, 336, 256, 3, 3, 1, 1, 1) == 1            # ConvFuser
    assert ok(24, 8, 22, 512, 512, 3, 3, 1, 1, 1) == 1              # widest 3x3 of ResNet-50: 576 pieces
    assert ok(24, 16, 44, 3072, 256, 1, 1, 1, 0, 1) == 1            # LSS-FPN lateral 1x1
    assert ok(1, 32, 32, 4096, 64, 3, 3, 1, 1, 1) == 0              # 4608 pieces along Cin
    assert ok(1, 32, 32, 64, 4096, 3, 3, 1, 1, 1) == 0              # 4608 pieces along Cout (data gradient)
    assert ok(1, 32, 32, 64, 65536, 1, 1, 1, 0, 1) == 0             # 16-bit channel field of the tap table
    assert ok(1, 32, 32, 60, 64, 3, 3, 1, 1, 1) == 0                # channels not a multiple of 8
    assert ok(1, 32, 32, 64, 64, 3, 3, 3, 1, 1) == 0                # stride not a power of two
