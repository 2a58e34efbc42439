#!/usr/bin/env bash
# Builds the REFERENCE's own CPU voxelization extension (unmodified sources, read in place
# from /root/reference) into oracle/_ref/voxel_layer_ref*.so.
#
# TEST INFRASTRUCTURE ONLY.  The output is used by tests/ and tools/make_golden.py to pin
# oracle/bevfusion_oracle.c against the real reference; it is never imported by the product.
# Nothing is copied into the repo: g++ reads the three translation units where they lie.
#   sources: projects/BEVFusion/bevfusion/ops/voxel/src/{voxelization.cpp,voxelization_cpu.cpp}
#   (scatter_points_cpu.cpp is declared in voxelization.h but never bound/called -> not needed)
# bev_pool has no CPU source in the reference (CUDA-only, bev_pool.cpp includes c10/cuda) -> unbuildable here.
set -euo pipefail
REF=${REFERENCE_ROOT:-/root/reference}
SRC=$REF/projects/BEVFusion/bevfusion/ops/voxel/src
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$SRC" ]; then
  echo "[build_ref] $SRC not present (GPU box?) - skipping, using prebuilt files if any" >&2
  exit 0
fi
mkdir -p "$OUT"
PY=${PYTHON:-python3}
TORCH_DIR=$($PY -c 'import torch,os;print(os.path.dirname(torch.__file__))')
PYINC=$($PY -c 'import sysconfig;print(sysconfig.get_paths()["include"])')
EXT=$($PY -c 'import sysconfig;print(sysconfig.get_config_var("EXT_SUFFIX"))')
TARGET=$OUT/voxel_layer_ref$EXT
if [ -f "$TARGET" ] && [ "$TARGET" -nt "$SRC/voxelization_cpu.cpp" ]; then
  echo "[build_ref] up to date: $TARGET"; exit 0
fi
g++ -O2 -std=c++17 -fPIC -shared -w \
  -DTORCH_EXTENSION_NAME=voxel_layer_ref -DTORCH_API_INCLUDE_EXTENSION_H \
  -D_GLIBCXX_USE_CXX11_ABI=1 \
  -I"$TORCH_DIR/include" -I"$TORCH_DIR/include/torch/csrc/api/include" -I"$PYINC" -I"$SRC" \
  "$SRC/voxelization.cpp" "$SRC/voxelization_cpu.cpp" \
  -L"$TORCH_DIR/lib" -Wl,-rpath,"$TORCH_DIR/lib" -ltorch -ltorch_cpu -lc10 -ltorch_python \
  -o "$TARGET"
echo "[build_ref] built $TARGET"
