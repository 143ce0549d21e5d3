#!/bin/bash
# Builds the `_gridencoder` torch extension (pybind11, host code only) in-tree against libucnerf_march.so:
#   ucnerf_amd/compat/native/_gridencoder<EXT_SUFFIX>     (rpath -> ../../csrc, so the snapshot that travels to the GPU box is self-contained)
set -euo pipefail
cd "$(dirname "$0")"
PY=${PYTHON:-python3}
read -r TORCH_DIR PY_INC EXT_SUFFIX CXX11 <<<"$($PY - <<'PYEOF'
import os, sysconfig, torch
print(os.path.dirname(torch.__file__), sysconfig.get_paths()['include'], sysconfig.get_config_var('EXT_SUFFIX'), int(torch._C._GLIBCXX_USE_CXX11_ABI))
PYEOF
)"
OUT=../../compat/native
mkdir -p $OUT
TARGET=$OUT/_gridencoder$EXT_SUFFIX
if [ -f "$TARGET" ] && [ "$TARGET" -nt gridencoder_bindings.cpp ] && [ "$TARGET" -nt ../../../include/ucnerf_march.h ]; then
  echo "up to date: $TARGET"; exit 0
fi
g++ -O2 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -D__HIP_PLATFORM_AMD__ -DUSE_ROCM -DTORCH_EXTENSION_NAME=_gridencoder -D_GLIBCXX_USE_CXX11_ABI=$CXX11 \
    -I../../../include -I"$TORCH_DIR/include" -I"$TORCH_DIR/include/torch/csrc/api/include" -I"$PY_INC" -I/opt/rocm/include \
    gridencoder_bindings.cpp -o "$TARGET" \
    -L.. -lucnerf_march -L"$TORCH_DIR/lib" -lc10 -lc10_hip -ltorch_cpu -ltorch -ltorch_python \
    -Wl,-rpath,'$ORIGIN/../../csrc' -Wl,-rpath,"$TORCH_DIR/lib"
echo "built $(cd $OUT && pwd)/_gridencoder$EXT_SUFFIX"
