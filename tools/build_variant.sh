#!/bin/bash
# Experiment build of ONE csrc file with an extra -D flag, linked against the in-tree objects of the others:
#   tools/build_variant.sh <file-without-.hip> <name> <-DFLAG...>  ->  tools/${UCN_EXP_DIR:-_exp}/<name>/lib.so   (UCN_TOOL_LIB for the tools)
set -euo pipefail
cd "$(dirname "$0")/../ucnerf_amd/csrc"
f=$1; name=$2; shift 2
mkdir -p ../../tools/${UCN_EXP_DIR:-_exp}/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -w "$@" -c $f.hip -o /tmp/${f}_$name.o \
    -Rpass-analysis=kernel-resource-usage 2> /tmp/${f}_$name.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/${UCN_EXP_DIR:-_exp}/$name/lib.so /tmp/${f}_$name.o $(ls _obj/*.o | grep -v "_obj/$f.o")
echo "built tools/${UCN_EXP_DIR:-_exp}/$name/lib.so"
