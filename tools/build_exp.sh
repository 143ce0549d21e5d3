#!/bin/bash
# Experiment builds of libucnerf_march.so for kernel diagnosis (NOT the product): copies csrc/ to
# tools/_exp/<name>/, applies a sed patch to the split-f16 engine and builds there.
#   nolo   : the lo half of every A pair is not read from LDS (halves LDS read traffic; wrong numerics)
#   onemf  : one MFMA per step instead of three, same LDS reads (wrong numerics)
#   nodma  : no weight DMA after chunk 0 (stale LDS contents; wrong numerics): what the DMA pieces cost the waves
#   nosb   : without the sched_barrier that pins the prefetch above the MFMAs
set -euo pipefail
cd "$(dirname "$0")/.."
for name in "$@"; do
  dst=tools/_exp/$name
  rm -rf "$dst"; mkdir -p "$dst/ucnerf_amd/csrc" "$dst/include"
  cp ucnerf_amd/csrc/*.hip ucnerf_amd/csrc/*.h ucnerf_amd/csrc/build.sh "$dst/ucnerf_amd/csrc/"
  cp include/*.h "$dst/include/"
  h=$dst/ucnerf_amd/csrc/mfma_chain_h.h
  case $name in
    nolo)  sed -i 's|    p.lo = group_h(ws, g + 1);|    p.lo = p.hi;|' "$h" ;;
    onemf) sed -i 's|    acc = mfma16h(p.hi, blo, acc);||; s|    acc = mfma16h(p.lo, bhi, acc);||; s|    acc = mfma16h(p.hi, bhi, acc);|    acc = mfma16h(p.hi + p.lo, bhi + blo, acc);|' "$h" ;;
    nosb)  sed -i 's|    __builtin_amdgcn_sched_barrier(0);.*||' "$h" ;;
    nodma) sed -i 's|    if (g % 4 == 0 \&\& |    if (false \&\& |' "$h" ;;
    noread) sed -i 's|    if (G0 + 2 \* d < GEND) pipe_fetch(G0 + 2 \* d, p, ws);|    { p.hi[d] = group_h(ws, G0 + 2 * d); p.lo[d] = group_h(ws, G0 + 2 * d + 1); }|; s|    p.hi\[(g / 2) % kDepth\] = group_h(ws, g);||; s|    p.lo\[(g / 2) % kDepth\] = group_h(ws, g + 1);||' "$h" ;;
    nodma_noread) sed -i 's|    if (g % 4 == 0 \&\& |    if (false \&\& |; s|    if (G0 + 2 \* d < GEND) pipe_fetch(G0 + 2 \* d, p, ws);|    { p.hi[d] = group_h(ws, G0 + 2 * d); p.lo[d] = group_h(ws, G0 + 2 * d + 1); }|; s|    p.hi\[(g / 2) % kDepth\] = group_h(ws, g);||; s|    p.lo\[(g / 2) % kDepth\] = group_h(ws, g + 1);||' "$h" ;;
    depth6) sed -i 's|constexpr int kDepth = 4;|constexpr int kDepth = 6;|' "$h" ;;
    depth8) sed -i 's|constexpr int kDepth = 4;|constexpr int kDepth = 8;|' "$h" ;;
    base)  ;;
    *) echo "unknown experiment $name"; exit 1 ;;
  esac
  bash "$dst/ucnerf_amd/csrc/build.sh" | tail -1
done
