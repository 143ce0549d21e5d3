#!/bin/bash
# Experiment builds of libucnerf_march.so for kernel diagnosis (NOT the product): copies csrc/ to
# tools/_exp/<name>/, applies a sed patch to the split-f16 engine (mfma_chain_h.h) and builds there.
# Run with tools/mlp_bench.py --lib tools/_exp/<name>/ucnerf_amd/csrc/libucnerf_march.so (numerics are
# wrong by construction in nodma / noread; they only answer "what does this part of the stream cost").
#   nodma   : no weight DMA after the first chunks (stale LDS contents)
#   noread  : A operands are read from LDS only for the first 16 groups (ring keeps stale values)
#   depth4 / depth8 : ring depth of the A-operand prefetch (default 6)
#   base    : unpatched copy
set -euo pipefail
cd "$(dirname "$0")/.."
for name in "$@"; do
  dst=tools/_exp/$name
  rm -rf "$dst"; mkdir -p "$dst/ucnerf_amd/csrc" "$dst/include"
  cp ucnerf_amd/csrc/*.hip ucnerf_amd/csrc/*.h ucnerf_amd/csrc/build.sh "$dst/ucnerf_amd/csrc/"
  cp include/*.h "$dst/include/"
  h=$dst/ucnerf_amd/csrc/mfma_chain_h.h
  case $name in
    nodma)  sed -i 's|    if constexpr (G % 4 == 0 \&\& |    if constexpr (false \&\& G % 4 == 0 \&\& |' "$h" ;;
    noread) sed -i 's|    p.hi\[(G / 2) % kDepth\] = __builtin_bit_cast(h8, ws.group(G));|    if constexpr (G < 16) p.hi[(G / 2) % kDepth] = __builtin_bit_cast(h8, ws.group(G));|; s|    p.lo\[(G / 2) % kDepth\] = __builtin_bit_cast(h8, ws.group(G + 1));|    if constexpr (G < 16) p.lo[(G / 2) % kDepth] = __builtin_bit_cast(h8, ws.group(G + 1));|' "$h" ;;
    depth4) sed -i 's|constexpr int kDepth = 6;|constexpr int kDepth = 4;|' "$h" ;;
    depth8) sed -i 's|constexpr int kDepth = 6;|constexpr int kDepth = 8;|' "$h" ;;
    base)   ;;
    *) echo "unknown experiment $name"; exit 1 ;;
  esac
  bash "$dst/ucnerf_amd/csrc/build.sh" | tail -1
done
