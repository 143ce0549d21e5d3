#!/bin/bash
cd $GRAFT_REPO_ROOT
F='ldx 2[6-9][0-9]\|ldx 3\|ldy 2[6-9][0-9]\|ldx 7\|ldx 9\|ldx 1[0-9][0-9]\|ldy 12\|ldy 4\b'
for s in 0 2 4 6 8 12; do echo "== UCN_H3_STAGGER=$s"; UCN_H3_STAGGER=$s python tools/h3_alias_probe.py 983040 2>&1 | grep "^M" | grep -v "$F"; done
echo "== default"; python tools/h3_alias_probe.py 983040 2>&1 | grep "^M" | grep -v "$F"
python tools/gemm_f32_bench.py 2>&1 | grep "^gemm"
