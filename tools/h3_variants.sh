#!/bin/bash
# times tools/h3_alias_probe.py's first rows with each experiment build of gemm_h3 (tools/_ab/h3_*/lib.so, UCN_LIB_PATH)
cd $GRAFT_REPO_ROOT
echo "== base"; python tools/h3_alias_probe.py 983040 2>&1 | grep "^M" | grep -v "ldx 2[6-9][0-9]\|ldx 3\|ldy 2[6-9][0-9]\|ldx 7\|ldx 9\|ldx 1[0-9][0-9]\|ldy 12"
for v in tools/_ab/h3_*; do
  echo "== $v"; UCN_LIB_PATH=$v/lib.so python tools/h3_alias_probe.py 983040 2>&1 | grep "^M" | grep -v "ldx 2[6-9][0-9]\|ldx 3\|ldy 2[6-9][0-9]\|ldx 7\|ldx 9\|ldx 1[0-9][0-9]\|ldy 12"
done
