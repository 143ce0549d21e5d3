#!/bin/bash
# ucn_field_mlp alone (tools/mlp_bench.py): two 4-wave workgroups per CU (default) / one 8-wave workgroup (UCN_MLP_WAVES=8), inline-asm operand split (in-tree) / compiler-visible split (tools/_ab/mlp_split_builtin)
cd $GRAFT_REPO_ROOT
for lib in "" tools/_ab/mlp_split_builtin/lib.so; do for w in 4 8; do
  echo "== lib=${lib:-in-tree} UCN_MLP_WAVES=$w"; UCN_MLP_WAVES=$w python tools/mlp_bench.py ${lib:+--lib $lib} --rays-fastest 2>&1 | tail -2
done; done
