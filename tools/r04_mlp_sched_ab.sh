#!/bin/bash
# Round 4 (second half): A/B of scheduling variants of mlp_bwd_split_f16_kernel (weight records requested ahead of the operand
# split, two GELU pairs side by side, bias / final-weight reads ahead of the activation blocks).  Variant libraries are built on
# the CPU side into permuto_sdf_amd/lib/variants/ (git-ignored) and selected with PSDF_LIB_PATH.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/r04; mkdir -p $O
for v in default $@; do
  if [ $v = default ]; then E=""; else E="PSDF_LIB_PATH=$R/permuto_sdf_amd/lib/variants/libpsdf_$v.so"; fi
  for d in 36-64-64-64-1 52-64-64-64-1; do env $E PSDF_MLP_BWD_SPLIT=f16 python tools/mlp_bwd_bench.py $d 2>&1 | grep "mlp_bwd" | head -1 | sed "s/^/$v: /"; done
  env $E python -m pytest tests/test_gpu_mlp.py -q -m gpu -k "split_f16_backward" 2>&1 | tail -1 | sed "s/^/$v: /"
done 2>&1 | tee $O/mlp_sched_ab.txt
