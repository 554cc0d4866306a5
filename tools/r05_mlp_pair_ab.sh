#!/bin/bash
# Round 5: wave-pair form of the split-fp16 MLP backward (mlp_bwd_split_f16_pair_kernel) against the one-wave form:
# the float64 tests under either form, then the kernel time on the BASELINE batch.  Output: gpurun_out/r05/mlp_pair_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
{
for form in ${FORMS:-one pair cd}; do
  echo "== form $form"
  PSDF_MLP_BWD_F16_FORM=$form timeout 600 python -m pytest tests/test_gpu_mlp.py -q -m gpu -k "split_f16_backward" -x 2>&1 | tail -5
  for d in 36-64-64-64-1 20-64-64-64-1; do
    PSDF_MLP_BWD_F16_FORM=$form PSDF_MLP_BWD_SPLIT=f16 timeout 300 python tools/mlp_bwd_bench.py $d 2>&1 | grep "mlp_bwd"
  done
done
} 2>&1 | tee $O/mlp_pair_ab.txt
