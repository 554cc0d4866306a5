#!/bin/bash
# timing of variant builds of the wave-pair MLP backward (permuto_sdf_amd/lib/variants/libpsdf_<v>.so): bash tools/r05_mlp_pair_variants.sh v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for v in default "$@"; do
  if [ $v = default ]; then E=""; else E="PSDF_LIB_PATH=$R/permuto_sdf_amd/lib/variants/libpsdf_$v.so"; fi
  for rep in 1 2; do
    env $E PSDF_MLP_BWD_F16_FORM=pair PSDF_MLP_BWD_SPLIT=f16 timeout 300 python tools/mlp_bwd_bench.py 36-64-64-64-1 2>&1 | grep "mlp_bwd" | head -1 | sed "s/^/pair $v: /"
  done
done 2>&1 | tee -a $O/mlp_pair_variants.txt
env PSDF_MLP_BWD_F16_FORM=one PSDF_MLP_BWD_SPLIT=f16 python tools/mlp_bwd_bench.py 36-64-64-64-1 2>&1 | grep "mlp_bwd" | head -1 | sed "s/^/one: /" | tee -a $O/mlp_pair_variants.txt
