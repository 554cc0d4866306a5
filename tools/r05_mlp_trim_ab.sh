#!/bin/bash
# Round 5: instruction trims of mlp_bwd_split_f16_kernel (|z| as a source modifier, cdf select as v_bfi, branch-free per-sample
# factors, packed elementwise products) against the build without the two GELU changes; bench-step time of the MLP backward.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
for v in default "$@"; do
  if [ $v = default ]; then E=""; else E="PSDF_LIB_PATH=$R/permuto_sdf_amd/lib/variants/libpsdf_$v.so"; fi
  for rep in 1 2; do
    env $E python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d['kernel_ms'])"
  done
done 2>&1 | tee -a $O/mlp_trim_ab.txt
