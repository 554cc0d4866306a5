#!/bin/bash
# bash tools/r05_env_ab.sh VAR v1 v2 ...: the bench step under VAR=v (two runs each; "-" = unset), kernel brackets printed.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/r05; mkdir -p $O
var=$1; shift
for v in "$@"; do
  for rep in 1 2; do
    if [ "$v" = "-" ]; then E=""; else E="$var=$v"; fi
    env $E python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'],4), d['kernel_ms'])"
  done
done 2>&1 | tee -a $O/env_ab.txt
