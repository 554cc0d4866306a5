#!/bin/bash
# Round 3: A/B of encode-backward build variants on the bench step (event bracket of the encode backward pair, whole step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
run() { PSDF_LIB_PATH=$2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1', 'ms/step %.4f' % d['ms_per_step'], d['kernel_ms'])"; }
run default $R/permuto_sdf_amd/lib/libpsdf_hip.so
for v in "$@"; do run $v $R/permuto_sdf_amd/lib/variants/libpsdf_$v.so; done
run default-again $R/permuto_sdf_amd/lib/libpsdf_hip.so
