#!/bin/bash
# Round 3: A/B of the MLP backward variants on the GPU box (bf16 three-piece product kernel vs the fp16 two-piece kernel and its
# build variants), accuracy test of the fp16 kernel, bench line with it.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_mlp.py -q -m gpu -k "split_f16 or split_bf16_backward" -s > $O/mlp_f16_test.log 2>&1; echo "pytest rc=$?"
grep -h "f16 split backward\|rows with\|passed\|failed" $O/mlp_f16_test.log | cut -c1-260
for v in bf16 f16; do for d in 36-64-64-64-1 52-64-64-64-1; do PSDF_MLP_BWD_SPLIT=$v python tools/mlp_bwd_bench.py $d 2>&1 | grep "mlp_bwd" | head -1 | sed "s/^/$v default-build: /"; done; done
for lib in pkgelu pkgelu_ilp ilp; do for d in 36-64-64-64-1 52-64-64-64-1; do PSDF_LIB_PATH=$R/permuto_sdf_amd/lib/variants/libpsdf_$lib.so PSDF_MLP_BWD_SPLIT=f16 python tools/mlp_bwd_bench.py $d 2>&1 | grep "mlp_bwd" | head -1 | sed "s/^/f16 $lib: /"; done; done
PSDF_MLP_BWD_SPLIT=f16 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err; python - <<PY
import json
d=json.load(open("$O/bench_f16.json")); print("bench with f16 backward: ms/step", d["ms_per_step"], d["kernel_ms"], "L24", d["extra"])
PY
