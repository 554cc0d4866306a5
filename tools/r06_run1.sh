#!/bin/bash
# round 6, first GPU call: accuracy of the round-to-nearest split, the full-batch per-ray test, smoke, bench
O=gpurun_out/r06_run1; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_fullbatch_radiance.py -x -q -m gpu -s > $O/fullbatch.txt 2>&1; echo "fullbatch rc=$?"
timeout 900 python -m pytest tests/test_gpu_hotpath_parity.py -q -m gpu -s -k "dense_gradient" > $O/dense_f64.txt 2>&1; echo "dense rc=$?"
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_optim.py -q -m gpu -x > $O/mlp_optim.txt 2>&1; echo "mlp rc=$?"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/fullbatch.txt $O/dense_f64.txt $O/mlp_optim.txt $O/smoke.txt
