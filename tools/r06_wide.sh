#!/bin/bash
O=gpurun_out/r06_wide; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mlp.py -x -q -m gpu -k "wide_net_backward" 2>&1 | tail -2
for f in ${FORMS:-f16}; do
  PSDF_MLP_WIDE_SPLIT=$f bash tools/kstats.sh $O/kstats_train_$f.txt -- python $R/tools/train_bench.py --manual --start-iter 20000 --repeats 1
  echo "== $f"; grep -i "wide_bwd" $O/kstats_train_$f.txt | cut -c1-70,100-180
done
