#!/bin/bash
# SQ counters of the chain / dW wave-pair MLP backward on the BASELINE batch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; export GRAFT_REPO_ROOT=$R
TAG=${1:-v1}
PSDF_MLP_BWD_F16_FORM=cd PSDF_MLP_BWD_SPLIT=f16 bash tools/pmc_sq.sh cd_kernel r05_mlp_cd_$TAG -- python $R/tools/mlp_bwd_bench.py 36-64-64-64-1
