#!/bin/bash
# SQ counters of the wave-pair MLP backward (and of the one-wave form beside it) on the BASELINE batch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; export GRAFT_REPO_ROOT=$R
TAG=${1:-v1}
PSDF_MLP_BWD_F16_FORM=pair PSDF_MLP_BWD_SPLIT=f16 bash tools/pmc_sq.sh pair_kernel r05_mlp_pair_$TAG -- python $R/tools/mlp_bwd_bench.py 36-64-64-64-1
