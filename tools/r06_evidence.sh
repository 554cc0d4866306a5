#!/bin/bash
# Round-6 evidence run (on the GPU box, from the repo root).  bench line as the driver runs it; rocprofv3 kernel traces of the bench
# step in each arithmetic; HBM traffic (FETCH_SIZE / WRITE_SIZE passes, both arithmetics, with the kernel-source hash); SQ counters of
# the 24-bit backward, the quad march and the split-fp16 wide backward; kernel trace of the hand-written training step; cfg-2 matrix;
# the training step through real RCCL on one rank.  SHORT=1: bench + kernel traces only.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
PSDF_BENCH_NO_24BIT=1 bash tools/kstats.sh $O/bench_kernel_stats.txt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra
PSDF_BENCH_NO_24BIT=1 PSDF_MLP_FWD_SPLIT=bf16 PSDF_MLP_BWD_SPLIT=bf16 bash tools/kstats.sh $O/bench_24bit_kernel_stats.txt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra
bash tools/kstats.sh $O/cfg4_manual_kernel_stats.txt -- python $R/tools/train_bench.py --manual --start-iter 20000 --repeats 1
if [ -z "$SHORT" ]; then
bash tools/pmc_hbm_traffic.sh r06 > $O/pmc_hbm.log 2>&1
PSDF_BENCH_NO_24BIT=1 PSDF_MLP_FWD_SPLIT=bf16 PSDF_MLP_BWD_SPLIT=bf16 bash tools/pmc_sq.sh "mlp_bwd_split_kernel" r06_mlpbwdbf16 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_mlp_bwd_bf16.log 2>&1
PSDF_BENCH_NO_24BIT=1 bash tools/pmc_sq.sh "mlp_bwd_split_f16_kernel" r06_mlpbwdf16 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_mlp_bwd_f16.log 2>&1
bash tools/pmc_sq.sh "march_quad_kernel" r06_marchquad -- python $R/tools/train_bench.py --manual --start-iter 20000 --repeats 1 --steps 10 --warmup 3 > $O/pmc_sq_march_quad.log 2>&1
bash tools/pmc_sq.sh "mlp_wide_bwd_f16_kernel<7" r06_widef16 -- python $R/tools/train_bench.py --manual --start-iter 20000 --repeats 1 --steps 10 --warmup 3 > $O/pmc_sq_wide_f16.log 2>&1
python tools/cfg2_matrix.py > $O/cfg2_matrix.jsonl 2> $O/cfg2_matrix.err
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 PSDF_DP_FORCE_COLLECTIVES=1 PSDF_DIST_BACKEND=nccl python tools/train_bench.py --manual --start-iter 20000 2> $O/cfg4_one_rank_rccl.err | grep "^{" > $O/cfg4_one_rank_rccl.json
rm -rf $R/gpurun_out/pmc_hbm_r06/FETCH_SIZE $R/gpurun_out/pmc_hbm_r06/WRITE_SIZE $R/gpurun_out/pmc_sq_r06_*/pass*
fi
tail -c 1500 $O/bench.json; echo; head -10 $O/bench_kernel_stats.txt | cut -c1-175; head -6 $O/bench_24bit_kernel_stats.txt | cut -c1-175
