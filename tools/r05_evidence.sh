#!/bin/bash
# Round-5 evidence run (on the GPU box, from the repo root): the bench line as the driver runs it, rocprofv3 kernel trace of the
# same command, HBM traffic (FETCH_SIZE / WRITE_SIZE passes), SQ counters of the dominant kernel and of the encode backward binning
# kernel, kernel trace of the hand-written training step.  SHORT=1: bench + kernel traces only.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python bench.py > $O/bench_final.json 2> $O/bench_final.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
python tools/rocpd_summary.py $O/prof_bench $O/bench_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_bench
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_manual -- python $R/tools/train_bench.py --manual --start-iter 20000 --repeats 1 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $O/prof_manual $O/cfg4_manual_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_manual
if [ -z "$SHORT" ]; then
bash tools/pmc_hbm_traffic.sh r05 > $O/pmc_hbm.log 2>&1
bash tools/pmc_sq.sh mlp_bwd_split_f16_kernel r05_mlpbwdf16 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_mlp_bwd_f16.log 2>&1
bash tools/pmc_sq.sh "encode_bwd_kernel" r05_encbwd -- python $R/bench.py --steps 12 --warmup 10 --no-cpu-baseline --no-extra > $O/pmc_sq_encode_bwd.log 2>&1
bash tools/pmc_sq.sh "encode_fwd_kernel" r05_encfwd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_encode_fwd.log 2>&1
bash tools/pmc_sq.sh "mlp_fwd_split_kernel" r05_mlpfwd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_mlp_fwd.log 2>&1
python tools/cfg2_matrix.py > $O/cfg2_matrix.jsonl 2> $O/cfg2_matrix.err
rm -rf $R/gpurun_out/pmc_hbm_r05/FETCH_SIZE $R/gpurun_out/pmc_hbm_r05/WRITE_SIZE $R/gpurun_out/pmc_sq_r05_mlpbwdf16/pass* $R/gpurun_out/pmc_sq_r05_encbwd/pass* $R/gpurun_out/pmc_sq_r05_encfwd/pass* $R/gpurun_out/pmc_sq_r05_mlpfwd/pass*
fi
tail -c 1200 $O/bench_final.json; echo; head -12 $O/bench_kernel_stats.txt | cut -c1-175; head -8 $O/cfg4_manual_kernel_stats.txt | cut -c1-175
