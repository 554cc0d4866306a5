#!/bin/bash
# Round-2 evidence run (on the GPU box, from the repo root): bench line, rocprofv3 kernel trace of the same command, HBM
# traffic (FETCH_SIZE / WRITE_SIZE passes), SQ counters of the dominant kernel, cfg-2 matrix, cfg 3 / cfg 5 benches.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
python tools/rocpd_summary.py $O/prof_bench $O/bench_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_bench
# SHORT=1: only what an encode / MLP kernel change moves (bench line, its kernel trace, cfg 2 matrix, cfg 3, cfg 4)
if [ -z "$SHORT" ]; then
bash tools/pmc_hbm_traffic.sh r02 > $O/pmc_hbm.log 2>&1
bash tools/pmc_sq.sh mlp_bwd_split_kernel mlpbwdsplit -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_mlp_bwd_split.log 2>&1
fi
python tools/cfg2_matrix.py > $O/cfg2_matrix.jsonl 2> $O/cfg2_matrix.err
python tools/cfg3_render.py > $O/cfg3.json 2> $O/cfg3.err
python tools/train_bench.py > $O/cfg4_final.json 2> $O/cfg4_final.err
if [ -n "$SHORT" ]; then tail -c 600 $O/bench_final.json; echo; head -12 $O/bench_kernel_stats.txt | cut -c1-170; tail -3 $O/cfg2_matrix.jsonl | cut -c1-300; cat $O/cfg3.json $O/cfg4_final.json | cut -c1-400; exit 0; fi
python tools/sphere_trace_bench.py > $O/cfg5.json 2> $O/cfg5.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_cfg4 -- python $R/tools/train_bench.py > $O/cfg4_under_rocprof.json 2> $O/cfg4_under_rocprof.err
cd $R
python tools/rocpd_summary.py $O/prof_cfg4 $O/cfg4_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_cfg4
python tools/dda_march_bench.py > $O/dda_march_bench.json 2> $O/dda_march_bench.err
python tools/trace_train_launches.py > $O/train_launch_sources.txt 2>&1
rm -rf $R/gpurun_out/pmc_hbm_r02/FETCH_SIZE $R/gpurun_out/pmc_hbm_r02/WRITE_SIZE $R/gpurun_out/pmc_sq_mlpbwdsplit/pass*
tail -c 600 $O/bench_final.json; echo; head -12 $O/bench_kernel_stats.txt | cut -c1-170; cat $R/gpurun_out/pmc_hbm_r02.json | head -60; cat $R/gpurun_out/pmc_sq_mlpbwdsplit/summary.txt; tail -3 $O/cfg2_matrix.jsonl | cut -c1-300; cat $O/cfg3.json $O/cfg5.json $O/cfg4_final.json | cut -c1-400
