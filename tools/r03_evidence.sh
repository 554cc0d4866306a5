#!/bin/bash
# Round-3 evidence run (on the GPU box, from the repo root): bench line, rocprofv3 kernel trace of the same command, HBM traffic
# (FETCH_SIZE / WRITE_SIZE passes), SQ counters of the dominant kernel and of the encode backward binning kernel, cfg-2 matrix,
# cfg 3 / cfg 4 / cfg 5 benches.  SHORT=1: bench + kernel trace + cfg 2/3/4 only.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python bench.py --steps 50 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
python tools/rocpd_summary.py $O/prof_bench $O/bench_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_bench
if [ -z "$SHORT" ]; then
bash tools/pmc_hbm_traffic.sh r03 > $O/pmc_hbm.log 2>&1
bash tools/pmc_sq.sh mlp_bwd_split_f16_kernel mlpbwdf16 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_mlp_bwd_f16.log 2>&1
bash tools/pmc_sq.sh "encode_bwd_kernel" encbwd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_encode_bwd.log 2>&1
fi
python tools/cfg2_matrix.py > $O/cfg2_matrix.jsonl 2> $O/cfg2_matrix.err
python tools/cfg3_render.py > $O/cfg3.json 2> $O/cfg3.err
python tools/train_bench.py > $O/cfg4_final.json 2> $O/cfg4_final.err
python tools/train_bench.py --manual > $O/cfg4_manual.json 2> $O/cfg4_manual.err
python tools/train_bench.py --manual --start-iter 20000 > $O/cfg4_manual_late.json 2> $O/cfg4_manual_late.err
python tools/train_bench.py --start-iter 20000 > $O/cfg4_late.json 2> $O/cfg4_late.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_manual -- python $R/tools/train_bench.py --manual --start-iter 20000 --repeats 1 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $O/prof_manual $O/cfg4_manual_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_manual
python tools/small_batch_bench.py > $O/small_batch_mlp.txt 2>&1
python tools/small_batch_enc_bench.py > $O/small_batch_enc.txt 2>&1
if [ -z "$SHORT" ]; then
python tools/sphere_trace_bench.py > $O/cfg5.json 2> $O/cfg5.err
rm -rf $R/gpurun_out/pmc_hbm_r03/FETCH_SIZE $R/gpurun_out/pmc_hbm_r03/WRITE_SIZE $R/gpurun_out/pmc_sq_mlpbwdf16/pass* $R/gpurun_out/pmc_sq_encbwd/pass*
cat $R/gpurun_out/pmc_hbm_r03.json | head -80; cat $R/gpurun_out/pmc_sq_mlpbwdf16/summary.txt $R/gpurun_out/pmc_sq_encbwd/summary.txt
fi
tail -c 700 $O/bench_final.json; echo; head -14 $O/bench_kernel_stats.txt | cut -c1-175; tail -3 $O/cfg2_matrix.jsonl | cut -c1-300; cat $O/cfg3.json $O/cfg4_final.json $O/cfg4_manual.json $O/cfg4_manual_late.json $O/cfg5.json 2>/dev/null | cut -c1-400; tail -4 $O/small_batch_mlp.txt $O/small_batch_enc.txt
