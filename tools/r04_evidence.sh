#!/bin/bash
# Round-4 evidence run (on the GPU box, from the repo root): the bench line as the driver runs it (train it/s, cfg 3, cfg 5 ride
# along as `extra`), rocprofv3 kernel trace of the same command, HBM traffic (FETCH_SIZE / WRITE_SIZE passes), SQ counters of
# the dominant kernel, of the encode backward binning kernel and -- new -- of the position-gradient kernel, the cfg-2 matrix,
# the kernel trace of the hand-written training step.  SHORT=1: bench + kernel traces only.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
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
bash tools/pmc_hbm_traffic.sh r04 > $O/pmc_hbm.log 2>&1
bash tools/pmc_sq.sh mlp_bwd_split_f16_kernel mlpbwdf16 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_mlp_bwd_f16.log 2>&1
bash tools/pmc_sq.sh "encode_bwd_kernel" encbwd -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_sq_encode_bwd.log 2>&1
PSDF_AB_ONLY=2M_L16 bash tools/pmc_sq.sh "encode_bwd_pos_kernel" encbwdpos -- python $R/tools/r04_enc_ab.py > $O/pmc_sq_encode_bwd_pos.log 2>&1
PSDF_AB_ONLY=2M_L16 PMC_SETS="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" bash tools/pmc_sq.sh "encode_bwd_pos_kernel" encbwdpos_mem -- python $R/tools/r04_enc_ab.py > $O/pmc_mem_encode_bwd_pos.log 2>&1
python tools/cfg2_matrix.py > $O/cfg2_matrix.jsonl 2> $O/cfg2_matrix.err
rm -rf $R/gpurun_out/pmc_hbm_r04/FETCH_SIZE $R/gpurun_out/pmc_hbm_r04/WRITE_SIZE $R/gpurun_out/pmc_sq_mlpbwdf16/pass* $R/gpurun_out/pmc_sq_encbwd/pass* $R/gpurun_out/pmc_sq_encbwdpos/pass* $R/gpurun_out/pmc_sq_encbwdpos_mem/pass*
cat $R/gpurun_out/pmc_sq_encbwdpos/summary.txt $R/gpurun_out/pmc_sq_encbwdpos_mem/summary.txt
fi
tail -c 1500 $O/bench_final.json; echo; head -14 $O/bench_kernel_stats.txt | cut -c1-175; head -12 $O/cfg4_manual_kernel_stats.txt | cut -c1-175; tail -3 $O/cfg2_matrix.jsonl 2>/dev/null | cut -c1-300
