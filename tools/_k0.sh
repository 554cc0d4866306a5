O=gpurun_out/r06; mkdir -p $O
R=$GRAFT_REPO_ROOT
bash tools/kstats.sh $O/kstats_cfg4_iter0.txt -- python $R/tools/train_bench.py --manual --start-iter 0 --repeats 1
head -24 $O/kstats_cfg4_iter0.txt | cut -c1-90,100-180
