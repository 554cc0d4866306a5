O=gpurun_out/r06; mkdir -p $O
R=$GRAFT_REPO_ROOT
PSDF_TRACE_WEIGHTS=sphere_init bash tools/kstats.sh $O/kstats_cfg5.txt -- python $R/tools/sphere_trace_bench.py
head -12 $O/kstats_cfg5.txt | cut -c1-90,100-180
bash tools/kstats.sh $O/kstats_cfg3.txt -- python $R/tools/cfg3_render.py
head -14 $O/kstats_cfg3.txt | cut -c1-90,100-180
