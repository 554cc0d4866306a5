cd $GRAFT_REPO_ROOT
PSDF_FUSE_REFERENCE_MLPS=1 timeout 1000 python -m cProfile -o /tmp/ref.prof tools/run_reference_on_gpu.py --sphere-iters 50 --train-iters 400 --out /tmp/ref.json > /tmp/ref.log 2>&1
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/ref.prof')
p.sort_stats('tottime').print_stats(45)
PY
