cd $GRAFT_REPO_ROOT
PSDF_FUSE_REFERENCE_MLPS=1 timeout 1000 python -m cProfile -o /tmp/ref.prof tools/run_reference_on_gpu.py --sphere-iters 50 --train-iters 400 --out /tmp/ref.json > /tmp/ref.log 2>&1
python - <<'PY'
import pstats, io
p = pstats.Stats('/tmp/ref.prof')
s = io.StringIO()
ps = pstats.Stats('/tmp/ref.prof', stream=s)
ps.sort_stats('cumulative').print_stats(140)
out = s.getvalue().splitlines()
keep = [l for l in out if ('_refcopy' in l or 'permuto_sdf_amd' in l or 'compat/' in l or 'run_backward' in l or 'optim' in l) ]
print("\n".join(keep[:90]))
PY
