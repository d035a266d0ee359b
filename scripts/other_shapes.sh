# usage (on the GPU box, from the repo root): bash scripts/other_shapes.sh [tag]   -> gpurun_out/<tag>_other_shapes.log
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-r02}
{
echo "# other BASELINE shapes, same library, one MI355X (bench.py --no-cpu --steps 20 --warmup 3)"
for args in "--S 256 --N 300 --walkers 256 --sz-only" "--S 256 --N 300 --walkers 1024 --sz-only" "--S 512 --N 500 --walkers 1024" "--S 512 --N 500 --walkers 4096" "--S 1024 --N 1000 --walkers 1024" "--S 1024 --N 1000 --walkers 1024 --dtype f32" "--S 512 --N 500 --walkers 1024 --dtype f32" "--S 171 --N 313 --walkers 1024" "--S 513 --N 500 --walkers 1024" "--S 257 --N 300 --walkers 1024"; do
  echo "## bench.py $args"
  python bench.py --no-cpu --no-full-map --no-f32 --steps 20 --warmup 3 $args | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({k:j[k] for k in ('metric','value','ms_per_step','stage_ms_per_step')})); print('conv_layout', j['config'].get('conv_layout')); o=j.get('operator_route') or {}; print('operator route: %.0f/s, %.4f ms per step, max rel diff vs map route %s' % (o.get('value',0), o.get('ms_per_step',0), o.get('max_rel_diff_vs_map_route')))"
done
echo "## context construction at 512^2/500"
python - <<'PY'
import time, sys
sys.path.insert(0, '.')
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
pb = datasets.synthetic_problem(S=512, N=500, seed=0)
t = time.time(); post = JoxszPosterior(pb, device=0); print('JoxszPosterior(...) took %.2f s' % (time.time() - t)); post.close()
t = time.time(); post = JoxszPosterior(pb, device=0); print('second context   took %.2f s' % (time.time() - t)); post.close()
PY
} > gpurun_out/${TAG}_other_shapes.log 2>&1
cat gpurun_out/${TAG}_other_shapes.log
