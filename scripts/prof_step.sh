# usage (on the GPU box, repo root): bash scripts/prof_step.sh <tag> [bench args]  -> gpurun_out/<tag>_bench.json, gpurun_out/<tag>_kernel_stats.csv
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; shift
export TMPDIR=/tmp
python bench.py --no-cpu --steps 100 --warmup 5 --regions 25 "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python -c "import json; j=json.load(open('gpurun_out/${TAG}_bench.json')); print('value %.0f  ms/step %.4f' % (j['value'], j['ms_per_step'])); print(j['stage_ms_per_step'])"
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace -d gpurun_out/prof_${TAG} -- python3 bench.py --no-cpu --no-full-map --no-f32 --no-other-configs --no-other-routes --no-host-pointer --steps 100 --warmup 5 --regions 5 "$@" > /dev/null 2>&1
DB=$(ls gpurun_out/prof_${TAG}/*/*results.db | head -1)
python scripts/kstats.py $DB gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}
