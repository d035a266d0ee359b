# usage (GPU box, repo root): bash scripts/prof_literal.sh <tag> [S N W reps]  -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 kernel trace of the literal route, conv = rocfft)
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace -d gpurun_out/prof_${TAG} -- python3 scripts/literal_prof.py "$@" > gpurun_out/${TAG}_run.log 2>&1
tail -2 gpurun_out/${TAG}_run.log
DB=$(ls gpurun_out/prof_${TAG}/*/*results.db | head -1)
test -n "$DB" && python3 scripts/kstats.py $DB gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}
