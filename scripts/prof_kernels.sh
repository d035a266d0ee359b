# usage (GPU box, repo root): bash scripts/prof_kernels.sh <tag> [step_only args]  -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 kernel trace of the timed step alone)
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace -d gpurun_out/prof_${TAG} -- python3 scripts/step_only.py "$@" > gpurun_out/${TAG}_step.log 2>&1
cat gpurun_out/${TAG}_step.log | tail -2
DB=$(ls gpurun_out/prof_${TAG}/*/*results.db | head -1)
python3 scripts/kstats.py $DB gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}
