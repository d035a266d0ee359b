# Everything profiles/<tag>_* is made from, in one pass on the GPU box (repo root):   bash scripts/collect_profiles.sh r05
# Writes under gpurun_out/; copy what is to be kept into profiles/.
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-r05}
export TMPDIR=/tmp
set -e
# 1. HBM bytes per launch of every kernel (two PMC passes), the full-map kernel included -- first, so that the bench lines below quote THIS build's traffic
python scripts/measure_traffic.py ${TAG} > gpurun_out/${TAG}_pmc_traffic.log 2>&1
rm -rf gpurun_out/traffic_${TAG}_fetch gpurun_out/traffic_${TAG}_write
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
# 1a. the same for the literal route (conv = rocfft) at 512^2 / 1024 walkers: HBM bytes of its kernels, SQ counters, kernel statistics, the three transform modes
python scripts/measure_traffic.py ${TAG} literal > gpurun_out/${TAG}_pmc_traffic_literal.log 2>&1
rm -rf gpurun_out/traffic_${TAG}_fetch gpurun_out/traffic_${TAG}_write
cp gpurun_out/${TAG}_pmc_traffic_literal.json profiles/${TAG}_pmc_traffic_literal.json
python scripts/sq_counters.py literal 512 500 1024 2 > gpurun_out/${TAG}_sq_counters_literal.log 2>&1
rm -rf gpurun_out/sq_pmc
bash scripts/prof_literal.sh ${TAG}_literal > /dev/null 2>&1
python scripts/literal_cols.py > gpurun_out/${TAG}_literal_transforms.log 2>&1
# 1b. where the wave cycles go (SQ counters, one pass) and the idle time between the kernels of a step under the profiler -- also ahead of the bench lines, which quote the shares
python scripts/sq_counters.py > gpurun_out/${TAG}_sq_counters.log 2>&1
mv gpurun_out/sq_counters.csv gpurun_out/${TAG}_sq_counters.csv
cp gpurun_out/${TAG}_sq_counters.csv profiles/${TAG}_sq_counters.csv
rm -rf gpurun_out/sq_pmc
python scripts/step_gaps.py > gpurun_out/${TAG}_step_gaps.log 2>&1
rm -rf gpurun_out/gaps
# 2. the bench line as the driver runs it (CPU leg included), then without the CPU leg + the kernel statistics of the same command (full-size launches only)
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash scripts/prof_step.sh ${TAG}_nocpu
mv gpurun_out/${TAG}_nocpu_bench.json gpurun_out/${TAG}_bench_nocpu.json
mv gpurun_out/${TAG}_nocpu_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
# 3. the north-star full-map kernel (S x S map of every walker written to HBM) under the profiler: kernel trace of a bench run that keeps it
rm -rf gpurun_out/prof_${TAG}_fm
rocprofv3 --kernel-trace -d gpurun_out/prof_${TAG}_fm -- python3 bench.py --no-cpu --no-f32 --no-other-configs --no-other-routes --no-host-pointer --steps 20 --warmup 3 --regions 3 > gpurun_out/${TAG}_fullmap_bench.json 2> /dev/null
python scripts/kstats.py $(ls gpurun_out/prof_${TAG}_fm/*/*results.db | head -1) gpurun_out/${TAG}_fullmap_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}_fm
# 5. fp64 FMA rate of the chip with uniform multipliers (the measured peak bench.py quotes)
timeout -k 10 120 scripts/ubench/fma_sgpr > gpurun_out/${TAG}_fma_sgpr.log 2>&1
# 6. the N > 1 plumbing rehearsed at N = 1 (RCCL communicator of one rank, gather, barrier, max over ranks, sharded sampler exchange): strict gather (the
#    headline mode) with the overlapped mode probed in the warm-up, then the overlapped mode pinned
JOXSZ_BENCH_FORCE_DIST=1 python bench.py --no-cpu --no-full-map --no-f32 --no-other-configs --no-other-routes > gpurun_out/${TAG}_bench_force_dist.json 2> gpurun_out/${TAG}_bench_force_dist.err
JOXSZ_BENCH_FORCE_DIST=1 JOXSZ_BENCH_OVERLAP_GATHER=1 python bench.py --no-cpu --no-full-map --no-f32 --no-other-configs --no-other-routes > gpurun_out/${TAG}_bench_force_dist_overlap.json 2> gpurun_out/${TAG}_bench_force_dist_overlap.err
# 7. the exact form against the rocFFT sequence and the oracle at every shape, the whole prior box, the arithmetic variants, the other shapes, the sampler
python scripts/exact_check.py > gpurun_out/${TAG}_exact_check.log 2>&1
python scripts/box_parity.py 512 500 400 > gpurun_out/${TAG}_box_parity.log 2>&1
python scripts/box_parity.py 256 300 400 >> gpurun_out/${TAG}_box_parity.log 2>&1
python scripts/dtype_sweep.py 512 500 1024 > gpurun_out/${TAG}_dtype_sweep.log 2>&1
python scripts/dtype_sweep.py 1024 1000 8192 >> gpurun_out/${TAG}_dtype_sweep.log 2>&1
python scripts/shapes.py > gpurun_out/${TAG}_shapes.log 2>&1
python scripts/sampler_rate.py > gpurun_out/${TAG}_sampler_rate.log 2>&1
# (no absolute paths of the build box in what gets committed)
sed -i "s#$PWD/##g; s#/tmp/[A-Za-z0-9_./-]*/repo/##g" gpurun_out/${TAG}_*.log
echo collected
