cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
JOXSZ_DEBUG_TKS=1 python scripts/mix_time.py mix 512 500 1024 2>&1 | grep "radial sub-grid" | head -1
for v in 0 64,256,18; do
  rm -rf gpurun_out/prof_ag
  JOXSZ_AG_SUBSAMPLE=$v rocprofv3 --kernel-trace -d gpurun_out/prof_ag -- python3 scripts/mix_time.py mix 512 500 1024 > /dev/null 2>&1
  python scripts/kstats.py $(ls gpurun_out/prof_ag/*/*results.db | head -1) gpurun_out/ag_$v.csv > /dev/null 2>&1
  echo "AG_SUBSAMPLE=$v"; python -c "
import csv
for r in list(csv.DictReader(open('gpurun_out/ag_$v.csv')))[:6]: print('  ', r['Name'][:44], r['FullSizeCalls'], r['AverageUs'])"
done
rm -rf gpurun_out/prof_ag
