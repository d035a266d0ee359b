# usage (GPU box): bash scripts/abl_stage1.sh -> stage-1 time of the default build, and of the diagnostic build with (1) no knot requests in the loop,
# (2) the scalar streams standing still (every scalar load hits), (3) both   [results of the diagnostic settings are wrong by construction]
cd ${GRAFT_REPO_ROOT:-.}
python scripts/mix_time.py mix 512 500 1024 | grep -o "stages (ms).*"
for d in 0 1 2 3; do
  JOXSZ_LIB=$PWD/joxsz_amd/csrc/libjoxsz_hip_abl.so JOXSZ_TRUNC_PROBE=0 JOXSZ_MIX_DBG=$d python scripts/mix_time.py mix 512 500 1024 | grep -o "stages (ms).*"
done
