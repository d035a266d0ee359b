# usage (GPU box, repo root): bash scripts/sub_scan.sh  -> stage times of the contracted route with and without the stage-1 sub-grid
cd ${GRAFT_REPO_ROOT:-.}
export JOXSZ_MIX_USPLIT=3 JOXSZ_MIX_SUBSAMPLE=40,160,12
for ks in 32 40 48 64; do
    JOXSZ_MIX_KSPLIT_USE=$ks timeout -k 10 120 python scripts/mix_time.py mix 512 500 1024 || exit 1
done
for ks in 32 64; do
    JOXSZ_MIX_KSPLIT_USE=$ks timeout -k 10 120 python scripts/mix_time.py mix 512 500 4096 || exit 1
    JOXSZ_MIX_KSPLIT_USE=$ks timeout -k 10 120 python scripts/mix_time.py mix 512 500 256 || exit 1
done
