# usage (GPU box, repo root): bash scripts/sub_scan.sh  -> stage times of the contracted route over the launch geometry of stage 1 (sub-grid on)
cd ${GRAFT_REPO_ROOT:-.}
for us in 2 3 4; do
  for wpb in 4 6 8 12 16; do
    JOXSZ_MIX_USPLIT=$us JOXSZ_MIX_WPB=$wpb timeout -k 10 120 python scripts/mix_time.py mix 512 500 1024 || exit 1
  done
done
