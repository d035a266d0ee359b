# timing ablations of stage 1 on the matrix cores (diagnostic build; results are wrong): bash scripts/abl_mfma.sh "<dbg values>"
export JOXSZ_LIB=$PWD/joxsz_amd/csrc/libjoxsz_hip_abl.so JOXSZ_TRUNC_PROBE=0
for d in ${1:-0 1 4 8 16 5 12 13 28 29}; do JOXSZ_MIX_DBG=$d python scripts/mix_time.py mix 512 500 ${W:-1024} 2>&1 | sed -e "s/.*stages (ms)//" | cut -c1-160; done
