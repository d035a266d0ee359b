#!/bin/bash
# usage (GPU box, repo root): bash scripts/build_variants.sh "<flags A>" "<flags B>" ...   -- rebuilds the library with each set
# of extra compiler flags (EXTRA=...), runs the headline bench without the side measurements, prints the stage times; ends with the default build
cd "$(dirname "$0")/.."
for ex in "$@" ""; do
  touch joxsz_amd/csrc/joxsz_hip.hip
  make -C joxsz_amd/csrc libjoxsz_hip.so EXTRA="$ex" > /dev/null 2>&1 || { echo "build failed: $ex"; continue; }
  timeout -k 10 120 python bench.py --no-cpu --no-f32 --no-full-map --steps 20 --warmup 5 ${BENCH_ARGS} > gpurun_out/var.json 2>gpurun_out/var.err || true
  python - "$ex" <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/var.json").read().strip().splitlines()[-1]); s=d["stage_ms_per_step"]
    print("EXTRA=%r: step %.4f ms | prep %.3f abel %.3f pass1 %.3f gemm %.3f pass3 %.3f tail %.3f" % (sys.argv[1], d["ms_per_step"], s["prep_ms"], s["abel_map_ms"], s["beam_fft_ms"], s["gemm_ms"], s["tf_fft_ms"]-s["gemm_ms"], s["tail_ms"]), flush=True)
except Exception as e:
    print("EXTRA=%r failed: %s" % (sys.argv[1], e)); print(open("gpurun_out/var.err").read()[-400:])
PY
done
