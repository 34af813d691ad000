#!/bin/bash
# needs the diagnostic build: run tools/build_diag.sh first; the switches below do nothing in the shipped library
export OPUSFE_LIB=${OPUSFE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/opus_amd/_build/libopusfe_diag.so}
# Cumulative cost of the phases of the node-parallel lowering (lower_par.h): runs bench.py with OPUSFE_PAR_STOP=n
# (the kernel leaves after phase n; everything then falls back to the sequential machine) and prints the kernel times.
BYTES=${1:-256e6}
for p in 1 2 3 4 5 6 7 8 0; do
  OPUSFE_PAR_STOP=$p timeout -k 10 200 python bench.py --steps 2 --warmup 1 --bytes $BYTES --no-cpu > gpurun_out/phase_$p.json 2> gpurun_out/phase_$p.err || { echo "phase $p failed"; tail -3 gpurun_out/phase_$p.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/phase_$p.json"))
k={x["name"]:round(x["ms_per_step"],2) for x in d["kernels"] if x["name"].startswith("lower_par")}
print("stop after phase $p:", k, "sum", round(sum(k.values()),2))
PY
done
