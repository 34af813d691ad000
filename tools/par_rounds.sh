#!/bin/bash
# needs the diagnostic build: run tools/build_diag.sh first; the switches below do nothing in the shipped library
export OPUSFE_LIB=${OPUSFE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/opus_amd/_build/libopusfe_diag.so}
# time of the node-parallel lowering kernels when the type fixpoint is cut after n rounds (then stop before the scans)
BYTES=${1:-256e6}
for r in 1 2 4 8 12 16 24 32 48 64 0; do
  OPUSFE_PAR_STOP=6 OPUSFE_PAR_ROUNDS=$r timeout -k 10 200 python bench.py --steps 2 --warmup 1 --bytes $BYTES --no-cpu > gpurun_out/rounds.json 2> gpurun_out/rounds.err || { echo "rounds $r failed"; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/rounds.json"))
k={x["name"]:round(x["ms_per_step"],2) for x in d["kernels"] if x["name"].startswith("lower_par")}
print("rounds $r: sum", round(sum(k.values()),2), "768:", k.get("lower_par_768"), "192:", k.get("lower_par_192"))
PY
done
