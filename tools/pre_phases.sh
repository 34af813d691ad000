#!/bin/bash
# Cumulative cost of the phases of k_flat_pre (lower_par.h): runs bench.py on the -DOPUSFE_DIAG build (tools/build_diag.sh) with
# OPUSFE_PRE_STOP=n, which makes the kernel leave after phase n (every function is then declined: only the kernel's own time means anything).
# usage (GPU box): tools/pre_phases.sh [bytes]
BYTES=${1:-256e6}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for p in 1 2 3 4 5 6 7 8 0; do
  OPUSFE_PRE_STOP=$p OPUSFE_LIB=$ROOT/opus_amd/_build/libopusfe_diag.so timeout -k 10 300 python $ROOT/bench.py --steps 2 --warmup 1 --bytes $BYTES --no-cpu > $ROOT/gpurun_out/pre_phase_$p.json 2> $ROOT/gpurun_out/pre_phase_$p.err || { echo "phase $p failed"; tail -3 $ROOT/gpurun_out/pre_phase_$p.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$ROOT/gpurun_out/pre_phase_$p.json").read().strip().split("\n")[-1])
k={x["name"]:round(x["ms_per_step"],2) for x in d["kernels"]}
print("stop after phase $p: lower_pre", k.get("lower_pre"))
PY
done
