#!/bin/bash
# A/B of one environment switch of the node-parallel lowering: tools/par_ab.sh VAR "v1 v2 ..." [bytes]
VAR=$1; VALS=$2; BYTES=${3:-256e6}
for v in $VALS; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --bytes $BYTES --no-cpu > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$VAR=$v failed"; tail -3 gpurun_out/ab.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab.json"))
k={x["name"]:round(x["ms_per_step"],2) for x in d["kernels"]}
print("$VAR=$v: step", d["ms_per_step"], "lower_par", k.get("lower_par"), "parse", k.get("parse_segments"), "lowering_functions", d.get("lowering_functions"))
PY
done
