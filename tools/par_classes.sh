#!/bin/bash
# per-class times of the node-parallel lowering for the values of one environment switch: tools/par_classes.sh VAR "v1 v2" [bytes]
VAR=$1; VALS=$2; BYTES=${3:-256e6}
for v in $VALS; do
  env OPUSFE_BENCH_RAW_KERNELS=1 $VAR=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --bytes $BYTES --no-cpu > gpurun_out/cls.json 2> gpurun_out/cls.err || { echo "$VAR=$v failed"; tail -3 gpurun_out/cls.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/cls.json"))
k={x["name"]:round(x["ms_per_step"],2) for x in d["kernels"] if x["name"].startswith("lower_par")}
print("$VAR=$v:", dict(sorted(k.items(), key=lambda kv: int(kv[0].split("_")[-1]) if kv[0].split("_")[-1].isdigit() else 0)), "sum", round(sum(k.values()),2))
PY
done
