#!/bin/bash
# needs the diagnostic build: run tools/build_diag.sh first; the switches below do nothing in the shipped library
export OPUSFE_LIB=${OPUSFE_LIB:-$(cd "$(dirname "$0")/.." && pwd)/opus_amd/_build/libopusfe_diag.so}
# lowering kernel times for several workgroup shapes (threads for the S,M,L size classes)
BYTES=${1:-256e6}
for sh in 64,128,256 64,256,512 64,128,512 128,256,512 64,64,128 64,64,256; do
  OPUSFE_PAR_SHAPE=$sh timeout -k 10 200 python bench.py --steps 3 --warmup 1 --bytes $BYTES --no-cpu > gpurun_out/shape.json 2> gpurun_out/shape.err || { echo "shape $sh failed"; tail -3 gpurun_out/shape.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/shape.json"))
k={x["name"]:round(x["ms_per_step"],2) for x in d["kernels"] if x["name"].startswith("lower_")}
print("shape $sh:", k, "par sum", round(sum(v for n,v in k.items() if n.startswith("lower_par")),2), "step", d["ms_per_step"])
PY
done
