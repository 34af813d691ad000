for lib in libopusfe.so libopusfe_w3.so libopusfe_w5.so libopusfe_w6.so; do
  OPUSFE_LIB=$PWD/opus_amd/_build/$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --bytes 256e6 --no-cpu > gpurun_out/wv.json 2> gpurun_out/wv.err || { echo "$lib failed"; tail -2 gpurun_out/wv.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/wv.json"))
k={x["name"]:round(x["ms_per_step"],2) for x in d["kernels"] if x["name"].startswith("lower_par")}
print("$lib: par sum", round(sum(k.values()),2), "step", d["ms_per_step"], k)
PY
done
