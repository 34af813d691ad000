#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" [bench args...]   -> gpurun_out/pmc_<tag>/   (one rocprofv3 --pmc pass, GPU box)
TAG=$1; CTRS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o p -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/run.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as o:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:40]:
        o.write(k + "\n")
        for c, v in sorted(d.items()): o.write(f"    {c:28s} {v:.4g}\n")
print(open(out + "/summary.txt").read())
PY
