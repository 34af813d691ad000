#!/bin/bash
# usage: tools/variants.sh <bytes> lib1.so lib2.so ...   (GPU box) -> one line per library variant: whole-path MB/s and the two machine kernels
BYTES=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for lib in "$@"; do
  OPUSFE_LIB=$lib timeout -k 10 200 python3 $ROOT/bench.py --bytes $BYTES --steps 2 --warmup 1 --no-cpu > /tmp/v.json 2> /tmp/v.err || { echo "$lib FAILED"; tail -3 /tmp/v.err; exit 1; }
  python3 - "$lib" <<'PY'
import json, sys
d = json.loads(open('/tmp/v.json').read().strip().splitlines()[-1])
k = {x['name']: x['ms_per_step'] for x in d['kernels']}
print(f"{sys.argv[1]:48s} {d['value']:8.1f} MB/s  step {d['ms_per_step']:7.1f} ms  lower_par {k.get('lower_par', 0):7.1f}  parse {k.get('parse_segments', 0):7.1f}  lex_emit {k.get('lex_emit', 0):6.2f}  lex_count {k.get('lex_count', 0):6.2f}")
PY
done
