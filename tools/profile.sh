#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats of bench.py plus two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "Counter limits").
# usage: tools/profile.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/pmc_fetch.log 2>&1 || exit 2
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/pmc_write.log 2>&1 || exit 3
find $OUT -name "*.csv" | head -20
