#!/bin/bash
# configuration 5 with different numbers of worker contexts and hardware queues: tools/c5_queues.sh "workers:queues ..."
for wq in ${1:-"8:4 8:8 12:12"}; do
  w=${wq%%:*}; q=${wq##*:}
  GPU_MAX_HW_QUEUES=$q OPUSFE_BATCH_WORKERS=$w timeout -k 10 200 python bench.py --workload c5 --files 3000 --steps 2 --warmup 1 --no-cpu > gpurun_out/c5q.json 2> gpurun_out/c5q.err || { echo "$wq failed"; tail -2 gpurun_out/c5q.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/c5q.json')); print('workers $w queues $q:', d['value'], 'MB/s', d['ms_per_step'], 'ms')"
done
