#!/bin/bash
# A/B runs of the fp32 C2 step under environment switches (one bench process each); prints ms per step.
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-modes --no-legs"
for v in "$@"; do
    n=${v%%:*}; e=${v#*:}
    env $e timeout -k 10 300 $B > gpurun_out/r3_ab_$n.json 2> gpurun_out/r3_ab_$n.err || { echo "$n FAILED"; tail -3 gpurun_out/r3_ab_$n.err; exit 1; }
    python -c "
import json
d=json.load(open('gpurun_out/r3_ab_$n.json')); print('$n $e', round(d['ms_per_step'],3))"
done
