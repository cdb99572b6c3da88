#!/bin/bash
# A/B of two builds of the library on one box: $1 = the other library (path inside the repo); C2 fp32, C2 bf16s and C3 bf16s steps, alternating.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
: > $O/ab_lib3.txt
for rep in 1 2; do
  for v in base other; do
    if [ $v = other ]; then export CAPDEC_LIBRARY="$GRAFT_REPO_ROOT/$1"; else unset CAPDEC_LIBRARY; fi
    for cfg in "" "--dtype bf16s" "--config c3 --dtype bf16s"; do
      timeout -k 10 300 python -u bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-modes --no-legs > $O/ab_tmp.json 2> $O/ab_tmp.err || exit 1
      python - "$v" "$cfg" <<'PY' >> $O/ab_lib3.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], repr(sys.argv[2]), "ms_per_step %.3f" % d["ms_per_step"], "gemm_all %.3f" % d["gemm_all"]["frac"], "vocab %.3f" % d["roofline"]["frac"], flush=True)
PY
    done
  done
done
cat $O/ab_lib3.txt
