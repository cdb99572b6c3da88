#!/bin/bash
# A/B on one box: vector epilogue (default) vs element form (CAPDEC_NO_VEC_EPI=1); C2 fp32, C2 bf16s, C3 bf16s.  Alternating runs.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
: > $O/ab_epi.txt
for rep in 1 2; do
  for v in vec elem; do
    if [ $v = elem ]; then export CAPDEC_NO_VEC_EPI=1; else unset CAPDEC_NO_VEC_EPI; fi
    for cfg in "" "--dtype bf16s" "--config c3 --dtype bf16s"; do
      timeout -k 10 300 python -u bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-modes --no-legs > $O/ab_tmp.json 2> $O/ab_tmp.err || exit 1
      python - "$v" "$cfg" <<'PY' >> $O/ab_epi.txt
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], repr(sys.argv[2]), "ms_per_step %.3f" % d["ms_per_step"], "gemm_all %.3f" % d["gemm_all"]["frac"], "vocab %.3f" % d["roofline"]["frac"], flush=True)
PY
    done
  done
done
cat $O/ab_epi.txt
