#!/bin/bash
# Round-3 (end) evidence run on one GPU box: default bench line, kernel traces of the fp32 C2 step, the C2 and C3
# bf16-storage steps, PMC passes over the C3 vocabulary forward on the eight-phase kernel, vendor comparison.  Outputs: gpurun_out/r3c_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python bench.py > $O/r3c_bench_default.json 2> $O/r3c_bench_default.err || exit 1
for leg in "fp32:" "c2s:--dtype bf16s" "c3:--config c3 --dtype bf16s"; do
  name=${leg%%:*}; args=${leg#*:}
  steps=20; [ $name = c3 ] && steps=6
  rm -rf $O/r3c_prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r3c_prof_$name -o trace -- python3 bench.py $args --steps $steps --warmup 3 --detail --no-cpu-baseline --no-decode --no-modes --no-legs > $O/r3c_bench_${name}_prof.json 2> $O/r3c_bench_${name}_prof.err || exit 1
  python tools/rocpd_stats.py $(ls $O/r3c_prof_$name/*.db | head -1) --timed > $O/r3c_stats_$name.txt || exit 1
  rm -rf $O/r3c_prof_$name
done
timeout -k 10 300 python -u tools/torch_gemm_ref_bf16.py > $O/r3c_vs_vendor.txt 2>&1 || exit 1
[ -n "$R3_SKIP_PMC" ] && { echo done; exit 0; }
i=0
for pmc in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf $O/r3c_pmc_$i
    GEMM_DTYPE=bf16s GEMM_CE=16 GEMM_TILE=256,256,1,32 timeout -s KILL 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/r3c_pmc_$i -o p -- python3 tools/gemm_one.py 0 1 5120 30000 2048 5 > $O/r3c_pmc_$i.log 2>&1 || exit 1
    python tools/pmc_summary.py $(find $O/r3c_pmc_$i -name "*counter_collection.csv" | head -1) gemm > $O/r3c_pmc_$i.txt || exit 1
    find $O/r3c_pmc_$i -name "*kernel_trace.csv" | head -1 | xargs -I{} python - {} >> $O/r3c_pmc_$i.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm" in r["Kernel_Name"]]
print("durations_us", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows])
PY
    rm -rf $O/r3c_pmc_$i
done
echo done
