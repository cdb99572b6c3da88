#!/bin/bash
# Round-3 evidence run (one GPU box): default bench line, kernel traces of the fp32 C2 step and the C3 bf16-storage leg,
# PMC passes over the fp32 ring attention kernels.  Outputs under gpurun_out/r3e_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python bench.py > $O/r3e_bench_default.json 2> $O/r3e_bench_default.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r3e_prof_fp32 -o trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-modes --no-legs > $O/r3e_bench_fp32_prof.json 2> $O/r3e_bench_fp32_prof.err || exit 1
python tools/rocpd_stats.py $(ls $O/r3e_prof_fp32/*.db | head -1) --timed > $O/r3e_stats_fp32.txt || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r3e_prof_c3 -o trace -- python3 bench.py --config c3 --dtype bf16s --steps 6 --warmup 2 --no-cpu-baseline --no-decode --no-modes --no-legs > $O/r3e_bench_c3_prof.json 2> $O/r3e_bench_c3_prof.err || exit 1
python tools/rocpd_stats.py $(ls $O/r3e_prof_c3/*.db | head -1) --timed > $O/r3e_stats_c3.txt || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/r3e_prof_c2s -o trace -- python3 bench.py --dtype bf16s --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-modes --no-legs > $O/r3e_bench_c2s_prof.json 2> $O/r3e_bench_c2s_prof.err || exit 1
python tools/rocpd_stats.py $(ls $O/r3e_prof_c2s/*.db | head -1) --timed > $O/r3e_stats_c2s.txt || exit 1
[ -n "$R3_SKIP_PMC" ] && { echo done; exit 0; }
i=0
for pmc in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/r3e_pmc_attn_$i -o p -- python3 tools/attn_one.py > $O/r3e_pmc_attn_$i.log 2>&1 || exit 1
done
echo done
