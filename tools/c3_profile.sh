#!/bin/bash
# Kernel trace of the C3 bf16-storage leg (one GPU box); outputs under gpurun_out/c3p_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/c3p_prof
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/c3p_prof -o trace -- python3 bench.py --config c3 --dtype bf16s --steps 6 --warmup 2 --detail --no-cpu-baseline --no-decode --no-modes --no-legs > $O/c3p_bench.json 2> $O/c3p_bench.err || exit 1
python tools/rocpd_stats.py $(ls $O/c3p_prof/*.db | head -1) --timed > $O/c3p_stats.txt || exit 1
rm -rf $O/c3p_prof
echo done
