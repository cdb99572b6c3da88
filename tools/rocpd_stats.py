"""Kernel-stats summary (name, calls, total, avg, %) from a rocprofv3 rocpd .db, plus GPU busy time
(union of kernel intervals) so overlapped side-stream work is visible.  usage: rocpd_stats.py file.db [steps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
# bench.py launches a one-workgroup capdec_probe_copy right before its timed region (after model set-up and GEMM plan
# measurement): with --timed only what follows that marker is counted
t_min = 0
if "--timed" in sys.argv:
    sys.argv.remove("--timed")
    m = db.execute("select min(start) from kernels where name like '%probe_copy_kernel%'").fetchone()[0]  # the first one
    t_min = m or 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
rows = db.execute("select name, start, end from kernels where start >= ? order by start", (t_min,)).fetchall()
by_grid = lambda: db.execute("select name, grid_x, grid_z, count(*), avg(end - start), min(end - start) from kernels "
                             "where name like '%gemm_%' and start >= ? group by name, grid_x, grid_z order by sum(end - start) desc",
                             (t_min,)).fetchall()
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(a[1] for a in agg.values())
busy, cur_s, cur_e = 0, None, None
for n, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
print(f"# {len(rows)} dispatches, sum of kernel durations {tot/1e6:.3f} ms, union (GPU busy) {busy/1e6:.3f} ms")
print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:100]:100s} {c:7d} {t/1e6:10.3f} {t/c/1e3:9.2f} {100*t/tot:6.2f}")
print()
print("# GEMM launches by (kernel, grid): the vocab-projection forward is gemm_f32_kernel<128,64,..,NT> with grid_x = 40*157*256")
print(f"{'kernel':72s} {'grid_x':>9s} {'gz':>3s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s}")
for n, gx, gz, c, avg, mn in by_grid()[:40]:
    short = n.replace("void capdec::", "").replace("(capdec::GemmParams)", "")
    print(f"{short[:72]:72s} {gx:9d} {gz:3d} {c:6d} {avg/1e3:9.2f} {mn/1e3:9.2f}")
