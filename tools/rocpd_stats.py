"""Kernel-stats summary (name, calls, total, avg, %) from a rocprofv3 rocpd .db, plus GPU busy time
(union of kernel intervals) so overlapped side-stream work is visible.  usage: rocpd_stats.py file.db [steps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels order by start").fetchall()
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
