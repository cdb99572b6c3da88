"""Per-kernel mean of each PMC counter from a rocprofv3 --pmc csv (counter_collection.csv)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "gemm" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
