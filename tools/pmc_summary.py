"""Per-kernel mean of each PMC counter from a rocprofv3 --pmc csv (counter_collection.csv); optional 2nd argument:
substring the kernel name must contain (default "gemm")."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if (sys.argv[2] if len(sys.argv) > 2 else "gemm") not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
