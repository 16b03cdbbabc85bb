"""Summarise a rocprofv3 kernel-trace csv: per-kernel count, mean, total."""
import csv, sys, collections, glob
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[n][0] += 1; agg[n][1] += d
tot = sum(v[1] for v in agg.values())
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t:12.1f} us  {c:6d} x {t/c:10.1f} us  {100*t/tot:5.1f}%  {n[:110]}")
print("total us", tot)
