"""Per-kernel microseconds per training step from rocprofv3 --kernel-trace --stats CSVs, side by side.
usage: python3 profiling/stats_table.py steps:dir/t_kernel_stats.csv [steps:dir2/...]   (ov_* kernels only)"""
import csv
import re
import sys

def label(path):
    parts = path.split("/")
    stem = parts[-1].replace("_kernel_stats.csv", "").replace(".csv", "")
    return parts[-2] if stem in ("t", "t_kernel_stats") and len(parts) > 1 else stem


cols, names = [], []
for arg in sys.argv[1:]:
    steps, path = arg.split(":", 1)
    d = {}
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if "ov_" not in n:
            continue
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)
        d[n] = (float(r["TotalDurationNs"]) / 1e3 / int(steps), int(r["Calls"]) / int(steps))
        if n not in names:
            names.append(n)
    cols.append(d)
names.sort(key=lambda n: -max(c.get(n, (0, 0))[0] for c in cols))
print("%-62s" % "kernel (us per step, launches per step)" + "".join("%18s" % label(a.split(":", 1)[1]) for a in sys.argv[1:]))
for n in names:
    print("%-62s" % n[:62] + "".join(("%12.1f x%-4.3g" % c[n]) if n in c else "%18s" % "-" for c in cols))
print("%-62s" % "sum" + "".join("%12.1f      " % sum(v[0] for v in c.values()) for c in cols))
