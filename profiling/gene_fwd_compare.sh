#!/bin/bash
# training step with the LDS-panel gene forward (flags 0) and with the 32-column-block one (flags 128), per batch size
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in "$@"; do
for fl in 0 128; do
  out=$R/gpurun_out/gfc_b${B}_f$fl
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/profiling/one_step.py 12 $B train $fl > /dev/null 2>&1 || exit 1
  python3 - "$out" "B=$B flags=$fl" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "ov_" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 12e3
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("ov_gene_lds") or n.startswith("ov_gene_cb"): print(f"{sys.argv[2]:18s} {n:24s} {float(r['AverageNs'])/1e3:9.1f} us   kernels per step {tot:.1f} us")
P
done
done
