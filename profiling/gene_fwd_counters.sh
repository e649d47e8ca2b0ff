#!/bin/bash
# HBM-side counters of the two gene forward kernels at a given batch (one --pmc pass per run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-8192}
for fl in 0 128; do
for pm in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  out=$R/gpurun_out/gp4_f${fl}_$(echo $pm | cut -d' ' -f1)
  timeout -s KILL 170 rocprofv3 --pmc $pm --output-format csv -d $out -o t -- python3 $R/profiling/one_step.py 3 $B train $fl > /dev/null 2>&1 || exit 1
  python3 - "$out" "$fl" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "ov_gene_lds" in r["Kernel_Name"] or "ov_gene_cb" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print("flags", sys.argv[2], k, f"{v[-1]:.4g}")
P
done
done
