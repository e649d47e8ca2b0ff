#!/bin/bash
# per-kernel times of the training step at the given batch sizes: bash profiling/step_kernels.sh 1024 8192
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in "$@"; do
  out=$R/gpurun_out/sk_b$B
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $R/profiling/one_step.py 12 $B train 0 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
  python3 - "$out" "$B" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "ov_" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 12e3
print(f"B={sys.argv[2]}: kernels per step {tot:.1f} us")
for r in rows: print(f"   {r['Name'].split('(')[0].replace('void ', '')[:44]:46s} {float(r['TotalDurationNs'])/12e3:9.1f}")
P
done
