#!/bin/bash
# HBM-side traffic of the backward gene kernel and of the launches that read its partials, at $1 columns, with one partial copy of
# the weight gradient per XCD (flags 0) and per column block (flags 64 = ONTOVAE_FLAG_PER_BLOCK_WGRAD): FETCH_SIZE / WRITE_SIZE,
# one rocprofv3 --pmc pass each (KB; FETCH_SIZE to be doubled on gfx950, see profiling/pmc_traffic.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=${1:-8192}; O=$R/gpurun_out/wgrad_copies; mkdir -p $O
for FL in 0 64; do for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/f${FL}_$C -o t -- python3 $R/profiling/one_step.py 3 $B train $FL > $O/f${FL}_$C.log 2>&1 || { echo "pass failed: $FL $C"; exit 1; }
  python3 - "$O/f${FL}_$C" "$FL" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("ov_bwd_gene", "ov_latent_bwd", "ov_dec_bwd_kernel", "ov_whT_finish")):
        acc[(n.split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    per_step = sum(v) / 3.0
    print(f"flags {sys.argv[2]:>2}  {k[0]:<46} {k[1]:<11} {per_step / 1024.0 * (2 if k[1] == 'FETCH_SIZE' else 1):9.1f} MB per step")
P
done; done
