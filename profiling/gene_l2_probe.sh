# L2 hit / miss requests and fabric-side fetch size (KB, to be doubled on gfx950) of the two gene-layer kernels of a training step at
# the batch sizes given (default 512 ... 8192), one rocprofv3 --pmc pass per counter group and size.
#   bash profiling/gene_l2_probe.sh 1024 8192 > gpurun_out/l2probe.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/l2probe; mkdir -p $O
for B in ${@:-512 1024 2048 4096 8192}; do for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | cut -c1-7)
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/b${B}_$T -o t -- python3 $R/profiling/one_step.py 3 $B train > $O/b${B}_$T.log 2>&1 || { echo "pass failed: $B $C"; exit 1; }
  python3 - "$O/b${B}_$T" "$B" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in ("ov_bwd_gene", "ov_gene_lds", "ov_gene_cb")):
        acc[(n.split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"B {sys.argv[2]:>5}  {k[0]:<42} {k[1]:<12} {sum(v) / 3.0:14.0f} per step")
P
done; done
