#!/bin/bash
# Hardware counters of ONE kernel of the training step (name substring $2) at batch $1, one rocprofv3 --pmc pass per counter
# group, each its own run of profiling/one_step.py.  $3 = ctx flags for one_step.py (measurement switches need the probe
# library: ONTOVAE_HIP_LIB=scratch/libontovae_probe.so, built with -DOV_MEASUREMENT_SWITCHES), $4 = output tag.
# usage (on the GPU box): bash profiling/kernel_counters.sh 8192 ov_bwd_gene 0 tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-8192}; K=${2:-ov_bwd_gene}; FL=${3:-0}; TAG=${4:-cnt}
O=$R/gpurun_out/$TAG
mkdir -p $O
i=0
for pm in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
          "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
          "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
          "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
          "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pm --output-format csv -d $O/p$i -o t -- python3 $R/profiling/one_step.py 3 $B train $FL > $O/p$i.log 2>&1 || { echo "pass $i failed: $pm"; tail -3 $O/p$i.log; continue; }
  python3 - "$O/p$i" "$K" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-34:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], f"{v[-1]:.5g}")
P
done
