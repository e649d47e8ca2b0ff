#!/bin/bash
# Hardware counters of ONE kernel of the training step (name substring $2) at batch $1, one rocprofv3 --pmc pass per counter
# group, each its own run of profiling/one_step.py.  $3 = ctx flags for one_step.py (measurement switches need the probe
# library: ONTOVAE_HIP_LIB=scratch/libontovae_probe.so, built with -DOV_MEASUREMENT_SWITCHES), $4 = output tag.
# usage (on the GPU box): bash profiling/kernel_counters.sh 8192 ov_bwd_gene 0 tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-8192}; K=${2:-ov_bwd_gene}; FL=${3:-0}; TAG=${4:-cnt}; SET=${5:-gather}     # $2 may hold several substrings joined by '|'; $5 = gather | gemm
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ "$SET" = gemm ]; then
  GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_IFETCH" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_DEP_WAIT SQ_WAIT_INST_ANY SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_WAVES")
else
  GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
          "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
          "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
          "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
          "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum")
fi
i=0
for pm in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pm --output-format csv -d $O/p$i -o t -- python3 $R/profiling/one_step.py 3 $B train $FL > $O/p$i.log 2>&1 || { echo "pass $i failed: $pm"; tail -3 $O/p$i.log; continue; }
  python3 - "$O/p$i" "$K" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if any(k in r["Kernel_Name"] for k in sys.argv[2].split("|")):
        acc[(r["Kernel_Name"].split("(")[0][-34:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print(k[0], k[1], f"{v[-1]:.5g}")
P
done
