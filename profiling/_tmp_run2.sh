cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/probe_cnt; mkdir -p $O
i=0
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
          "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
          "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
          "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -s KILL 100 rocprofv3 --pmc $pm --output-format csv -d $O/p$i -o t -- $R/scratch/mfma_probe/mfma_probe > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; continue; }
  python3 - "$O/p$i" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "chunkloop" in n and "true" not in n:
        acc[(n.split("(")[0][-24:], r["Counter_Name"])].append(float(r["Counter_Value"]))
want = ["chunkloop<3, false>", "chunkloop<4, false>", "chunkloop<11, false>", "chunkloop<5, false>", "chunkloop<13, false>"]
names = sorted({k[1] for k in acc})
print("%-36s" % "counter (last launch)" + "".join("%14s" % w[9:-8] for w in want))
for c in names:
    print("%-36s" % c + "".join("%14.4g" % (acc[(next((k[0] for k in acc if k[0].endswith(w)), ""), c)] or [float("nan")])[0 if w != "chunkloop<5, false>" else 0] for w in want))
P
done
