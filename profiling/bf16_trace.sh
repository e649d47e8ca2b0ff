#!/bin/bash
# Kernel traces of the bf16-MFMA variant's training step (ctx flag 32) at 1024 / 2048 / 8192 rows.  Outputs under gpurun_out/$1/.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-bf16}
mkdir -p $O
for B in 1024 2048 8192; do
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_b$B -o t -- python3 profiling/one_step.py 12 $B train 32 > $O/train_b$B.log 2>&1
done
python3 profiling/stats_table.py 12:$O/train_b1024/t_kernel_stats.csv 12:$O/train_b2048/t_kernel_stats.csv 12:$O/train_b8192/t_kernel_stats.csv > $O/kernel_table.txt
