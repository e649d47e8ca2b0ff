#!/bin/bash
# Round-3 measurement run (on the GPU box, from the repo root): rocprofv3 kernel traces of the training step at batch 128
# (BASELINE configs[1]) and at the per-GPU shapes of configs[2] (1024 / 2048 / 4096 / 8192 rows); outputs under gpurun_out/r03/.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-r03}
mkdir -p $O
for B in 128 1024 2048 4096 8192; do
  n=30; [ $B -ge 4096 ] && n=12
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_b$B -o t -- python3 profiling/one_step.py $n $B train > $O/train_b$B.log 2>&1
  echo trace $B done
done
