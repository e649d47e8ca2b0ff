#!/bin/bash
# Round-2 measurement run (on the GPU box, from the repo root): kernel trace + the two HBM counter passes of the training step at
# batch 128 (BASELINE configs[1]) and the kernel trace of inference; outputs under gpurun_out/r02/.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -o t -- python3 profiling/one_step.py 60 128 train > $O/train_trace.log 2>&1
echo trace done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 profiling/one_step.py 6 128 train > $O/pmc_f.log 2>&1
echo fetch done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 profiling/one_step.py 6 128 train > $O/pmc_w.log 2>&1
echo write done
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer_trace -o t -- python3 profiling/one_step.py 20 4096 infer > $O/infer_trace.log 2>&1
echo infer trace done
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train8192_trace -o t -- python3 profiling/one_step.py 12 8192 train > $O/train8192_trace.log 2>&1
echo 8192 trace done
