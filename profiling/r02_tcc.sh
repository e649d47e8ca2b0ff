#!/bin/bash
# L2 (TCC) counters of the training step at batch 128: hit / miss / requests, and the memory-side read requests split by width.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_tcc -o h -- python3 profiling/one_step.py 6 128 train > $O/pmc_h.log 2>&1 || exit 1
echo hit done
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/pmc_rd -o r -- python3 profiling/one_step.py 6 128 train > $O/pmc_r.log 2>&1 || exit 1
echo rd done
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmc_wr -o w2 -- python3 profiling/one_step.py 6 128 train > $O/pmc_w2.log 2>&1 || exit 1
echo wr done
