#!/bin/bash
# Round-3 measurement run on the final tree (on the GPU box, from the repo root).  Outputs under gpurun_out/r03_final/:
#   train_b{128,1024,2048,4096,8192}/  rocprofv3 --kernel-trace --stats of profiling/one_step.py (training steps, GOBP-syn)
#   infer_b16384/                      the same of inference batches
#   bench_trace/                       rocprofv3 --kernel-trace --stats of `python3 bench.py --no-extras --no-cpu-baseline` (the SAME command
#                                      whose JSON line is bench_trace.json: its dw0 average must agree with roofline.kernel_ms_per_step)
#   pmc_fetch/, pmc_write/             the two HBM counter passes at batch 128 (one --pmc group per run)
#   bench_default.json, bench_b8192.json, bench_bf16_b8192.json, bench_fullgo_*.json, bench_infer.json   bench.py lines
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-r03_final}
mkdir -p $O
for B in 128 1024 2048 4096 8192; do
  n=30; [ $B -ge 4096 ] && n=12
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_b$B -o t -- python3 profiling/one_step.py $n $B train > $O/train_b$B.log 2>&1
done
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer_b16384 -o t -- python3 profiling/one_step.py 12 16384 infer > $O/infer_b16384.log 2>&1
echo traces done
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o t -- python3 bench.py --no-extras --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.log
echo bench trace done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 profiling/one_step.py 6 128 train > $O/pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 profiling/one_step.py 6 128 train > $O/pmc_w.log 2>&1
echo pmc done
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 200 python3 bench.py --global-batch 8192 --steps 30 --warmup 6 --no-cpu-baseline > $O/bench_b8192.json 2> $O/bench_b8192.err
for B in 1024 2048 4096; do timeout -k 10 200 python3 bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_b$B.json 2> $O/bench_b$B.err; done
timeout -k 10 200 python3 bench.py --global-batch 8192 --steps 30 --warmup 6 --no-cpu-baseline --dtype bf16 > $O/bench_bf16_b8192.json 2> $O/bench_bf16_b8192.err
timeout -k 10 200 python3 bench.py --global-batch 1024 --steps 50 --warmup 10 --no-cpu-baseline --ontology fullgo > $O/bench_fullgo_f32_b1024.json 2> $O/bench_fullgo_f32.err
timeout -k 10 200 python3 bench.py --global-batch 1024 --steps 50 --warmup 10 --no-cpu-baseline --ontology fullgo --dtype bf16 > $O/bench_fullgo_bf16_b1024.json 2> $O/bench_fullgo_bf16.err
timeout -k 10 200 python3 bench.py --mode infer > $O/bench_infer.json 2> $O/bench_infer.err
timeout -k 10 200 python3 bench.py --mode infer --dtype bf16 > $O/bench_infer_bf16.json 2> $O/bench_infer_bf16.err
echo benches done
