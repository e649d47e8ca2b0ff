#!/bin/bash
# Copy what one run of profiling/r03_final.sh left under gpurun_out/$1 into profiles/r03/ (the committed summaries) and rebuild the
# derived files (kernel_table.txt, pmc_traffic.json).  Run from the repo root in the build container, on a clean tree, so that
# pmc_traffic.json's stamp names the commit that was measured.
set -e
S=gpurun_out/${1:-r03_final}; D=profiles/r03
for B in 128 1024 2048 4096 8192; do cp $S/train_b$B/t_kernel_stats.csv $D/train_b${B}_kernel_stats.csv; done
cp $S/infer_b16384/t_kernel_stats.csv $D/infer_b16384_kernel_stats.csv
cp $S/bench_trace/t_kernel_stats.csv $D/bench_trace_kernel_stats.csv
cp $S/bench_trace.json $D/bench_under_rocprof.json
cp $S/pmc_fetch/f_counter_collection.csv $D/pmc_fetch_size.csv
cp $S/pmc_write/w_counter_collection.csv $D/pmc_write_size.csv
for f in bench_default bench_b1024 bench_b2048 bench_b4096 bench_b8192 bench_bf16_b8192 bench_fullgo_f32_b1024 bench_fullgo_bf16_b1024 bench_infer bench_infer_bf16; do cp $S/$f.json $D/$f.json; done
python3 profiling/stats_table.py 30:$D/train_b128_kernel_stats.csv 30:$D/train_b1024_kernel_stats.csv 30:$D/train_b2048_kernel_stats.csv 12:$D/train_b4096_kernel_stats.csv 12:$D/train_b8192_kernel_stats.csv > $D/kernel_table.txt
python3 profiling/pmc_traffic.py $D/pmc_fetch_size.csv $D/pmc_write_size.csv > /dev/null
cp profiles/pmc_traffic.json $D/pmc_traffic.json
