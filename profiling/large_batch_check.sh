set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pipefix; mkdir -p $O
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_gpu_configs.py tests/test_gpu_parity.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for B in 1024 2048 8192; do
  n=30; [ $B -ge 4096 ] && n=12
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_b$B -o t -- python3 profiling/one_step.py $n $B train > $O/train_b$B.log 2>&1
done
python3 profiling/stats_table.py 30:$O/train_b1024/t_kernel_stats.csv 30:$O/train_b2048/t_kernel_stats.csv 12:$O/train_b8192/t_kernel_stats.csv > $O/table.txt
