cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/noriders; mkdir -p $O
for B in 1024 8192; do
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -o t -- python3 profiling/one_step.py 8 $B train 16 > $O/b$B.log 2>&1
done
python3 profiling/stats_table.py 8:$O/b1024/t_kernel_stats.csv 8:$O/b8192/t_kernel_stats.csv > $O/table.txt
