# Same-box A/B of two builds of the library: traces `one_step.py 12 <rows> train` with the in-tree library ("base") and with
# scratch/libontovae_alt.so ("alt", through ONTOVAE_HIP_LIB) and puts the two kernel tables side by side.
#   bash profiling/ab_trace.sh <tag> <rows> [<rows> ...]   ->  gpurun_out/ab_<tag>/table_<rows>.txt
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/ab_$1; mkdir -p $O
shift
for B in "$@"; do
for V in base alt; do
  if [ $V = base ]; then unset ONTOVAE_HIP_LIB; else export ONTOVAE_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libontovae_alt.so; fi
  timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${V}_b$B -o t -- python3 profiling/one_step.py 12 $B train > $O/${V}_b$B.log 2>&1
done
python3 profiling/stats_table.py 12:$O/base_b$B/t_kernel_stats.csv 12:$O/alt_b$B/t_kernel_stats.csv > $O/table_$B.txt
done
echo done
