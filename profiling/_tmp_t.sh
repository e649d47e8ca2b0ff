set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_gpu_parity.py tests/test_gpu_configs.py > gpurun_out/t_acc.log 2>&1 || { tail -30 gpurun_out/t_acc.log; exit 1; }
tail -2 gpurun_out/t_acc.log
bash profiling/_tmp_ab.sh acc 512 1024 2048
