cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "chain" > gpurun_out/t4.log 2>&1
tail -n 3 gpurun_out/t4.log
