cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_cli.py -x -q -k "generat or incremental or sampling or cli" > gpurun_out/t7.log 2>&1
tail -n 8 gpurun_out/t7.log
