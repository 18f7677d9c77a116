cd /root/repo
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -n 6 gpurun_out/t_all.log
