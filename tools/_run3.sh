cd /root/repo
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parallel.py tests/test_gpu_model.py tests/test_gpu_stack.py -x -q -k "full_size or config1 or many_ranks or L70 or self_launches or backward_equals" > gpurun_out/t3.log 2>&1
tail -n 15 gpurun_out/t3.log
