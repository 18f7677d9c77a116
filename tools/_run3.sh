cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "chain" > gpurun_out/t4.log 2>&1
tail -n 3 gpurun_out/t4.log
for b in 1 2; do for c in 0 1; do
WN_NN_CHAIN=$c timeout -k 10 300 python bench.py --batch $b --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b chain $c %.3f ms/step nn %.1f us x %d' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['launches_per_step']))"
done; done
KB_ROWS=16000 timeout -k 10 300 python tools/chain_ab.py 2>&1 | grep -v amdgpu.ids
