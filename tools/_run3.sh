cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_fastgen_default.py -x -q > gpurun_out/t6.log 2>&1
tail -n 4 gpurun_out/t6.log
for p in 0 1; do echo -n "persist=$p: "; WN_FASTGEN_PERSIST=$p timeout -k 10 200 python tools/fastgen_time.py 16000 2>&1 | grep -v amdgpu; done
WN_LIB_PATH=$PWD/tensorflow-wavenet_amd/build/ab/lib_stamps.so timeout -k 10 200 python tools/fgp_stamps.py 2000 2>&1 | grep -v amdgpu
