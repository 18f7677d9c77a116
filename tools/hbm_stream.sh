#!/bin/bash
# the practical streaming rates the stack launches are priced against:
#   tools/hbm_stream.sh <tag>   -> gpurun_out/<tag>_hbm_stream.txt on the GPU box (the only
#   directory gpurun brings back); copy it to profiles/<tag>_hbm_stream.txt, where bench.py reads it
set -e
TAG=${1:-r04}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_stream tools/ubench/hbm_stream.hip
mkdir -p gpurun_out
timeout -k 10 120 /tmp/hbm_stream | tee gpurun_out/${TAG}_hbm_stream.txt
