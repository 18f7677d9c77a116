#!/bin/bash
# the practical streaming rates the stack launches are priced against:
#   tools/hbm_stream.sh <tag>   -> profiles/<tag>_hbm_stream.txt  (run on the GPU box)
set -e
TAG=${1:-r04}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_stream tools/ubench/hbm_stream.hip
mkdir -p gpurun_out
timeout -k 10 120 /tmp/hbm_stream | tee gpurun_out/${TAG}_hbm_stream.txt
