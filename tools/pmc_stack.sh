#!/bin/bash
# Dynamic instruction mix of the persistent stack kernels of one library build
# (run through gpurun from the repo root):  tools/pmc_stack.sh <lib.so> [bwd|fwd]
# Prints chip totals per launch; divide by 200000 (tiles x layers at B=8,
# T=16000) for per-tile figures.
LIB=$(realpath ${1:-tensorflow-wavenet_amd/libwavenet_hip.so}); WHAT=${2:-bwd}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp KB_ONLY=$WHAT KB_REPS=2
rm -rf /tmp/pmc_stack
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_stack -- python3 $R/tools/stack_ab.py $LIB > /tmp/pmc_stack.log 2>&1
python3 $R/tools/pmc_any.py /tmp/pmc_stack/*/*_counter_collection.csv stack_$WHAT
