#!/bin/bash
# Collect the per-round evidence on the GPU box (run through gpurun from the
# repo root):  tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*
# Counters are collected in their own passes (never with --sys-trace); the
# program directly follows `--` (no env / shell wrapper).
set -e -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 3 --no-cpu-baseline --no-secondary"
python3 $R/bench.py > $O/${TAG}_bench.log 2> $O/${TAG}_bench.err
tail -1 $O/${TAG}_bench.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py $ARGS > /dev/null 2>&1
cp $O/${TAG}_stats/*/*kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $O/${TAG}_fetch/*/*_counter_collection.csv $O/${TAG}_write/*/*_counter_collection.csv $O/${TAG}_pmc_traffic.json
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_mfma -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 $R/tools/pmc_mfma.py $O/${TAG}_mfma/*/*_counter_collection.csv $O/${TAG}_mfma_util.json
# issue slots of the stack launches (their bound): a pass of its own
# (SQ_INST_CYCLES_VMEM / SQ_VALU_MFMA_COEXEC_CYCLES do not count on gfx950: dropped in round 6)
ISS="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $ISS --output-format csv -d $O/${TAG}_issue -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-secondary > $O/${TAG}_issue.log 2>&1
python3 $R/tools/pmc_issue.py $O/${TAG}_issue/*/*_counter_collection.csv $O/${TAG}_issue.json
# the microbenchmark behind "f32 MFMA and VALU are one resource"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu $R/tools/ubench/mfma_valu.hip && timeout -k 10 120 /tmp/mfma_valu > $O/${TAG}_mfma_valu.txt 2>&1 || true
# one lone wave's issue rate and the DPP / LDS / readlane mat-vec variants (fast generation's chain)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_matvec $R/tools/ubench/dpp_matvec.hip && timeout -k 10 120 /tmp/dpp_matvec > $O/${TAG}_dpp_matvec.txt 2>&1 || true
# raw traces are large: keep the summaries only
rm -rf $O/${TAG}_stats $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_mfma $O/${TAG}_issue
