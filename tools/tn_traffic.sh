#!/bin/bash
# L2 fetch bytes (rocprofv3 --pmc FETCH_SIZE) of the three TN GEMM launches of
# a step for each library build given (default build/ab/lib_*.so), with and
# without the column sums:   tools/tn_traffic.sh [lib.so ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tn_traffic
mkdir -p $O
LIBS=${@:-$(ls tensorflow-wavenet_amd/build/ab/lib_*.so | grep -v stamps)}
export KB_ONLY=tn KB_REPS=1
for lib in $LIBS; do
  for cs in 0 1; do
    tag=$(basename $lib .so)_cs$cs
    KB_CS=$cs timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag -o p -- python3 tools/gemm_ab.py $lib > $O/$tag.log 2>&1 || exit 1
    f=$(ls $O/$tag/*counter_collection.csv | head -1)
    echo "== $tag"
    python3 tools/pmc_any.py $f gemm_tn | grep -v "^ *$"
  done
done
