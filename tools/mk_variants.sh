#!/bin/bash
# build named variants of ONE source file for tools/stack_ab.py / gemm_ab.py / ab_lib.sh:
#   tools/mk_variants.sh wn_stack.hip base: xnt:-DSB_X_AUX=2 "both:-DSB_X_AUX=2 -DSB_OWN_ST=1"
# -> tensorflow-wavenet_amd/build/ab/lib_<name>.so (the other objects from build/)
set -e
PK=tensorflow-wavenet_amd
F=$1; shift
N=${F%.hip}
mkdir -p $PK/build/ab /tmp/ab_src
rm -f $PK/build/ab/lib_*.so
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
OTHERS=$(ls $PK/build/*.o | grep -v "/$N.o")
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FL $defs -c $PK/csrc/$F -o /tmp/ab_src/$name.o
  /opt/rocm/bin/hipcc $FL -shared -o $PK/build/ab/lib_$name.so /tmp/ab_src/$name.o $OTHERS
done
ls -la $PK/build/ab/
