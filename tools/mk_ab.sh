#!/bin/bash
# build two variants of the library for tools/ab_lib.sh:
#   tools/mk_ab.sh <file.hip>   -> build/ab/lib_base.so (the file as of git HEAD)
#                                  build/ab/lib_new.so  (the working tree)
set -e
PK=tensorflow-wavenet_amd
F=${1:-wn_stack.hip}
N=${F%.hip}
mkdir -p $PK/build/ab /tmp/ab_src
rm -f $PK/build/ab/lib_*.so
mkdir -p /tmp/ab_src/base
git show HEAD:$PK/csrc/$F > /tmp/ab_src/base/$F
git show HEAD:$PK/csrc/wn_common.h > /tmp/ab_src/base/wn_common.h
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
/opt/rocm/bin/hipcc $FL -c /tmp/ab_src/base/$F -o /tmp/ab_src/base.o
/opt/rocm/bin/hipcc $FL -c $PK/csrc/$F -o /tmp/ab_src/new.o
OTHERS=$(ls $PK/build/*.o | grep -v "/$N.o")
/opt/rocm/bin/hipcc $FL -shared -o $PK/build/ab/lib_base.so /tmp/ab_src/base.o $OTHERS
/opt/rocm/bin/hipcc $FL -shared -o $PK/build/ab/lib_new.so /tmp/ab_src/new.o $OTHERS
ls -la $PK/build/ab/
