#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh <kbench args...>
# (variants under tensorflow-wavenet_amd/build/ab/lib_*.so, interleaved 3x)
for rep in 1 2 3; do
  for lib in tensorflow-wavenet_amd/build/ab/lib_*.so; do
    echo "== $(basename $lib) rep $rep"
    WN_LIB_PATH=$PWD/$lib python tools/kbench.py "$@" 2>&1 | grep -v amdgpu
  done
done
