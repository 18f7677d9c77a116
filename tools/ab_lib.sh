#!/bin/bash
# same-box A/B of library builds on the full training step:
#   tools/ab_lib.sh [bench.py args...]   (variants: tensorflow-wavenet_amd/build/ab/lib_*.so, interleaved 3x)
for rep in 1 2 3; do
  for lib in tensorflow-wavenet_amd/build/ab/lib_*.so; do
    echo -n "$(basename $lib) rep $rep: "
    WN_LIB_PATH=$PWD/$lib timeout -k 10 180 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  loss %s' % (d['ms_per_step'], d['config'].get('final_loss')))" || exit 1
  done
done
