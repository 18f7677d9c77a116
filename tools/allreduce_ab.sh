#!/bin/bash
# A/B of the gradient all-reduce on an N-GPU node (default 8): one call of the
# whole 6.06 MB bucket before the update vs two calls (the skip / post-processing
# tail from inside the backward pass, beside the backward stack), each under
# NCCL_ALGO=Ring and Tree.  Three interleaved runs per setting; prints ms/step,
# the HIP-event time around the (last) all-reduce and the global loss.
#   tools/allreduce_ab.sh [N] [bench.py args...]
# (Not runnable through the one-GPU harness: for whoever has the node.)
N=${1:-8}; shift
for rep in 1 2 3; do
  for algo in Ring Tree; do
    for two in True False; do
      echo -n "NCCL_ALGO=$algo dp_overlap_allreduce=$two rep $rep: "
      NCCL_ALGO=$algo timeout -k 10 600 python bench.py --gpus $N --steps 20 --warmup 5 \
          --set dp_overlap_allreduce=$two "$@" 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  allreduce %.0f us  loss %s' % (d['ms_per_step'], d.get('allreduce_us_per_step') or 0, d['config'].get('global_loss')))" || exit 1
    done
  done
done
