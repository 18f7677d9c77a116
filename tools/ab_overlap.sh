#!/bin/bash
run() { echo -n "$1: "; env $1 timeout -k 10 180 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  loss %s' % (d['ms_per_step'], d['config'].get('final_loss')))"; }
for rep in 1 2 3; do
run "WN_X=0"
run "WN_OVERLAP_TN=1"
run "WN_OVERLAP_TN=1 WN_STACK_BWD_WAVES=4"
run "WN_STACK_BWD_WAVES=4"
done
