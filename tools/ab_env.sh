#!/bin/bash
# same-box A/B of one model attribute on the full training step:
#   tools/ab_env.sh stack_fwd "False True" [bench.py args...]
# (three interleaved runs per value; prints ms/step and the final loss)
var=$1; vals=$2; shift 2
for rep in 1 2 3; do
  for v in $vals; do
    echo -n "$var=$v rep $rep: "
    timeout -k 10 180 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --set "$var=$v" "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  loss %s' % (d['ms_per_step'], d['config'].get('final_loss')))" || exit 1
  done
done
