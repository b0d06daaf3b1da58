#!/bin/bash
# training step under several environments, same box, two rounds:  tools/ab_stepN.sh [steps] "<env A>" "<env B>" ...
STEPS=$1; shift
for rep in 1 2; do
  for E in "$@"; do
    env $E python bench.py --steps $STEPS --warmup 2 --no-decode --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$E]', d['ms_per_step'])"
  done
done
