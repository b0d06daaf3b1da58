#!/bin/bash
# Same-box A/B of the training step: tools/ab_step.sh "<env A>" "<env B>" [steps]   (each arm twice, interleaved; prints ms/step)
A="$1"; B="$2"; STEPS="${3:-8}"
for rep in 1 2; do
  for arm in A B; do
    if [ $arm = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --steps $STEPS --warmup 2 --no-decode --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$arm [$E]', d['ms_per_step'])"
  done
done
