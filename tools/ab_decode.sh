#!/bin/bash
# Same-box A/B of the decode leg: tools/ab_decode.sh "<env A>" "<env B>"   (each arm twice, interleaved; prints captions/s)
A="$1"; B="$2"
for rep in 1 2; do
  for arm in A B; do
    if [ $arm = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$arm [$E]', d['greedy_captions_per_sec'])"
  done
done
