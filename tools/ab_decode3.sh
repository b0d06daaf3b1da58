#!/bin/bash
# decode leg under several environments, same box, two rounds:  tools/ab_decode3.sh "<env A>" "<env B>" "<env C>" ...
for rep in 1 2; do
  for E in "$@"; do
    env $E python bench.py --steps 1 --warmup 1 --batch 256 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$E]', d['greedy_captions_per_sec'])"
  done
done
