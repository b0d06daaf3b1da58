#!/bin/bash
# decode leg over (captions per batch, concurrent batches):  tools/sweep_decode.sh "4096 3" "4096 4" ...
for cfg in "$@"; do
  set -- $cfg
  python bench.py --steps 1 --warmup 1 --batch 256 --no-cpu-baseline --no-kernel-timing --decode-batch $1 --decode-streams $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$cfg]', d['greedy_captions_per_sec'])"
done
