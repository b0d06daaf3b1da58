for mt in 40 50 100; do
  I2T_G256_MIN_TILES=$mt timeout -k 10 300 python bench.py --steps 1 --warmup 1 --batch 512 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_tiles=$mt captions/s', d['greedy_captions_per_sec'])"
done
