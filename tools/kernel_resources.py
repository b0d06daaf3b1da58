#!/usr/bin/env python3
"""Register / spill summary per kernel of one HIP source (compile only; no GPU needed).

    python tools/kernel_resources.py image2text_amd/csrc/gemm.hip [name-regex] [extra hipcc flags...]
"""
import re
import subprocess
import sys

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else '.'
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'] + sys.argv[3:], capture_output=True, text=True).stderr
cur, rows = None, {}
for ln in out.splitlines():
    m = re.search(r'Function Name: (\S+)', ln)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r'(SGPRs Spill|VGPRs Spill|VGPRs|AGPRs|ScratchSize|Occupancy)[^:]*: (\d+)', ln)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    name = name.replace('(anonymous namespace)::', '')
    if not re.search(filt, name):
        continue
    g = lambda key: v.get(key, -1)
    print(f'{name[:80]:80s} vgpr {g("VGPRs"):3d} agpr {g("AGPRs"):3d} sgpr-spill {g("SGPRs Spill"):3d} vgpr-spill {g("VGPRs Spill"):3d} '
          f'scratch {g("ScratchSize"):4d} occ {g("Occupancy")}')
