#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / occupancy / LDS of every kernel in libgnnloop (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gnnkeras_amd', 'csrc', 'gnnloop.hip')
res = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off',
                      '-Rpass-analysis=kernel-resource-usage', '-c', '-o', '/tmp/_kr.o', src] + [a for a in sys.argv[1:] if a.startswith('-D')],
                     capture_output=True, text=True)
blocks = re.split(r'remark: [^\n]*Function Name: ', res.stderr)[1:]
want = [a for a in sys.argv[1:] if not a.startswith('-D')]
for b in blocks:
    name = b.split('\n')[0].strip().split(' ')[0]
    try: dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception: dn = name
    dn = re.sub(r'^void gnn::', '', dn).split('(')[0]
    if want and not any(w in dn for w in want): continue
    g = lambda k: re.search(k + r': (\S+)', b).group(1)
    print(dn[:64].ljust(64), 'VGPR', g('VGPRs').rjust(3), 'AGPR', g('AGPRs').rjust(3), 'SGPR', g('SGPRs').rjust(3), 'scratch',
          g(r'ScratchSize \[bytes/lane\]').rjust(4), 'occ', g(r'Occupancy \[waves/SIMD\]'), 'LDS', g(r'LDS Size \[bytes/block\]'))
