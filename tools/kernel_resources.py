#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy figures of the compiled kernels, from the compiler's own remarks
(`-Rpass-analysis=kernel-resource-usage`) on the COMMITTED sources with the Makefile's flags - so that the numbers quoted in
DESIGN.md and profiles/ are printed, not remembered (they drifted in three rounds running).

    python tools/kernel_resources.py                    # every kernel of every translation unit (slow: recompiles all)
    python tools/kernel_resources.py gn.hip refill      # one file, kernels whose name contains 'refill'
    python tools/kernel_resources.py --md gn.hip        # a markdown table
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dex-ct-sim_amd', 'csrc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Rpass-analysis=kernel-resource-usage',
         '--cuda-device-only', '-c', '-o', '/dev/null']
FIELDS = [('VGPRs', r'VGPRs: (\d+)'), ('AGPRs', r'AGPRs: (\d+)'), ('SGPRs', r'SGPRs: (\d+)'),
          ('scratch B/lane', r'ScratchSize \[bytes/lane\]: (\d+)'), ('VGPR spills', r'VGPRs Spill: (\d+)'),
          ('SGPR spills', r'SGPRs Spill: (\d+)'), ('LDS B/block', r'LDS Size \[bytes/block\]: (\d+)'),
          ('waves/SIMD', r'Occupancy \[waves/SIMD\]: (\d+)')]


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
        return out.splitlines()
    except OSError:
        return names


def resources(src):
    p = subprocess.run(['/opt/rocm/bin/hipcc', *FLAGS, os.path.join(CSRC, src)], capture_output=True, text=True)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-3000:])
    blocks = re.split(r'remark: [^\n]*Function Name: ', p.stderr)[1:]
    res = []
    for b in blocks:
        name = b.split()[0]
        row = {'kernel': name}
        for key, pat in FIELDS:
            m = re.search(pat, b)
            row[key] = int(m.group(1)) if m else None
        res.append(row)
    for row, nm in zip(res, demangle([r['kernel'] for r in res])):
        row['kernel'] = re.sub(r'^void dexct::', '', nm).split('(')[0]
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    md = '--md' in sys.argv
    files = [a for a in args if a.endswith('.hip')] or sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    pats = [a for a in args if not a.endswith('.hip')]
    head = ['kernel'] + [k for k, _ in FIELDS]
    if md:
        print('| ' + ' | '.join(head) + ' |\n|' + '---|' * len(head))
    for f in files:
        for row in resources(f):
            if pats and not any(p in row['kernel'] for p in pats):
                continue
            if md:
                print('| `' + row['kernel'] + '` | ' + ' | '.join(str(row[k]) for k, _ in FIELDS) + ' |')
            else:
                print(f"{f}: {row['kernel']}\n    " + ', '.join(f'{k} {row[k]}' for k, _ in FIELDS))


if __name__ == '__main__':
    main()
