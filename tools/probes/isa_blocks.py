"""Instruction counts per basic block of one kernel in a hipcc -S listing (tools for profiles/r03_gn_isa.md).

    python tools/probes/isa_blocks.py <file.s> <mangled-kernel-name-substring> [--dump BLOCK]
"""
import re
import sys


def kernel_text(path, key):
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    return lines[start:end + 1]


def blocks(lines):
    out, cur = [], ['entry', '', []]
    for ln in lines:
        m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?', ln)
        if m:
            out.append(cur)
            cur = [m.group(1), (m.group(2) or ''), []]
        elif re.match(r'^\s+[a-z]', ln) and not ln.strip().startswith(('.', ';')):
            cur[2].append(ln.strip())
        elif ln.strip().startswith(';') and ('Loop' in ln):
            cur[1] += ' ' + ln.strip()
    out.append(cur)
    return out


if __name__ == '__main__':
    txt = kernel_text(sys.argv[1], sys.argv[2])
    bl = blocks(txt)
    if '--dump' in sys.argv:
        want = sys.argv[sys.argv.index('--dump') + 1]
        for name, note, ins in bl:
            if name.endswith('_' + want) or name == want:
                print(name, note)
                print('\n'.join('    ' + i for i in ins))
        sys.exit(0)
    tot = [0, 0, 0, 0]
    for name, note, ins in bl:
        v = sum(i.startswith('v_') for i in ins)
        sa = sum(i.startswith('s_') for i in ins)
        d = sum(i.startswith('ds_') for i in ins)
        m = sum(i.startswith(('scratch_', 'global_', 'buffer_', 'flat_')) for i in ins)
        depth = 'D2' if 'Depth=2' in note else ('D1' if ('Depth=1' in note or 'Header=' in note) else '  ')
        print(f'{name:12s} {depth} valu {v:4d} salu {sa:4d} ds {d:3d} vmem {m:3d}')
        for k, x in enumerate((v, sa, d, m)):
            tot[k] += x
    print('total', tot)
