#!/usr/bin/env python3
"""Dev tool: per-basic-block instruction histogram of one kernel in a hipcc -S listing.
usage: isa_blocks.py file.s <substring of the kernel's mangled name> [min block size]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = [i for i, l in enumerate(lines) if key in l and l.startswith('_Z') and ': ' in l][0]
end = [i for i, l in enumerate(lines) if 's_endpgm' in l and i > start][0]
def cat(op):
    if re.match(r'v_(fma|mul|add|fmac)_f64', op): return 'f64'
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
    if 'dpp' in op: return 'dpp'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_mov'): return 'mov'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'scratch_', 'buffer_')): return 'vmem'
    return 'other'
blocks, cur = [], ['entry', start, []]
for i in range(start + 1, end + 1):
    s = lines[i].strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        blocks.append(cur); cur = [m.group(1), i, []]; continue
    if not s or s[0] in ';.':
        continue
    op = s.split()[0]
    full = s
    cur[2].append(op + ('_dpp' if 'row_' in full or 'quad_perm' in full else ''))
blocks.append(cur)
tot = collections.Counter()
for name, ln, ops in blocks:
    c = collections.Counter(cat(o) for o in ops)
    tot += c
    if len(ops) >= minsz:
        print(f"{name:12s} line {ln:6d} n={len(ops):4d}", dict(sorted(c.items())))
print('total', dict(sorted(tot.items())), sum(tot.values()))
