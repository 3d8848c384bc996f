#!/usr/bin/env python3
"""Instruction mix, registers and spills of one kernel from a device assembly file (hipcc --cuda-device-only -S).
usage: python tools/isa_stats.py <file.s> <mangled-name substring> [--dump out.s]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n', s, re.M)
name = m.group(1)
start = m.end()
end = s.index('.Lfunc_end', start)
body = s[start:end]
if '--dump' in sys.argv:
    open(sys.argv[sys.argv.index('--dump') + 1], 'w').write(body)
lines = [l.split(';')[0].strip() for l in body.split('\n')]
lines = [l for l in lines if l and not l.startswith('.') and not l.endswith(':')]
print(name[:80])
print('instructions', len(lines))
c = collections.Counter(l.split()[0] for l in lines)
groups = collections.Counter()
for op, n in c.items():
    g = ('valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'scratch' if op.startswith('scratch_') else 'other')
    groups[g] += n
print(dict(groups))
print(', '.join('%s %d' % kv for kv in c.most_common(40)))
i = s.index('.name:           ' + name)
meta = s[max(0, i - 2500):i + 1200]
for k in ('vgpr_count', 'vgpr_spill_count', 'sgpr_count', 'sgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size', 'agpr_count'):
    mm = re.findall(r'\.' + k + r':\s+(\d+)', meta)
    print(k, mm)
