"""Instruction mix per basic block of one kernel in a hipcc -S listing: tools/isa_blocks.py file.s <mangled-name regex> [min-instructions]"""
import re, sys
s = open(sys.argv[1]).read()
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = re.search(r'^(' + sys.argv[2] + r'\w*):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
print(m.group(1))
blocks, cur = [], ['entry', []]
for l in m.group(2).split('\n'):
    l = l.strip()
    if l.startswith('.LBB'):
        blocks.append(cur); cur = [l.split(':')[0] + (' loop' if 'Loop' in l else ''), []]
    elif l and not l.startswith(';') and not l.startswith('.'):
        cur[1].append(l)
blocks.append(cur)
tot = dict(n=0, mfma=0, valu=0, salu=0, ds=0, vmem=0)
for name, ins in blocks:
    c = dict(n=len(ins), mfma=sum(i.startswith('v_mfma') for i in ins),
             valu=sum(i.startswith('v_') and not i.startswith('v_mfma') for i in ins), salu=sum(i.startswith('s_') for i in ins),
             ds=sum(i.startswith('ds_') for i in ins), vmem=sum(i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) for i in ins))
    for k in tot: tot[k] += c[k]
    if c['n'] >= mn: print(f"{name:18s}", ' '.join(f"{k}={v:4d}" for k, v in c.items()))
print('total', tot)
