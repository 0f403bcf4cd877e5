"""Instruction histogram per large basic block of one kernel in an ISA listing (hipcc -S --cuda-device-only)."""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
L = open(path).read().split('\n')
st = [i for i, l in enumerate(L) if l.startswith(sym + ':')][0]
en = [i for i, l in enumerate(L) if i > st and '.end_amdhsa_kernel' in l][0]
blocks, cur, name = [], [], 'entry'
for l in L[st:en]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append((name, cur)); cur = []; name = m.group(1)
    elif l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.'):
        cur.append(l.split()[0])
blocks.append((name, cur))
for n, b in blocks:
    if len(b) > 300:
        c = collections.Counter(b)
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        print(n, len(b), 'valu', valu, 'mfma', sum(v for k, v in c.items() if k.startswith('v_mfma')), 'exp', c['v_exp_f32_e32'])
        print('   ', c.most_common(18))
