"""Runs of non-MFMA instructions between MFMAs for one basic block: python tools/isa_runs.py ke.s SYMBOL .LBBx_y"""
import sys, collections
path, sym, lab = sys.argv[1:4]
L = open(path).read().split('\n')
st = [i for i, l in enumerate(L) if l.startswith(sym + ':')][0]
i0 = [i for i, l in enumerate(L) if i > st and l.startswith(lab + ':')][0]
ops = []
for l in L[i0 + 1:]:
    t = l.strip()
    if t.startswith('.LBB') and t.split(':')[0] != lab: break
    if not t or t.startswith(';') or t.startswith('.'): continue
    ops.append(t.split()[0])
out, run = [], []
def flush():
    global run
    if run:
        c = collections.Counter('V' if o.startswith('v_') else 'D' if o.startswith('ds_') else 'G' if o.startswith('global') or o.startswith('scratch') else 'W' if o == 's_waitcnt' else 'N' if o == 's_nop' else 's' for o in run)
        out.append(' '.join(f'{k}{v}' for k, v in sorted(c.items())))
        run = []
for o in ops:
    if o.startswith('v_mfma'):
        flush(); out.append('M')
    else:
        run.append(o)
flush()
print(' | '.join(out))
