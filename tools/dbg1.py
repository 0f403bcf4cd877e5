import sys; sys.path.insert(0,'/root/repo')
import torch
from tests.test_gpu_denoiser import *
from diffalign_amd import engine, _lib as L
from diffalign_amd.engine import _i
B,n=3,44
X, E, y, nm, am, pos, mol = inputs(B, n, seed=11, dx=256)
src = torch.empty(B, n, dtype=torch.int32, device=D)
L.check(L.lib().da_align_index(L.ptr(am.to(D)), L.ptr(mol.to(D)), L.ptr(src), _i(B), _i(n), L.stream()), 'align')
print(src.cpu()[0]); print(O.align_index(am, mol)[0])
m, sd = build(TINY, 2, seed=21)
X, E, y, nm, am, pos, mol = inputs(4, 24, seed=31, dx=64)
with torch.no_grad():
    r = m(X.to(D), E.to(D), y.to(D), nm.to(D), am.to(D), pos.to(D), mol.to(D))
o = O.denoiser_forward(sd, X, E, y, nm, 2, 4, (KX, KE, 1), am, pos, mol, True)
print('relX', rel_l2(r[0],o[0]), 'relE', rel_l2(r[1],o[1]))
ag = (r[0].argmax(-1).cpu() == o[0].argmax(-1))
print(ag[nm].float().mean(), ag.shape, nm.sum())
bad = (~ag & nm).nonzero()
for b,i in bad.tolist()[:5]:
    t = o[0][b,i].topk(2); print(b,i,t.values.tolist(), r[0][b,i].cpu()[t.indices].tolist())
