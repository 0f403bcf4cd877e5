"""How long does the batched symmetric eigendecomposition of the extra features take on the device?"""
import time, torch
D = torch.device('cuda:0')
torch.manual_seed(0)
for B, n in [(64, 300), (16, 140), (8, 64)]:
    A = (torch.rand(B, n, n, device=D) < 0.01).double(); A = torch.triu(A, 1); A = A + A.transpose(1, 2)
    L = torch.diag_embed(A.sum(-1)) - A
    for dt in (torch.float64, torch.float32):
        Ld = L.to(dt)
        for name, fn in (('eigh', torch.linalg.eigh), ('eigvalsh', torch.linalg.eigvalsh)):
            fn(Ld); torch.cuda.synchronize()
            t0 = time.time(); fn(Ld); torch.cuda.synchronize(); dtm = time.time() - t0
            print(f'B={B} n={n} {dt} {name}: {dtm*1e3:.1f} ms', flush=True)
    Af = A.float()
    torch.bmm(Af, Af); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10): k2 = torch.bmm(Af, Af); k3 = torch.bmm(k2, Af)
    torch.cuda.synchronize(); print(f'B={B} n={n} two fp32 bmm: {(time.time()-t0)*100:.3f} ms', flush=True)
Lc = L.cpu()
t0 = time.time(); torch.linalg.eigh(Lc); print(f'CPU eigh fp64 B=8 n=64: {(time.time()-t0)*1e3:.1f} ms')
