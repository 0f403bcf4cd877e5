"""rocSOLVER's strided-batched symmetric eigensolvers called directly (ctypes) vs torch.linalg.eigh on the same batch."""
import ctypes as C, time, torch
D = torch.device('cuda:0')
rs = C.CDLL('/opt/rocm/lib/librocsolver.so'); rb = C.CDLL('/opt/rocm/lib/librocblas.so')
h = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(h)) == 0
rb.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.manual_seed(0)
for B, n in [(64, 300), (16, 140)]:
    A = (torch.rand(B, n, n, device=D) < 0.01).double(); A = torch.triu(A, 1); A = A + A.transpose(1, 2)
    L = (torch.diag_embed(A.sum(-1)) - A).contiguous()
    torch.linalg.eigh(L); torch.cuda.synchronize()
    t0 = time.time(); w_ref, v_ref = torch.linalg.eigh(L); torch.cuda.synchronize(); t_torch = time.time() - t0
    for name in ('rocsolver_dsyevd_strided_batched', 'rocsolver_dsyevj_strided_batched', 'rocsolver_dsyevdj_strided_batched'):
        fn = getattr(rs, name)
        Aw = L.clone(); W = torch.empty(B, n, dtype=torch.float64, device=D); E = torch.empty(B, n, dtype=torch.float64, device=D)
        info = torch.zeros(B, dtype=torch.int32, device=D)
        resid = torch.zeros(B, dtype=torch.float64, device=D); nsw = torch.zeros(B, dtype=torch.int32, device=D)
        def call(Aw):
            p = lambda t: C.c_void_p(t.data_ptr())
            if name.endswith('dsyevd_strided_batched'):
                return fn(h, C.c_int(211), C.c_int(122), C.c_int(n), p(Aw), C.c_int(n), C.c_longlong(n * n), p(W), C.c_longlong(n), p(E), C.c_longlong(n), p(info), C.c_int(B))
            if name.endswith('dsyevj_strided_batched'):   # handle, esort(ascending=242?), evect, uplo, n, A, lda, strideA, abstol, residual, max_sweeps, n_sweeps, W, strideW, info, batch
                return fn(h, C.c_int(252), C.c_int(211), C.c_int(122), C.c_int(n), p(Aw), C.c_int(n), C.c_longlong(n * n), C.c_double(0.0), p(resid), C.c_int(100), p(nsw), p(W), C.c_longlong(n), p(info), C.c_int(B))
            return fn(h, C.c_int(211), C.c_int(122), C.c_int(n), p(Aw), C.c_int(n), C.c_longlong(n * n), p(W), C.c_longlong(n), p(info), C.c_int(B))
        try:
            rc = call(Aw); torch.cuda.synchronize()
            Aw = L.clone(); torch.cuda.synchronize()
            t0 = time.time(); rc = call(Aw); torch.cuda.synchronize(); dt = time.time() - t0
            err = (W - w_ref).abs().max().item()
            # eigenvectors: columns of the column-major result = rows of the row-major tensor
            V = Aw.transpose(1, 2)
            res = (L @ V - V * W.unsqueeze(1)).abs().max().item()
            print(f'B={B} n={n} {name}: rc={rc} {dt*1e3:.2f} ms (torch.linalg.eigh {t_torch*1e3:.2f} ms)  max|w - w_torch| {err:.2e}  residual {res:.2e}  info {int(info.abs().max())}', flush=True)
        except Exception as e:
            print(name, 'failed', e, flush=True)
