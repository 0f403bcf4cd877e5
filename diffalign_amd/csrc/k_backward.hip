// Training path (diffusion_abstract.py:134-225 + training/helpers.py:139-142): the pieces of the fp32 training
// forward / backward that are not plain dense contractions.  The contractions themselves (every nn.Linear, forward and both
// backward products) are library GEMMs under torch.autograd; these kernels are wrapped as torch.autograd.Functions in
// diffalign_amd/autograd_ops.py:
//   k_attn_fwd / k_attn_bwd     the per-pair, per-channel core of NodeEdgeBlock (layers.py:225-265): Y = Q_i K_j / sqrt(df) (E1+1) + E2,
//                               FiLM by y, softmax over keys per channel without key mask, weighted V
//   k_ln_bwd                    LayerNorm backward (normX/E/y 1,2, layers.py:341-359)
//   k_pool_fwd / k_pool_bwd     Xtoy / Etoy pooled statistics [mean, min, max, unbiased std] (layers.py:15-47)
//   k_vb_grad                   d elbo_batch_quick / d logits (elbo.py:273-334, log-space posterior helpers.py:772-781)
//   k_adamw                     AdamW with amsgrad, one fused pass over a flat parameter buffer (utils/setup.py:356)
// Training batches are small (8-16 graphs, n <= 140: align_absorbing.yaml:49-88), so 256-wide per-pair tensors are materialised
// here in fp32 (321 MB each at B = 16, n = 140), as the reference does; everything is HBM-bound streaming.
#include "mfma_util.h"
#include <cmath>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------------------------------------
// NodeEdgeBlock core.  One workgroup per query row (b,i), one thread per channel (strided when dx > blockDim).
// Q, K, V (B,n,dx) already masked; E1 = e_mul(E), E2 = e_add(E) raw linear outputs (B,n,n,dx), masked here by m_i m_j.
// Z = ye1 + (ye2 + 1) Y is the input of e_out; wV the input of the y_x FiLM; (smax, ssum) the softmax statistics kept for backward.
__global__ void __launch_bounds__(256) k_attn_fwd(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                  const float* __restrict__ E1, const float* __restrict__ E2, const uint8_t* __restrict__ nm,
                                                  const float* __restrict__ ye1, const float* __restrict__ ye2, float qscale,
                                                  float* __restrict__ Z, float* __restrict__ wV, float* __restrict__ smax,
                                                  float* __restrict__ ssum, int n, int dx) {
  const int b = blockIdx.y, i = blockIdx.x;
  const size_t row = (size_t)b * n + i;
  const bool mi = nm[row] != 0;
  for (int c = threadIdx.x; c < dx; c += blockDim.x) {
    const float q = Q[row * dx + c], f1 = ye1[(size_t)b * dx + c], f2 = ye2[(size_t)b * dx + c] + 1.f;
    float m = -INFINITY, l = 0.f, a = 0.f;
    for (int j = 0; j < n; ++j) {
      const size_t kj = ((size_t)b * n + j) * dx + c, off = (row * n + j) * dx + c;
      const bool pm = mi && nm[(size_t)b * n + j] != 0;
      const float e1 = pm ? E1[off] : 0.f, e2 = pm ? E2[off] : 0.f;
      const float y = (q * K[kj]) * qscale * (e1 + 1.f) + e2;
      Z[off] = f1 + f2 * y;
      const float mn = fmaxf(m, y), sc = expf(m - mn), p = expf(y - mn);
      l = l * sc + p; a = a * sc + p * V[kj]; m = mn;
    }
    wV[row * dx + c] = a / l; smax[row * dx + c] = m; ssum[row * dx + c] = l;
  }
}

// Backward of the same: inputs dZ (B,n,n,dx), dwV (B,n,dx).  One workgroup per (graph, block of ATT_IB query rows); threads =
// (channel, key slice): slice s walks the keys j = s, s + ATT_JS, ...  Inside the key loop the ATT_IB rows share K_j, V_j and sum
// their dK_j, dV_j contributions in registers, so that one fp32 atomic per (block, key, channel) reaches dK / dV (zero-filled by the
// caller; the only order-dependent sums of the training path); dQ, dye1, dye2 are folded over the slices in LDS, then one atomic per
// block for dye1, dye2 (B,dx).
template <int ATT_IB, int ATT_JS>
__global__ void __launch_bounds__(256 * ATT_JS) k_attn_bwd(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                   const float* __restrict__ E1, const float* __restrict__ E2, const uint8_t* __restrict__ nm,
                                                   const float* __restrict__ ye2, float qscale, const float* __restrict__ wV,
                                                   const float* __restrict__ smax, const float* __restrict__ ssum,
                                                   const float* __restrict__ dZ, const float* __restrict__ dwV, float* __restrict__ dQ,
                                                   float* __restrict__ dK, float* __restrict__ dV, float* __restrict__ dE1,
                                                   float* __restrict__ dE2, float* __restrict__ dye1, float* __restrict__ dye2, int n, int dx) {
  extern __shared__ float s_red[];                                        // [ATT_JS][ATT_IB + 2][CW]
  const int CW = blockDim.x / ATT_JS;                                     // channels per pass
  const int cl = threadIdx.x % CW, js = threadIdx.x / CW;
  const int b = blockIdx.y, i0 = blockIdx.x * ATT_IB;
  const int ni = min(ATT_IB, n - i0);
  for (int c0 = 0; c0 < dx; c0 += CW) {
    const int c = c0 + cl;
    const bool cok = c < dx;
    float dq[ATT_IB], d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int t = 0; t < ATT_IB; ++t) dq[t] = 0.f;
    if (cok) {
      const float f2 = ye2[(size_t)b * dx + c] + 1.f;
      float q[ATT_IB], m[ATT_IB], inv_l[ATT_IB], g[ATT_IB], D[ATT_IB];
      bool mi[ATT_IB];
#pragma unroll
      for (int t = 0; t < ATT_IB; ++t) {
        const size_t row = ((size_t)b * n + min(i0 + t, n - 1)) * dx + c;
        q[t] = Q[row]; m[t] = smax[row]; inv_l[t] = 1.f / ssum[row]; g[t] = dwV[row]; D[t] = g[t] * wV[row];   // D = sum_j attn_j (dwV V_j)
        mi[t] = t < ni && nm[(size_t)b * n + i0 + t] != 0;
      }
      for (int j = js; j < n; j += ATT_JS) {
        const size_t kj = ((size_t)b * n + j) * dx + c;
        const bool mj = nm[(size_t)b * n + j] != 0;
        const float k = K[kj], v = V[kj];
        float dk = 0.f, dv = 0.f;
#pragma unroll
        for (int t = 0; t < ATT_IB; ++t) {
          if (t < ni) {
            const size_t off = (((size_t)b * n + i0 + t) * n + j) * dx + c;
            const bool pm = mi[t] && mj;
            const float e1 = (pm ? E1[off] : 0.f) + 1.f, e2 = pm ? E2[off] : 0.f;
            const float qk = (q[t] * k) * qscale;
            const float y = qk * e1 + e2;
            const float attn = expf(y - m[t]) * inv_l[t];
            const float dz = dZ[off];
            const float dy = dz * f2 + attn * (g[t] * v - D[t]);
            d1 += dz; d2 += dz * y;
            dE2[off] = pm ? dy : 0.f;
            dE1[off] = pm ? dy * qk : 0.f;
            const float dqk = dy * e1 * qscale;
            dq[t] += dqk * k;
            dk += dqk * q[t];
            dv += attn * g[t];
          }
        }
        atomicAdd(dK + kj, dk);
        atomicAdd(dV + kj, dv);
      }
    }
    __syncthreads();                                                      // s_red of the previous channel pass has been read
    float* mine = s_red + (size_t)js * (ATT_IB + 2) * CW + cl;
#pragma unroll
    for (int t = 0; t < ATT_IB; ++t) mine[t * CW] = dq[t];
    mine[ATT_IB * CW] = d1; mine[(ATT_IB + 1) * CW] = d2;
    __syncthreads();
    if (js == 0 && cok) {
      for (int s2 = 1; s2 < ATT_JS; ++s2) {
        const float* o = s_red + (size_t)s2 * (ATT_IB + 2) * CW + cl;
#pragma unroll
        for (int t = 0; t < ATT_IB; ++t) dq[t] += o[t * CW];
        d1 += o[ATT_IB * CW]; d2 += o[(ATT_IB + 1) * CW];
      }
#pragma unroll
      for (int t = 0; t < ATT_IB; ++t)
        if (t < ni) dQ[((size_t)b * n + i0 + t) * dx + c] = dq[t];
      atomicAdd(dye1 + (size_t)b * dx + c, d1);
      atomicAdd(dye2 + (size_t)b * dx + c, d2);
    }
  }
}

// The same two kernels with FOUR channels per thread (16-byte loads and stores: a wave instruction moves 1 KiB instead of 256 B of
// the 321 MB streams, which is what bounds them) for dx % 4 == 0 with dx / 4 dividing 256.  A workgroup of 256 threads = dx / 4 channel
// quads x JS key slices (slice s walks the keys j = s, s + JS, ...); the slices' softmax states (forward) and dQ / dye sums
// (backward) meet in LDS.  Same arithmetic per element; sums over keys are grouped by slice.
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// the per-pair streams (E1, E2, Z and their gradients) in bf16 storage (mixed-precision training: the reference trains under fp16
// autocast, diffusion_abstract.py:181-183): four channels = 8 bytes; arithmetic stays fp32
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return f32x4{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)};
}
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) { *reinterpret_cast<uint2*>(p) = uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])}; }
template <class TE>
__global__ void __launch_bounds__(256) k_attn_fwd4(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                   const TE* __restrict__ E1, const TE* __restrict__ E2, const uint8_t* __restrict__ nm,
                                                   const float* __restrict__ ye1, const float* __restrict__ ye2, float qscale,
                                                   TE* __restrict__ Z, float* __restrict__ wV, float* __restrict__ smax,
                                                   float* __restrict__ ssum, int n, int dx) {
  extern __shared__ float s_st[];                                         // [3][JS][dx]: running max, sum, weighted sum per slice
  const int CW4 = dx >> 2, JS = 256 / CW4;
  const int c = (threadIdx.x % CW4) * 4, js = threadIdx.x / CW4;
  const int b = blockIdx.y, i = blockIdx.x;
  const size_t row = (size_t)b * n + i;
  const bool mi = nm[row] != 0;
  const f32x4 q = ld4(Q + row * dx + c) * qscale, f1 = ld4(ye1 + (size_t)b * dx + c), f2 = ld4(ye2 + (size_t)b * dx + c) + 1.f;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l = {0.f, 0.f, 0.f, 0.f}, a = l;
  // every operand of a key is loaded unconditionally (a masked pair's E1 / E2 are dropped by a select AFTER the load: a load behind an
  // exec-masked branch gets its own s_waitcnt vmcnt(0) and the four streams of a key are fetched one after the other), and the next
  // key's operands are in flight while this one is evaluated
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto off_of = [&](int j) { return (row * n + min(j, n - 1)) * dx + c; };
  auto kj_of = [&](int j) { return ((size_t)b * n + min(j, n - 1)) * dx + c; };
  f32x4 e1n = ld4(E1 + off_of(js)), e2n = ld4(E2 + off_of(js)), kn = ld4(K + kj_of(js)), vn = ld4(V + kj_of(js));
  uint8_t mjn = nm[(size_t)b * n + min(js, n - 1)];
  for (int j = js; j < n; j += JS) {
    const size_t off = (row * n + j) * dx + c;
    const bool pm = mi && mjn != 0;
    const f32x4 e1 = pm ? e1n : zero, e2 = pm ? e2n : zero, v = vn, k = kn;
    e1n = ld4(E1 + off_of(j + JS)); e2n = ld4(E2 + off_of(j + JS)); kn = ld4(K + kj_of(j + JS)); vn = ld4(V + kj_of(j + JS));
    mjn = nm[(size_t)b * n + min(j + JS, n - 1)];
    const f32x4 y = (q * k) * (e1 + 1.f) + e2;
    st4(Z + off, f1 + f2 * y);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float mn = fmaxf(m[t], y[t]), sc = __expf(m[t] - mn), p = __expf(y[t] - mn);
      l[t] = l[t] * sc + p; a[t] = a[t] * sc + p * v[t]; m[t] = mn;
    }
  }
  float* sm = s_st + (size_t)js * dx + c;
  st4(sm, m); st4(sm + (size_t)JS * dx, l); st4(sm + (size_t)2 * JS * dx, a);
  __syncthreads();
  if (js == 0) {
    f32x4 M = m;
    for (int s2 = 1; s2 < JS; ++s2) {
      const f32x4 o = ld4(s_st + (size_t)s2 * dx + c);
#pragma unroll
      for (int t = 0; t < 4; ++t) M[t] = fmaxf(M[t], o[t]);
    }
    f32x4 L = {0.f, 0.f, 0.f, 0.f}, A = L;
    for (int s2 = 0; s2 < JS; ++s2) {
      const float* o = s_st + (size_t)s2 * dx + c;
      const f32x4 ms = ld4(o), ls = ld4(o + (size_t)JS * dx), as = ld4(o + (size_t)2 * JS * dx);
#pragma unroll
      for (int t = 0; t < 4; ++t) { const float w = expf(ms[t] - M[t]); L[t] += ls[t] * w; A[t] += as[t] * w; }
    }
    st4(wV + row * dx + c, A / L); st4(smax + row * dx + c, M); st4(ssum + row * dx + c, L);
  }
}
// dK, dV = sum over the row blocks' partial results (block order: deterministic)
__global__ void __launch_bounds__(256) k_attn_fold(const float* __restrict__ pK, const float* __restrict__ pV, float* __restrict__ dK,
                                                   float* __restrict__ dV, int nblk, long long quads) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= quads) return;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  for (int g = 0; g < nblk; ++g) { a += ld4(pK + ((size_t)g * quads + t) * 4); b += ld4(pV + ((size_t)g * quads + t) * 4); }
  st4(dK + t * 4, a); st4(dV + t * 4, b);
}
template <int IB, class TE>
__global__ void __launch_bounds__(256) k_attn_bwd4(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                   const TE* __restrict__ E1, const TE* __restrict__ E2, const uint8_t* __restrict__ nm,
                                                   const float* __restrict__ ye2, float qscale, const float* __restrict__ wV,
                                                   const float* __restrict__ smax, const float* __restrict__ ssum,
                                                   const TE* __restrict__ dZ, const float* __restrict__ dwV, float* __restrict__ dQ,
                                                   float* __restrict__ dKp, float* __restrict__ dVp, TE* __restrict__ dE1,
                                                   TE* __restrict__ dE2, float* __restrict__ dye1, float* __restrict__ dye2, int n, int dx) {
  // dKp, dVp (gridDim.x, B, n, dx): this row block's contribution to dK, dV, every element written once (k_attn_fold sums the blocks
  // in order).  With fp32 atomics on dK / dV instead, the kernel's time was proportional to the number of atomics (2.25 / 1.19 / 0.74 ms
  // for 1 / 2 / 4 rows per block against ~0.2 ms of streaming).
  extern __shared__ float s_red[];                                        // [JS][IB + 2][dx]
  const int CW4 = dx >> 2, JS = 256 / CW4;
  const int c = (threadIdx.x % CW4) * 4, js = threadIdx.x / CW4;
  const int b = blockIdx.y, i0 = blockIdx.x * IB;
  const int ni = min(IB, n - i0);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 f2 = ld4(ye2 + (size_t)b * dx + c) + 1.f;
  f32x4 q[IB], m[IB], inv_l[IB], g[IB], D[IB], dq[IB], d1 = zero, d2 = zero;
  bool mi[IB];
#pragma unroll
  for (int t = 0; t < IB; ++t) {
    const size_t row = ((size_t)b * n + min(i0 + t, n - 1)) * dx + c;
    q[t] = ld4(Q + row); m[t] = ld4(smax + row); inv_l[t] = 1.f / ld4(ssum + row); g[t] = ld4(dwV + row); D[t] = g[t] * ld4(wV + row);
    dq[t] = zero; mi[t] = t < ni && nm[(size_t)b * n + i0 + t] != 0;
  }
  // The 3 IB row operands of a key are loaded up front and unconditionally (rows past the block clamped, masked pairs dropped by a
  // select after the load): behind exec-masked branches every one of them had its own s_waitcnt vmcnt(0).  With bf16 storage (half the
  // registers per operand) the NEXT key's operands are in flight while this one is evaluated.
  constexpr bool AHEAD = sizeof(TE) == 2;
  f32x4 e1n[IB], e2n[IB], dzn[IB], kn, vn;
  uint8_t mjn = 0;
  auto fetch = [&](int j, f32x4 (&e1)[IB], f32x4 (&e2)[IB], f32x4 (&dz)[IB], f32x4& kk, f32x4& vv, uint8_t& mj) {
    const int jc = min(j, n - 1);
    const size_t kj = ((size_t)b * n + jc) * dx + c;
    kk = ld4(K + kj); vv = ld4(V + kj); mj = nm[(size_t)b * n + jc];
#pragma unroll
    for (int t = 0; t < IB; ++t) {
      const size_t off = (((size_t)b * n + i0 + min(t, ni - 1)) * n + jc) * dx + c;
      e1[t] = ld4(E1 + off); e2[t] = ld4(E2 + off); dz[t] = ld4(dZ + off);
    }
  };
  if constexpr (AHEAD) fetch(js, e1n, e2n, dzn, kn, vn, mjn);
  for (int j = js; j < n; j += JS) {
    f32x4 e1r[IB], e2r[IB], dzr[IB], k, v;
    uint8_t mjb;
    if constexpr (AHEAD) {
#pragma unroll
      for (int t = 0; t < IB; ++t) { e1r[t] = e1n[t]; e2r[t] = e2n[t]; dzr[t] = dzn[t]; }
      k = kn; v = vn; mjb = mjn;
      fetch(j + JS, e1n, e2n, dzn, kn, vn, mjn);
    } else {
      fetch(j, e1r, e2r, dzr, k, v, mjb);
    }
    const bool mj = mjb != 0;
    f32x4 dk = zero, dv = zero;
#pragma unroll
    for (int t = 0; t < IB; ++t) {
      if (t < ni) {
        const size_t off = (((size_t)b * n + i0 + t) * n + j) * dx + c;
        const bool pm = mi[t] && mj;
        const f32x4 e1 = (pm ? e1r[t] : zero) + 1.f, e2 = pm ? e2r[t] : zero, dz = dzr[t];
        const f32x4 qk = (q[t] * k) * qscale;
        const f32x4 y = qk * e1 + e2;
        f32x4 attn;
#pragma unroll
        for (int u = 0; u < 4; ++u) attn[u] = __expf(y[u] - m[t][u]);
        attn *= inv_l[t];
        const f32x4 dy = dz * f2 + attn * (g[t] * v - D[t]);
        d1 += dz; d2 += dz * y;
        st4(dE2 + off, pm ? dy : zero);
        st4(dE1 + off, pm ? dy * qk : zero);
        const f32x4 dqk = dy * e1 * qscale;
        dq[t] += dqk * k;
        dk += dqk * q[t];
        dv += attn * g[t];
      }
    }
    const size_t pj = (((size_t)blockIdx.x * gridDim.y + b) * n + j) * dx + c;
    st4(dKp + pj, dk); st4(dVp + pj, dv);
  }
  float* mine = s_red + (size_t)js * (IB + 2) * dx + c;
#pragma unroll
  for (int t = 0; t < IB; ++t) st4(mine + (size_t)t * dx, dq[t]);
  st4(mine + (size_t)IB * dx, d1); st4(mine + (size_t)(IB + 1) * dx, d2);
  __syncthreads();
  if (js == 0) {
    for (int s2 = 1; s2 < JS; ++s2) {
      const float* o = s_red + (size_t)s2 * (IB + 2) * dx + c;
#pragma unroll
      for (int t = 0; t < IB; ++t) dq[t] += ld4(o + (size_t)t * dx);
      d1 += ld4(o + (size_t)IB * dx); d2 += ld4(o + (size_t)(IB + 1) * dx);
    }
#pragma unroll
    for (int t = 0; t < IB; ++t)
      if (t < ni) st4(dQ + ((size_t)b * n + i0 + t) * dx + c, dq[t]);
#pragma unroll
    for (int u = 0; u < 4; ++u) { atomicAdd(dye1 + (size_t)b * dx + c + u, d1[u]); atomicAdd(dye2 + (size_t)b * dx + c + u, d2[u]); }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm backward, eps 1e-5.  s = the normalised input (a + b of the residual).  A row of D floats is held by D/4 lanes as one
// float4 each (16-byte loads), so a wave covers 256/D rows per pass; grid-stride over rows.  dgamma / dbeta: per-workgroup partial
// sums (pg, pb: (gridDim.x, D)), folded afterwards by k_wgrad_fold -- thousands of atomics on the same 2 D addresses serialise in L2
// (measured: 0.27 ms of a 0.3 ms launch).
template <int D>
__global__ void __launch_bounds__(256) k_ln_bwd(const float* __restrict__ s, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                float* __restrict__ ds, float* __restrict__ pg, float* __restrict__ pb, long long rows) {
  __shared__ float s_g[4][D], s_b[4][D];
  constexpr int LPR = D / 4, RPW = 64 / LPR;                              // lanes per row, rows per wave pass
  const int lane = threadIdx.x & 63, lr = lane % LPR, rw = lane / LPR;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
  const f32x4 gm = {gamma[4 * lr], gamma[4 * lr + 1], gamma[4 * lr + 2], gamma[4 * lr + 3]};   // parameters may sit at any 4-byte offset
  f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
  auto rsum = [](float v) {                                               // over the LPR lanes of one row
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  for (long long r0 = wave * RPW; r0 < rows; r0 += nwave * RPW) {
    const long long r = r0 + rw;
    const bool ok = r < rows;
    f32x4 x = {0.f, 0.f, 0.f, 0.f}, g = x;
    if (ok) { x = *reinterpret_cast<const f32x4*>(s + r * D + 4 * lr); g = *reinterpret_cast<const f32x4*>(dy + r * D + 4 * lr); }
    const float mean = rsum(x[0] + x[1] + x[2] + x[3]) * (1.f / D);
    x -= mean;
    const float rstd = rsqrtf(rsum(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]) * (1.f / D) + 1e-5f);
    x *= rstd;                                                            // xhat
    ag += g * x; ab += g;
    g *= gm;                                                              // dxhat
    const float s1 = rsum(g[0] + g[1] + g[2] + g[3]) * (1.f / D);
    const float s2 = rsum(g[0] * x[0] + g[1] * x[1] + g[2] * x[2] + g[3] * x[3]) * (1.f / D);
    if (ok) *reinterpret_cast<f32x4*>(ds + r * D + 4 * lr) = rstd * (g - s1 - x * s2);
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) { ag[k] += __shfl_xor(ag[k], o); ab[k] += __shfl_xor(ab[k], o); }
  const int wv = threadIdx.x >> 6;
  if (rw == 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_g[wv][4 * lr + k] = ag[k]; s_b[wv][4 * lr + k] = ab[k]; }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    pg[(size_t)blockIdx.x * D + c] = (s_g[0][c] + s_g[1][c]) + (s_g[2][c] + s_g[3][c]);
    pb[(size_t)blockIdx.x * D + c] = (s_b[0][c] + s_b[1][c]) + (s_b[2][c] + s_b[3][c]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Xtoy / Etoy: T (B,N,d) -> z (B,4,d) = [mean, min, max, unbiased std] over the N rows (padding included), arg (B,2,d) = row of the
// min / max (first occurrence).  Stage 1: workgroup = (64 channels, graph, slice of rows), 256 threads = 4 row groups x 64 channels,
// partial (sum, sum of squares) in fp64 and (min, max, their rows) per slice; stage 2 folds the slices in row order.  With one
// slice the two stages are one launch (S == 1 writes the result directly).
struct PoolPart { double s, q; float mn, mx; int imn, imx; };
static_assert(sizeof(PoolPart) == 32, "PoolPart layout");
__device__ __forceinline__ void pool_write(const PoolPart& p, float* z, int* arg, int b, int c, int N, int d) {
  const double mean = p.s / N;
  const double var = N > 1 ? fmax(0.0, (p.q - p.s * mean) / (N - 1)) : 0.0;
  float* o = z + (size_t)b * 4 * d;
  o[c] = (float)mean; o[d + c] = p.mn; o[2 * d + c] = p.mx; o[3 * d + c] = (float)sqrt(var);
  arg[((size_t)b * 2) * d + c] = p.imn; arg[((size_t)b * 2 + 1) * d + c] = p.imx;
}
__global__ void __launch_bounds__(256) k_pool_fwd(const float* __restrict__ T, float* __restrict__ z, int* __restrict__ arg,
                                                  PoolPart* __restrict__ part, int N, int d, int rows_per_slice) {
  __shared__ double s_s[4][64], s_q[4][64];
  __shared__ float s_mn[4][64], s_mx[4][64];
  __shared__ int s_imn[4][64], s_imx[4][64];
  const int b = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.z * rows_per_slice, r1 = min(N, r0 + rows_per_slice);
  PoolPart p{0.0, 0.0, INFINITY, -INFINITY, r0, r0};
  if (c < d)
    for (int r = r0 + rg; r < r1; r += 4) {
      const float v = T[((size_t)b * N + r) * d + c];
      p.s += v; p.q += (double)v * v;
      if (v < p.mn) { p.mn = v; p.imn = r; }
      if (v > p.mx) { p.mx = v; p.imx = r; }
    }
  s_s[rg][cl] = p.s; s_q[rg][cl] = p.q; s_mn[rg][cl] = p.mn; s_mx[rg][cl] = p.mx; s_imn[rg][cl] = p.imn; s_imx[rg][cl] = p.imx;
  __syncthreads();
  if (rg == 0 && c < d) {
    for (int g = 1; g < 4; ++g) {
      p.s += s_s[g][cl]; p.q += s_q[g][cl];
      if (s_mn[g][cl] < p.mn || (s_mn[g][cl] == p.mn && s_imn[g][cl] < p.imn)) { p.mn = s_mn[g][cl]; p.imn = s_imn[g][cl]; }
      if (s_mx[g][cl] > p.mx || (s_mx[g][cl] == p.mx && s_imx[g][cl] < p.imx)) { p.mx = s_mx[g][cl]; p.imx = s_imx[g][cl]; }
    }
    if (gridDim.z == 1) pool_write(p, z, arg, b, c, N, d);
    else part[((size_t)b * gridDim.z + blockIdx.z) * d + c] = p;
  }
}
__global__ void __launch_bounds__(64) k_pool_fold(const PoolPart* __restrict__ part, float* __restrict__ z, int* __restrict__ arg, int S, int N,
                                                 int d) {
  const int b = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= d) return;
  PoolPart p = part[((size_t)b * S) * d + c];
  for (int s = 1; s < S; ++s) {                                           // slices in row order: a strict comparison keeps the first row
    const PoolPart o = part[((size_t)b * S + s) * d + c];
    p.s += o.s; p.q += o.q;
    if (o.mn < p.mn) { p.mn = o.mn; p.imn = o.imn; }
    if (o.mx > p.mx) { p.mx = o.mx; p.imx = o.imx; }
  }
  pool_write(p, z, arg, b, c, N, d);
}
// dT = dmean / N + dstd (x - mean) / ((N-1) std) + [row == argmin] dmin + [row == argmax] dmax   (std = 0: that term is dropped)
__global__ void __launch_bounds__(256) k_pool_bwd(const float* __restrict__ T, const float* __restrict__ z, const int* __restrict__ arg,
                                                  const float* __restrict__ dz, float* __restrict__ dT, int N, int d, long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % d);
  const long long br = t / d;
  const int r = (int)(br % N), b = (int)(br / N);
  const float* zb = z + (size_t)b * 4 * d; const float* gb = dz + (size_t)b * 4 * d;
  const float mean = zb[c], sd = zb[3 * d + c];
  float g = gb[c] / N;
  if (sd > 0.f && N > 1) g += gb[3 * d + c] * (T[t] - mean) / ((float)(N - 1) * sd);
  if (arg[((size_t)b * 2) * d + c] == r) g += gb[d + c];
  if (arg[((size_t)b * 2 + 1) * d + c] == r) g += gb[2 * d + c];
  dT[t] = g;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// d elbo_batch_quick / d logits (elbo.py:273-334), avg_over_batch = True.  Per free row of a graph with t > 1 the term is the
// log-space KL(q || p) of compute_Lt (helpers.py:772-781):  a_k = log(v_k + 1e-30) + log(u_k + 1e-30), u = softmax(logits) Qbar_{t-1},
// v = z_t Q_t^T, logp = a - logsumexp(a);  dKL/da = softmax(a) - q;  dKL/ds_j = sum_k (P_k - q_k) Qsb[j][k] / (u_k + 1e-30);
// dKL/dlogit_i = s_i (g_i - sum_j s_j g_j).  Rows of graphs with t == 1 carry -log p(x_0): gradient softmax - onehot.  Weight w =
// weight / #free rows of the batch.  Not-free rows: zero.
template <int K, bool IS_E>
__global__ void __launch_bounds__(256) k_vb_grad(const float* __restrict__ tr, const float* __restrict__ zt, const float* __restrict__ lg,
                                                 const float* __restrict__ Qt, const float* __restrict__ Qsb, const uint8_t* __restrict__ t_is_1,
                                                 const uint8_t* __restrict__ fm, const float* __restrict__ sums, float scale,
                                                 float* __restrict__ grad, int B, int n) {
  __shared__ float sQ[2 * K * K];
  __shared__ float s_w;
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < K * K; t += blockDim.x) { sQ[t] = Qt[(size_t)b * K * K + t]; sQ[K * K + t] = Qsb[(size_t)b * K * K + t]; }
  if (threadIdx.x == 0) {
    double cnt = 0.0;
    for (int g = 0; g < B; ++g) cnt += sums[g * 4 + (IS_E ? 3 : 1)];
    s_w = cnt > 0.0 ? (float)((double)scale / cnt) : 0.f;
  }
  __syncthreads();
  const float w = s_w;
  const int M = IS_E ? n * n : n;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const bool free_ = IS_E ? (fm[b * n + p / n] != 0 && fm[b * n + p % n] != 0) : fm[b * n + p] != 0;
  const size_t r = ((size_t)b * M + p) * K;
  if (!free_) {
#pragma unroll
    for (int k = 0; k < K; ++k) grad[r + k] = 0.f;
    return;
  }
  float s[K];
  float mx = -INFINITY;
  int x0 = 0, xt = 0; float b0 = tr[r], bt = zt[r];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    s[k] = lg[r + k]; mx = fmaxf(mx, s[k]);
    const float t0 = tr[r + k], t1 = zt[r + k];
    if (t0 > b0) { b0 = t0; x0 = k; }
    if (t1 > bt) { bt = t1; xt = k; }
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { s[k] = expf(s[k] - mx); sum += s[k]; }
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] /= sum;
  if (t_is_1[b]) {
#pragma unroll
    for (int k = 0; k < K; ++k) grad[r + k] = w * (s[k] - (k == x0 ? 1.f : 0.f));
    return;
  }
  const float* Qa = sQ; const float* Qb = sQ + K * K;
  float q[K], a[K], u[K];
  float sq = 0.f, am = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float v = Qa[k * K + xt];
    q[k] = v * Qb[x0 * K + k]; sq += q[k];
    float bp = 0.f;
    for (int j = 0; j < K; ++j) bp += s[j] * Qb[j * K + k];
    u[k] = bp + 1e-30f;
    a[k] = logf(v + 1e-30f) + logf(u[k]);
    am = fmaxf(am, a[k]);
  }
  float sa = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { a[k] = expf(a[k] - am); sa += a[k]; }
#pragma unroll
  for (int k = 0; k < K; ++k) a[k] = (a[k] / sa - q[k] / sq) / u[k];                    // (P_k - q_k) / u_k
  float g[K];
  float sg = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    float t = 0.f;
    for (int k = 0; k < K; ++k) t += a[k] * Qb[j * K + k];
    g[j] = t; sg += s[j] * t;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) grad[r + k] = w * s[k] * (g[k] - sg);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight / bias gradient of a Linear applied to every pair: dW (OUT,IN) = dY^T X, db = column sums of dY, with X (R,IN), dY (R,OUT)
// and R = B n n rows (313,600 at B = 16, n = 140) -- a contraction with a tiny result and a huge reduction axis, which the library
// runs on a handful of workgroups (0.58 - 0.73 ms each, 30 % of the training step: profiles/r02/train_kernel_stats_a.csv).
// Here: split over R.  Each workgroup (4 waves) owns a slice of rows and the whole (OUT,IN) result in accumulators, as 32x32 tiles
// of v_mfma_f32_32x32x2_f32 (true fp32).  For that instruction lane l supplies A[m = l % 32][k = l / 32] and B[k = l / 32][n = l % 32]:
// with m = output feature, n = input feature and k = two consecutive rows, both operands are plain coalesced row reads from HBM
// (no LDS, no transposes).  Stage 2 sums the slices' partial results in slice order (deterministic).
#define WG_U 8                                                            // row pairs per software-pipeline stage
__device__ __forceinline__ float ld1(const float* p) { return *p; }
// bf16 operands stay RAW (the zero-extended 16 bits) in the pipeline registers and are widened where the MFMA consumes them: a
// conversion at the load would pin an s_waitcnt to the load stage and serialise the software pipeline.
__device__ __forceinline__ unsigned ld_raw(const bf16_t* p) { return *reinterpret_cast<const unsigned short*>(p); }
__device__ __forceinline__ float ld_raw(const float* p) { return *p; }
__device__ __forceinline__ float widen(unsigned r) { return __builtin_bit_cast(float, r << 16); }
__device__ __forceinline__ float widen(float r) { return r; }
template <class T> struct RawOf { using type = float; };
template <> struct RawOf<bf16_t> { using type = unsigned; };
template <int IN, int OUT, int WI, int WJ, class TX = float, class TY = float>
__global__ void __launch_bounds__(256) k_wgrad_partial(const TX* __restrict__ X, const TY* __restrict__ dY, float* __restrict__ part,
                                                       float* __restrict__ bpart, long long R, int rows_per_wg) {
  static_assert(WI * WJ == 4 && (IN / 32) % WI == 0 && (OUT / 32) % WJ == 0, "tiling");
  constexpr int TI = IN / 32 / WI, TJ = OUT / 32 / WJ;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l32 = lane & 31, lh = lane >> 5;
  const int wi = w / WJ, wj = w % WJ;
  const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
  f32x16 acc[TJ][TI];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[tj][ti][v] = 0.f;
  float bs[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) bs[tj] = 0.f;
  const TX* xp = X + (wi * TI) * 32 + l32;
  const TY* yp = dY + (wj * TJ) * 32 + l32;
  typename RawOf<TY>::type a[2][WG_U][TJ];
  typename RawOf<TX>::type bb[2][WG_U][TI];
  // rows past the slice are CLAMPED to its last row at the load (no exec-masked branch around a load: that would put a wait inside the
  // load stage) and their dY operand is zeroed where it is consumed, which removes them from dW and db alike
  auto load = [&](int buf, long long r) {
#pragma unroll
    for (int u = 0; u < WG_U; ++u) {
      const long long rr = min(r + 2 * u + lh, r1 - 1);
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) a[buf][u][tj] = ld_raw(yp + rr * OUT + tj * 32);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) bb[buf][u][ti] = ld_raw(xp + rr * IN + ti * 32);
    }
  };
  auto compute = [&](int buf, long long r) {
#pragma unroll
    for (int u = 0; u < WG_U; ++u) {
      const bool ok = r + 2 * u + lh < r1;
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const float av = ok ? widen(a[buf][u][tj]) : 0.f;
        bs[tj] += av;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) acc[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, widen(bb[buf][u][ti]), acc[tj][ti], 0, 0, 0);
      }
    }
  };
  load(0, r0);
  for (long long r = r0; r < r1; r += 4 * WG_U) {                         // two stages per trip: the loads of one run under the MFMAs of the other
    load(1, r + 2 * WG_U);
    __builtin_amdgcn_sched_barrier(0);                                    // keep the stages apart: the scheduler otherwise sinks every load
    compute(0, r);                                                        // to just before its MFMA and waits for it there
    __builtin_amdgcn_sched_barrier(0);
    load(0, r + 4 * WG_U);
    __builtin_amdgcn_sched_barrier(0);
    compute(1, r + 2 * WG_U);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* pw = part + (size_t)blockIdx.x * OUT * IN;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int o = (wj * TJ + tj) * 32 + 8 * (v >> 2) + 4 * lh + (v & 3), i = (wi * TI + ti) * 32 + l32;
        pw[(size_t)o * IN + i] = acc[tj][ti][v];
      }
  if (wi == 0) {
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const float t = bs[tj] + __shfl_xor(bs[tj], 32);
      if (lh == 0) bpart[(size_t)blockIdx.x * OUT + (wj * TJ + tj) * 32 + l32] = t;
    }
  }
}
// Both operands in bf16 storage: the same split over rows on the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16: 16 rows per
// instruction instead of 2, 1/16 of the matrix-pipe time of the fp32 form, which bounds that kernel at 75 us for 64 -> 256).  That
// instruction wants 8 consecutive k per lane, and k is the ROW here while memory is contiguous along the features, so a 32-row x
// 32-feature tile is first loaded the natural way -- lane (row, half) reads 16 contiguous bytes: the A-operand layout with k =
// feature -- and turned by two identity MFMAs into accumulator layout (feature on the lane, 16 rows in registers); packed back to bf16
// (exact: the values were bf16) its two register halves are the k-steps of the product.  dY and X tiles go through the same
// transposition, so the row permutation it leaves (crow) is the same on both operands and the sum over rows does not see it.
// Per 32 rows a wave issues 8 16-byte loads per lane (128 scalar loads in the fp32 form), 8 transposing and 8 multiplying MFMAs.
template <int IN, int OUT, int WI, int WJ>
__global__ void __launch_bounds__(256) k_wgrad_partial_bf16(const bf16_t* __restrict__ X, const bf16_t* __restrict__ dY, float* __restrict__ part,
                                                            float* __restrict__ bpart, long long R, int rows_per_wg) {
  static_assert(WI * WJ == 4 && (IN / 32) % WI == 0 && (OUT / 32) % WJ == 0, "tiling");
  constexpr int TI = IN / 32 / WI, TJ = OUT / 32 / WJ;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, l32 = lane & 31, lh = lane >> 5;
  const int wi = w / WJ, wj = w % WJ;
  const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
  f32x16 acc[TJ][TI];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) acc[tj][ti] = splat16(0.f);
  float bs[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) bs[tj] = 0.f;
  bf16x8 idn[2];                                                          // identity operand of k-step s: element j = (16 s + 8 h + j == n)
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int j = 0; j < 8; ++j) idn[s2][j] = (16 * s2 + 8 * lh + j == l32) ? (short)0x3F80 : (short)0;
  const bf16_t* xp = X + (wi * TI) * 32 + 8 * lh;
  const bf16_t* yp = dY + (wj * TJ) * 32 + 8 * lh;
  uint4 ya[2][TJ][2], xa[2][TI][2];                                       // [buffer][tile][k-step of the transposition]
  auto load = [&](int buf, long long r) {
    const long long rr = min(r + l32, r1 - 1);                            // clamped, zeroed below
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) ya[buf][tj][s2] = *reinterpret_cast<const uint4*>(yp + rr * OUT + tj * 32 + 16 * s2);
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) xa[buf][ti][s2] = *reinterpret_cast<const uint4*>(xp + rr * IN + ti * 32 + 16 * s2);
  };
  auto compute = [&](int buf, long long r) {
    const bool ok = r + l32 < r1;
    const uint4 z4 = {0u, 0u, 0u, 0u};
    bf16x8 yt[TJ][2], xt[TI][2];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      f32x16 t = splat16(0.f);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) t = mfma32(frag_from_u4(ok ? ya[buf][tj][s2] : z4), idn[s2], t);   // rows past the slice: zero in dY
      float sb = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) sb += t[v];
      bs[tj] += sb;
      yt[tj][0] = pack_acc(t, 0); yt[tj][1] = pack_acc(t, 1);
    }
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      f32x16 t = splat16(0.f);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) t = mfma32(frag_from_u4(xa[buf][ti][s2]), idn[s2], t);
      xt[ti][0] = pack_acc(t, 0); xt[ti][1] = pack_acc(t, 1);
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[tj][ti] = mfma32(yt[tj][q], xt[ti][q], acc[tj][ti]);
  };
  load(0, r0);
  for (long long r = r0; r < r1; r += 64) {                               // two 32-row stages per trip, the loads of one under the MFMAs of the other
    load(1, r + 32);
    __builtin_amdgcn_sched_barrier(0);
    compute(0, r);
    __builtin_amdgcn_sched_barrier(0);
    load(0, r + 64);
    __builtin_amdgcn_sched_barrier(0);
    if (r + 32 < r1) compute(1, r + 32);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* pw = part + (size_t)blockIdx.x * OUT * IN;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int o = (wj * TJ + tj) * 32 + crow(v, lh), i = (wi * TI + ti) * 32 + l32;
        pw[(size_t)o * IN + i] = acc[tj][ti][v];
      }
  if (wi == 0) {                                                          // column sums of dY: lane = feature, both halves hold half of the rows
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const float t = bs[tj] + __shfl_xor(bs[tj], 32);
      if (lh == 0) bpart[(size_t)blockIdx.x * OUT + (wj * TJ + tj) * 32 + l32] = t;
    }
  }
}
// Fold of the G partial results (bias columns appended: partial g = nW weights then, in a second array, nB biases): workgroup = 32
// outputs x 8 slice groups; a thread sums its group's partials in slice order, the 8 groups are then added in group order (LDS).
__global__ void __launch_bounds__(256) k_wgrad_fold(const float* __restrict__ part, const float* __restrict__ bpart, float* __restrict__ dW,
                                                    float* __restrict__ db, int G, int nW, int nB) {
  __shared__ float s_p[8][32];
  const int ol = threadIdx.x & 31, gg = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + ol;
  const int per = (G + 7) / 8, g0 = gg * per, g1 = min(G, g0 + per);
  const bool isW = t < nW, isB = !isW && t < nW + nB && db != nullptr;
  const float* src = isW ? part + t : bpart + (t - nW);
  const size_t stride = isW ? (size_t)nW : (size_t)nB;
  float sum = 0.f;
  if (isW || isB)
    for (int g = g0; g < g1; g += 16) {                 // sixteen partials in flight, added in slice order (same sum, bit for bit)
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = g + q < g1 ? src[(size_t)(g + q) * stride] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) if (g + q < g1) sum += v[q];
    }
  s_p[gg][ol] = sum;
  __syncthreads();
  if (gg == 0 && (isW || isB)) {
    for (int k = 1; k < 8; ++k) sum += s_p[k][ol];
    if (isW) dW[t] = sum; else db[t - nW] = sum;
  }
}
#define WGRAD_MAX_SLICES 512
template <int IN, int OUT, int WI, int WJ, class TX = float, class TY = float>
static void launch_wgrad(const TX* X, const TY* dY, float* dW, float* db, float* ws, long long R, hipStream_t s) {
  long long per = (R + WGRAD_MAX_SLICES - 1) / WGRAD_MAX_SLICES;
  per = (per + 1) & ~1LL;                                                 // rows are consumed in pairs
  if (per < 64) per = 64;
  const int G = (int)((R + per - 1) / per);
  float* bpart = ws + (size_t)G * OUT * IN;
  k_wgrad_partial<IN, OUT, WI, WJ, TX, TY><<<G, 256, 0, s>>>(X, dY, ws, bpart, R, (int)per);
  k_wgrad_fold<<<(IN * OUT + OUT + 31) / 32, 256, 0, s>>>(ws, bpart, dW, db, G, IN * OUT, OUT);
}

template <int IN, int OUT, int WI, int WJ>
static void launch_wgrad_bf16(const bf16_t* X, const bf16_t* dY, float* dW, float* db, float* ws, long long R, hipStream_t s) {
  long long per = (R + WGRAD_MAX_SLICES - 1) / WGRAD_MAX_SLICES;
  per = (per + 31) & ~31LL;                                               // rows are consumed 32 at a time
  if (per < 64) per = 64;
  const int G = (int)((R + per - 1) / per);
  float* bpart = ws + (size_t)G * OUT * IN;
  k_wgrad_partial_bf16<IN, OUT, WI, WJ><<<G, 256, 0, s>>>(X, dY, ws, bpart, R, (int)per);
  k_wgrad_fold<<<(IN * OUT + OUT + 31) / 32, 256, 0, s>>>(ws, bpart, dW, db, G, IN * OUT, OUT);
}
static int wgrad_bf16(const bf16_t* X, const bf16_t* dY, float* dW, float* db, float* ws, long long R, int in_features, int out_features, hipStream_t s) {
  switch (in_features * 1000 + out_features) {
    case 64256: launch_wgrad_bf16<64, 256, 1, 4>(X, dY, dW, db, ws, R, s); break;
    case 256064: launch_wgrad_bf16<256, 64, 4, 1>(X, dY, dW, db, ws, R, s); break;
    case 64128: launch_wgrad_bf16<64, 128, 1, 4>(X, dY, dW, db, ws, R, s); break;
    case 128064: launch_wgrad_bf16<128, 64, 4, 1>(X, dY, dW, db, ws, R, s); break;
    case 64064: launch_wgrad_bf16<64, 64, 2, 2>(X, dY, dW, db, ws, R, s); break;
    case 128128: launch_wgrad_bf16<128, 128, 2, 2>(X, dY, dW, db, ws, R, s); break;
    default: da_set_error("da_linear_wgrad_mixed: %d -> %d features not built", in_features, out_features); return DA_ERR_UNSUPPORTED;
  }
  return DA_OK;
}

template <class TX, class TY>
static int wgrad_typed(const TX* X, const TY* dY, float* dW, float* db, float* ws, long long R, int in_features, int out_features, hipStream_t s) {
  switch (in_features * 1000 + out_features) {
    case 64256: launch_wgrad<64, 256, 1, 4, TX, TY>(X, dY, dW, db, ws, R, s); break;
    case 256064: launch_wgrad<256, 64, 4, 1, TX, TY>(X, dY, dW, db, ws, R, s); break;
    case 64128: launch_wgrad<64, 128, 1, 4, TX, TY>(X, dY, dW, db, ws, R, s); break;
    case 128064: launch_wgrad<128, 64, 4, 1, TX, TY>(X, dY, dW, db, ws, R, s); break;
    case 64064: launch_wgrad<64, 64, 2, 2, TX, TY>(X, dY, dW, db, ws, R, s); break;
    case 128128: launch_wgrad<128, 128, 2, 2, TX, TY>(X, dY, dW, db, ws, R, s); break;
    default: da_set_error("da_linear_wgrad_mixed: %d -> %d features not built", in_features, out_features); return DA_ERR_UNSUPPORTED;
  }
  return DA_OK;
}
// ---------------------------------------------------------------------------------------------------------------------------------
// AdamW(amsgrad) on a flat fp32 buffer (torch.optim.AdamW semantics: decoupled weight decay, bias corrections by step).
__global__ void __launch_bounds__(256) k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                               float* __restrict__ vmax, long long N, float lr, float b1, float b2, float eps, float wd,
                                               float bc1, float bc2_sqrt, float grad_scale, const uint8_t* __restrict__ slot_active) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  if (slot_active && !slot_active[t >> 6]) return;      // a parameter without a gradient this step: torch skips it (no decay, state frozen)
  const float gr = g[t] * grad_scale;
  float x = p[t];
  x *= 1.f - lr * wd;
  const float mm = b1 * m[t] + (1.f - b1) * gr, vv = b2 * v[t] + (1.f - b2) * gr * gr;
  m[t] = mm; v[t] = vv;
  float den;
  if (vmax) { const float vm = fmaxf(vmax[t], vv); vmax[t] = vm; den = sqrtf(vm) / bc2_sqrt + eps; }
  else den = sqrtf(vv) / bc2_sqrt + eps;
  p[t] = x - (lr / bc1) * mm / den;
}

// The same update with the step number in device memory (*step holds the number of updates done so far): the bias corrections are
// computed on the device, so the launch carries no per-step host value and can be replayed from a HIP graph.  k_adamw_tick follows it.
__global__ void __launch_bounds__(256) k_adamw_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, long long N, float lr, float b1, float b2, float eps, float wd,
                                                   const int* __restrict__ step, float grad_scale, const uint8_t* __restrict__ slot_active) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  if (slot_active && !slot_active[t >> 6]) return;
  const float st = (float)(*step + 1);
  const float bc1 = 1.f - powf(b1, st), bc2_sqrt = sqrtf(1.f - powf(b2, st));
  const float gr = g[t] * grad_scale;
  float x = p[t];
  x *= 1.f - lr * wd;
  const float mm = b1 * m[t] + (1.f - b1) * gr, vv = b2 * v[t] + (1.f - b2) * gr * gr;
  m[t] = mm; v[t] = vv;
  float den;
  if (vmax) { const float vm = fmaxf(vmax[t], vv); vmax[t] = vm; den = sqrtf(vm) / bc2_sqrt + eps; }
  else den = sqrtf(vv) / bc2_sqrt + eps;
  p[t] = x - (lr / bc1) * mm / den;
}
__global__ void k_adamw_tick(int* step) { *step += 1; }

static inline unsigned grid_for(long long n, int threads) { return (unsigned)((n + threads - 1) / threads); }

template <int K, bool IS_E>
static void launch_vb(const float* tr, const float* zt, const float* lg, const float* Qt, const float* Qsb, const uint8_t* t1, const uint8_t* fm,
                      const float* sums, float scale, float* grad, int B, int n, hipStream_t s) {
  const int M = IS_E ? n * n : n;
  k_vb_grad<K, IS_E><<<dim3((M + 255) / 256, B), 256, 0, s>>>(tr, zt, lg, Qt, Qsb, t1, fm, sums, scale, grad, B, n);
}
#define VB_DISPATCH(K_, IS_E_, ...)                                                                            \
  switch (K_) {                                                                                                \
    case 4: launch_vb<4, IS_E_>(__VA_ARGS__); break;   case 5: launch_vb<5, IS_E_>(__VA_ARGS__); break;         \
    case 7: launch_vb<7, IS_E_>(__VA_ARGS__); break;   case 29: launch_vb<29, IS_E_>(__VA_ARGS__); break;       \
    default: da_set_error("unsupported class count %d (built for 4,5,7,29)", K_); return DA_ERR_UNSUPPORTED;   \
  }

extern "C" {

int da_attn_core_fwd(const float* Q, const float* K, const float* V, const float* E1, const float* E2, const uint8_t* node_mask,
                     const float* ye1, const float* ye2, float qscale, float* Z, float* wV, float* smax, float* ssum, int B, int n, int dx,
                     void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && dx >= 1, "da_attn_core_fwd: sizes");
  if (B == 0 || n == 0) return DA_OK;
  const bool quad = dx % 4 == 0 && dx / 4 <= 256 && 256 % (dx / 4) == 0 && getenv("DA_ATTN_SCALAR") == nullptr &&
                    ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)E1 | (uintptr_t)E2 | (uintptr_t)ye1 | (uintptr_t)ye2 | (uintptr_t)Z |
                       (uintptr_t)wV | (uintptr_t)smax | (uintptr_t)ssum) & 15) == 0);
  if (quad) {
    const int JS = 256 / (dx / 4);
    k_attn_fwd4<float><<<dim3(n, B), 256, sizeof(float) * 3 * JS * dx, (hipStream_t)stream>>>(Q, K, V, E1, E2, node_mask, ye1, ye2, qscale, Z, wV, smax, ssum, n, dx);
    DA_CHECK_LAUNCH("da_attn_core_fwd");
    return DA_OK;
  }
  k_attn_fwd<<<dim3(n, B), dx < 256 ? ((dx + 63) / 64) * 64 : 256, 0, (hipStream_t)stream>>>(Q, K, V, E1, E2, node_mask, ye1, ye2, qscale, Z, wV,
                                                                                         smax, ssum, n, dx);
  DA_CHECK_LAUNCH("da_attn_core_fwd");
  return DA_OK;
}

// Workspace of da_attn_core_bwd's fast path: per-row-block partial dK, dV.
long long da_attn_core_bwd_workspace(int B, int n, int dx) {
  if (B < 0 || n < 0 || dx < 1) return -1;
  return (long long)sizeof(float) * 2 * ((n + 3) / 4) * B * n * dx;
}
int da_attn_core_bwd(const float* Q, const float* K, const float* V, const float* E1, const float* E2, const uint8_t* node_mask,
                     const float* ye2, float qscale, const float* wV, const float* smax, const float* ssum, const float* dZ, const float* dwV,
                     float* dQ, float* dK, float* dV, float* dE1, float* dE2, float* dye1, float* dye2, void* workspace, long long workspace_bytes,
                     int B, int n, int dx, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && dx >= 1, "da_attn_core_bwd: sizes");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t nb = sizeof(float) * (size_t)B * n * dx, yb = sizeof(float) * (size_t)B * dx;
  if (hipMemsetAsync(dye1, 0, yb, s) != hipSuccess || hipMemsetAsync(dye2, 0, yb, s) != hipSuccess) {
    da_set_error("da_attn_core_bwd: memset failed"); return DA_ERR_HIP;
  }
  static int variant = -1;
  if (variant < 0) { const char* e = getenv("DA_ATTN_BWD"); variant = e ? atoi(e) : 2; }
  constexpr int QIB = 4;                                                    // query rows per workgroup of the quad kernel
  const long long need = (long long)sizeof(float) * 2 * ((n + QIB - 1) / QIB) * B * n * dx;
  const bool quad = variant < 10 && dx % 4 == 0 && dx / 4 <= 256 && 256 % (dx / 4) == 0 && workspace && workspace_bytes >= need &&
                    (((uintptr_t)workspace | (uintptr_t)dK | (uintptr_t)dV) & 15) == 0 &&
                    ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)E1 | (uintptr_t)E2 | (uintptr_t)ye2 | (uintptr_t)wV | (uintptr_t)smax |
                       (uintptr_t)ssum | (uintptr_t)dZ | (uintptr_t)dwV | (uintptr_t)dQ | (uintptr_t)dE1 | (uintptr_t)dE2) & 15) == 0);
  if (quad) {                                                               // dK, dV: per-block partials + ordered fold, no atomics
    const int JS = 256 / (dx / 4), nblk = (n + QIB - 1) / QIB;
    float* pK = (float*)workspace; float* pV = pK + (size_t)nblk * B * n * dx;
    k_attn_bwd4<QIB, float><<<dim3(nblk, B), 256, sizeof(float) * JS * (QIB + 2) * dx, s>>>(Q, K, V, E1, E2, node_mask, ye2, qscale, wV, smax, ssum, dZ, dwV, dQ,
                                                                                 pK, pV, dE1, dE2, dye1, dye2, n, dx);
    const long long quads = (long long)B * n * dx / 4;
    k_attn_fold<<<grid_for(quads, 256), 256, 0, s>>>(pK, pV, dK, dV, nblk, quads);
    DA_CHECK_LAUNCH("da_attn_core_bwd");
    return DA_OK;
  }
  if (hipMemsetAsync(dK, 0, nb, s) != hipSuccess || hipMemsetAsync(dV, 0, nb, s) != hipSuccess) { da_set_error("da_attn_core_bwd: memset failed"); return DA_ERR_HIP; }
  const int cw = dx < 256 ? ((dx + 63) / 64) * 64 : 256;
  const int scalar_variant = variant < 10 ? 42 : variant;                                                // DA_ATTN_BWD=<rows per block><key slices>, e.g. 42 (tuning only)
#define ATT_LAUNCH(IB_, JS_)                                                                                                      \
  k_attn_bwd<IB_, JS_><<<dim3((n + IB_ - 1) / IB_, B), cw * JS_, sizeof(float) * JS_ * (IB_ + 2) * cw, s>>>(                       \
      Q, K, V, E1, E2, node_mask, ye2, qscale, wV, smax, ssum, dZ, dwV, dQ, dK, dV, dE1, dE2, dye1, dye2, n, dx)
  switch (scalar_variant) {
    case 11: ATT_LAUNCH(1, 1); break;  case 12: ATT_LAUNCH(1, 2); break;  case 21: ATT_LAUNCH(2, 1); break;  case 22: ATT_LAUNCH(2, 2); break;
    case 24: ATT_LAUNCH(2, 4); break;  case 41: ATT_LAUNCH(4, 1); break;  case 44: ATT_LAUNCH(4, 4); break;  case 82: ATT_LAUNCH(8, 2); break;
    case 84: ATT_LAUNCH(8, 4); break;  default: ATT_LAUNCH(4, 2); break;
  }
#undef ATT_LAUNCH
  DA_CHECK_LAUNCH("da_attn_core_bwd");
  return DA_OK;
}

// The attention core with the per-pair streams in bf16 storage (E1, E2, Z / dZ, dE1, dE2); everything per node and per graph stays
// fp32, as does the arithmetic.  dx % 4 == 0 with dx / 4 dividing 256 (the quad kernels); pair streams 8-byte aligned.
static bool quad_ok(int dx) { return dx % 4 == 0 && dx / 4 <= 256 && 256 % (dx / 4) == 0; }
int da_attn_core_fwd_bf16(const float* Q, const float* K, const float* V, const void* E1, const void* E2, const uint8_t* node_mask,
                          const float* ye1, const float* ye2, float qscale, void* Z, float* wV, float* smax, float* ssum, int B, int n, int dx,
                          void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && quad_ok(dx), "da_attn_core_fwd_bf16: dx = %d must be a multiple of 4 with dx / 4 dividing 256", dx);
  DA_CHECK_ARG(((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)ye1 | (uintptr_t)ye2 | (uintptr_t)wV | (uintptr_t)smax | (uintptr_t)ssum) & 15) == 0) &&
               ((((uintptr_t)E1 | (uintptr_t)E2 | (uintptr_t)Z) & 7) == 0), "da_attn_core_fwd_bf16: unaligned tensor");
  if (B == 0 || n == 0) return DA_OK;
  const int JS = 256 / (dx / 4);
  k_attn_fwd4<bf16_t><<<dim3(n, B), 256, sizeof(float) * 3 * JS * dx, (hipStream_t)stream>>>(Q, K, V, (const bf16_t*)E1, (const bf16_t*)E2, node_mask, ye1,
                                                                                      ye2, qscale, (bf16_t*)Z, wV, smax, ssum, n, dx);
  DA_CHECK_LAUNCH("da_attn_core_fwd_bf16");
  return DA_OK;
}
int da_attn_core_bwd_bf16(const float* Q, const float* K, const float* V, const void* E1, const void* E2, const uint8_t* node_mask,
                          const float* ye2, float qscale, const float* wV, const float* smax, const float* ssum, const void* dZ, const float* dwV,
                          float* dQ, float* dK, float* dV, void* dE1, void* dE2, float* dye1, float* dye2, void* workspace, long long workspace_bytes,
                          int B, int n, int dx, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && quad_ok(dx), "da_attn_core_bwd_bf16: dx = %d must be a multiple of 4 with dx / 4 dividing 256", dx);
  if (B == 0 || n == 0) return DA_OK;
  constexpr int QIB = 4;
  DA_CHECK_ARG(workspace && workspace_bytes >= da_attn_core_bwd_workspace(B, n, dx), "da_attn_core_bwd_bf16: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t yb = sizeof(float) * (size_t)B * dx;
  if (hipMemsetAsync(dye1, 0, yb, s) != hipSuccess || hipMemsetAsync(dye2, 0, yb, s) != hipSuccess) { da_set_error("da_attn_core_bwd_bf16: memset failed"); return DA_ERR_HIP; }
  const int JS = 256 / (dx / 4), nblk = (n + QIB - 1) / QIB;
  float* pK = (float*)workspace; float* pV = pK + (size_t)nblk * B * n * dx;
  k_attn_bwd4<QIB, bf16_t><<<dim3(nblk, B), 256, sizeof(float) * JS * (QIB + 2) * dx, s>>>(Q, K, V, (const bf16_t*)E1, (const bf16_t*)E2, node_mask, ye2, qscale,
                                                                                      wV, smax, ssum, (const bf16_t*)dZ, dwV, dQ, pK, pV, (bf16_t*)dE1,
                                                                                      (bf16_t*)dE2, dye1, dye2, n, dx);
  const long long quads = (long long)B * n * dx / 4;
  k_attn_fold<<<grid_for(quads, 256), 256, 0, s>>>(pK, pV, dK, dV, nblk, quads);
  DA_CHECK_LAUNCH("da_attn_core_bwd_bf16");
  return DA_OK;
}

#define LN_BWD_MAX_WG 1024
int da_layernorm_bwd(const float* s, const float* gamma, const float* dy, float* ds, float* dgamma, float* dbeta, void* workspace,
                     long long workspace_bytes, long long rows, int D, void* stream) {
  DA_CHECK_ARG(rows >= 0, "da_layernorm_bwd: rows");
  DA_CHECK_ARG((((uintptr_t)s | (uintptr_t)dy | (uintptr_t)ds) & 15) == 0, "da_layernorm_bwd: s, dy, ds must be 16-byte aligned");
  DA_CHECK_ARG(workspace && workspace_bytes >= (long long)sizeof(float) * 2 * LN_BWD_MAX_WG * D, "da_layernorm_bwd: workspace needs 8 * 1024 * D bytes");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (hipMemsetAsync(dgamma, 0, sizeof(float) * D, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * D, st) != hipSuccess) {
      da_set_error("da_layernorm_bwd: memset failed"); return DA_ERR_HIP;
    }
    return DA_OK;
  }
  long long g = (rows * D / 256 + 3) / 4; if (g < 1) g = 1; if (g > LN_BWD_MAX_WG) g = LN_BWD_MAX_WG;   // a wave pass covers 256 floats
  float* pg = (float*)workspace; float* pb = pg + (size_t)g * D;
  switch (D) {
    case 32: k_ln_bwd<32><<<(unsigned)g, 256, 0, st>>>(s, gamma, dy, ds, pg, pb, rows); break;
    case 64: k_ln_bwd<64><<<(unsigned)g, 256, 0, st>>>(s, gamma, dy, ds, pg, pb, rows); break;
    case 128: k_ln_bwd<128><<<(unsigned)g, 256, 0, st>>>(s, gamma, dy, ds, pg, pb, rows); break;
    case 256: k_ln_bwd<256><<<(unsigned)g, 256, 0, st>>>(s, gamma, dy, ds, pg, pb, rows); break;
    default: da_set_error("da_layernorm_bwd: width %d not built (32, 64, 128, 256)", D); return DA_ERR_UNSUPPORTED;
  }
  k_wgrad_fold<<<(2 * D + 31) / 32, 256, 0, st>>>(pg, pb, dgamma, dbeta, (int)g, D, D);
  DA_CHECK_LAUNCH("da_layernorm_bwd");
  return DA_OK;
}

int da_pool_stats_fwd(const float* T, float* z, int* arg, void* workspace, long long workspace_bytes, int B, int N, int d, void* stream) {
  DA_CHECK_ARG(B >= 0 && N >= 1 && d >= 1, "da_pool_stats_fwd: sizes");
  if (B == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  int S = (N + 255) / 256;                                                // >= 256 rows per slice
  if (S > 64) S = 64;
  if (!workspace || workspace_bytes < (long long)sizeof(PoolPart) * B * S * d) S = 1;   // no room for partials: one slice per graph
  const int per = (N + S - 1) / S;
  S = (N + per - 1) / per;
  k_pool_fwd<<<dim3((d + 63) / 64, B, S), 256, 0, s>>>(T, z, arg, (PoolPart*)workspace, N, d, per);
  if (S > 1) k_pool_fold<<<dim3((d + 63) / 64, B), 64, 0, s>>>((const PoolPart*)workspace, z, arg, S, N, d);
  DA_CHECK_LAUNCH("da_pool_stats_fwd");
  return DA_OK;
}
int da_pool_stats_bwd(const float* T, const float* z, const int* arg, const float* dz, float* dT, int B, int N, int d, void* stream) {
  DA_CHECK_ARG(B >= 0 && N >= 1 && d >= 1, "da_pool_stats_bwd: sizes");
  if (B == 0) return DA_OK;
  const long long total = (long long)B * N * d;
  k_pool_bwd<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(T, z, arg, dz, dT, N, d, total);
  DA_CHECK_LAUNCH("da_pool_stats_bwd");
  return DA_OK;
}

long long da_linear_wgrad_workspace(long long R, int in_features, int out_features) {
  (void)R;
  return (long long)sizeof(float) * WGRAD_MAX_SLICES * ((long long)in_features * out_features + out_features);
}
int da_linear_wgrad(const float* X, const float* dY, float* dW, float* db, void* workspace, long long workspace_bytes, long long R,
                    int in_features, int out_features, void* stream) {
  DA_CHECK_ARG(R >= 1 && X && dY && dW && workspace, "da_linear_wgrad: arguments");
  DA_CHECK_ARG(workspace_bytes >= da_linear_wgrad_workspace(R, in_features, out_features), "da_linear_wgrad: workspace of %lld bytes is too small",
               workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int key = in_features * 1000 + out_features;
  switch (key) {
    case 64256: launch_wgrad<64, 256, 1, 4>(X, dY, dW, db, ws, R, s); break;      // e_mul, e_add
    case 256064: launch_wgrad<256, 64, 4, 1>(X, dY, dW, db, ws, R, s); break;     // e_out
    case 64128: launch_wgrad<64, 128, 1, 4>(X, dY, dW, db, ws, R, s); break;      // linE1, mlp_out_E.0
    case 128064: launch_wgrad<128, 64, 4, 1>(X, dY, dW, db, ws, R, s); break;     // linE2, mlp_in_E.2
    case 64064: launch_wgrad<64, 64, 2, 2>(X, dY, dW, db, ws, R, s); break;
    case 128128: launch_wgrad<128, 128, 2, 2>(X, dY, dW, db, ws, R, s); break;
    default: da_set_error("da_linear_wgrad: %d -> %d features not built (64->256, 256->64, 64->128, 128->64, 64->64, 128->128)", in_features,
                          out_features); return DA_ERR_UNSUPPORTED;
  }
  DA_CHECK_LAUNCH("da_linear_wgrad");
  return DA_OK;
}

// da_linear_wgrad with X and / or dY in bf16 storage (x_bf16 / dy_bf16 != 0); the products and the reduction stay fp32 (the elements
// are widened on load and go through the same v_mfma_f32_32x32x2_f32 tiles): what changes is the bytes of the (R, 256) stream.
int da_linear_wgrad_mixed(const void* X, int x_bf16, const void* dY, int dy_bf16, float* dW, float* db, void* workspace, long long workspace_bytes,
                          long long R, int in_features, int out_features, void* stream) {
  DA_CHECK_ARG(R >= 1 && X && dY && dW && workspace, "da_linear_wgrad_mixed: arguments");
  DA_CHECK_ARG(workspace_bytes >= da_linear_wgrad_workspace(R, in_features, out_features), "da_linear_wgrad_mixed: workspace of %lld bytes is too small",
               workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  int rc;
  if (x_bf16 && dy_bf16 && (((uintptr_t)X | (uintptr_t)dY) & 15) != 0) { da_set_error("da_linear_wgrad_mixed: bf16 operands must be 16-byte aligned"); return DA_ERR_ARG; }
  if (x_bf16 && dy_bf16) rc = wgrad_bf16((const bf16_t*)X, (const bf16_t*)dY, dW, db, ws, R, in_features, out_features, s);
  else if (x_bf16) rc = wgrad_typed((const bf16_t*)X, (const float*)dY, dW, db, ws, R, in_features, out_features, s);
  else if (dy_bf16) rc = wgrad_typed((const float*)X, (const bf16_t*)dY, dW, db, ws, R, in_features, out_features, s);
  else rc = wgrad_typed((const float*)X, (const float*)dY, dW, db, ws, R, in_features, out_features, s);
  if (rc) return rc;
  DA_CHECK_LAUNCH("da_linear_wgrad_mixed");
  return DA_OK;
}

int da_vb_grad(const float* true_X, const float* true_E, const float* zt_X, const float* zt_E, const float* logits_X, const float* logits_E,
               const float* Qt_x, const float* Qsb_x, const float* Qt_e, const float* Qsb_e, const uint8_t* t_is_1, const uint8_t* mask_x,
               const uint8_t* mask_e, const float* sums, float weight_X, float weight_E, float* grad_X, float* grad_E, int B, int n, int kx,
               int ke, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && sums && mask_x && mask_e && t_is_1, "da_vb_grad: arguments");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  if (grad_X) { VB_DISPATCH(kx, false, true_X, zt_X, logits_X, Qt_x, Qsb_x, t_is_1, mask_x, sums, weight_X, grad_X, B, n, s) }
  if (grad_E) { VB_DISPATCH(ke, true, true_E, zt_E, logits_E, Qt_e, Qsb_e, t_is_1, mask_e, sums, weight_E, grad_E, B, n, s) }
  DA_CHECK_LAUNCH("da_vb_grad");
  return DA_OK;
}

int da_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long long N, float lr, float beta1,
             float beta2, float eps, float weight_decay, int step, float grad_scale, const uint8_t* slot_active, void* stream) {
  DA_CHECK_ARG(N >= 0 && step >= 1 && param && grad && exp_avg && exp_avg_sq, "da_adamw: arguments");
  if (N == 0) return DA_OK;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = sqrtf(1.f - powf(beta2, (float)step));
  k_adamw<<<grid_for(N, 256), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, N, lr, beta1, beta2, eps, weight_decay,
                                                          bc1, bc2, grad_scale, slot_active);
  DA_CHECK_LAUNCH("da_adamw");
  return DA_OK;
}

// da_adamw with the step counter in device memory (int, starts at 0, incremented here): for optimiser steps replayed from a HIP graph.
int da_adamw_devstep(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long long N, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int* step_counter, float grad_scale, const uint8_t* slot_active,
                     void* stream) {
  DA_CHECK_ARG(N >= 0 && step_counter && param && grad && exp_avg && exp_avg_sq, "da_adamw_devstep: arguments");
  if (N > 0)
    k_adamw_dev<<<grid_for(N, 256), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, N, lr, beta1, beta2, eps,
                                                                  weight_decay, step_counter, grad_scale, slot_active);
  k_adamw_tick<<<1, 1, 0, (hipStream_t)stream>>>(step_counter);
  DA_CHECK_LAUNCH("da_adamw_devstep");
  return DA_OK;
}

}  // extern "C"
