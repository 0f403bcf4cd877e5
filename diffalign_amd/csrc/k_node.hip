// Node-side (X) and global (y) kernels of the denoiser.  The X path is a handful of small dense contractions
// (B*n rows x 256) -> MFMA bf16 with fp32 accumulate; the y path is per-graph vector work -> fp32 VALU.
#include "mfma_util.h"
#include <climits>

// ---------------------------------------------------------------------------------------------------
// out[M,N] = act(A[M,K] @ W[N,K]^T + bias) (* rowmask).  A fp32 (lda), W bf16 row-major [Npad][Kpad] (zero padded to
// multiples of 32 / 16 by diffalign_amd/pack.py), out fp32 (ldc).  One wave = 32 rows x (32*NTILE) columns; fragments
// are loaded straight from global/L2 (the operands are small and hot), no LDS.
// W is pre-packed in MFMA B-fragment order (diffalign_amd/pack.py bf16_weight): fragment (nt, ks) = 64 lanes x 8 bf16,
// lane (r,h) element j = W[32nt + r][16ks + 8h + j]; fragments are stored [nt][ks], 1 KiB each, so a wave's operand
// read is one coalesced 1 KiB line set.  KSMAX*16 >= Kpad: the wave's 32 activation rows are converted to bf16
// A fragments once and stay in registers while it walks its NTILE output tiles.
// The loops run KSMAX k-steps without branches (steps >= KS multiply a zero activation fragment by a clamped, valid
// weight fragment): with a branch per step every load waits for the previous one (no hoisting across basic blocks).
// VEC: rows are 16-byte aligned and K is a multiple of 8 (float4 loads); otherwise guarded scalar loads.
// One wave owns 32 rows and walks `ntiles` 32-column output tiles (blockIdx.y selects the column group); the weight
// fragments of the next tile are fetched while the current tile's MFMAs and stores run (two register buffers).
template <int KSMAX, bool RELU, bool VEC>
__global__ void __launch_bounds__(256, 2) k_linear(const float* __restrict__ A, int lda, const uint4* __restrict__ W, int KS,
                                                   const float* __restrict__ bias, const uint8_t* __restrict__ rowmask,
                                                   float* __restrict__ out, int ldc, int M, int N, int K, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  const int nt0 = blockIdx.y * ntiles;
  if (m0 >= M) return;
  const int arow = min(m0 + r, M - 1);
  const float* ap = A + (size_t)arow * lda;
  bf16x8 af[KSMAX];
  if (VEC) {
#pragma unroll
    for (int s0 = 0; s0 < KSMAX; s0 += 8) {            // eight k-steps (16 float4) in flight at a time
      float4 x0[8], x1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u < KSMAX) {
          const int kb = 16 * min(s0 + u, KS - 1) + 8 * h;
#ifdef EXP_LIN_NOA
          x0[u] = float4{(float)kb, 1.f, 2.f, (float)lane}; x1[u] = x0[u];
#else
          x0[u] = *reinterpret_cast<const float4*>(ap + kb);
          x1[u] = *reinterpret_cast<const float4*>(ap + kb + 4);
#endif
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u < KSMAX) {
          const bool on = s0 + u < KS;
          const uint4 au = {pack_bf16(x0[u].x, x0[u].y), pack_bf16(x0[u].z, x0[u].w), pack_bf16(x1[u].x, x1[u].y), pack_bf16(x1[u].z, x1[u].w)};
          const uint4 z = {0u, 0u, 0u, 0u};
          af[s0 + u] = frag_from_u4(on ? au : z);
        }
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KSMAX; ++s) {
      const int kb = 16 * s + 8 * h;
      float av[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = ap[min(kb + j, K - 1)];         // unconditional; zeroed below
#pragma unroll
      for (int j = 0; j < 8; ++j) if (kb + j >= K) av[j] = 0.f;
      const uint4 au = {pack_bf16(av[0], av[1]), pack_bf16(av[2], av[3]), pack_bf16(av[4], av[5]), pack_bf16(av[6], av[7])};
      af[s] = frag_from_u4(au);
    }
  }
  // Every load of this kernel is unconditional (clamped index, value selected afterwards) and issued with its batch: a load inside a
  // branch gets a wait of its own -- the 16 row-mask bytes were 16 round trips, and a bias load inside the column guard put an
  // s_waitcnt vmcnt(0) in front of each of a tile's 16 stores, which also waited for the NEXT tile's prefetched weights (the double
  // buffer overlapped nothing): 47 -> 20 us for 7,200 rows x 256 -> 256.
  uint8_t rm[16];
  {
    const uint8_t* mp = rowmask ? rowmask : reinterpret_cast<const uint8_t*>(bias ? (const void*)bias : (const void*)A);   // any valid address
    const int mmax = rowmask ? M - 1 : 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) rm[i] = mp[min(m0 + crow(i, h), mmax)];
    if (!rowmask) {
#pragma unroll
      for (int i = 0; i < 16; ++i) rm[i] = 1;
    }
  }
  const int NTall = (N + 31) / 32;
  const float* bp = bias ? bias : A;                   // (valid address; the value is dropped when there is no bias)
  const int bmax = bias ? N - 1 : 0;
  auto load_w = [&](uint4 (&wf)[KSMAX], float& bv, int t) {
    const uint4* wp = W + ((size_t)min(nt0 + t, NTall - 1) * KS) * 64 + lane;
#pragma unroll
#ifdef EXP_LIN_NOW
    for (int s = 0; s < KSMAX; ++s) wf[s] = uint4{(unsigned)t, (unsigned)s, 0x3f803f80u, (unsigned)lane};
#else
    for (int s = 0; s < KSMAX; ++s) wf[s] = wp[min(s, KS - 1) * 64];
#endif
    bv = bp[min((nt0 + t) * 32 + r, bmax)];
  };
  auto tile = [&](const uint4 (&wf)[KSMAX], float bvl, int t) {
    f32x16 acc = splat16(0.f);
#pragma unroll
    for (int s = 0; s < KSMAX; ++s) acc = mfma32(af[s], frag_from_u4(wf[s]), acc);
    const int col = (nt0 + t) * 32 + r;
    if (col < N) {
      const float bv = bias ? bvl : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + crow(i, h);
        if (row >= M) continue;
        float v = acc[i] + bv;
        if (RELU) v = fmaxf(v, 0.f);
        v = rm[i] ? v : 0.f;
#ifdef EXP_LIN_NOSTORE
        if (v == 1234.5678f) out[(size_t)row * ldc + col] = v;
#else
        out[(size_t)row * ldc + col] = v;
#endif
      }
    }
  };
  uint4 wA[KSMAX], wB[KSMAX];
  float bA, bB;
  load_w(wA, bA, 0);
  for (int t = 0; t < ntiles; t += 2) {
    load_w(wB, bB, t + 1);
    tile(wA, bA, t);
    load_w(wA, bA, t + 2);
    if (t + 1 < ntiles) tile(wB, bB, t + 1);
  }
}

// out = LayerNorm(a + b) * g + beta, one wave per row (D <= 1024)
__global__ void __launch_bounds__(256) k_add_layernorm(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ g, const float* __restrict__ beta,
                                                       float* __restrict__ out, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[16];
  float s = 0.f;
  const int per = (D + 63) / 64;
#pragma unroll 4
  for (int k = 0; k < per; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < D ? a[(size_t)row * D + c] + (b ? b[(size_t)row * D + c] : 0.f) : 0.f;
    s += v[k];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / D;
  float q = 0.f;
  for (int k = 0; k < per; ++k) { const int c = lane + 64 * k; const float d = c < D ? v[k] - mean : 0.f; q = fmaf(d, d, q); }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / D + 1e-5f);
  for (int k = 0; k < per; ++k) {
    const int c = lane + 64 * k;
    if (c < D) out[(size_t)row * D + c] = fmaf((v[k] - mean) * rstd, g[c], beta[c]);
  }
}

// after mlp_in_X: mask rows, encode_no_element on the hidden features, add positional encodings
// (neuralnet/transformer_model_with_y.py:661,666)
__global__ void k_x_in_finish(float* X, const uint8_t* nm, const float* pos, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float m = nm[row] ? 1.f : 0.f;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += X[(size_t)row * D + c] * m;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  for (int c = lane; c < D; c += 64) {
    float v = X[(size_t)row * D + c] * m;
    if (c == 0 && s == 0.f) v = 1.f;
    if (pos) v += pos[(size_t)row * D + c];
    X[(size_t)row * D + c] = v;
  }
}

// X_o = mlp_out_X(X_h) + X_in[..., :Kx]  (+ s1 * X_in[src]) -- transformer_model_with_y.py:676,699-701
__global__ void k_x_out_finish(const float* lin, int ldl, const float* Xin, int dxin, const int* src, const float* s1p,
                               float* out, int B, int n, int Kx) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * n * Kx) return;
  const int c = t % Kx; const size_t row = t / Kx; const int b = row / n;
  float v = lin[row * ldl + c] + Xin[row * dxin + c];
  if (src) { const int sidx = src[row]; if (sidx >= 0) v += s1p[0] * Xin[((size_t)b * n + sidx) * dxin + c]; }
  out[t] = v;
}

// alignment gather index (semantics of transformer_model_with_y.py:506-555, alignment_type='correct'):
// src[b,r] = product node with the same atom-map number as reactant-side node r, else -1.  One block per graph.
// The graph's atom-map numbers and a product flag per node are staged in LDS first (LDS_N > 0): the search loop of a node then walks
// LDS instead of issuing up to n dependent global loads (38 -> ~8 us at B = 64, n = 300).
__global__ void k_align_index(const int64_t* am, const int64_t* mol, int* src, int n, int lds_n) {
  extern __shared__ long long s_am[];                 // [lds_n] atom-map number of product nodes, 0 elsewhere
  const int b = blockIdx.x;
  __shared__ int s_prod;
  if (threadIdx.x == 0) s_prod = INT_MIN;
  __syncthreads();
  int mx = INT_MIN;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mx = max(mx, (int)mol[(size_t)b * n + i]);
  atomicMax(&s_prod, mx);
  __syncthreads();
  const long long prod = s_prod;
  if (lds_n >= n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_am[i] = mol[(size_t)b * n + i] == prod ? am[(size_t)b * n + i] : 0;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int res = -1;
    const long long a = am[(size_t)b * n + i];
    if (a > 0 && mol[(size_t)b * n + i] != prod) {
      if (lds_n >= n) {
        for (int j = 0; j < n; ++j)
          if (s_am[j] == a) { res = j; break; }       // (a > 0, so a product node whose number is 0 never matches)
      } else {
        for (int j = 0; j < n; ++j)
          if (mol[(size_t)b * n + j] == prod && am[(size_t)b * n + j] == a) { res = j; break; }
      }
    }
    src[(size_t)b * n + i] = res;
  }
}

// ---------------------------------------------------------------------------------------------------
// y path.  One workgroup (256 threads) per graph; vectors live in LDS.
// mat-vec helpers.  Every load chain is independent (no load->reduce->next-load serialisation):
//  * block_matvec:   W row-major [O][K], one THREAD per output, 16-byte loads along K (K % 4 == 0), for small K
//  * block_matvec_T: W stored transposed [K][O] (pack.py), threads = (output, K-slice): coalesced along O, partial
//                    sums combined through LDS `scratch` (>= blockDim.x floats), for large K
__device__ __forceinline__ void block_matvec(const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                                             const float* v, float* out, int O, int K, bool relu, float addc = 0.f) {
  for (int o = threadIdx.x; o < O; o += blockDim.x) {
    const float* w = W + (size_t)o * ldw;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    const bool vec = ((ldw & 3) == 0) && ((((uintptr_t)W) & 15) == 0);
    for (; vec && k + 4 <= K; k += 4) {
      const float4 ww = *reinterpret_cast<const float4*>(w + k);
      s0 = fmaf(ww.x, v[k], s0); s1 = fmaf(ww.y, v[k + 1], s1); s2 = fmaf(ww.z, v[k + 2], s2); s3 = fmaf(ww.w, v[k + 3], s3);
    }
    for (; k < K; ++k) s0 = fmaf(w[k], v[k], s0);
    float s = (s0 + s1) + (s2 + s3) + (bias ? bias[o] : 0.f) + addc;
    out[o] = relu ? fmaxf(s, 0.f) : s;
  }
}
__device__ __forceinline__ void block_matvec_T(const float* __restrict__ WT, const float* __restrict__ bias, const float* v,
                                               float* out, float* scratch, int O, int K, bool relu) {
  // requires blockDim.x % O == 0
  const int slices = blockDim.x / O, o = threadIdx.x % O, sl = threadIdx.x / O;
  const int per = (K + slices - 1) / slices, k0 = sl * per, k1 = min(K, k0 + per);
  float acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.f;
  int k = k0;
  for (; k + 16 <= k1; k += 16) {
    float w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = WT[(size_t)(k + u) * O + o];     // 16 independent loads in flight
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = fmaf(w[u], v[k + u], acc[u]);
  }
  for (; k < k1; ++k) acc[0] = fmaf(WT[(size_t)k * O + o], v[k], acc[0]);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) { s0 += acc[u]; s1 += acc[8 + u]; }
  scratch[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < O) {
    float s = bias ? bias[o] : 0.f;
    for (int q = 0; q < slices; ++q) s += scratch[q * O + o];
    out[o] = relu ? fmaxf(s, 0.f) : s;
  }
}
// out[o] = act(sum_k WT[k][o] v[k] + bias[o] + addc), WT (K,O) k-major, any O: one output per thread and pass, every load of a wave
// is one contiguous 256-byte run (a (O,K) row-major weight read row-per-thread touches 64 cache lines per wave instruction:
// lin_y1 took 15 us that way against 5.4 us for the equally large lin_y2 in this form).  K % 16 == 0 or the tail loop takes the rest.
__device__ __forceinline__ void block_matvec_kmajor(const float* __restrict__ WT, const float* __restrict__ bias, const float* v,
                                                    float* out, int O, int K, bool relu, float addc = 0.f) {
  for (int o = threadIdx.x; o < O; o += blockDim.x) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 16 <= K; k += 16) {
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = WT[(size_t)(k + u) * O + o];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = fmaf(w[u], v[k + u], acc[u & 3]);
    }
    for (; k < K; ++k) acc[0] = fmaf(WT[(size_t)k * O + o], v[k], acc[0]);
    const float s = (acc[0] + acc[1]) + (acc[2] + acc[3]) + (bias ? bias[o] : 0.f) + addc;
    out[o] = relu ? fmaxf(s, 0.f) : s;
  }
}
__device__ __forceinline__ void block_layernorm(const float* x, const float* g, const float* b, float* out, int D, float* red) {
  // out = LN(x), D <= 256, x / out in LDS; red: 2 floats of LDS.  The first wave reduces (strided partial sums, then shuffles), the
  // others wait: two barriers, no thread sums D values alone.  Ends synchronised.
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int k = threadIdx.x; k < D; k += 64) s += x[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / D;
    float q = 0.f;
    for (int k = threadIdx.x; k < D; k += 64) { const float d = x[k] - mean; q = fmaf(d, d, q); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    if (threadIdx.x == 0) { red[0] = mean; red[1] = rsqrtf(q / D + 1e-5f); }
  }
  __syncthreads();
  const float mean = red[0], rstd = red[1];
  for (int k = threadIdx.x; k < D; k += blockDim.x) out[k] = fmaf((x[k] - mean) * rstd, g[k], b[k]);
  __syncthreads();
}

struct YInArgs { const float* y; const float* W0; const float* b0; const float* W1; const float* b1; float* out; int din, dh, dy; };
// y_h = relu(W1 relu(W0 y + b0) + b1)   (mlp_in_y, transformer_model_with_y.py:590-591)
__global__ void __launch_bounds__(256) k_y_in(YInArgs a) {
  extern __shared__ float sm[];
  float* v = sm; float* hbuf = sm + a.din;
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < a.din; k += blockDim.x) v[k] = a.y[(size_t)b * a.din + k];
  __syncthreads();
  block_matvec(a.W0, a.din, a.b0, v, hbuf, a.dh, a.din, true);
  __syncthreads();
  block_matvec(a.W1, a.dh, a.b1, hbuf, a.out + (size_t)b * a.dy, a.dy, a.dh, true);
}

// Diagnostic build only (EXP_Y_STAMP): wall-clock stamps (100 MHz) of the y chain's stages, workgroup 0 (tools/stamp_y.py)
#ifdef EXP_Y_STAMP
__device__ long long g_ystamps[32];
#define YSTAMP(i) { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_ystamps[i] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int da_debug_read_ystamps(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ystamps), sizeof(long long) * 32); }
#else
#define YSTAMP(i)
#endif
struct YFilmArgs {
  const float* y;                                     // (B,dy)
  const float *We_add, *be_add, *We_mul, *be_mul;     // y_e_add / y_e_mul  weights TRANSPOSED (dy,dx)
  const float *Wx_add, *bx_add, *Wx_mul, *bx_mul;     // y_x_add / y_x_mul  weights TRANSPOSED (dy,dx)
  const float *Wout, *bout;                           // e_out.weight TRANSPOSED (dx,de), e_out.bias
  float *ye2p1, *ebias, *yx1, *yx2p1;
  int dy, dx, de;
};
// FiLM vectors of one layer (layers.py:245-247,268-270); the additive edge FiLM is folded through e_out:
//   e_out(ye1 + (ye2+1)*Y) = e_out.W((ye2+1)*Y) + [e_out.b + e_out.W ye1]
// v: y of graph b in LDS (dy floats); ye1: dx floats of LDS followed by blockDim.x floats of scratch
__device__ __forceinline__ void y_film_body(const YFilmArgs& a, const float* v, float* ye1, int b) {
  // the four projections as ONE pass over 4 dx outputs (every thread busy at dx = 256 and 1,024 threads), weights k-major
  for (int u = threadIdx.x; u < 4 * a.dx; u += blockDim.x) {
    const int which = u / a.dx, o = u - which * a.dx;
    const float* WT = which == 0 ? a.We_add : which == 1 ? a.We_mul : which == 2 ? a.Wx_add : a.Wx_mul;
    const float* bb = which == 0 ? a.be_add : which == 1 ? a.be_mul : which == 2 ? a.bx_add : a.bx_mul;
    float* dst = which == 0 ? ye1 : which == 1 ? a.ye2p1 + (size_t)b * a.dx : which == 2 ? a.yx1 + (size_t)b * a.dx : a.yx2p1 + (size_t)b * a.dx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 16 <= a.dy; k += 16) {
      float w[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) w[q] = WT[(size_t)(k + q) * a.dx + o];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = fmaf(w[q], v[k + q], acc[q & 3]);
    }
    for (; k < a.dy; ++k) acc[0] = fmaf(WT[(size_t)k * a.dx + o], v[k], acc[0]);
    dst[o] = (acc[0] + acc[1]) + (acc[2] + acc[3]) + bb[o] + ((which & 1) ? 1.f : 0.f);
  }
  __syncthreads();
  YSTAMP(8)
  block_matvec_T(a.Wout, a.bout, ye1, a.ebias + (size_t)b * a.de, ye1 + a.dx, a.de, a.dx, false);   // Wout = e_out.weight^T (dx,de)
}
__global__ void __launch_bounds__(256) k_y_film(YFilmArgs a) {
  extern __shared__ float sm[];
  float* v = sm; float* ye1 = sm + a.dy;
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < a.dy; k += blockDim.x) v[k] = a.y[(size_t)b * a.dy + k];
  __syncthreads();
  y_film_body(a, v, ye1, b);
}
struct YUpdArgs {
  const float* y; const float* xstats; const float* estats;   // partials (B,nchunk,4,d): sum, sum of squares, min, max
  const float *WxyT, *bxy, *WeyT, *bey;               // x_y.lin^T (4dx,dy), e_y.lin^T (4de,dy)
  const float *Wo0, *bo0, *Wo2, *bo2;                 // y_out.0 / .2, weights TRANSPOSED (dy,dy)
  const float *n1g, *n1b, *W1, *b1, *W2T, *b2, *n2g, *n2b;   // norm_y1, lin_y1^T (dy,dff), lin_y2^T (dff,dy), norm_y2
  float* out;
  int n, dx, de, dy, dff, nchunk_x, nchunk_e;
  const float* xy;                                    // (B,dy) x_y.lin(Xtoy statistics) precomputed by k_y_pre, or NULL: computed here
};
// chunk partials -> (mean, min, max, unbiased std) per channel in dst[4*d].  Threads = (channel, chunk slice): every slice
// folds its chunks with independent loads, the slices meet in LDS `scr` (>= 4 * blockDim.x floats).  Ends synchronised.
__device__ __forceinline__ void pooled_stats(const float* part, int nchunk, int d, double N, float* dst, float* scr) {
  const int slices = max(1, (int)blockDim.x / d), c = threadIdx.x % d, sl = threadIdx.x / d;
  float s = 0.f, q = 0.f, mn = INFINITY, mx = -INFINITY;
  if (sl < slices) {
#pragma unroll 4
    for (int k = sl; k < nchunk; k += slices) {
      const float* p = part + (size_t)k * 4 * d;
      s += p[c]; q += p[d + c]; mn = fminf(mn, p[2 * d + c]); mx = fmaxf(mx, p[3 * d + c]);
    }
  }
  scr[threadIdx.x] = s; scr[blockDim.x + threadIdx.x] = q; scr[2 * blockDim.x + threadIdx.x] = mn; scr[3 * blockDim.x + threadIdx.x] = mx;
  __syncthreads();
  if (threadIdx.x < d) {
    double S = 0.0, Q = 0.0; float MN = INFINITY, MX = -INFINITY;
    for (int u = 0; u < slices; ++u) {
      const int t = u * d + c;
      S += scr[t]; Q += scr[blockDim.x + t]; MN = fminf(MN, scr[2 * blockDim.x + t]); MX = fmaxf(MX, scr[3 * blockDim.x + t]);
    }
    const double mean = S / N;
    double var = N > 1 ? (Q - N * mean * mean) / (N - 1) : 0.0;
    if (var < 0) var = 0;
    dst[c] = (float)mean; dst[d + c] = MN; dst[2 * d + c] = MX; dst[3 * d + c] = (float)sqrt(var);
  }
  __syncthreads();
}
// new_y = y_out(y + Xtoy(X) + Etoy(E)); y = LN(y+new_y); y = LN(y + lin_y2(relu(lin_y1(y))))
// (layers.py:15-47,277-281,346-359).  Statistics run over ALL n nodes / n^2 pairs (padding included), std unbiased.
// `film`: NULL, or the NEXT layer's FiLM arguments -- its vectors are then produced here from the new y while it is still in LDS
// (one launch and one dependent kernel start less on the chain between two edge kernels).
__device__ __forceinline__ void y_update_body(const YUpdArgs& a, const YFilmArgs* film, float* sm) {
  float* sx = sm;                       // 4*dx
  float* se = sx + 4 * a.dx;            // 4*de
  float* v0 = se + 4 * a.de;            // dy
  float* v1 = v0 + a.dy;                // dy
  float* v2 = v1 + a.dy;                // dy
  float* hb = v2 + a.dy;                // dff
  float* scr = hb + a.dff;              // 4 * blockDim.x
  const int b = blockIdx.x, n = a.n;
  YSTAMP(0)
  if (a.xy == nullptr) pooled_stats(a.xstats + (size_t)b * a.nchunk_x * 4 * a.dx, a.nchunk_x, a.dx, (double)n, sx, scr);
  pooled_stats(a.estats + (size_t)b * a.nchunk_e * 4 * a.de, a.nchunk_e, a.de, (double)n * n, se, scr);
  for (int k = threadIdx.x; k < a.dy; k += blockDim.x) v0[k] = a.y[(size_t)b * a.dy + k];
  __syncthreads();
  YSTAMP(1)
  if (a.xy == nullptr) block_matvec_T(a.WxyT, a.bxy, sx, v1, scr, a.dy, 4 * a.dx, false);
  else for (int k = threadIdx.x; k < a.dy; k += blockDim.x) v1[k] = a.xy[(size_t)b * a.dy + k];
  __syncthreads();
  block_matvec_T(a.WeyT, a.bey, se, v2, scr, a.dy, 4 * a.de, false);
  __syncthreads();
  YSTAMP(2)
  for (int k = threadIdx.x; k < a.dy; k += blockDim.x) v1[k] = v0[k] + v1[k] + v2[k];
  __syncthreads();
  block_matvec_T(a.Wo0, a.bo0, v1, v2, scr, a.dy, a.dy, true);       // y_out.0 / .2, weights k-major, K split over the thread slices
  __syncthreads();
  block_matvec_T(a.Wo2, a.bo2, v2, v1, scr, a.dy, a.dy, false);
  __syncthreads();
  YSTAMP(3)
  for (int k = threadIdx.x; k < a.dy; k += blockDim.x) v1[k] += v0[k];
  __syncthreads();
  block_layernorm(v1, a.n1g, a.n1b, v0, a.dy, scr);  // v0 = y after norm_y1
  YSTAMP(4)
  block_matvec_kmajor(a.W1, a.b1, v0, hb, a.dff, a.dy, true);
  __syncthreads();
  YSTAMP(5)
  block_matvec_T(a.W2T, a.b2, hb, v1, scr, a.dy, a.dff, false);
  __syncthreads();
  YSTAMP(6)
  for (int k = threadIdx.x; k < a.dy; k += blockDim.x) v1[k] += v0[k];
  __syncthreads();
  block_layernorm(v1, a.n2g, a.n2b, v2, a.dy, scr);  // v2 = the layer's output y (LDS)
  for (int k = threadIdx.x; k < a.dy; k += blockDim.x) a.out[(size_t)b * a.dy + k] = v2[k];
  YSTAMP(7)
  if (film != nullptr) y_film_body(*film, v2, sx, b);
  YSTAMP(9) // sx .. : dx floats + blockDim.x floats of scratch (4 dx + 4 de + ... >= dx + 1024)
}
__global__ void __launch_bounds__(1024) k_y_update(YUpdArgs a) {
  extern __shared__ float sm[];
  y_update_body(a, nullptr, sm);
}
__global__ void __launch_bounds__(1024) k_y_update_film(YUpdArgs a, YFilmArgs f) {
  extern __shared__ float sm[];
  y_update_body(a, &f, sm);
}
// The part of y_update that does not depend on the edge kernel: x_y.lin of the Xtoy statistics of the layer's input X -> xy (B,dy).
// Launched beside q/k/v, off the chain between two edge kernels (its 4 dx -> dy contraction is the largest of the y branch).
__global__ void __launch_bounds__(1024) k_y_pre(const float* xstats, const float* WxyT, const float* bxy, float* xy, int n, int dx, int dy,
                                                int nchunk_x) {
  extern __shared__ float sm[];
  float* sx = sm; float* scr = sx + 4 * dx;
  const int b = blockIdx.x;
  pooled_stats(xstats + (size_t)b * nchunk_x * 4 * dx, nchunk_x, dx, (double)n, sx, scr);
  block_matvec_T(WxyT, bxy, sx, xy + (size_t)b * dy, scr, dy, 4 * dx, false);
}

// pooled statistics partials of X (Xtoy, layers.py:15-29): out[b][chunk][4][D] over the chunk's rows (padding included)
__global__ void __launch_bounds__(256) k_x_stats(const float* X, float* out, int n, int D, int nchunk) {
  const int b = blockIdx.y, ch = blockIdx.x;
  const int per = (n + nchunk - 1) / nchunk, i0 = ch * per, i1 = min(n, i0 + per);
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float s = 0.f, q = 0.f, mn = INFINITY, mx = -INFINITY;
    const float* xp = X + ((size_t)b * n) * D + c;
#pragma unroll 4
    for (int i = i0; i < i1; ++i) { const float x = xp[(size_t)i * D]; s += x; q = fmaf(x, x, q); mn = fminf(mn, x); mx = fmaxf(mx, x); }
    float* o = out + ((size_t)(b * nchunk + ch) * 4) * D;
    o[c] = s; o[D + c] = q; o[2 * D + c] = mn; o[3 * D + c] = mx;
  }
}

// ===================================================================================================
extern "C" {

int da_linear(const float* A, int lda, const void* W_bf16, int Kpad, const float* bias, const uint8_t* rowmask, float* out,
              int ldc, int M, int N, int K, int relu, void* stream) {
  DA_CHECK_ARG(Kpad % 16 == 0 && Kpad >= K, "da_linear: Kpad must be a multiple of 16 and >= K");
  if (M == 0 || N == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* W = (const bf16_t*)W_bf16;
  const int Npad = (N + 31) / 32 * 32;
  const int gm = (M + 127) / 128;
  const int KS = Kpad / 16;
  const uint4* Wf = (const uint4*)W;
  const bool vec = (lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);   // with K == Kpad: float4 loads of whole k-steps
  // column groups: enough waves to fill the chip (about one per SIMD) without re-reading the activations more than needed
  const int NTall = Npad / 32;
  const long long rowtiles = (M + 31) / 32;
  int groups = 1;
  while (groups < NTall && rowtiles * groups < 1024 && NTall % (groups * 2) == 0) groups *= 2;
  const int ntiles = NTall / groups;
#define DA_LIN2(KSMAX_, RELU_, VEC_)                                                                                       \
  k_linear<KSMAX_, RELU_, VEC_><<<dim3(gm, groups), 256, 0, s>>>(A, lda, Wf, KS, bias, rowmask, out, ldc, M, N, K, ntiles)
#define DA_LIN(KSMAX_)                                                                                                     \
  do {                                                                                                                     \
    if (vec && K == Kpad) { if (relu) DA_LIN2(KSMAX_, true, true); else DA_LIN2(KSMAX_, false, true); }                    \
    else { if (relu) DA_LIN2(KSMAX_, true, false); else DA_LIN2(KSMAX_, false, false); }                                   \
  } while (0)
  if (KS <= 4) DA_LIN(4);
  else if (KS <= 16) DA_LIN(16);
  else {
    da_set_error("da_linear: K=%d not built (K <= 256)", K);
    return DA_ERR_UNSUPPORTED;
  }
#undef DA_LIN
#undef DA_LIN2
#undef DA_LIN
  DA_CHECK_LAUNCH("da_linear");
  return DA_OK;
}

int da_add_layernorm(const float* a, const float* b, const float* gamma, const float* beta, float* out, int rows, int D,
                     void* stream) {
  DA_CHECK_ARG(D >= 1 && D <= 1024, "da_add_layernorm: D out of range");
  if (rows == 0) return DA_OK;
  k_add_layernorm<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(a, b, gamma, beta, out, rows, D);
  DA_CHECK_LAUNCH("da_add_layernorm");
  return DA_OK;
}

int da_x_in_finish(float* X, const uint8_t* node_mask, const float* pos, int rows, int D, void* stream) {
  if (rows == 0) return DA_OK;
  k_x_in_finish<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(X, node_mask, pos, rows, D);
  DA_CHECK_LAUNCH("da_x_in_finish");
  return DA_OK;
}

int da_x_out_finish(const float* lin, int ldl, const float* X_in, int dx_in, const int* src, const float* skip_scale,
                    float* out, int B, int n, int kx, void* stream) {
  if (B == 0 || n == 0) return DA_OK;
  size_t tot = (size_t)B * n * kx;
  k_x_out_finish<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(lin, ldl, X_in, dx_in, src, skip_scale, out, B, n, kx);
  DA_CHECK_LAUNCH("da_x_out_finish");
  return DA_OK;
}

int da_align_index(const int64_t* atom_map_numbers, const int64_t* mol_assignments, int* src, int B, int n, void* stream) {
  if (B == 0 || n == 0) return DA_OK;
  const int lds_n = n <= 6144 ? n : 0;             // 8 bytes per node in LDS; larger graphs search global memory
  k_align_index<<<B, 256, (size_t)lds_n * sizeof(long long), (hipStream_t)stream>>>(atom_map_numbers, mol_assignments, src, n, lds_n);
  DA_CHECK_LAUNCH("da_align_index");
  return DA_OK;
}

int da_y_in(const float* y, const float* W0, const float* b0, const float* W1, const float* b1, float* out, int B, int din,
            int dh, int dy, void* stream) {
  if (B == 0) return DA_OK;
  YInArgs a{y, W0, b0, W1, b1, out, din, dh, dy};
  k_y_in<<<B, 256, (din + dh) * sizeof(float), (hipStream_t)stream>>>(a);
  DA_CHECK_LAUNCH("da_y_in");
  return DA_OK;
}

int da_y_film(const float* y, const float* const* w /* 10 pointers, YFilmArgs order */, float* ye2p1, float* ebias,
              float* yx1, float* yx2p1, int B, int dy, int dx, int de, void* stream) {
  if (B == 0) return DA_OK;
  YFilmArgs a{y, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], ye2p1, ebias, yx1, yx2p1, dy, dx, de};
  DA_CHECK_ARG(de > 0 && 256 % de == 0, "da_y_film: de=%d must divide 256", de);
  k_y_film<<<B, 256, (dy + dx + 256) * sizeof(float), (hipStream_t)stream>>>(a);
  DA_CHECK_LAUNCH("da_y_film");
  return DA_OK;
}

int da_x_stats(const float* X, float* partials, int B, int n, int D, int nchunk, void* stream) {
  DA_CHECK_ARG(nchunk >= 1, "da_x_stats: nchunk < 1");
  if (B == 0 || n == 0) return DA_OK;
  k_x_stats<<<dim3(nchunk, B), 256, 0, (hipStream_t)stream>>>(X, partials, n, D, nchunk);
  DA_CHECK_LAUNCH("da_x_stats");
  return DA_OK;
}

int da_y_update(const float* y, const float* xstats, const float* estats,
                const float* const* w /* 16 pointers, YUpdArgs order */, float* out, int B, int n, int dx, int de, int dy,
                int dff, int nchunk_x, int nchunk_e, void* stream) {
  DA_CHECK_ARG(dy > 0 && 1024 % dy == 0, "da_y_update: dy=%d must divide 1024", dy);
  if (B == 0) return DA_OK;
  YUpdArgs a{y, xstats, estats, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13],
             w[14], w[15], out, n, dx, de, dy, dff, nchunk_x, nchunk_e, nullptr};
  size_t sh = (size_t)(4 * dx + 4 * de + 3 * dy + dff + 4 * 1024) * sizeof(float);
  k_y_update<<<B, 1024, sh, (hipStream_t)stream>>>(a);
  DA_CHECK_LAUNCH("da_y_update");
  return DA_OK;
}

// x_y.lin(Xtoy statistics) of a layer's input X (layers.py:15-29, 277): xy (B,dy) from the chunk partials da_x_stats wrote.  The part
// of da_y_update that does not depend on the layer's edge kernel; w_upd: da_y_update's pointer array (WxyT, bxy are its first two).
int da_y_pre(const float* xstats, const float* const* w_upd, float* xy, int B, int n, int dx, int dy, int nchunk_x, void* stream) {
  DA_CHECK_ARG(dy > 0 && 1024 % dy == 0 && 1024 % dx == 0, "da_y_pre: dy=%d and dx=%d must divide 1024", dy, dx);
  if (B == 0) return DA_OK;
  k_y_pre<<<B, 1024, (size_t)(4 * dx + 4 * 1024) * sizeof(float), (hipStream_t)stream>>>(xstats, w_upd[0], w_upd[1], xy, n, dx, dy, nchunk_x);
  DA_CHECK_LAUNCH("da_y_pre");
  return DA_OK;
}

// da_y_update with the Xtoy term given (xy from da_y_pre; same values, same bits as computing it here) and, when film_w is not NULL,
// da_y_film of the NEXT layer appended to the same launch (layers.py:241-262 of layer l+1 behind :277-281,346-359 of layer l).
int da_y_update_film(const float* y, const float* xy, const float* estats, const float* const* w_upd, float* out,
                     const float* const* film_w /* 10 pointers, YFilmArgs order, or NULL */, float* ye2p1, float* ebias, float* yx1,
                     float* yx2p1, int B, int n, int dx, int de, int dy, int dff, int nchunk_e, void* stream) {
  DA_CHECK_ARG(dy > 0 && 1024 % dy == 0 && de > 0 && 1024 % de == 0, "da_y_update_film: dy=%d and de=%d must divide 1024", dy, de);
  DA_CHECK_ARG(xy && estats && out, "da_y_update_film: null pointer");
  if (B == 0) return DA_OK;
  const float* const* w = w_upd;
  YUpdArgs a{y, nullptr, estats, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13],
             w[14], w[15], out, n, dx, de, dy, dff, 0, nchunk_e, xy};
  size_t sh = (size_t)(4 * dx + 4 * de + 3 * dy + dff + 4 * 1024) * sizeof(float);
  if (film_w) {
    DA_CHECK_ARG(ye2p1 && ebias && yx1 && yx2p1, "da_y_update_film: FiLM outputs");
    const float* const* fw = film_w;
    YFilmArgs f{nullptr, fw[0], fw[1], fw[2], fw[3], fw[4], fw[5], fw[6], fw[7], fw[8], fw[9], ye2p1, ebias, yx1, yx2p1, dy, dx, de};
    k_y_update_film<<<B, 1024, sh, (hipStream_t)stream>>>(a, f);
  } else {
    k_y_update<<<B, 1024, sh, (hipStream_t)stream>>>(a);
  }
  DA_CHECK_LAUNCH("da_y_update_film");
  return DA_OK;
}

}  // extern "C"
