// Fused residual blocks of the training step (layers.py:341-359: X/E/y = norm(X/E/y + dropout(branch)) of XEyTransformerLayer, and
// the ReLU + dropout between the two Linears of each feed-forward branch), forward and backward.
//
// In the unfused step each block is a chain of elementwise passes over the residual stream -- cast, mask, dropout, add, LayerNorm,
// cast again for the next GEMM -- which for the per-pair stream E (B n n rows x 64 channels) is 2.5 GB of traffic per layer.  Here:
//   k_res_ln_fwd   s = res + dropout(mask(u));  out = LayerNorm(s);  optionally out also as bf16 (the next GEMM's operand)
//   k_res_ln_bwd   (d out [+ d out_bf16]) -> d res = d s,  d u = mask(dropout'(d s)),  partial d gamma / d beta
//   k_relu_drop    h = dropout(relu(h)) in place;   k_relu_drop_bwd   d = (h != 0) ? scale * dy : 0
// Dropout keeps no mask in memory: the keep bits of element quad q are a hash of (q, seed, call-site salt), regenerated in the
// backward; `seed` is read from device memory so that a captured HIP graph draws new masks at every replay.  Not torch's Philox stream
// -- dropout masks are random either way; with p = 0 the blocks are exactly the reference's arithmetic.
// A row of D floats is held by D/4 lanes as one float4 each (16-byte accesses), 256/D rows per wave pass; grid-stride over rows.
#include "common.h"

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// four 16-bit uniform fields for element quad q (elements 4q .. 4q+3 of the tensor)
__device__ __forceinline__ uint2 drop_fields(unsigned long long q, unsigned key) {
  const unsigned a = mix32((unsigned)q ^ key);
  const unsigned b = mix32(a + 0x9e3779b9u + (unsigned)(q >> 32));
  return uint2{a, b};
}
__device__ __forceinline__ unsigned drop_key(const long long* seed, unsigned salt) {
  const unsigned long long s = seed ? (unsigned long long)*seed : 0ull;
  return mix32((unsigned)s ^ mix32((unsigned)(s >> 32) + salt * 0x85ebca6bu + 0x27d4eb2fu));
}
// keep factors (0 or scale) of the quad; thr = round(p * 65536), a field below thr drops its element
__device__ __forceinline__ f32x4 drop_factors(unsigned long long q, unsigned key, unsigned thr, float scale) {
  if (thr == 0) return f32x4{1.f, 1.f, 1.f, 1.f};
  const uint2 f = drop_fields(q, key);
  return f32x4{(f.x & 0xffffu) >= thr ? scale : 0.f, (f.x >> 16) >= thr ? scale : 0.f, (f.y & 0xffffu) >= thr ? scale : 0.f,
               (f.y >> 16) >= thr ? scale : 0.f};
}

__device__ __forceinline__ f32x4 ldq(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ldq(const bf16_t* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return f32x4{bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y)};
}
__device__ __forceinline__ void stq(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void stq(bf16_t* p, f32x4 v) { *reinterpret_cast<uint2*>(p) = uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])}; }

// mask_mode 0: none; 1: per row (row r of B n rows is kept when nm[r]); 2: per pair (row r = (b n + i) n + j is kept when
// nm[b n + i] and nm[b n + j])
__device__ __forceinline__ bool row_kept(const uint8_t* __restrict__ nm, int mode, int n, long long r) {
  if (mode == 0) return true;
  if (mode == 1) return nm[r] != 0;
  const long long bi = r / n;
  const int j = (int)(r - bi * n);
  return nm[bi] != 0 && nm[bi - bi % n + j] != 0;
}

template <int D, class TU>
__global__ void __launch_bounds__(256) k_res_ln_fwd(const float* __restrict__ res, const TU* __restrict__ u, const uint8_t* __restrict__ nm,
                                                    int mode, int n, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const long long* __restrict__ seed, unsigned salt, unsigned thr, float scale,
                                                    float* __restrict__ s_out, float* __restrict__ out, bf16_t* __restrict__ out_b, long long rows) {
  constexpr int LPR = D / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, lr = lane % LPR, rw = lane / LPR;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
  const f32x4 gm = {gamma[4 * lr], gamma[4 * lr + 1], gamma[4 * lr + 2], gamma[4 * lr + 3]};
  const f32x4 bt = {beta[4 * lr], beta[4 * lr + 1], beta[4 * lr + 2], beta[4 * lr + 3]};
  const unsigned key = thr ? drop_key(seed, salt) : 0u;
  auto rsum = [](float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  for (long long r0 = wave * RPW; r0 < rows; r0 += nwave * RPW) {
    const long long r = r0 + rw;
    const bool ok = r < rows;
    const long long rc = ok ? r : rows - 1;                               // clamped: every lane loads, the tail stores nothing
    const size_t off = (size_t)rc * D + 4 * lr;
    f32x4 x = ldq(res + off);
    f32x4 b = ldq(u + off);
    if (!row_kept(nm, mode, n, rc)) b = f32x4{0.f, 0.f, 0.f, 0.f};
    b *= drop_factors((unsigned long long)rc * LPR + lr, key, thr, scale);
    x += b;
    if (ok) stq(s_out + off, x);
    const float mean = rsum(x[0] + x[1] + x[2] + x[3]) * (1.f / D);
    x -= mean;
    const float rstd = rsqrtf(rsum(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]) * (1.f / D) + 1e-5f);
    x = x * rstd * gm + bt;
    if (ok) {
      stq(out + off, x);
      if (out_b) stq(out_b + off, x);
    }
  }
}

template <int D, class TU>
__global__ void __launch_bounds__(256) k_res_ln_bwd(const float* __restrict__ s, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                    const bf16_t* __restrict__ dyb, const uint8_t* __restrict__ nm, int mode, int n,
                                                    const long long* __restrict__ seed, unsigned salt, unsigned thr, float scale,
                                                    float* __restrict__ ds, TU* __restrict__ du, float* __restrict__ pg, float* __restrict__ pb,
                                                    long long rows) {
  __shared__ float s_g[4][D], s_b[4][D];
  constexpr int LPR = D / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, lr = lane % LPR, rw = lane / LPR;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
  const f32x4 gm = {gamma[4 * lr], gamma[4 * lr + 1], gamma[4 * lr + 2], gamma[4 * lr + 3]};
  const unsigned key = thr ? drop_key(seed, salt) : 0u;
  f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
  auto rsum = [](float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  for (long long r0 = wave * RPW; r0 < rows; r0 += nwave * RPW) {
    const long long r = r0 + rw;
    const bool ok = r < rows;
    const long long rc = ok ? r : rows - 1;
    const size_t off = (size_t)rc * D + 4 * lr;
    f32x4 x = ldq(s + off);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (dy) g = ldq(dy + off);
    if (dyb) g += ldq(dyb + off);
    if (!ok) g = f32x4{0.f, 0.f, 0.f, 0.f};
    const float mean = rsum(x[0] + x[1] + x[2] + x[3]) * (1.f / D);
    x -= mean;
    const float rstd = rsqrtf(rsum(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]) * (1.f / D) + 1e-5f);
    x *= rstd;
    ag += g * x; ab += g;
    g *= gm;
    const float s1 = rsum(g[0] + g[1] + g[2] + g[3]) * (1.f / D);
    const float s2 = rsum(g[0] * x[0] + g[1] * x[1] + g[2] * x[2] + g[3] * x[3]) * (1.f / D);
    f32x4 d = rstd * (g - s1 - x * s2);
    if (ok) stq(ds + off, d);
    if (du) {
      if (!row_kept(nm, mode, n, rc)) d = f32x4{0.f, 0.f, 0.f, 0.f};
      d *= drop_factors((unsigned long long)rc * LPR + lr, key, thr, scale);
      if (ok) stq(du + off, d);
    }
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
// d gamma / d beta: the G partial rows folded in a fixed order (deterministic).  Workgroup = 32 of the 2 D outputs x 8 groups of
// partial rows; a thread sums its group's rows sixteen loads at a time, the 8 group sums are added in group order through LDS
// (one thread per output walking all G rows took 30 us a call, 52 calls a step).
__global__ void __launch_bounds__(256) k_col_fold(const float* __restrict__ pg, const float* __restrict__ pb, float* __restrict__ dg,
                                                  float* __restrict__ db, int G, int D) {
  __shared__ float s_p[8][32];
  const int ol = threadIdx.x & 31, gg = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + ol;
  const bool live = t < 2 * D;
  const float* src = t < D ? pg + t : pb + (t - D);
  const int per = (G + 7) / 8, g0 = gg * per, g1 = min(G, g0 + per);
  float sum = 0.f;
  if (live)
    for (int g = g0; g < g1; g += 16) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = g + q < g1 ? src[(size_t)(g + q) * D] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += v[q];
    }
  s_p[gg][ol] = sum;
  __syncthreads();
  if (gg == 0 && live) {
    for (int k = 1; k < 8; ++k) sum += s_p[k][ol];
    if (t < D) dg[t] = sum; else db[t - D] = sum;
  }
}

// ReLU + dropout in place, 8 elements (two quads) per lane for bf16, 4 for fp32
template <class T>
__global__ void __launch_bounds__(256) k_relu_drop(T* __restrict__ h, long long quads, const long long* __restrict__ seed, unsigned salt,
                                                   unsigned thr, float scale) {
  const unsigned key = thr ? drop_key(seed, salt) : 0u;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long long)gridDim.x * blockDim.x) {
    f32x4 v = ldq(h + 4 * q);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    v *= drop_factors((unsigned long long)q, key, thr, scale);
    stq(h + 4 * q, v);
  }
}
template <class T>
__global__ void __launch_bounds__(256) k_relu_drop_bwd(const T* __restrict__ h, const T* __restrict__ dy, T* __restrict__ dx, long long quads, float scale) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long long)gridDim.x * blockDim.x) {
    const f32x4 o = ldq(h + 4 * q);
    f32x4 g = ldq(dy + 4 * q);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = o[k] != 0.f ? g[k] * scale : 0.f;
    stq(dx + 4 * q, g);
  }
}

#define RES_LN_MAX_WG 1024
static bool res_ln_width_ok(int D) { return D == 16 || D == 32 || D == 64 || D == 128 || D == 256; }
static void drop_consts(float p, unsigned* thr, float* scale) {
  unsigned t = (unsigned)(p * 65536.f + 0.5f);
  if (t > 65535u) t = 65535u;
  *thr = t;
  *scale = 65536.f / (float)(65536u - t);                                 // the exact keep probability of the 16-bit threshold
}
static long long res_ln_grid(long long rows, int D) {
  long long g = (rows * D / 256 + 3) / 4;                                  // a wave pass covers 256 floats
  if (g < 1) g = 1;
  if (g > RES_LN_MAX_WG) g = RES_LN_MAX_WG;
  return g;
}

extern "C" {

int da_res_ln_fwd(const float* res, const void* u, int u_bf16, const uint8_t* node_mask, int mask_mode, int n, const float* gamma,
                  const float* beta, const long long* seed, int salt, float p, float* s_out, float* out, void* out_bf16, long long rows,
                  int D, void* stream) {
  DA_CHECK_ARG(rows >= 0 && res_ln_width_ok(D), "da_res_ln_fwd: width %d not built (16, 32, 64, 128, 256)", D);
  DA_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2 && (mask_mode == 0 || (node_mask && n >= 1)), "da_res_ln_fwd: mask_mode / node_mask");
  DA_CHECK_ARG(mask_mode != 2 || rows % ((long long)n * n) == 0, "da_res_ln_fwd: rows is not a multiple of n * n");
  DA_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || seed), "da_res_ln_fwd: p in [0,1), seed required when p > 0");
  DA_CHECK_ARG((((uintptr_t)res | (uintptr_t)s_out | (uintptr_t)out) & 15) == 0 && (((uintptr_t)u | (uintptr_t)out_bf16) & 7) == 0 &&
                   (u_bf16 || ((uintptr_t)u & 15) == 0), "da_res_ln_fwd: operands must be 16-byte (fp32) / 8-byte (bf16) aligned");
  if (rows == 0) return DA_OK;
  unsigned thr; float scale;
  drop_consts(p, &thr, &scale);
  const unsigned g = (unsigned)res_ln_grid(rows, D);
  hipStream_t st = (hipStream_t)stream;
#define FWD(DD)                                                                                                                          \
  case DD:                                                                                                                                \
    if (u_bf16) k_res_ln_fwd<DD, bf16_t><<<g, 256, 0, st>>>(res, (const bf16_t*)u, node_mask, mask_mode, n, gamma, beta, seed, (unsigned)salt, thr, scale, \
                                                          s_out, out, (bf16_t*)out_bf16, rows);                                          \
    else k_res_ln_fwd<DD, float><<<g, 256, 0, st>>>(res, (const float*)u, node_mask, mask_mode, n, gamma, beta, seed, (unsigned)salt, thr, scale, s_out,   \
                                                   out, (bf16_t*)out_bf16, rows);                                                        \
    break;
  switch (D) { FWD(16) FWD(32) FWD(64) FWD(128) FWD(256) }
#undef FWD
  DA_CHECK_LAUNCH("da_res_ln_fwd");
  return DA_OK;
}

long long da_res_ln_bwd_workspace(int D) { return (long long)sizeof(float) * 2 * RES_LN_MAX_WG * D; }

int da_res_ln_bwd(const float* s, const float* gamma, const float* d_out, const void* d_out_bf16, const uint8_t* node_mask, int mask_mode, int n,
                  const long long* seed, int salt, float p, float* d_res, void* d_u, int u_bf16, float* dgamma, float* dbeta,
                  void* workspace, long long workspace_bytes, long long rows, int D, void* stream) {
  DA_CHECK_ARG(rows >= 0 && res_ln_width_ok(D), "da_res_ln_bwd: width %d not built (16, 32, 64, 128, 256)", D);
  DA_CHECK_ARG(d_out || d_out_bf16, "da_res_ln_bwd: no incoming gradient");
  DA_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2 && (mask_mode == 0 || (node_mask && n >= 1)), "da_res_ln_bwd: mask_mode / node_mask");
  DA_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || seed), "da_res_ln_bwd: p in [0,1), seed required when p > 0");
  DA_CHECK_ARG(workspace && workspace_bytes >= da_res_ln_bwd_workspace(D), "da_res_ln_bwd: workspace smaller than da_res_ln_bwd_workspace(D)");
  DA_CHECK_ARG((((uintptr_t)s | (uintptr_t)d_out | (uintptr_t)d_res) & 15) == 0 && (((uintptr_t)d_out_bf16 | (uintptr_t)d_u) & 7) == 0 &&
                   (u_bf16 || ((uintptr_t)d_u & 15) == 0), "da_res_ln_bwd: operands must be 16-byte (fp32) / 8-byte (bf16) aligned");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    if (hipMemsetAsync(dgamma, 0, sizeof(float) * D, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * D, st) != hipSuccess) {
      da_set_error("da_res_ln_bwd: memset failed"); return DA_ERR_HIP;
    }
    return DA_OK;
  }
  unsigned thr; float scale;
  drop_consts(p, &thr, &scale);
  const unsigned g = (unsigned)res_ln_grid(rows, D);
  float* pg = (float*)workspace; float* pb = pg + (size_t)g * D;
#define BWD(DD)                                                                                                                          \
  case DD:                                                                                                                                \
    if (u_bf16) k_res_ln_bwd<DD, bf16_t><<<g, 256, 0, st>>>(s, gamma, d_out, (const bf16_t*)d_out_bf16, node_mask, mask_mode, n, seed, (unsigned)salt, thr, \
                                                          scale, d_res, (bf16_t*)d_u, pg, pb, rows);                                      \
    else k_res_ln_bwd<DD, float><<<g, 256, 0, st>>>(s, gamma, d_out, (const bf16_t*)d_out_bf16, node_mask, mask_mode, n, seed, (unsigned)salt, thr, scale,  \
                                                   d_res, (float*)d_u, pg, pb, rows);                                                     \
    break;
  switch (D) { BWD(16) BWD(32) BWD(64) BWD(128) BWD(256) }
#undef BWD
  k_col_fold<<<(2 * D + 31) / 32, 256, 0, st>>>(pg, pb, dgamma, dbeta, (int)g, D);
  DA_CHECK_LAUNCH("da_res_ln_bwd");
  return DA_OK;
}

int da_relu_dropout(void* h, int is_bf16, long long numel, const long long* seed, int salt, float p, void* stream) {
  DA_CHECK_ARG(numel >= 0 && numel % 4 == 0, "da_relu_dropout: numel must be a multiple of 4");
  DA_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || seed), "da_relu_dropout: p in [0,1), seed required when p > 0");
  DA_CHECK_ARG(((uintptr_t)h & (is_bf16 ? 7 : 15)) == 0, "da_relu_dropout: alignment");
  if (numel == 0) return DA_OK;
  unsigned thr; float scale;
  drop_consts(p, &thr, &scale);
  const long long quads = numel / 4;
  long long g = (quads + 255) / 256; if (g > 16384) g = 16384;
  if (is_bf16) k_relu_drop<bf16_t><<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((bf16_t*)h, quads, seed, (unsigned)salt, thr, scale);
  else k_relu_drop<float><<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((float*)h, quads, seed, (unsigned)salt, thr, scale);
  DA_CHECK_LAUNCH("da_relu_dropout");
  return DA_OK;
}

int da_relu_dropout_bwd(const void* h, const void* dy, void* dx, int is_bf16, long long numel, float p, void* stream) {
  DA_CHECK_ARG(numel >= 0 && numel % 4 == 0, "da_relu_dropout_bwd: numel must be a multiple of 4");
  DA_CHECK_ARG(p >= 0.f && p < 1.f, "da_relu_dropout_bwd: p in [0,1)");
  DA_CHECK_ARG((((uintptr_t)h | (uintptr_t)dy | (uintptr_t)dx) & (is_bf16 ? 7 : 15)) == 0, "da_relu_dropout_bwd: alignment");
  if (numel == 0) return DA_OK;
  unsigned thr; float scale;
  drop_consts(p, &thr, &scale);
  const long long quads = numel / 4;
  long long g = (quads + 255) / 256; if (g > 16384) g = 16384;
  if (is_bf16) k_relu_drop_bwd<bf16_t><<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((const bf16_t*)h, (const bf16_t*)dy, (bf16_t*)dx, quads, scale);
  else k_relu_drop_bwd<float><<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((const float*)h, (const float*)dy, (float*)dx, quads, scale);
  DA_CHECK_LAUNCH("da_relu_dropout_bwd");
  return DA_OK;
}

}  // extern "C"
