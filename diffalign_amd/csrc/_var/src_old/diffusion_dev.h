// Device functions shared by the diffusion kernels (k_diffusion.hip) and the fused output stage (k_edge_io.hip):
// posterior over a K-class table, exponential-race draw, counter-based Exp(1) noise.  Floating-point contraction is
// switched off inside each function, so that both translation units produce the same bits whatever their flags are
// (draws are compared bit for bit between the fused and the separate reverse step).
#pragma once
#include "common.h"

template <int K>
__device__ __forceinline__ int onehot_argmax(const float* x) {
  int best = 0; float bv = x[0];
#pragma unroll
  for (int k = 1; k < K; ++k) if (x[k] > bv) { bv = x[k]; best = k; }
  return best;
}
// exponential race: argmax_k p[k] / q[k], first index wins ties (torch.multinomial on CPU)
template <int K>
__device__ __forceinline__ int race(const float* p, const float* q) {
#pragma clang fp contract(off)
  int best = 0; float bv = p[0] / q[0];
#pragma unroll
  for (int k = 1; k < K; ++k) { float v = p[k] / q[k]; if (v > bv) { bv = v; best = k; } }
  return best;
}

// The per-(xt, k0, k) factor (Qt[k][xt] * Qsb[k0][k]) / d[k0] of posterior_elem as a table M[xt][k0][k] (K^3 floats, LDS).
template <int K>
__device__ __forceinline__ void posterior_table_fill(const float* qt, const float* qsb, const float* qtb, float* M, int tid, int nthreads) {
#pragma clang fp contract(off)
  for (int t = tid; t < K * K * K; t += nthreads) {
    const int xt = t / (K * K), k0 = (t / K) % K, k = t % K;
    float d = qtb[k0 * K + xt];
    if (d == 0.f) d = 1e-6f;
    M[t] = (qt[k * K + xt] * qsb[k0 * K + k]) / d;
  }
}
// posterior_elem with that factor taken from the table, which was filled with exactly the expression posterior_elem
// evaluates: same operations on the same values in the same order => same bits.
template <int K>
__device__ __forceinline__ void posterior_elem_tab(const float* __restrict__ logit, int xt, const float* M, float temp, float* u) {
#pragma clang fp contract(off)
  float l[K];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) { l[k] = logit[k]; mx = fmaxf(mx, l[k]); }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += expf(l[k] - mx);
  float lse = mx + logf(s);
  float mx2 = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) { l[k] = (l[k] - lse) * temp; mx2 = fmaxf(mx2, l[k]); }
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { l[k] = expf(l[k] - mx2); s2 += l[k]; }
#pragma unroll
  for (int k = 0; k < K; ++k) u[k] = 0.f;
  const float* Mx = M + xt * K * K;
#pragma unroll
  for (int k0 = 0; k0 < K; ++k0) {
    float ph = l[k0] / s2;
#pragma unroll
    for (int k = 0; k < K; ++k) u[k] += ph * Mx[k0 * K + k];
  }
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) tot += u[k];
  if (tot == 0.f) {
#pragma unroll
    for (int k = 0; k < K; ++k) u[k] = 1e-5f;
    tot = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) tot += u[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) u[k] = u[k] / tot;
}

// 24 random bits of (seed, stream, element) -> u in (0,1) -> Exp(1).  splitmix64's finaliser over a Weyl-stepped key; the same
// integers as oracle.counter_exp_noise (numpy uint64).
__device__ __forceinline__ float counter_exp(unsigned long long seed, unsigned long long stream, unsigned long long elem) {
#pragma clang fp contract(off)
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (stream + 1ull);
  z ^= elem * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = ((float)(unsigned)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
  return -__logf(u);
}
