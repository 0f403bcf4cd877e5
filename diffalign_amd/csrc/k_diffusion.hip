// Diffusion-side kernels: container masking, free-to-diffuse masks, apply_mask, posterior + categorical
// draw + apply_mask (fused), forward noising.  All HBM-bound integer/float elementwise work.
// Reference semantics: see include/diffalign_hip.h for the file:line each entry point replaces.
#include "common.h"
#include "diffusion_dev.h"

#define MAXK 32   // max number of classes handled in registers

// ---------------------------------------------------------------------------------------------------
// PlaceHolder.mask (utils/placeholder.py:81-102), in place.  One thread per feature vector.
__global__ void k_mask_dense_X(float* X, const uint8_t* nm, int rows, int d) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float* x = X + (size_t)r * d;
  float m = nm[r] ? 1.f : 0.f, s = 0.f;
  for (int c = 0; c < d; ++c) { float v = x[c] * m; x[c] = v; s += v; }
  if (s == 0.f) x[0] = 1.f;
}
__global__ void k_mask_dense_E(float* E, const uint8_t* nm, int B, int n, int d) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t tot = (size_t)B * n * n;
  if (p >= tot) return;
  int j = p % n; int i = (p / n) % n; int b = p / ((size_t)n * n);
  float m = (nm[b * n + i] && nm[b * n + j] && i != j) ? 1.f : 0.f;
  float* e = E + p * d;
  float s = 0.f;
  for (int c = 0; c < d; ++c) { float v = e[c] * m; e[c] = v; s += v; }
  if (s == 0.f) e[0] = 1.f;
}
// the same for feature widths that are a multiple of 4 with d / 4 lanes dividing a wave (hidden features: d = 64 -> 16 lanes per pair,
// 16-byte accesses; the per-thread row walk above moves 4 bytes per lane and instruction: 372 us against 45 us at B = 16, n = 140)
template <int LPR>
__global__ void __launch_bounds__(256) k_mask_dense_E4(float* E, const uint8_t* nm, int B, int n) {
  const size_t tot = (size_t)B * n * n;
  const int lr = threadIdx.x % LPR;
  for (size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR; p < (tot + 63) / 64 * 64; p += (size_t)gridDim.x * blockDim.x / LPR) {
    const bool ok = p < tot;
    const size_t pc = ok ? p : tot - 1;
    const int j = pc % n; const int i = (pc / n) % n; const int b = pc / ((size_t)n * n);
    const float m = (nm[b * n + i] && nm[b * n + j] && i != j) ? 1.f : 0.f;
    float4* e = reinterpret_cast<float4*>(E + pc * (4 * LPR)) + lr;
    float4 v = *e;
    v.x *= m; v.y *= m; v.z *= m; v.w *= m;
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (s == 0.f && lr == 0) v.x = 1.f;
    if (ok) *e = v;
  }
}
// collapse=True: argmax then zero masked (utils/placeholder.py:90-94)
__global__ void k_collapse_X(const float* X, const uint8_t* nm, int64_t* out, int rows, int d) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* x = X + (size_t)r * d;
  int best = 0; float bv = x[0];
  for (int c = 1; c < d; ++c) if (x[c] > bv) { bv = x[c]; best = c; }
  out[r] = nm[r] ? best : 0;
}
__global__ void k_collapse_E(const float* E, const uint8_t* nm, int64_t* out, int B, int n, int d) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t tot = (size_t)B * n * n;
  if (p >= tot) return;
  int j = p % n; int i = (p / n) % n; int b = p / ((size_t)n * n);
  const float* e = E + p * d;
  int best = 0; float bv = e[0];
  for (int c = 1; c < d; ++c) if (e[c] > bv) { bv = e[c]; best = c; }
  out[p] = (nm[b * n + i] && nm[b * n + j]) ? best : 0;
}

// ---------------------------------------------------------------------------------------------------
// free-to-diffuse node masks, closed form of utils/graph.py:462-619 (one block per graph)
// mode: 0 product, 1 sn, 2 product_and_sn, 3 reactant, 4 reactant_and_sn, 5 atom_mapping
__global__ void k_free_masks(const float* X, const uint8_t* nm, const int64_t* am, int n, int kx, int mode,
                             int suno_idx, int include_sn, const int* states, int n_states, uint8_t* mx, uint8_t* me) {
  int b = blockIdx.x;
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = -1;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float* x = X + ((size_t)b * n + i) * kx;
    int best = 0; float bv = x[0];
    for (int c = 1; c < kx; ++c) if (x[c] > bv) { bv = x[c]; best = c; }
    if (best == suno_idx) atomicMax(&s_last, i);
  }
  __syncthreads();
  int last = s_last;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float* x = X + ((size_t)b * n + i) * kx;
    int best = 0; float bv = x[0];
    for (int c = 1; c < kx; ++c) if (x[c] > bv) { bv = x[c]; best = c; }
    bool prod = (i <= last) && (i != n - 1);
    bool sn = include_sn ? (i != last) : (nm[b * n + i] != 0);
    bool fx, fe;
    switch (mode) {
      case 0: fx = fe = prod; break;
      case 1: fx = fe = sn; break;
      case 2: fx = fe = prod && sn; break;
      case 3: fx = fe = !(prod && sn); break;
      case 4: fx = fe = (!(prod && sn)) && sn; break;
      default: fe = prod && sn; fx = fe && (am[b * n + i] == 0); break;
    }
    bool st = true;
    for (int s = 0; s < n_states; ++s) st = st && (best != states[s]);
    mx[b * n + i] = (fx && st) ? 1 : 0;
    me[b * n + i] = (fe && st) ? 1 : 0;
  }
}

// apply_mask (utils/graph.py:440-455): z = free ? z : orig (*100 when as_logits).  In place on z.
__global__ void k_apply_mask_X(float* Z, const float* O, const uint8_t* mx, int rows, int d, float scale) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)rows * d) return;
  if (!mx[t / d]) Z[t] = O[t] * scale;
}
__global__ void k_apply_mask_E(float* Z, const float* O, const uint8_t* me, int B, int n, int d, float scale) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t tot = (size_t)B * n * n * d;
  if (t >= tot) return;
  size_t p = t / d;
  int j = p % n; int i = (p / n) % n; int b = p / ((size_t)n * n);
  if (!(me[b * n + i] && me[b * n + j])) Z[t] = O[t] * scale;
}

// ---------------------------------------------------------------------------------------------------
// posterior p(z_s | z_t) for one element (utils/diffusion/helpers.py:193-229,785-807), K classes.
//   phat = softmax(pred);  a[k] = Qt[k][xt];  d[k0] = Qtb[k0][xt] (0 -> 1e-6)
//   u[k] = sum_k0 phat[k0] * (a[k] * Qsb[k0][k] / d[k0]);  all-zero u -> 1e-5;  normalise
// pred = log_softmax(logits) * temperature (diffusion/sampling.py:82).
template <int K>
__device__ __forceinline__ void posterior_elem(const float* __restrict__ logit, int xt, const float* Qt, const float* Qsb,
                                               const float* Qtb, float temp, float* u) {
  float l[K];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) { l[k] = logit[k]; mx = fmaxf(mx, l[k]); }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += expf(l[k] - mx);
  float lse = mx + logf(s);
  // pred = (l - lse) * temp ; phat = softmax(pred)
  float mx2 = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) { l[k] = (l[k] - lse) * temp; mx2 = fmaxf(mx2, l[k]); }
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { l[k] = expf(l[k] - mx2); s2 += l[k]; }
#pragma unroll
  for (int k = 0; k < K; ++k) u[k] = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < K; ++k0) {
    float ph = l[k0] / s2;
    float d = Qtb[k0 * K + xt];
    if (d == 0.f) d = 1e-6f;
#pragma unroll
    for (int k = 0; k < K; ++k) u[k] += ph * ((Qt[k * K + xt] * Qsb[k0 * K + k]) / d);
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

// nodes: one thread per (b,i)
template <int K>
__global__ void k_posterior_X(const float* logits, const float* zt, const uint8_t* nm, const float* Qt, const float* Qsb,
                              const float* Qtb, float temp, const float* noise, const float* orig, const uint8_t* mx,
                              float* prob_out, float* zs_out, int B, int n) {
  extern __shared__ float sQ[];  // 3*K*K
  int b = blockIdx.y;
  for (int t = threadIdx.x; t < 3 * K * K; t += blockDim.x) {
    int w = t / (K * K), o = t % (K * K);
    const float* src = w == 0 ? Qt : (w == 1 ? Qsb : Qtb);
    sQ[t] = src[(size_t)b * K * K + o];
  }
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  size_t r = (size_t)b * n + i;
  bool valid = nm[r] != 0;
  float u[K];
  int xt = onehot_argmax<K>(zt + r * K);
  posterior_elem<K>(logits + r * K, xt, sQ, sQ + K * K, sQ + 2 * K * K, temp, u);
  if (!valid) {  // prob_s.mask(): zero row -> encode_no_element -> class 0
#pragma unroll
    for (int k = 0; k < K; ++k) u[k] = k == 0 ? 1.f : 0.f;
  }
  if (prob_out) {
#pragma unroll
    for (int k = 0; k < K; ++k) prob_out[r * K + k] = u[k];
  }
  if (zs_out) {
    int cls = 0;
    if (valid) cls = race<K>(u, noise + r * K);
    bool free_ = mx ? mx[r] != 0 : true;
#pragma unroll
    for (int k = 0; k < K; ++k) zs_out[r * K + k] = free_ ? (k == cls ? 1.f : 0.f) : orig[r * K + k];
  }
}

// nodes, K > 16 classes: 32 lanes per (b,i), lane k owns class k.  One thread per node walks K x K quotients one after the other
// (841 divisions for the 29 atom classes: 75 - 130 us for a few thousand nodes, all latency); here a lane does its own column's K.
// Every value is produced by the same operations in the same order as posterior_elem / race do (the softmax sums and the total are
// accumulated in class order, by every lane alike or through shuffles): same bits.  A draw-only call skips nodes that are not free
// to diffuse -- their row is the condition's.
template <int K>
__global__ void __launch_bounds__(256) k_posterior_X_lanes(const float* logits, const float* zt, const uint8_t* nm, const float* Qt,
                                                           const float* Qsb, const float* Qtb, float temp, const float* noise,
                                                           const float* orig, const uint8_t* mx, float* prob_out, float* zs_out, int B, int n) {
  static_assert(K <= 32, "one class per lane of a half wave");
  extern __shared__ float sQ[];  // 3*K*K
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < 3 * K * K; t += blockDim.x) {
    int w = t / (K * K), o = t % (K * K);
    const float* src = w == 0 ? Qt : (w == 1 ? Qsb : Qtb);
    sQ[t] = src[(size_t)b * K * K + o];
  }
  __syncthreads();
  const float* sQt = sQ; const float* sQsb = sQ + K * K; const float* sQtb = sQ + 2 * K * K;
  const int kl = threadIdx.x & 31, k = min(kl, K - 1), half = (threadIdx.x & 63) & 32;
  const int i = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
  const bool inside = i < n;
  const size_t r = (size_t)b * n + min(i, n - 1);
  const bool valid = nm[r] != 0;
  const bool free_ = mx ? mx[r] != 0 : true;
  if (!prob_out && !free_) {                               // (uniform over the 32 lanes of the node)
    if (inside && kl < K) zs_out[r * K + k] = orig[r * K + k];
    return;
  }
  const int xt = onehot_argmax<K>(zt + r * K);
  float l[K];
  float m1 = -INFINITY;
#pragma unroll
  for (int c = 0; c < K; ++c) { l[c] = logits[r * K + c]; m1 = fmaxf(m1, l[c]); }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < K; ++c) s += expf(l[c] - m1);
  const float lse = m1 + logf(s);
  float m2 = -INFINITY;
#pragma unroll
  for (int c = 0; c < K; ++c) { l[c] = (l[c] - lse) * temp; m2 = fmaxf(m2, l[c]); }
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < K; ++c) { l[c] = expf(l[c] - m2); s2 += l[c]; }
  float u = 0.f;
  const float a = sQt[k * K + xt];
#pragma unroll
  for (int k0 = 0; k0 < K; ++k0) {
    const float ph = l[k0] / s2;
    float d = sQtb[k0 * K + xt];
    if (d == 0.f) d = 1e-6f;
    u += ph * ((a * sQsb[k0 * K + k]) / d);
  }
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < K; ++c) tot += __shfl(u, half + c);
  if (tot == 0.f) {
    u = 1e-5f;
    tot = 0.f;
#pragma unroll
    for (int c = 0; c < K; ++c) tot += 1e-5f;
  }
  u = u / tot;
  if (!valid) u = k == 0 ? 1.f : 0.f;                       // prob_s.mask(): zero row -> encode_no_element -> class 0
  const bool wr = inside && kl < K;
  if (prob_out && wr) prob_out[r * K + k] = u;
  if (zs_out) {
    int cls = 0;
    if (valid) {
      const float v = u / noise[r * K + k];
      float bv = __shfl(v, half);
#pragma unroll
      for (int c = 1; c < K; ++c) { const float vc = __shfl(v, half + c); if (vc > bv) { bv = vc; cls = c; } }
    }
    if (wr) zs_out[r * K + k] = free_ ? (k == cls ? 1.f : 0.f) : orig[r * K + k];
  }
}

// edges: one thread per ordered pair (b,i,j); the draw of the unordered pair comes from (min,max)
template <int K>
__global__ void k_posterior_E(const float* logits, const float* zt, const uint8_t* nm, const float* Qt, const float* Qsb,
                              const float* Qtb, float temp, const float* noise, const float* orig, const uint8_t* me,
                              float* prob_out, float* zs_out, int B, int n) {
  __shared__ float sQ[3 * K * K];
  int b = blockIdx.y;
  for (int t = threadIdx.x; t < 3 * K * K; t += blockDim.x) {
    int w = t / (K * K), o = t % (K * K);
    const float* src = w == 0 ? Qt : (w == 1 ? Qsb : Qtb);
    sQ[t] = src[(size_t)b * K * K + o];
  }
  __syncthreads();
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n * n) return;
  int i = p / n, j = p % n;
  size_t base = (size_t)b * n * n;
  bool pairv = nm[b * n + i] && nm[b * n + j];
  float u[K];
  if (prob_out) {   // probabilities of the ordered pair itself
    size_t r = base + p;
    int xt = onehot_argmax<K>(zt + r * K);
    posterior_elem<K>(logits + r * K, xt, sQ, sQ + K * K, sQ + 2 * K * K, temp, u);
    if (!pairv || i == j) {
#pragma unroll
      for (int k = 0; k < K; ++k) u[k] = k == 0 ? 1.f : 0.f;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) prob_out[r * K + k] = u[k];
  }
  if (zs_out) {
    int cls = 0;
    if (pairv && i != j) {
      int a = min(i, j), c = max(i, j);
      size_t r = base + (size_t)a * n + c;     // upper-triangle element drives both (i,j) and (j,i)
      int xt = onehot_argmax<K>(zt + r * K);
      posterior_elem<K>(logits + r * K, xt, sQ, sQ + K * K, sQ + 2 * K * K, temp, u);
      cls = race<K>(u, noise + r * K);
    }
    size_t r = base + p;
    bool free_ = me ? (me[b * n + i] && me[b * n + j]) : true;
#pragma unroll
    for (int k = 0; k < K; ++k) zs_out[r * K + k] = free_ ? (k == cls ? 1.f : 0.f) : orig[r * K + k];
  }
}

// edges, draw only (no probabilities out): one workgroup per pair of mirrored 32x32 tiles (it <= jt).  Each unordered pair is
// evaluated ONCE, by its upper-triangle element; the class goes through LDS to both output tiles, which are written
// row-wise.  Same arithmetic per element as k_posterior_E (bit-identical draws), half the evaluations, no transposed reads.
template <int K>
__global__ void __launch_bounds__(256) k_posterior_E_tiles(const float* __restrict__ logits, const float* __restrict__ zt,
                                                           const uint8_t* __restrict__ nm, const float* __restrict__ Qt,
                                                           const float* __restrict__ Qsb, const float* __restrict__ Qtb, float temp,
                                                           const float* __restrict__ noise, const float* __restrict__ orig,
                                                           const uint8_t* __restrict__ me, float* __restrict__ zs_out, int n) {
  int it, jt;
  tri_tile(blockIdx.x, (n + 31) / 32, it, jt);
  const int b = blockIdx.y;
  __shared__ float sM[K * K * K];
  __shared__ uint8_t cls[32][33];
  posterior_table_fill<K>(Qt + (size_t)b * K * K, Qsb + (size_t)b * K * K, Qtb + (size_t)b * K * K, sM, threadIdx.x, 256);
  __syncthreads();
  const size_t base = (size_t)b * n * n;
  for (int t = threadIdx.x; t < 32 * 32; t += 256) {
    const int ri = t >> 5, cj = t & 31, i = it * 32 + ri, j = jt * 32 + cj;
    int c = 0;
    if (i < j && j < n && nm[b * n + i] && nm[b * n + j]) {
      const size_t r = base + (size_t)i * n + j;
      float u[K];
      const int xt = onehot_argmax<K>(zt + r * K);
      posterior_elem_tab<K>(logits + r * K, xt, sM, temp, u);
      c = race<K>(u, noise + r * K);
    }
    cls[ri][cj] = (uint8_t)c;
  }
  __syncthreads();
  constexpr int W = 32 * K;
  for (int t = threadIdx.x; t < 32 * W; t += 256) {                // tile (it, jt), rows i
    const int ri = t / W, o = t - ri * W, cj = o / K, k = o - cj * K, i = it * 32 + ri, j = jt * 32 + cj;
    if (i >= n || j >= n) continue;
    const int c = i < j ? cls[ri][cj] : (i > j ? cls[cj][ri] : 0); // i > j only inside a diagonal tile
    const size_t q = (base + (size_t)i * n + j) * K + k;
    const bool free_ = me ? (me[b * n + i] && me[b * n + j]) : true;
    zs_out[q] = free_ ? (k == c ? 1.f : 0.f) : orig[q];
  }
  if (jt != it)
    for (int t = threadIdx.x; t < 32 * W; t += 256) {              // tile (jt, it), rows j
      const int rj = t / W, o = t - rj * W, ci = o / K, k = o - ci * K, j = jt * 32 + rj, i = it * 32 + ci;
      if (i >= n || j >= n) continue;
      const int c = cls[ci][rj];
      const size_t q = (base + (size_t)j * n + i) * K + k;
      const bool free_ = me ? (me[b * n + i] && me[b * n + j]) : true;
      zs_out[q] = free_ ? (k == c ? 1.f : 0.f) : orig[q];
    }
}

// sample_discrete_features (utils/diffusion/helpers.py:706-752) from explicit probabilities + Exp(1) noise.
// Bit-exact w.r.t. the reference given identical prob and noise tensors.  Output: class indices (int64).
__global__ void k_sample_X(const float* prob, const uint8_t* nm, const float* noise, int64_t* out, int rows, int K) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int cls = 0;
  if (nm[r]) {
    const float* p = prob + (size_t)r * K; const float* q = noise + (size_t)r * K;
    float bv = p[0] / q[0];
    for (int k = 1; k < K; ++k) { float v = p[k] / q[k]; if (v > bv) { bv = v; cls = k; } }
  }
  out[r] = cls;
}
__global__ void k_sample_E(const float* prob, const uint8_t* nm, const float* noise, int64_t* out, int B, int n, int K) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t tot = (size_t)B * n * n;
  if (p >= tot) return;
  int j = p % n; int i = (p / n) % n; int b = p / ((size_t)n * n);
  int cls = 0;
  if (i != j && nm[b * n + i] && nm[b * n + j]) {
    int a = min(i, j), c = max(i, j);
    size_t r = ((size_t)b * n + a) * n + c;
    const float* pp = prob + r * K; const float* q = noise + r * K;
    float bv = pp[0] / q[0];
    for (int k = 1; k < K; ++k) { float v = pp[k] / q[k]; if (v > bv) { bv = v; cls = k; } }
  }
  out[p] = cls;
}
__global__ void k_onehot(const int64_t* idx, float* out, size_t rows, int K) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * K) return;
  out[t] = (int)(t % K) == (int)idx[t / K] ? 1.f : 0.f;
}

// forward noising probabilities (diffusion/diffusion_abstract.py:253-257): prob = onehot/dense row @ Qbar_t, .mask()
__global__ void k_noise_prob_X(const float* X, const uint8_t* nm, const float* Qtb, float* out, int B, int n, int K) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B * n) return;
  int b = r / n;
  const float* x = X + (size_t)r * K; const float* Q = Qtb + (size_t)b * K * K;
  float s = 0.f;
  bool v = nm[r] != 0;
  for (int k = 0; k < K; ++k) {
    float acc = 0.f;
    for (int c = 0; c < K; ++c) acc += x[c] * Q[c * K + k];
    acc = v ? acc : 0.f;
    out[(size_t)r * K + k] = acc; s += acc;
  }
  if (s == 0.f) out[(size_t)r * K] = 1.f;
}
__global__ void k_noise_prob_E(const float* E, const uint8_t* nm, const float* Qtb, float* out, int B, int n, int K) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t tot = (size_t)B * n * n;
  if (p >= tot) return;
  int j = p % n; int i = (p / n) % n; int b = p / ((size_t)n * n);
  const float* x = E + p * K; const float* Q = Qtb + (size_t)b * K * K;
  bool v = nm[b * n + i] && nm[b * n + j] && i != j;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    float acc = 0.f;
    for (int c = 0; c < K; ++c) acc += x[c] * Q[c * K + k];
    acc = v ? acc : 0.f;
    out[p * K + k] = acc; s += acc;
  }
  if (s == 0.f) out[p * K] = 1.f;
}

// ===================================================================================================
static inline dim3 g1(size_t tot, int bs) { return dim3((unsigned)((tot + bs - 1) / bs)); }

template <int K>
static int launch_post_X(const float* logits, const float* zt, const uint8_t* nm, const float* Qt, const float* Qsb,
                         const float* Qtb, float temp, const float* noise, const float* orig, const uint8_t* mx,
                         float* prob, float* zs, int B, int n, hipStream_t s) {
  if constexpr (K > 16 && K <= 32) {
    k_posterior_X_lanes<K><<<dim3(cdiv(n, 8), B), 256, 3 * K * K * sizeof(float), s>>>(logits, zt, nm, Qt, Qsb, Qtb, temp, noise, orig, mx,
                                                                                    prob, zs, B, n);
    return 0;
  }
  dim3 grid(cdiv(n, 128), B);
  k_posterior_X<K><<<grid, 128, 3 * K * K * sizeof(float), s>>>(logits, zt, nm, Qt, Qsb, Qtb, temp, noise, orig, mx, prob,
                                                               zs, B, n);
  return 0;
}
template <int K>
static int launch_post_E(const float* logits, const float* zt, const uint8_t* nm, const float* Qt, const float* Qsb,
                         const float* Qtb, float temp, const float* noise, const float* orig, const uint8_t* me,
                         float* prob, float* zs, int B, int n, hipStream_t s) {
  if constexpr (K <= 8) {
    if (!prob && zs) {
      const int nt = cdiv(n, 32);
      k_posterior_E_tiles<K><<<dim3(nt * (nt + 1) / 2, B), 256, 0, s>>>(logits, zt, nm, Qt, Qsb, Qtb, temp, noise, orig, me, zs, n);
      return 0;
    }
  }
  dim3 grid(cdiv(n * n, 256), B);
  k_posterior_E<K><<<grid, 256, 0, s>>>(logits, zt, nm, Qt, Qsb, Qtb, temp, noise, orig, me, prob, zs, B, n);
  return 0;
}

#define DISPATCH_K(K_, fn, ...)                                                         \
  switch (K_) {                                                                         \
    case 4: fn<4>(__VA_ARGS__); break;   case 5: fn<5>(__VA_ARGS__); break;             \
    case 7: fn<7>(__VA_ARGS__); break;   case 29: fn<29>(__VA_ARGS__); break;           \
    default: da_set_error("unsupported class count %d (built for 4,5,7,29)", K_); return DA_ERR_UNSUPPORTED; \
  }

extern "C" {

int da_mask_dense(float* X, float* E, const uint8_t* node_mask, int B, int n, int dx, int de, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && dx >= 0 && de >= 0, "da_mask_dense: negative size");
  hipStream_t s = (hipStream_t)stream;
  if (B == 0 || n == 0) return DA_OK;
  if (X && dx > 0) k_mask_dense_X<<<g1((size_t)B * n, 256), 256, 0, s>>>(X, node_mask, B * n, dx);
  if (E && de == 64 && (reinterpret_cast<uintptr_t>(E) & 15) == 0) {
    size_t g = ((size_t)B * n * n * 16 + 255) / 256; if (g > 8192) g = 8192;
    k_mask_dense_E4<16><<<(unsigned)g, 256, 0, s>>>(E, node_mask, B, n);
  } else if (E && de > 0) k_mask_dense_E<<<g1((size_t)B * n * n, 256), 256, 0, s>>>(E, node_mask, B, n, de);
  DA_CHECK_LAUNCH("da_mask_dense");
  return DA_OK;
}

int da_collapse(const float* X, const float* E, const uint8_t* node_mask, int64_t* Xi, int64_t* Ei, int B, int n, int dx,
                int de, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && dx > 0 && de > 0, "da_collapse: bad size");
  hipStream_t s = (hipStream_t)stream;
  if (B == 0 || n == 0) return DA_OK;
  k_collapse_X<<<g1((size_t)B * n, 256), 256, 0, s>>>(X, node_mask, Xi, B * n, dx);
  k_collapse_E<<<g1((size_t)B * n * n, 256), 256, 0, s>>>(E, node_mask, Ei, B, n, de);
  DA_CHECK_LAUNCH("da_collapse");
  return DA_OK;
}

int da_free_masks(const float* X_orig, const uint8_t* node_mask, const int64_t* atom_map, int B, int n, int kx, int mode,
                  int suno_idx, int include_supernode, const int* states_dev, int n_states, uint8_t* mask_x,
                  uint8_t* mask_e, void* stream) {
  DA_CHECK_ARG(mode >= 0 && mode <= 5, "da_free_masks: unknown mode %d", mode);
  DA_CHECK_ARG(mode != 5 || atom_map != nullptr, "da_free_masks: atom_mapping mode needs atom_map_numbers");
  DA_CHECK_ARG(n_states == 0 || states_dev != nullptr, "da_free_masks: states pointer is null");
  if (B == 0 || n == 0) return DA_OK;
  k_free_masks<<<B, 256, 0, (hipStream_t)stream>>>(X_orig, node_mask, atom_map, n, kx, mode, suno_idx, include_supernode,
                                                     states_dev, n_states, mask_x, mask_e);
  DA_CHECK_LAUNCH("da_free_masks");
  return DA_OK;
}

int da_apply_mask(float* zX, float* zE, const float* origX, const float* origE, const uint8_t* mask_x,
                  const uint8_t* mask_e, int B, int n, int kx, int ke, int as_logits, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (B == 0 || n == 0) return DA_OK;
  float scale = as_logits ? 100.f : 1.f;
  if (zX) k_apply_mask_X<<<g1((size_t)B * n * kx, 256), 256, 0, s>>>(zX, origX, mask_x, B * n, kx, scale);
  if (zE) k_apply_mask_E<<<g1((size_t)B * n * n * ke, 256), 256, 0, s>>>(zE, origE, mask_e, B, n, ke, scale);
  DA_CHECK_LAUNCH("da_apply_mask");
  return DA_OK;
}

// Fused reverse step: posterior (a14) + categorical draw (a15) + apply_mask (a2).
// prob_* and zs_* are optional outputs (null = skip).  noise_* required when zs_* requested.
int da_posterior_sample(const float* logits_X, const float* logits_E, const float* zt_X, const float* zt_E,
                        const uint8_t* node_mask, const float* Qt_x, const float* Qsb_x, const float* Qtb_x,
                        const float* Qt_e, const float* Qsb_e, const float* Qtb_e, float temp_x, float temp_e,
                        const float* noise_X, const float* noise_E, const float* orig_X, const float* orig_E,
                        const uint8_t* mask_x, const uint8_t* mask_e, float* prob_X, float* prob_E, float* zs_X,
                        float* zs_E, int B, int n, int kx, int ke, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0, "da_posterior_sample: negative size");
  DA_CHECK_ARG(!(zs_X && !noise_X) && !(zs_E && !noise_E), "da_posterior_sample: draws requested without noise");
  DA_CHECK_ARG(!(mask_x && !orig_X) && !(mask_e && !orig_E), "da_posterior_sample: mask given without orig");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  if (prob_X || zs_X) { DISPATCH_K(kx, launch_post_X, logits_X, zt_X, node_mask, Qt_x, Qsb_x, Qtb_x, temp_x, noise_X, orig_X, mask_x, prob_X, zs_X, B, n, s) }
  if (prob_E || zs_E) { DISPATCH_K(ke, launch_post_E, logits_E, zt_E, node_mask, Qt_e, Qsb_e, Qtb_e, temp_e, noise_E, orig_E, mask_e, prob_E, zs_E, B, n, s) }
  DA_CHECK_LAUNCH("da_posterior_sample");
  return DA_OK;
}

int da_sample_discrete(const float* prob_X, const float* prob_E, const uint8_t* node_mask, const float* noise_X,
                       const float* noise_E, int64_t* Xs, int64_t* Es, float* X1h, float* E1h, int B, int n, int kx,
                       int ke, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && kx > 0 && ke > 0, "da_sample_discrete: bad size");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  k_sample_X<<<g1((size_t)B * n, 256), 256, 0, s>>>(prob_X, node_mask, noise_X, Xs, B * n, kx);
  k_sample_E<<<g1((size_t)B * n * n, 256), 256, 0, s>>>(prob_E, node_mask, noise_E, Es, B, n, ke);
  if (X1h) k_onehot<<<g1((size_t)B * n * kx, 256), 256, 0, s>>>(Xs, X1h, (size_t)B * n, kx);
  if (E1h) k_onehot<<<g1((size_t)B * n * n * ke, 256), 256, 0, s>>>(Es, E1h, (size_t)B * n * n, ke);
  DA_CHECK_LAUNCH("da_sample_discrete");
  return DA_OK;
}

int da_noise_probs(const float* X, const float* E, const uint8_t* node_mask, const float* Qtb_x, const float* Qtb_e,
                   float* prob_X, float* prob_E, int B, int n, int kx, int ke, void* stream) {
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  k_noise_prob_X<<<g1((size_t)B * n, 128), 128, 0, s>>>(X, node_mask, Qtb_x, prob_X, B, n, kx);
  k_noise_prob_E<<<g1((size_t)B * n * n, 256), 256, 0, s>>>(E, node_mask, Qtb_e, prob_E, B, n, ke);
  DA_CHECK_LAUNCH("da_noise_probs");
  return DA_OK;
}

}  // extern "C"
