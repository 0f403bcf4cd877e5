// ELBO / CE terms of the diffusion objective (diffalign/diffusion/elbo.py, utils/diffusion/helpers.py): per-node and
// per-pair KL / log-likelihood rows, and the masked means over the free-to-diffuse elements.  HBM-bound elementwise
// work: one thread per node (b,i) or ordered pair (b,i,j); the per-graph transition tables sit in LDS.
// Arithmetic follows the reference's order (this file is built with -ffp-contract=off like k_diffusion.hip).
#include "common.h"
#include <cmath>

enum ElboMode { LT_PROB = 0, LT_LOG = 1, KL_PRIOR = 2, RECON = 3, LOGP = 4 };

template <int K>
__device__ __forceinline__ int argmax_row(const float* p) {
  int best = 0; float bv = p[0];
#pragma unroll
  for (int k = 1; k < K; ++k) { const float v = p[k]; if (v > bv) { bv = v; best = k; } }
  return best;
}
template <int K>
__device__ __forceinline__ void log_softmax_row(const float* x, float* out) {
  float m = x[0];
#pragma unroll
  for (int k = 1; k < K; ++k) m = fmaxf(m, x[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += expf(x[k] - m);
  const float ls = logf(s);
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = (x[k] - m) - ls;
}
// F.kl_div(input, target, reduction='none').sum(-1):  sum_k xlogy(t,t) - t*input
template <int K>
__device__ __forceinline__ float kl_row(const float* target, const float* input) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float t = target[k];
    const float a = (t == 0.f) ? 0.f : t * logf(t);
    s += a - t * input[k];
  }
  return s;
}

// One row.  tr = one-hot truth, zt = one-hot noisy state, lg = logits; tables row-major (K,K).
//  LT_PROB / LT_LOG: elbo.py:85-162 (compute_Lt, log=False / True) with helpers.py:754-783
//  KL_PRIOR:         elbo.py:197-238 + helpers.py:115-137   (Qa = Qbar_T, lim = limit distribution)
//  RECON:            elbo.py:182-193 + helpers.py:139-151   (logits apply_mask(as_logits)-ed, log_softmax, sum true*logp)
//  LOGP:             sum true * log_softmax(logits) without masking (the CE rows of helpers.py:153-191, negated)
template <int K>
__device__ __forceinline__ float elbo_row(int mode, const float* tr, const float* zt, const float* lg, const float* Qa,
                                          const float* Qb, const float* lim, bool free_, float eps) {
  float t[K];
#pragma unroll
  for (int k = 0; k < K; ++k) t[k] = tr[k];
  if (mode == RECON || mode == LOGP) {
    float x[K], lp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = (mode == LOGP || free_) ? lg[k] : t[k] * 100.f;
    log_softmax_row<K>(x, lp);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += t[k] * lp[k];
    return s;
  }
  const int x0 = argmax_row<K>(t);
  if (mode == KL_PRIOR) {
    float p[K], l[K], in[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { p[k] = free_ ? Qa[x0 * K + k] : t[k]; l[k] = free_ ? lim[k] : t[k]; in[k] = logf(p[k] + eps); }
    return kl_row<K>(l, in);
  }
  // Lt: Qa = Q_t, Qb = Qbar_{t-1}
  float z[K];
#pragma unroll
  for (int k = 0; k < K; ++k) z[k] = zt[k];
  const int xt = argmax_row<K>(z);
  float a[K], q[K];
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { a[k] = Qa[k * K + xt]; q[k] = a[k] * Qb[x0 * K + k]; sq += q[k]; }
#pragma unroll
  for (int k = 0; k < K; ++k) q[k] = free_ ? q[k] / sq : t[k];
  float in[K];
  if (mode == LT_PROB) {
    float p0[K];
    {
      float m = lg[0];
#pragma unroll
      for (int k = 1; k < K; ++k) m = fmaxf(m, lg[k]);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) { p0[k] = expf(lg[k] - m); s += p0[k]; }
#pragma unroll
      for (int k = 0; k < K; ++k) p0[k] = p0[k] / s;
    }
    float p[K];
    float sp = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float bp = 0.f;
      for (int j = 0; j < K; ++j) bp += p0[j] * Qb[j * K + k];
      p[k] = a[k] * bp; sp += p[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) in[k] = logf((free_ ? p[k] / sp : t[k]) + eps);
  } else {
    float ml[K], ln[K];
    {
      float x[K];
#pragma unroll
      for (int k = 0; k < K; ++k) x[k] = lg[k];
      log_softmax_row<K>(x, ml);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) ml[k] = expf(ml[k]);
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float bp = 0.f;
      for (int j = 0; j < K; ++j) bp += ml[j] * Qb[j * K + k];
      ln[k] = logf(a[k] + 1e-30f) + logf(bp + 1e-30f);
      m = fmaxf(m, ln[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += expf(ln[k] - m);
    const float lse = m + logf(s);
    float lp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) lp[k] = free_ ? ln[k] - lse : t[k] * 100.f;
    log_softmax_row<K>(lp, in);
  }
  return kl_row<K>(q, in);
}

// rows of one graph: M = n (nodes, IS_E = false) or n*n (ordered pairs)
template <int K, bool IS_E>
__global__ void __launch_bounds__(256) k_elbo_rows(int mode, const float* tr, const float* zt, const float* lg, const float* Qa,
                                                   const float* Qb, int q_per_graph, const float* lim, const uint8_t* fm,
                                                   float eps, float* out, int n) {
  __shared__ float sQ[2 * K * K + K];
  const int b = blockIdx.y;
  for (int t = threadIdx.x; t < K * K; t += blockDim.x) {
    sQ[t] = Qa ? Qa[(q_per_graph ? (size_t)b * K * K : 0) + t] : 0.f;
    sQ[K * K + t] = Qb ? Qb[(q_per_graph ? (size_t)b * K * K : 0) + t] : 0.f;
  }
  for (int t = threadIdx.x; t < K; t += blockDim.x) sQ[2 * K * K + t] = lim ? lim[t] : 0.f;
  __syncthreads();
  const int M = IS_E ? n * n : n;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  bool free_ = true;
  if (fm) free_ = IS_E ? (fm[b * n + p / n] != 0 && fm[b * n + p % n] != 0) : fm[b * n + p] != 0;
  const size_t r = (size_t)b * M + p;
  out[r] = elbo_row<K>(mode, tr + r * K, zt ? zt + r * K : tr + r * K, lg ? lg + r * K : tr + r * K, sQ, sQ + K * K,
                       sQ + 2 * K * K, free_, eps);
}

// helpers.mean_without_masked (helpers.py:59-106): per graph  (sum over free nodes, #free nodes, sum over free pairs, #free pairs)
__global__ void __launch_bounds__(256) k_masked_sums(const float* tX, const float* tE, const uint8_t* mx, const uint8_t* me,
                                                     float* out, int n) {
  __shared__ double red[4][256];
  const int b = blockIdx.x;
  double sX = 0, cX = 0, sE = 0, cE = 0;
  for (int i = threadIdx.x; i < n; i += 256)
    if (mx[b * n + i]) { sX += tX[(size_t)b * n + i]; cX += 1; }
  for (int p = threadIdx.x; p < n * n; p += 256)
    if (me[b * n + p / n] && me[b * n + p % n]) { sE += tE[(size_t)b * n * n + p]; cE += 1; }
  red[0][threadIdx.x] = sX; red[1][threadIdx.x] = cX; red[2][threadIdx.x] = sE; red[3][threadIdx.x] = cE;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) out[b * 4 + threadIdx.x] = (float)red[threadIdx.x][0];
}

template <int K, bool IS_E>
static int launch_rows(int mode, const float* tr, const float* zt, const float* lg, const float* Qa, const float* Qb, int qpg,
                       const float* lim, const uint8_t* fm, float eps, float* out, int B, int n, hipStream_t s) {
  const int M = IS_E ? n * n : n;
  k_elbo_rows<K, IS_E><<<dim3((M + 255) / 256, B), 256, 0, s>>>(mode, tr, zt, lg, Qa, Qb, qpg, lim, fm, eps, out, n);
  return 0;
}
#define ELBO_DISPATCH(K_, IS_E_, ...)                                                                     \
  switch (K_) {                                                                                           \
    case 4: launch_rows<4, IS_E_>(__VA_ARGS__); break;   case 5: launch_rows<5, IS_E_>(__VA_ARGS__); break;  \
    case 7: launch_rows<7, IS_E_>(__VA_ARGS__); break;   case 29: launch_rows<29, IS_E_>(__VA_ARGS__); break; \
    default: da_set_error("unsupported class count %d (built for 4,5,7,29)", K_); return DA_ERR_UNSUPPORTED; \
  }

extern "C" {

// mode: 0 L_t rows (probability space), 1 L_t rows (log space), 2 KL-prior rows, 3 reconstruction log-p rows,
// 4 plain sum true*log_softmax(logits) rows.  Q tables: modes 0/1 Qa = Q_t, Qb = Qbar_{t-1} per graph (B,K,K);
// mode 2 Qa = Qbar_T as ONE (K,K) table (q_per_graph = 0) with lim_* the limit distributions.  Unused pointers may be NULL.
int da_elbo_rows(int mode, const float* true_X, const float* true_E, const float* zt_X, const float* zt_E,
                 const float* logits_X, const float* logits_E, const float* Qa_x, const float* Qb_x, const float* Qa_e,
                 const float* Qb_e, int q_per_graph, const float* lim_x, const float* lim_e, const uint8_t* mask_x,
                 const uint8_t* mask_e, float eps, float* rows_X, float* rows_E, int B, int n, int kx, int ke, void* stream) {
  DA_CHECK_ARG(mode >= 0 && mode <= 4, "da_elbo_rows: mode %d", mode);
  DA_CHECK_ARG(B >= 0 && n >= 0, "da_elbo_rows: negative size");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  if (rows_X) { ELBO_DISPATCH(kx, false, mode, true_X, zt_X, logits_X, Qa_x, Qb_x, q_per_graph, lim_x, mask_x, eps, rows_X, B, n, s) }
  if (rows_E) { ELBO_DISPATCH(ke, true, mode, true_E, zt_E, logits_E, Qa_e, Qb_e, q_per_graph, lim_e, mask_e, eps, rows_E, B, n, s) }
  DA_CHECK_LAUNCH("da_elbo_rows");
  return DA_OK;
}

// out (B,4) = per graph: sum of rows_X over free nodes, their count, sum of rows_E over free pairs, their count
int da_masked_sums(const float* rows_X, const float* rows_E, const uint8_t* mask_x, const uint8_t* mask_e, float* out, int B,
                   int n, void* stream) {
  if (B == 0) return DA_OK;
  k_masked_sums<<<B, 256, 0, (hipStream_t)stream>>>(rows_X, rows_E, mask_x, mask_e, out, n);
  DA_CHECK_LAUNCH("da_masked_sums");
  return DA_OK;
}

}  // extern "C"
