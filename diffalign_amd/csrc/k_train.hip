// Training step (diffusion_abstract.py:134-225), first stage of the backward: the gradient of the CE loss with
// respect to the denoiser's logits and the gradients of the output-stage skip scalings.  HBM-bound row kernels, fp32.
#include "common.h"

// F.cross_entropy(reduction='mean') over the free rows (helpers.py:176-189):  d loss / d logit[r][k] =
//   w * (softmax(logit[r])[k] - [k == argmax true[r]])   on free rows,   0 elsewhere,
// with w = wX / #free nodes of the batch (nodes) or wE / #free pairs (pairs).  sums (B,4) comes from da_masked_sums.
template <int K, bool IS_E>
__global__ void __launch_bounds__(256) k_ce_grad(const float* __restrict__ tr, const float* __restrict__ lg,
                                                 const uint8_t* __restrict__ fm, const float* __restrict__ sums, float scale,
                                                 float* __restrict__ grad, int B, int n) {
  __shared__ float s_w;
  if (threadIdx.x == 0) {
    double cnt = 0.0;
    for (int b = 0; b < B; ++b) cnt += sums[b * 4 + (IS_E ? 3 : 1)];
    s_w = cnt > 0.0 ? (float)((double)scale / cnt) : 0.f;
  }
  __syncthreads();
  const float w = s_w;
  const int b = blockIdx.y, M = IS_E ? n * n : n;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const bool free_ = IS_E ? (fm[b * n + p / n] != 0 && fm[b * n + p % n] != 0) : fm[b * n + p] != 0;
  const size_t r = ((size_t)b * M + p) * K;
  if (!free_) {
#pragma unroll
    for (int k = 0; k < K; ++k) grad[r + k] = 0.f;
    return;
  }
  float x[K];
  float m = -INFINITY;
  int cls = 0; float bv = tr[r];
#pragma unroll
  for (int k = 0; k < K; ++k) { x[k] = lg[r + k]; m = fmaxf(m, x[k]); const float t = tr[r + k]; if (t > bv) { bv = t; cls = k; } }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { x[k] = expf(x[k] - m); s += x[k]; }
  const float inv = 1.f / s;
#pragma unroll
  for (int k = 0; k < K; ++k) grad[r + k] = w * (x[k] * inv - (k == cls ? 1.f : 0.f));
}

template <int K, bool IS_E>
static void launch_ce_grad(const float* tr, const float* lg, const uint8_t* fm, const float* sums, float scale, float* grad, int B,
                           int n, hipStream_t s) {
  const int M = IS_E ? n * n : n;
  k_ce_grad<K, IS_E><<<dim3((M + 255) / 256, B), 256, 0, s>>>(tr, lg, fm, sums, scale, grad, B, n);
}
#define CE_DISPATCH(K_, IS_E_, ...)                                                                               \
  switch (K_) {                                                                                                   \
    case 4: launch_ce_grad<4, IS_E_>(__VA_ARGS__); break;   case 5: launch_ce_grad<5, IS_E_>(__VA_ARGS__); break;  \
    case 7: launch_ce_grad<7, IS_E_>(__VA_ARGS__); break;   case 29: launch_ce_grad<29, IS_E_>(__VA_ARGS__); break; \
    default: da_set_error("unsupported class count %d (built for 4,5,7,29)", K_); return DA_ERR_UNSUPPORTED;      \
  }

extern "C" {

int da_ce_grad(const float* true_X, const float* true_E, const float* logits_X, const float* logits_E, const uint8_t* mask_x,
               const uint8_t* mask_e, const float* sums, float weight_X, float weight_E, float* grad_X, float* grad_E, int B, int n,
               int kx, int ke, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0, "da_ce_grad: negative size");
  DA_CHECK_ARG(sums != nullptr && mask_x != nullptr && mask_e != nullptr, "da_ce_grad: sums / masks are null");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  if (grad_X) { CE_DISPATCH(kx, false, true_X, logits_X, mask_x, sums, weight_X, grad_X, B, n, s) }
  if (grad_E) { CE_DISPATCH(ke, true, true_E, logits_E, mask_e, sums, weight_E, grad_E, B, n, s) }
  DA_CHECK_LAUNCH("da_ce_grad");
  return DA_OK;
}

}  // extern "C"
