// Probe of the gfx950 MFMA 32x32x16 bf16 lane maps this repo's kernels rely on.
// Exact small-integer data, so every check is bit-exact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __host__ inline unsigned short f2bf(float f) {
  union { float f; uint32_t u; } v; v.f = f;
  uint32_t u = v.u; u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16);
}
__device__ inline int crow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// test 1: plain C = A(32x16) * B(16x32) with the documented maps
__global__ void k_plain(const float* A, const float* B, float* C) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A[r * 16 + 8 * h + j]); b[j] = f2bf(B[(8 * h + j) * 32 + r]); }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[crow(i, h) * 32 + r] = c[i];
}
// test 2: X = A1*B1 (32x32, K=16), then Y = W(32x32) * X using X's accumulators as B operand,
// with W supplied in "C-order" k permutation.  Y[m][n] = sum_k W[m][k] X[k][n]
__global__ void k_chain_B(const float* A1, const float* B1, const float* W, float* Y) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A1[r * 16 + 8 * h + j]); b[j] = f2bf(B1[(8 * h + j) * 32 + r]); }
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 y = {0};
  for (int s = 0; s < 2; ++s) {
    bf16x8 xb, wa;
    for (int j = 0; j < 8; ++j) {
      xb[j] = f2bf(x[8 * s + j]);
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      wa[j] = f2bf(W[r * 32 + k]);
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, y, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) Y[crow(i, h) * 32 + r] = y[i];
}
// test 3: transposition via permuted identity: T = X^T using X accumulators as A operand
__global__ void k_transpose(const float* A1, const float* B1, float* T) {
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = f2bf(A1[r * 16 + 8 * h + j]); b[j] = f2bf(B1[(8 * h + j) * 32 + r]); }
  f32x16 x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 t = {0};
  for (int s = 0; s < 2; ++s) {
    bf16x8 xa, ib;
    for (int j = 0; j < 8; ++j) {
      xa[j] = f2bf(x[8 * s + j]);
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      ib[j] = f2bf(k == r ? 1.0f : 0.0f);
    }
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, ib, t, 0, 0, 0);
  }
  // t should be X^T in C layout: t[i] = T[row=crow(i,h)][col=r] = X[r][crow(i,h)]
  for (int i = 0; i < 16; ++i) T[crow(i, h) * 32 + r] = t[i];
}
// test 4: permlane32_swap semantics
__global__ void k_swap(unsigned* out) {
  int l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  auto rr = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = rr[0]; out[64 + l] = rr[1];
}

static int check(const char* name, const std::vector<float>& got, const std::vector<float>& ref) {
  int bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) if (got[i] != ref[i]) { if (bad < 5) printf("  %s mismatch at %zu: got %g want %g\n", name, i, got[i], ref[i]); ++bad; }
  printf("%s: %s (%d bad)\n", name, bad ? "FAIL" : "PASS", bad);
  return bad;
}

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), W(32 * 32);
  srand(1);
  for (auto& v : A) v = (float)(rand() % 7 - 3);
  for (auto& v : B) v = (float)(rand() % 5 - 2);
  for (auto& v : W) v = (float)(rand() % 3 - 1);
  std::vector<float> X(32 * 32, 0.f), Y(32 * 32, 0.f), XT(32 * 32);
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += A[m * 16 + k] * B[k * 32 + n]; X[m * 32 + n] = s; }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += W[m * 32 + k] * X[k * 32 + n]; Y[m * 32 + n] = s; XT[m * 32 + n] = X[n * 32 + m]; }
  float *dA, *dB, *dW, *dO; unsigned* dS;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dO, 32 * 32 * 4); hipMalloc(&dS, 128 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> got(32 * 32);
  int bad = 0;
  k_plain<<<1, 64>>>(dA, dB, dO); hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost); bad += check("plain_AB_C_maps", got, X);
  k_chain_B<<<1, 64>>>(dA, dB, dW, dO); hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost); bad += check("acc_as_B_operand", got, Y);
  k_transpose<<<1, 64>>>(dA, dB, dO); hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost); bad += check("identity_transpose", got, XT);
  k_swap<<<1, 64>>>(dS); std::vector<unsigned> sw(128); hipMemcpy(sw.data(), dS, 128 * 4, hipMemcpyDeviceToHost);
  int sb = 0;
  for (int l = 0; l < 64; ++l) {
    unsigned wa = l < 32 ? 1000 + l : 2000 + (l - 32);   // lanes 32-63 of a <- lanes 0-31 of b
    unsigned wb = l < 32 ? 1000 + (l + 32) : 2000 + l;   // lanes 0-31 of b <- lanes 32-63 of a
    if (sw[l] != wa || sw[64 + l] != wb) { if (sb < 4) printf("  swap lane %d: got (%u,%u) want (%u,%u)\n", l, sw[l], sw[64 + l], wa, wb); ++sb; }
  }
  printf("permlane32_swap: %s\n", sb ? "FAIL" : "PASS"); bad += sb;
  hipError_t e = hipDeviceSynchronize();
  printf("hip status: %s\n", hipGetErrorString(e));
  return bad ? 1 : 0;
}
