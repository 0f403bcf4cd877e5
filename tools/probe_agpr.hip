// Does the register class of the MFMA accumulator change how much VALU work a SIMD can issue beside the matrix pipe?
// One loop iteration = NM x v_mfma_f32_32x32x16_bf16, each followed by NV v_fma_f32 (independent), for one and two waves per SIMD,
// accumulators in VGPRs ("v") or AGPRs ("a").  Prints cycles per MFMA.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int NV, bool AGPR, int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS) k(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + threadIdx.x % 3); b[j] = (short)0x3F80; }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 1.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
  const float c0 = 1.0001f, c1 = 0.0001f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (AGPR) {
        if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
      } else {
        if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m * NV + u) % 16]) : "v"(c0), "v"(c1));
        if (KIND == 1) { f32x2& p = *reinterpret_cast<f32x2*>(&v[2 * ((m * NV + u) % 8)]); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1})); }
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(m * NV + u) % 16]));
      }
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, bool AGPR, int KIND, int THREADS>
void run(const char* what) {
  float* out; long long* cyc; long long h = 0;
  hipMalloc(&out, 1024 * THREADS * sizeof(float)); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int r = 0; r < 2; ++r) k<NV, AGPR, KIND, THREADS><<<1024, THREADS>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-9s acc in %s, %d waves/SIMD, %2d %s per MFMA: %6.1f cycles per MFMA\n", what, AGPR ? "AGPR" : "VGPR", THREADS / 256, NV,
         KIND == 0 ? "v_fma_f32" : KIND == 1 ? "v_pk_fma_f32" : "v_exp_f32", (double)h / iters / 4);
  hipFree(out); hipFree(cyc);
}
#define ROW(NV, KIND) run<NV, false, KIND, 256>(""); run<NV, true, KIND, 256>(""); run<NV, false, KIND, 512>(""); run<NV, true, KIND, 512>("");
int main() {
  ROW(0, 0) ROW(4, 0) ROW(8, 0) ROW(12, 0) ROW(16, 0)
  ROW(4, 1) ROW(8, 1)
  ROW(2, 2) ROW(4, 2)
  return 0;
}
