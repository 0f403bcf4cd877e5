// How do MFMA and VALU work of ONE wave, and of two waves sharing a SIMD, overlap on gfx950?
// Loop body: 1 x v_mfma_f32_32x32x16_bf16 (dependent accumulator chain) + NV independent VALU instructions of a kind.
// Prints cycles per iteration (s_memtime) for 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int NV, int KIND, int NM, bool IND = false>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + threadIdx.x % 3); b[j] = (short)0x3F80; }
  f32x16 acc, accB;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accB[i] = 1.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
  const float c0 = 1.0001f, c1 = 0.0001f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (IND && (m & 1)) accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accB, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      if (IND) {                                   // VALU spread behind each MFMA
#pragma unroll
        for (int u = 0; u < NV / NM; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m * (NV / NM) + u) % 16]) : "v"(c0), "v"(c1));
      }
    }
#pragma unroll
    for (int u = 0; u < (IND ? 0 : NV); ++u) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u % 16]) : "v"(c0), "v"(c1));
      if (KIND == 1) { f32x2& p = *reinterpret_cast<f32x2*>(&v[2 * (u % 8)]); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1})); }
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u % 16]));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i] + accB[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// role split: waves 0-3 (one per SIMD) issue only MFMAs, waves 4-7 (their SIMD partners) only VALU
template <int KIND>
__global__ void __launch_bounds__(512) k_roles(float* out, long long* cyc, int iters, int mode) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + threadIdx.x % 3); b[j] = (short)0x3F80; }
  f32x16 acc, accB;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accB[i] = 1.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
  const float c0 = 1.0001f, c1 = 0.0001f;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0) ? wave < 4 : (mode == 1 ? true : false);
  const bool do_valu = (mode == 0) ? wave >= 4 : (mode == 1 ? false : true);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accB, 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u]) : "v"(c0), "v"(c1));
        if (KIND == 1) { f32x2& p = *reinterpret_cast<f32x2*>(&v[2 * (u % 8)]); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1})); }
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i] + accB[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

template <int NV, int KIND, int NM, bool IND = false>
void run(const char* kind, float* out, long long* cyc) {
  const int iters = 2000;
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k<NV, KIND, NM, IND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-8s MFMA/iter %d  VALU/iter %2d  waves/SIMD %d : %7.1f memtime-ticks/iter\n", kind, NM, NV, threads / 256, (double)c / iters);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  // calibration: memtime ticks vs wall
  run<0, 0, 1>("mfma", out, cyc);
  run<8, 0, 0>("fma", out, cyc); run<8, 1, 0>("pk_fma", out, cyc); run<8, 2, 0>("exp", out, cyc);
  run<4, 0, 1>("fma", out, cyc); run<8, 0, 1>("fma", out, cyc); run<12, 0, 1>("fma", out, cyc); run<16, 0, 1>("fma", out, cyc);
  run<4, 1, 1>("pk_fma", out, cyc); run<8, 1, 1>("pk_fma", out, cyc); run<12, 1, 1>("pk_fma", out, cyc);
  run<4, 2, 1>("exp", out, cyc); run<8, 2, 1>("exp", out, cyc);
  printf("two independent accumulators, VALU spread behind each MFMA:\n");
  run<0, 0, 2, true>("ind", out, cyc); run<4, 0, 2, true>("ind", out, cyc); run<8, 0, 2, true>("ind", out, cyc);
  run<12, 0, 2, true>("ind", out, cyc); run<16, 0, 2, true>("ind", out, cyc); run<24, 0, 2, true>("ind", out, cyc);
  printf("role split on each SIMD (2 MFMA per iteration in waves 0-3, 16 VALU per iteration in waves 4-7):\n");
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL((k_roles<0>), dim3(256), dim3(512), 0, 0, out, cyc, 2000, mode);
    hipDeviceSynchronize();
    long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("  fma    mode %d (0 split, 1 all waves MFMA, 2 all waves VALU): wave0 %.1f  wave4 %.1f ticks/iter\n", mode, c[0] / 2000.0, c[1] / 2000.0);
    hipLaunchKernelGGL((k_roles<1>), dim3(256), dim3(512), 0, 0, out, cyc, 2000, mode);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("  pk_fma mode %d: wave0 %.1f  wave4 %.1f ticks/iter\n", mode, c[0] / 2000.0, c[1] / 2000.0);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((k<0, 0, 1>), dim3(256), dim3(256), 0, 0, out, cyc, 200000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("calibration: %lld ticks in %.3f ms -> %.1f MHz tick rate; 200000 MFMA -> %.2f ns each\n", c, ms, c / ms / 1e3, ms * 1e6 / 200000);
  return 0;
}
