// SIMD-level issue throughput on gfx950 with 1..4 waves per SIMD: pure VALU streams and a realistic MFMA + VALU mix
// (per MFMA: NF v_fma_f32, NP v_pk_fma_f32, NE v_exp_f32, NC v_cvt_pk_bf16_f32).  Long unrolled bodies, so that loop
// overhead does not matter.  hipcc --offload-arch=gfx950 -O3 -w -o tools/bin/probe_issue tools/probe_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int NM, int NF, int NP, int NE, int NC, int REP>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + threadIdx.x % 3); b[j] = (short)0x3F80; }
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 1.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
  unsigned w[4] = {0, 0, 0, 0};
  const float c0 = 1.0001f, c1 = 0.0001f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < REP; ++rep) {
      if (NM) acc[rep & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[rep & 1], 0, 0, 0);
      int q = rep * 5;
#pragma unroll
      for (int u = 0; u < NF; ++u, ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q % 16]) : "v"(c0), "v"(c1));
#pragma unroll
      for (int u = 0; u < NP; ++u, ++q) { f32x2& p = *reinterpret_cast<f32x2*>(&v[2 * (q % 8)]); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1})); }
#pragma unroll
      for (int u = 0; u < NE; ++u, ++q) asm volatile("v_exp_f32 %0, %0" : "+v"(v[q % 16]));
#pragma unroll
      for (int u = 0; u < NC; ++u, ++q) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[u % 4]) : "v"(v[q % 16]), "v"(v[(q + 1) % 16]));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(w[0] ^ w[1] ^ w[2] ^ w[3]);
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NM, int NF, int NP, int NE, int NC>
void run(const char* what, float* out, long long* cyc) {
  constexpr int REP = 16;
  const int iters = 500;
  printf("%-32s", what);
  for (int wps = 1; wps <= 4; ++wps) {
    hipLaunchKernelGGL((k<NM, NF, NP, NE, NC, REP>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c[16]; hipMemcpy(c, cyc, 8 * 4 * wps, hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 60;
    for (int q = 0; q < 4 * wps; ++q) { mx = c[q] > mx ? c[q] : mx; mn = c[q] < mn ? c[q] : mn; }
    // slowest wave's cycles per group; SIMD cycles per group = that / waves per SIMD (all waves run the same stream)
    printf("  %dw: %5.1f-%5.1f /%5.1f", wps, (double)mn / iters / REP, (double)mx / iters / REP, (double)mx / iters / REP / wps);
  }
  printf("\n");
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16 * 8);
  run<0, 8, 0, 0, 0>("8 fma", out, cyc);
  run<0, 0, 8, 0, 0>("8 pk_fma", out, cyc);
  run<0, 0, 0, 8, 0>("8 exp", out, cyc);
  run<0, 0, 0, 0, 8>("8 cvt_pk", out, cyc);
  run<1, 0, 0, 0, 0>("1 mfma", out, cyc);
  run<1, 4, 0, 0, 0>("1 mfma + 4 fma", out, cyc);
  run<1, 8, 0, 0, 0>("1 mfma + 8 fma", out, cyc);
  run<1, 12, 0, 0, 0>("1 mfma + 12 fma", out, cyc);
  run<1, 6, 2, 1, 1>("1 mfma + 6 fma 2 pk 1 exp 1 cvt", out, cyc);
  run<1, 10, 0, 1, 1>("1 mfma + 10 fma 1 exp 1 cvt", out, cyc);
  run<1, 2, 4, 1, 1>("1 mfma + 2 fma 4 pk 1 exp 1 cvt", out, cyc);
  run<0, 6, 2, 1, 1>("6 fma 2 pk 1 exp 1 cvt", out, cyc);
  run<0, 10, 0, 1, 1>("10 fma 1 exp 1 cvt", out, cyc);
  return 0;
}
