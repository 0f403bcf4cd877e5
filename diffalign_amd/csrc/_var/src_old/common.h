// Shared device/host helpers for the diffalign_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DA_OK 0
#define DA_ERR_ARG -1
#define DA_ERR_HIP -2
#define DA_ERR_UNSUPPORTED -3
#define DA_MAX_DEVICES 64

void da_set_error(const char* fmt, ...);

#define DA_CHECK_ARG(cond, ...)                  \
  do {                                           \
    if (!(cond)) {                               \
      da_set_error(__VA_ARGS__);                 \
      return DA_ERR_ARG;                         \
    }                                            \
  } while (0)

#define DA_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      da_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return DA_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 hwbf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// row of accumulator register i (0..15) in lane-half h for the 32x32 MFMA C/D layout
__device__ __forceinline__ constexpr int crow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  hwbf16x2 b = __builtin_convertvector(v, hwbf16x2);  // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, ((unsigned)b) << 16); }

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// (it, jt), it <= jt: the idx-th pair of mirrored tiles, row-major over the upper triangle of an nt x nt tile grid
// (grids of nt (nt + 1) / 2 workgroups instead of nt x nt with the lower half exiting at once)
__device__ __forceinline__ void tri_tile(int idx, int nt, int& it, int& jt) {
  it = 0;
  while (idx >= nt - it) { idx -= nt - it; ++it; }
  jt = it + idx;
}
