// MFMA 32x32x16 bf16 helpers (gfx950).  Lane maps verified on hardware by tools/probe_mfma.hip:
//   A operand: lane l (r=l&31, h=l>>5) holds A[row r][k = 8h + j], j=0..7
//   B operand: lane l holds B[k = 8h + j][col r]
//   C/D:       lane l, reg i holds D[row crow(i,h)][col r],  crow(i,h) = (i&3) + 8(i>>2) + 4h
// "C-order": when 8 consecutive accumulator registers 8s..8s+7 are packed to bf16 and fed back as the A or B operand
// of k-step s, element j of lane-half h carries k index  corder(s,h,j) = 16s + 8(j>>2) + 4h + (j&3).
// Every weight operand that multiplies such a fragment is pre-packed on the host in the same order
// (diffalign_amd/pack.py), so weight fragments are plain lane-linear 16-byte reads.
#pragma once
#include "common.h"

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = v;
  return r;
}
__device__ __forceinline__ bf16x8 frag_from_u4(uint4 u) { return __builtin_bit_cast(bf16x8, u); }

// pack accumulator registers 8s..8s+7 into a bf16 fragment (C-order k)
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s) {
  uint4 u;
  u.x = pack_bf16(a[8 * s + 0], a[8 * s + 1]);
  u.y = pack_bf16(a[8 * s + 2], a[8 * s + 3]);
  u.z = pack_bf16(a[8 * s + 4], a[8 * s + 5]);
  u.w = pack_bf16(a[8 * s + 6], a[8 * s + 7]);
  return frag_from_u4(u);
}
// lane-linear 16-byte fragment read from a (1 KiB per fragment) LDS image
__device__ __forceinline__ bf16x8 lds_frag(const uint4* base, int frag, int lane) { return frag_from_u4(base[frag * 64 + lane]); }

// exchange between the two lane halves: lanes 32-63 of a <-> lanes 0-31 of b
__device__ __forceinline__ void half_swap(unsigned& a, unsigned& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
// 16 bytes at [16u + 8h .. +8) of a row, both halves  ->  C-order fragment of k-step u (and back: involution)
__device__ __forceinline__ void row16_to_corder(uint4& v) { half_swap(v.x, v.z); half_swap(v.y, v.w); }

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

// identity operand for the in-register transpose  T = X^T  (X's accumulators as the A operand)
__device__ __forceinline__ bf16x8 ident_frag(int s, int r, int h) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (r == 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) ? (short)0x3F80 : (short)0;
  return f;
}

// K/V operand words of the edge-layer kernel: 8 consecutive accumulator-order values -> 4 dwords.  Words (2m, 2m+1)
// carry elements (4m, 4m+1) in their low halves and (4m+2, 4m+3) in their high halves.  The kernel reads a high half as an
// fp32 WITHOUT masking the low half off, so the high half is chosen such that the whole word, seen as fp32, is the value
// nearest to the element (error <= half a bf16 ulp, as for round-to-nearest-even).
__device__ __forceinline__ uint4 kv_words(const float (&v)[8]) {
  unsigned w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int mm = q >> 1, t2 = q & 1;
    const unsigned lo = pack_bf16(v[4 * mm + t2], 0.f) & 0xffffu;
    const unsigned tb = __builtin_bit_cast(unsigned, v[4 * mm + 2 + t2]);
    const unsigned sgn = tb & 0x80000000u, mag = tb & 0x7fffffffu;
    const unsigned hm = mag >= lo ? (mag - lo + 0x8000u) >> 16 : 0u;
    w[q] = sgn | (hm << 16) | lo;
  }
  return uint4{w[0], w[1], w[2], w[3]};
}
