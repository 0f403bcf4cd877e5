// Node side of one XEyTransformerLayer after the attention (neuralnet/layers.py:268-273 x_out + mask, :343-351 the X branch):
//   newX = x_out(newX_in) * m;  X1 = LN1(X + newX);  X' = LN2(X1 + linX2(relu(linX1(X1))))
// fused into one kernel: the three 256-wide GEMMs are chained through MFMA accumulators exactly like the edge FFN
// (activations transposed: channel in registers, row on the lane; out^T = W in^T), so the row tile is read once
// (newX_in, X) and written once (X'); the separate kernels moved five fp32 (R,dx) tensors through HBM.
// One wave = 32 rows, weight fragments stream from L2 (pre-packed 1 KiB lane-linear fragments), one wave per SIMD
// (the chain keeps ~300 registers live).
#include "mfma_util.h"

template <int DX, int DFF>
struct XPostCfg {
  static constexpr int MT = DX / 32, KSX = DX / 16, HT = DFF / 32, KSF = DFF / 16;
  static constexpr int FR_XO = 0, FR_W1 = MT * KSX, FR_W2 = FR_W1 + HT * KSX, N_FR = FR_W2 + MT * KSF;
  // fp32 params after the fragments
  static constexpr int P_BXO = 0, P_G1 = DX, P_B1N = 2 * DX, P_BF1 = 3 * DX, P_BF2 = 3 * DX + DFF, P_G2 = 4 * DX + DFF,
                       P_B2N = 5 * DX + DFF, N_PAR = 6 * DX + DFF;
  static constexpr size_t BLOB_BYTES = (size_t)N_FR * 1024 + (size_t)N_PAR * 4;
};

#ifndef XP_WPB_
#define XP_WPB_ 4
#endif
// waves per workgroup.  Alone, 600 one-wave workgroups spread over all CUs are faster (35.6 us against 40.1 us for 4 waves on 150 CUs),
// but every one of them takes a whole SIMD's registers, and the y chain's 1,024-thread workgroups that run beside this kernel then find
// no CU to start on (engine._layers_node_on_main): with 4 the step is 1 - 2.6 % shorter.
constexpr int XP_WPB = XP_WPB_;
#ifdef EXP_XP_NOW            // timing hook (tools/build_variant.sh): every weight fragment is fragment 0 -> the kernel without its weight stream
#define XP_FR(i) 0
#else
#define XP_FR(i) (i)
#endif
template <int DX, int DFF>
__global__ void __launch_bounds__(64 * XP_WPB) k_x_post(const float* __restrict__ newX_in, const float* __restrict__ X,
                                                const uint8_t* __restrict__ mask, const uint4* __restrict__ blob,
                                                float* __restrict__ out, int R) {
  using C = XPostCfg<DX, DFF>;
  __shared__ __attribute__((aligned(16))) float sP[C::N_PAR];
  {
    const float* gp = reinterpret_cast<const float*>(blob + (size_t)C::N_FR * 64);
    for (int t = threadIdx.x; t < C::N_PAR; t += 64 * XP_WPB) sP[t] = gp[t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = (blockIdx.x * XP_WPB + wave) * 32;
  if (m0 >= R) return;
  const int row = min(m0 + r, R - 1);
  const bool rvalid = m0 + r < R;
  const uint4* fr = blob + lane;

  // ---- activation operand of x_out: this lane's row, natural k order
  bf16x8 inB[C::KSX];
  {
    const float* ap = newX_in + (size_t)row * DX + 8 * h;
#pragma unroll
    for (int s = 0; s < C::KSX; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * s), a1 = *reinterpret_cast<const float4*>(ap + 16 * s + 4);
      const uint4 u = {pack_bf16(a0.x, a0.y), pack_bf16(a0.z, a0.w), pack_bf16(a1.x, a1.y), pack_bf16(a1.z, a1.w)};
      inB[s] = frag_from_u4(u);
    }
  }
  const float mrow = mask[row] ? 1.f : 0.f;
  // ---- x_out, mask, residual
  f32x16 x1[C::MT];
  float sum = 0.f;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    f32x16 acc = splat16(0.f);
#pragma unroll
    for (int s = 0; s < C::KSX; ++s) acc = mfma32(frag_from_u4(fr[XP_FR(C::FR_XO + mt * C::KSX + s) * 64]), inB[s], acc);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * mt + 8 * g + 4 * h;
      const float4 xr = *reinterpret_cast<const float4*>(X + (size_t)row * DX + c);
      const float4 bv = *reinterpret_cast<const float4*>(sP + C::P_BXO + c);
      acc[4 * g + 0] = fmaf(mrow, acc[4 * g + 0] + bv.x, xr.x);
      acc[4 * g + 1] = fmaf(mrow, acc[4 * g + 1] + bv.y, xr.y);
      acc[4 * g + 2] = fmaf(mrow, acc[4 * g + 2] + bv.z, xr.z);
      acc[4 * g + 3] = fmaf(mrow, acc[4 * g + 3] + bv.w, xr.w);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[i];
    x1[mt] = acc;
  }
  // ---- LN1
  {
    const float mean = xhalf_sum(sum) * (1.f / DX);
    float vs = 0.f;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float d = x1[mt][i] - mean; vs = fmaf(d, d, vs); }
    const float rstd = rsqrtf(xhalf_sum(vs) * (1.f / DX) + 1e-5f);
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * mt + 8 * g + 4 * h;
        const float4 gv = *reinterpret_cast<const float4*>(sP + C::P_G1 + c), bv = *reinterpret_cast<const float4*>(sP + C::P_B1N + c);
        x1[mt][4 * g + 0] = fmaf((x1[mt][4 * g + 0] - mean) * rstd, gv.x, bv.x);
        x1[mt][4 * g + 1] = fmaf((x1[mt][4 * g + 1] - mean) * rstd, gv.y, bv.y);
        x1[mt][4 * g + 2] = fmaf((x1[mt][4 * g + 2] - mean) * rstd, gv.z, bv.z);
        x1[mt][4 * g + 3] = fmaf((x1[mt][4 * g + 3] - mean) * rstd, gv.w, bv.w);
      }
  }
  // ---- FFN chained through accumulators: H^T = relu(W1 X1^T + b1), O^T = W2 H^T + b2
  bf16x8 hB[C::KSF];
  {
    bf16x8 xb[C::KSX];
#pragma unroll
    for (int s = 0; s < C::KSX; ++s) xb[s] = pack_acc(x1[s >> 1], s & 1);
#pragma unroll
    for (int ht = 0; ht < C::HT; ++ht) {
      f32x16 hacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(sP + C::P_BF1 + 32 * ht + 8 * g + 4 * h);
        hacc[4 * g + 0] = bv.x; hacc[4 * g + 1] = bv.y; hacc[4 * g + 2] = bv.z; hacc[4 * g + 3] = bv.w;
      }
#pragma unroll
      for (int s = 0; s < C::KSX; ++s) hacc = mfma32(frag_from_u4(fr[XP_FR(C::FR_W1 + ht * C::KSX + s) * 64]), xb[s], hacc);
      // relu on the packed bf16 words (as 16-bit integers negative floats are negative)
      hB[2 * ht + 0] = __builtin_elementwise_max(pack_acc(hacc, 0), (bf16x8)(short)0);
      hB[2 * ht + 1] = __builtin_elementwise_max(pack_acc(hacc, 1), (bf16x8)(short)0);
    }
  }
  float sum2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    f32x16 o;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(sP + C::P_BF2 + 32 * mt + 8 * g + 4 * h);
      o[4 * g + 0] = bv.x; o[4 * g + 1] = bv.y; o[4 * g + 2] = bv.z; o[4 * g + 3] = bv.w;
    }
#pragma unroll
    for (int s = 0; s < C::KSF; ++s) o = mfma32(frag_from_u4(fr[XP_FR(C::FR_W2 + mt * C::KSF + s) * 64]), hB[s], o);
#pragma unroll
    for (int i = 0; i < 16; ++i) { x1[mt][i] += o[i]; sum2 += x1[mt][i]; }
  }
  // ---- LN2 and store
  const float mean2 = xhalf_sum(sum2) * (1.f / DX);
  float vs2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = x1[mt][i] - mean2; vs2 = fmaf(d, d, vs2); }
  const float rstd2 = rsqrtf(xhalf_sum(vs2) * (1.f / DX) + 1e-5f);
  if (rvalid) {
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * mt + 8 * g + 4 * h;
        const float4 gv = *reinterpret_cast<const float4*>(sP + C::P_G2 + c), bv = *reinterpret_cast<const float4*>(sP + C::P_B2N + c);
        float4 y;
        y.x = fmaf((x1[mt][4 * g + 0] - mean2) * rstd2, gv.x, bv.x);
        y.y = fmaf((x1[mt][4 * g + 1] - mean2) * rstd2, gv.y, bv.y);
        y.z = fmaf((x1[mt][4 * g + 2] - mean2) * rstd2, gv.z, bv.z);
        y.w = fmaf((x1[mt][4 * g + 3] - mean2) * rstd2, gv.w, bv.w);
        *reinterpret_cast<float4*>(out + (size_t)row * DX + c) = y;
      }
  }
}

// ---- 16-row tiles (round 3).  The 32-row kernel above holds ~300 registers per wave: one wave per SIMD, whose matrix work (384
// dependent-chain MFMAs), vector work (two LayerNorms, packs, residuals) and memory latencies (row loads, the 384 KB weight stream)
// all run one after the other -- 23 us even with the weight stream removed, whatever the row count.  On v_mfma_f32_16x16x32_bf16 a
// wave owns 16 rows: half the accumulators (x1: 64 registers), half of every chain, and TWO waves per SIMD that fill each other's
// waits.  Lane l = (r = l & 15, g = l >> 4):
//   A operand (weights):      lane holds W[16 tile + r][k(s, g, e)], e = 0..7
//   B operand (activations):  lane holds in[row r][k(s, g, e)]
//   C / D:                    lane holds out[channel 16 tile + 4 g + i][row r], i = 0..3
// natural k(s, g, e) = 32 s + 8 g + e (rows read from memory); C-order k(s, g, e) = 32 s + 16 (e >> 2) + 4 g + (e & 3) (two
// consecutive accumulator tiles packed to bf16); the host packs every weight in the order its operand uses (pack.x_post16_blob).
// The three GEMMs' fragments are consecutive in the blob, 8 per output tile: the prefetch of "the next tile" runs across the
// phase boundaries.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x8 pack2x4(const f32x4& a, const f32x4& b) {
  const uint4 u = {pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
  return frag_from_u4(u);
}
__device__ __forceinline__ float xquad_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

template <int DX>
struct XPost16Cfg {
  static constexpr int MT = DX / 16, KS = DX / 32, N_TILES = 3 * MT, N_FR = N_TILES * KS;
  static constexpr int P_BXO = 0, P_G1 = DX, P_B1N = 2 * DX, P_BF1 = 3 * DX, P_BF2 = 4 * DX, P_G2 = 5 * DX, P_B2N = 6 * DX, N_PAR = 7 * DX;
  static constexpr size_t BLOB_BYTES = (size_t)N_FR * 1024 + (size_t)N_PAR * 4;
};

constexpr int XP16_WPB = 4;
template <int DX>
__global__ void __launch_bounds__(64 * XP16_WPB, 2) k_x_post16(const float* __restrict__ newX_in, const float* __restrict__ X,
                                                               const uint8_t* __restrict__ mask, const uint4* __restrict__ blob,
                                                               float* __restrict__ out, int R) {
  using C = XPost16Cfg<DX>;
  constexpr int MT = C::MT, KS = C::KS;
  __shared__ __attribute__((aligned(16))) float sP[C::N_PAR];
  {
    const float* gp = reinterpret_cast<const float*>(blob + (size_t)C::N_FR * 64);
    for (int t = threadIdx.x; t < C::N_PAR; t += 64 * XP16_WPB) sP[t] = gp[t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = (blockIdx.x * XP16_WPB + wave) * 16;
  if (m0 >= R) return;
  const int row = min(m0 + r, R - 1);
  const bool rvalid = m0 + r < R;
  const uint4* fr = blob + lane;
  const float* xrow = X + (size_t)row * DX + 4 * g;
  // weights of output tile T (all three GEMMs numbered through), and for the first GEMM the residual's four entries of that tile: a
  // load issued behind the prefetch would return behind it
  auto ldw = [&](uint4 (&w)[KS], f32x4& xr, int T, float dep) {
    const int Tc = min(T, C::N_TILES - 1);
    // The byte offsets pass through an opaque statement that also names `dep`, a value of the tile finished last: loads from
    // __restrict__ const memory have no ordering edge of their own, and in this fully unrolled kernel instruction selection
    // otherwise moves ALL of them to the top (1,255 spilled registers).  Nothing is executed for it.
    unsigned xo = 64u * (unsigned)min(Tc, MT - 1), wo = (unsigned)Tc * KS * 1024u;
    asm volatile("" : "+v"(xo), "+v"(wo) : "v"(dep));
    xr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xrow) + xo);
    const uint4* wp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(fr) + wo);
#pragma unroll
    for (int s = 0; s < KS; ++s) w[s] = wp[s * 64];
  };
  auto gemm = [&](const uint4 (&w)[KS], const bf16x8 (&b)[KS], f32x4 acc) {
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma16(frag_from_u4(w[s]), b[s], acc);
    return acc;
  };
  auto par4 = [&](int base, int t) { return *reinterpret_cast<const f32x4*>(sP + base + 16 * t + 4 * g); };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  uint4 wA[KS], wB[KS];
  f32x4 xrA, xrB;
  ldw(wA, xrA, 0, 0.f);
  // ---- activation operand of x_out: this lane's row, natural k
  bf16x8 act[KS];
  {
    const float* ap = newX_in + (size_t)row * DX + 8 * g;
    f32x4 a0[KS], a1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a0[s] = *reinterpret_cast<const f32x4*>(ap + 32 * s); a1[s] = *reinterpret_cast<const f32x4*>(ap + 32 * s + 4); }
#pragma unroll
    for (int s = 0; s < KS; ++s) act[s] = pack2x4(a0[s], a1[s]);
  }
  const float mrow = mask[row] ? 1.f : 0.f;
  // ---- x_out, mask, residual
  f32x4 x1[MT];
  float sum = 0.f;
  auto epi1 = [&](int mt, f32x4 acc, const f32x4& xr) {
    const f32x4 bv = par4(C::P_BXO, mt);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = fmaf(mrow, acc[i] + bv[i], xr[i]); sum += acc[i]; }
    return acc;
  };
#pragma unroll
  for (int mt = 0; mt < MT; mt += 2) {
    ldw(wB, xrB, mt + 1, sum);
    __builtin_amdgcn_sched_barrier(0);
    x1[mt] = epi1(mt, gemm(wA, act, zero4), xrA);
    __builtin_amdgcn_sched_barrier(0);
    ldw(wA, xrA, mt + 2, sum);
    __builtin_amdgcn_sched_barrier(0);
    x1[mt + 1] = epi1(mt + 1, gemm(wB, act, zero4), xrB);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- LN1
  {
    const float mean = xquad_sum(sum) * (1.f / DX);
    float vs = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float d = x1[mt][i] - mean; vs = fmaf(d, d, vs); }
    const float rstd = rsqrtf(xquad_sum(vs) * (1.f / DX) + 1e-5f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 gv = par4(C::P_G1, mt), bv = par4(C::P_B1N, mt);
#pragma unroll
      for (int i = 0; i < 4; ++i) x1[mt][i] = fmaf((x1[mt][i] - mean) * rstd, gv[i], bv[i]);
    }
  }
  // ---- FFN chained through accumulators: H^T = relu(W1 X1^T + b1), O^T = W2 H^T + b2
#pragma unroll
  for (int s = 0; s < KS; ++s) act[s] = pack2x4(x1[2 * s], x1[2 * s + 1]);
  bf16x8 hB[KS];
  float dep = x1[MT - 1][0];
#pragma unroll
  for (int ht = 0; ht < MT; ht += 2) {
    ldw(wB, xrB, MT + ht + 1, dep);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 h0 = gemm(wA, act, par4(C::P_BF1, ht));
    __builtin_amdgcn_sched_barrier(0);
    ldw(wA, xrA, MT + ht + 2, h0[0]);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 h1 = gemm(wB, act, par4(C::P_BF1, ht + 1));
    dep = h1[0];
    // relu on the packed bf16 words (as 16-bit integers negative floats are negative)
    hB[ht >> 1] = __builtin_elementwise_max(pack2x4(h0, h1), (bf16x8)(short)0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float sum2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; mt += 2) {
    ldw(wB, xrB, 2 * MT + mt + 1, dep);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 o0 = gemm(wA, hB, par4(C::P_BF2, mt));
    __builtin_amdgcn_sched_barrier(0);
    ldw(wA, xrA, 2 * MT + mt + 2, o0[0]);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 o1 = gemm(wB, hB, par4(C::P_BF2, mt + 1));
    dep = o1[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x1[mt][i] += o0[i]; x1[mt + 1][i] += o1[i]; sum2 += x1[mt][i] + x1[mt + 1][i]; }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- LN2 and store
  const float mean2 = xquad_sum(sum2) * (1.f / DX);
  float vs2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = x1[mt][i] - mean2; vs2 = fmaf(d, d, vs2); }
  const float rstd2 = rsqrtf(xquad_sum(vs2) * (1.f / DX) + 1e-5f);
  if (rvalid) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 gv = par4(C::P_G2, mt), bv = par4(C::P_B2N, mt);
      f32x4 y;
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = fmaf((x1[mt][i] - mean2) * rstd2, gv[i], bv[i]);
      *reinterpret_cast<f32x4*>(out + (size_t)row * DX + 16 * mt + 4 * g) = y;
    }
  }
}

extern "C" {

long long da_x_post_blob_bytes(int dx, int dffx) {
  if (dx == 256 && dffx == 256) return (long long)XPostCfg<256, 256>::BLOB_BYTES;
  if (dx == 64 && dffx == 64) return (long long)XPostCfg<64, 64>::BLOB_BYTES;
  return -1;
}

// X' = LN2(X1 + linX2(relu(linX1(X1)))),  X1 = LN1(X + x_out(newX_in) * mask);  rows = B*n, all fp32 (R, dx) row-major
long long da_x_post16_blob_bytes(int dx, int dffx) {
  if (dx == 256 && dffx == 256) return (long long)XPost16Cfg<256>::BLOB_BYTES;
  if (dx == 64 && dffx == 64) return (long long)XPost16Cfg<64>::BLOB_BYTES;
  return -1;
}

int da_x_post16(const float* newX_in, const float* X, const uint8_t* node_mask, const void* blob, float* X_out, int rows, int dx,
                int dffx, void* stream) {
  DA_CHECK_ARG(rows >= 0, "da_x_post16: negative size");
  DA_CHECK_ARG(X != X_out && newX_in != X_out, "da_x_post16: in-place not supported");
  if (rows == 0) return DA_OK;
  const int grid = (rows + 16 * XP16_WPB - 1) / (16 * XP16_WPB);
  hipStream_t s = (hipStream_t)stream;
  if (dx == 256 && dffx == 256) k_x_post16<256><<<grid, 64 * XP16_WPB, 0, s>>>(newX_in, X, node_mask, (const uint4*)blob, X_out, rows);
  else if (dx == 64 && dffx == 64) k_x_post16<64><<<grid, 64 * XP16_WPB, 0, s>>>(newX_in, X, node_mask, (const uint4*)blob, X_out, rows);
  else { da_set_error("da_x_post16: dims (dx=%d, dffX=%d) not built; built: (256,256), (64,64)", dx, dffx); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_x_post16");
  return DA_OK;
}

int da_x_post(const float* newX_in, const float* X, const uint8_t* node_mask, const void* blob, float* X_out, int rows, int dx,
              int dffx, void* stream) {
  DA_CHECK_ARG(rows >= 0, "da_x_post: negative size");
  DA_CHECK_ARG(X != X_out && newX_in != X_out, "da_x_post: in-place not supported");
  if (rows == 0) return DA_OK;
  const int grid = (rows + 32 * XP_WPB - 1) / (32 * XP_WPB);
  hipStream_t s = (hipStream_t)stream;
  if (dx == 256 && dffx == 256) k_x_post<256, 256><<<grid, 64 * XP_WPB, 0, s>>>(newX_in, X, node_mask, (const uint4*)blob, X_out, rows);
  else if (dx == 64 && dffx == 64) k_x_post<64, 64><<<grid, 64 * XP_WPB, 0, s>>>(newX_in, X, node_mask, (const uint4*)blob, X_out, rows);
  else { da_set_error("da_x_post: dims (dx=%d, dffX=%d) not built; built: (256,256), (64,64)", dx, dffx); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_x_post");
  return DA_OK;
}

}  // extern "C"
