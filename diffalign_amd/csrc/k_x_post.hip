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

extern "C" {

long long da_x_post_blob_bytes(int dx, int dffx) {
  if (dx == 256 && dffx == 256) return (long long)XPostCfg<256, 256>::BLOB_BYTES;
  if (dx == 64 && dffx == 64) return (long long)XPostCfg<64, 64>::BLOB_BYTES;
  return -1;
}

// X' = LN2(X1 + linX2(relu(linX1(X1)))),  X1 = LN1(X + x_out(newX_in) * mask);  rows = B*n, all fp32 (R, dx) row-major
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
