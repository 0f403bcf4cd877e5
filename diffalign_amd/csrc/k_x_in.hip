// Node input stage (neuralnet/transformer_model_with_y.py:584-585, 661-666): Xh = relu(W1 relu(W0 x + b0) + b1), hidden-feature mask
// with encode_no_element (utils/graph.py PlaceHolder.mask), + positional encodings -- fused into one kernel.
// Replaces two da_linear launches and da_x_in_finish (56 + 50 + 21 us alone, 148 + 70 us beside the HBM-bound edge input stage whose
// bandwidth they share: two (R,256) fp32 round trips) by one pass that reads the (R, dx_in) input rows and the positional encodings
// and writes Xh once.  Same scheme as k_x_post: one wave = 32 rows, activations transposed (channel in registers, row on the lane),
// the two GEMMs chained through MFMA accumulators, weight fragments streamed from L2.
#include "mfma_util.h"

constexpr int XIN_KS0_MAX = 6;                        // input width up to 96 (29 classes + 8 extra features + 29 aligned = 66)

template <int HX, int DX>
struct XInCfg {
  static constexpr int HT = HX / 32, MT = DX / 32, KSF = HX / 16;
  // fragments: W0 [HT][ks0] (natural k order, K padded to 16 ks0), then W1 [MT][KSF] (C order); fp32 params: b0[HX] b1[DX]
  static constexpr int N_PAR = HX + DX;
  static size_t blob_bytes(int ks0) { return (size_t)(HT * ks0 + MT * KSF) * 1024 + (size_t)N_PAR * 4; }
};

template <int HX, int DX>
__global__ void __launch_bounds__(64) k_x_in(const float* __restrict__ X, int dx_in, int ks0, const uint8_t* __restrict__ mask,
                                             const float* __restrict__ pos, const uint4* __restrict__ blob, float* __restrict__ out, int R) {
  using C = XInCfg<HX, DX>;
  __shared__ __attribute__((aligned(16))) float sP[C::N_PAR];
  const int n_fr = C::HT * ks0 + C::MT * C::KSF;
  {
    const float* gp = reinterpret_cast<const float*>(blob + (size_t)n_fr * 64);
    for (int t = threadIdx.x; t < C::N_PAR; t += 64) sP[t] = gp[t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32;
  if (m0 >= R) return;
  const int row = min(m0 + r, R - 1);
  const bool rvalid = m0 + r < R;
  const uint4* fr = blob + lane;
  // ---- this lane's input row as the B operand, natural k order (zero beyond dx_in).  The loads are unconditional on clamped indices and
  // the zero is a select afterwards: behind `k < dx_in` branches each of the 48 loads had a wait of its own.
  bf16x8 inB[XIN_KS0_MAX];
  {
    const float* ap = X + (size_t)row * dx_in;
    float v[XIN_KS0_MAX][8];
#pragma unroll
    for (int s = 0; s < XIN_KS0_MAX; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[s][j] = ap[min(16 * s + 8 * h + j, dx_in - 1)];
#pragma unroll
    for (int s = 0; s < XIN_KS0_MAX; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (16 * s + 8 * h + j >= dx_in) v[s][j] = 0.f;
      const uint4 u = {pack_bf16(v[s][0], v[s][1]), pack_bf16(v[s][2], v[s][3]), pack_bf16(v[s][4], v[s][5]), pack_bf16(v[s][6], v[s][7])};
      inB[s] = frag_from_u4(u);
    }
  }
  // Weight fragments of a whole column tile are fetched into registers before its MFMAs, the next tile's while this one runs (the
  // compiler's own order was load, wait, MFMA for each of the 152 fragments: 75 us of L2 round trips for 2.3 us of matrix work).
  // ---- H^T = relu(W0 X^T + b0), kept as bf16 operands of the second GEMM
  bf16x8 hB[C::KSF];
  {
    uint4 w0[2][XIN_KS0_MAX];
    auto load_w0 = [&](uint4 (&w)[XIN_KS0_MAX], int ht) {
      const int hc = min(ht, C::HT - 1);
#pragma unroll
      for (int s = 0; s < XIN_KS0_MAX; ++s) w[s] = fr[(size_t)(hc * ks0 + min(s, ks0 - 1)) * 64];
    };
    load_w0(w0[0], 0);
#pragma unroll
    for (int ht = 0; ht < C::HT; ++ht) {
      load_w0(w0[(ht + 1) & 1], ht + 1);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 hacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(sP + 32 * ht + 8 * g + 4 * h);
        hacc[4 * g + 0] = bv.x; hacc[4 * g + 1] = bv.y; hacc[4 * g + 2] = bv.z; hacc[4 * g + 3] = bv.w;
      }
#pragma unroll
      for (int s = 0; s < XIN_KS0_MAX; ++s)
        if (s < ks0) hacc = mfma32(frag_from_u4(w0[ht & 1][s]), inB[s], hacc);
      hB[2 * ht + 0] = __builtin_elementwise_max(pack_acc(hacc, 0), (bf16x8)(short)0);
      hB[2 * ht + 1] = __builtin_elementwise_max(pack_acc(hacc, 1), (bf16x8)(short)0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- O^T = relu(W1 H^T + b1) * mask, stored tile by tile (+ positional encodings)
  const float mrow = mask[row] ? 1.f : 0.f;
  const uint4* fr1 = fr + (size_t)C::HT * ks0 * 64;
  float sum = 0.f;
  uint4 w1[2][C::KSF];
  auto load_w1 = [&](uint4 (&w)[C::KSF], int mt) {
    const int mc = min(mt, C::MT - 1);
#pragma unroll
    for (int s = 0; s < C::KSF; ++s) w[s] = fr1[(size_t)(mc * C::KSF + s) * 64];
  };
  load_w1(w1[0], 0);
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    load_w1(w1[(mt + 1) & 1], mt + 1);
    float4 pv[4];
    if (pos) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pv[g] = *reinterpret_cast<const float4*>(pos + (size_t)row * DX + 32 * mt + 8 * g + 4 * h);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(sP + HX + 32 * mt + 8 * g + 4 * h);
      acc[4 * g + 0] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
    }
#pragma unroll
    for (int s = 0; s < C::KSF; ++s) acc = mfma32(frag_from_u4(w1[mt & 1][s]), hB[s], acc);
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = fmaxf(acc[i], 0.f) * mrow; sum += acc[i]; }
    if (rvalid) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * mt + 8 * g + 4 * h;
        float4 y = {acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        if (pos) { y.x += pv[g].x; y.y += pv[g].y; y.z += pv[g].z; y.w += pv[g].w; }
        *reinterpret_cast<float4*>(out + (size_t)row * DX + c) = y;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // encode_no_element on the hidden features: a row whose masked features sum to zero gets 1 in channel 0 (rewritten after the fact:
  // the tiles are stored as they are produced; same-lane stores to one address keep their order)
  sum = xhalf_sum(sum);
  if (sum == 0.f && h == 0 && rvalid) out[(size_t)row * DX] = 1.f + (pos ? pos[(size_t)row * DX] : 0.f);
}

extern "C" {

// Bytes of da_x_in's weight blob (diffalign_amd/pack.py x_in_blob) for an input width dx_in; -1 when the widths are not built.
long long da_x_in_blob_bytes(int dx_in, int hx, int dx) {
  const int ks0 = (dx_in + 15) / 16;
  if (dx_in < 1 || ks0 > XIN_KS0_MAX) return -1;
  if (hx == 256 && dx == 256) return (long long)XInCfg<256, 256>::blob_bytes(ks0);
  if (hx == 64 && dx == 64) return (long long)XInCfg<64, 64>::blob_bytes(ks0);
  return -1;
}

// Xh = mask(relu(mlp_in_X.2(relu(mlp_in_X.0(X))))) [+ pos]  (transformer_model_with_y.py:661-666); X (rows, dx_in) fp32 row-major,
// pos (rows, dx) or NULL, Xh (rows, dx).
int da_x_in(const float* X, int dx_in, const uint8_t* node_mask, const float* pos, const void* blob, float* Xh, int rows, int hx, int dx,
            void* stream) {
  DA_CHECK_ARG(rows >= 0 && X && node_mask && blob && Xh, "da_x_in: arguments");
  const int ks0 = (dx_in + 15) / 16;
  DA_CHECK_ARG(dx_in >= 1 && ks0 <= XIN_KS0_MAX, "da_x_in: input width %d not in 1..%d", dx_in, 16 * XIN_KS0_MAX);
  if (rows == 0) return DA_OK;
  const int grid = (rows + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  if (hx == 256 && dx == 256) k_x_in<256, 256><<<grid, 64, 0, s>>>(X, dx_in, ks0, node_mask, pos, (const uint4*)blob, Xh, rows);
  else if (hx == 64 && dx == 64) k_x_in<64, 64><<<grid, 64, 0, s>>>(X, dx_in, ks0, node_mask, pos, (const uint4*)blob, Xh, rows);
  else { da_set_error("da_x_in: dims (hidden %d, dx %d) not built; built: (256,256), (64,64)", hx, dx); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_x_in");
  return DA_OK;
}

}  // extern "C"
