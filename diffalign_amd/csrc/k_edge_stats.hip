// Pooled statistics of the hidden edge tensor for Etoy (neuralnet/layers.py:31-47): one streaming pass, HBM-bound.
#include "mfma_util.h"

// per-graph pooled statistics of the bf16 edge tensor (Etoy, neuralnet/layers.py:31-47): partial sums per chunk
//   out[b][chunk][4][DE] = (sum, sum of squares, min, max) over the chunk's pairs (padding and diagonal included)
template <int DE>
__global__ void __launch_bounds__(256) k_e_stats(const bf16_t* E, float* out, int n, int nchunk, const uint8_t* skip) {
  constexpr int G = DE / 8;            // 16-byte groups per row
  constexpr int RL = 256 / G;          // rows handled concurrently
  constexpr int U = 4;                 // independent 16-byte loads in flight per thread (2, 8, 16 measured: no faster)
  const int b = blockIdx.y, ch = blockIdx.x;
  if (skip && skip[b] && skip[gridDim.y + b]) return;  // graph served by da_edge_stats_table (flags of da_edge_in_sym)
  const size_t N = (size_t)n * n;
  const size_t per = (N + nchunk - 1) / nchunk;
  const size_t p0 = ch * per, p1 = min(N, p0 + per);
  const int g = threadIdx.x % G, rl = threadIdx.x / G;
  float s[8], ss[8], mn[8], mx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; mn[e] = INFINITY; mx[e] = -INFINITY; }
  const bf16_t* base = E + (size_t)b * N * DE + 8 * g;
  auto fold = [&](const uint4& u) {
    const unsigned d[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (e & 1) ? bf16_hi(d[e >> 1]) : bf16_lo(d[e >> 1]);
      s[e] += v; ss[e] = fmaf(v, v, ss[e]); mn[e] = fminf(mn[e], v); mx[e] = fmaxf(mx[e], v);
    }
  };
  size_t p = p0 + rl;
  for (; p + (U - 1) * RL < p1; p += U * RL) {
    uint4 u[U];
#pragma unroll
    for (int k = 0; k < U; ++k) u[k] = *reinterpret_cast<const uint4*>(base + (p + k * RL) * DE);
#pragma unroll
    for (int k = 0; k < U; ++k) fold(u[k]);
  }
  for (; p < p1; p += RL) fold(*reinterpret_cast<const uint4*>(base + p * DE));
  // fold the row-lanes: lanes with equal channel group g inside a wave (xor 8, 16, 32 for DE = 64), then the 4 waves via LDS
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] += __shfl_xor(s[e], off); ss[e] += __shfl_xor(ss[e], off);
      mn[e] = fminf(mn[e], __shfl_xor(mn[e], off)); mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], off));
    }
  }
  __shared__ float red[4][4][DE];                  // [wave][stat][channel]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < G) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[wave][0][8 * lane + e] = s[e]; red[wave][1][8 * lane + e] = ss[e];
      red[wave][2][8 * lane + e] = mn[e]; red[wave][3][8 * lane + e] = mx[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < DE) {
    const int c = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = INFINITY, a3 = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a0 += red[w][0][c]; a1 += red[w][1][c]; a2 = fminf(a2, red[w][2][c]); a3 = fmaxf(a3, red[w][3][c]); }
    float* o = out + ((size_t)(b * nchunk + ch) * 4) * DE;
    o[c] = a0; o[DE + c] = a1; o[2 * DE + c] = a2; o[3 * DE + c] = a3;
  }
}

extern "C" {

static int edge_stats_launch(const void* E_bf16, float* partials, const uint8_t* skip, int B, int n, int de, int nchunk, void* stream) {
  DA_CHECK_ARG(nchunk >= 1, "da_edge_stats: nchunk < 1");
  if (B == 0 || n == 0) return DA_OK;
  dim3 grid(nchunk, B);
  if (de == 64) k_e_stats<64><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)E_bf16, partials, n, nchunk, skip);
  else if (de == 32) k_e_stats<32><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)E_bf16, partials, n, nchunk, skip);
  else { da_set_error("da_edge_stats: de=%d not built (32, 64)", de); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_edge_stats");
  return DA_OK;
}
int da_edge_stats(const void* E_bf16, float* partials, int B, int n, int de, int nchunk, void* stream) {
  return edge_stats_launch(E_bf16, partials, nullptr, B, n, de, nchunk, stream);
}
// da_edge_stats for the graphs whose two flags (skip[b], skip[B + b]: da_edge_in_sym's scratch) are not both set; the
// partials of the others are left untouched (da_edge_stats_table writes them).
int da_edge_stats_skip(const void* E_bf16, float* partials, const uint8_t* skip, int B, int n, int de, int nchunk, void* stream) {
  DA_CHECK_ARG(skip != nullptr, "da_edge_stats_skip: skip flags are null");
  return edge_stats_launch(E_bf16, partials, skip, B, n, de, nchunk, stream);
}

}  // extern "C"
