// q, k, v projections of one layer (neuralnet/layers.py:217-223: Q = q(X)*m, K = k(X)*m, V = v(X)*m) in one kernel that
// writes Q as fp32 rows and K, V directly in the bf16 fragment layout the edge-layer kernel reads (k_kv_pack's layout):
// one wave = one key tile (b, jt) of 32 rows, so its MFMA accumulators ARE the fragment (row in registers, channel on the
// lane).  Replaces da_linear (59 MB of fp32 q|k|v written) + da_kv_pack (39 MB read back).
#include "mfma_util.h"

constexpr int QKV_PARTS = 3;   // waves per key tile: one per projection (38 -> 32 us; 6: 39 us, 12: 59 us)
template <int DX>
__global__ void __launch_bounds__(256, 2) k_qkv_frag(const float* __restrict__ X, const uint4* __restrict__ W,
                                                     const float* __restrict__ bias, const uint8_t* __restrict__ mask,
                                                     float* __restrict__ Q, uint4* __restrict__ kvf, int B, int n) {
  constexpr int KS = DX / 16, NCB = DX / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int NT = (n + 31) / 32;
  // one wave = one key tile (b, jt) and ONE of the three projections (1,800 waves at the benchmark size instead of 600:
  // the kernel is bound by the latency of a single wave's weight stream, not by the re-read of the 32 activation rows)
  const int gwp = blockIdx.x * 4 + wave;
  if (gwp >= QKV_PARTS * B * NT) return;
  const int part = gwp % QKV_PARTS, gw = gwp / QKV_PARTS;
  const int b = gw / NT, jt = gw - b * NT;
  const float* ap = X + ((size_t)b * n + min(32 * jt + r, n - 1)) * DX + 8 * h;
  bf16x8 af[KS];
#pragma unroll
  for (int s0 = 0; s0 < KS; s0 += 8) {
    float4 x0[8], x1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < KS) { x0[u] = *reinterpret_cast<const float4*>(ap + 16 * (s0 + u)); x1[u] = *reinterpret_cast<const float4*>(ap + 16 * (s0 + u) + 4); }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < KS) {
        const uint4 au = {pack_bf16(x0[u].x, x0[u].y), pack_bf16(x0[u].z, x0[u].w), pack_bf16(x1[u].x, x1[u].y), pack_bf16(x1[u].z, x1[u].w)};
        af[s0 + u] = frag_from_u4(au);
      }
  }
  // rows of this lane's accumulator registers: key/query row 32jt + crow(i,h); dead rows (j >= n) and masked nodes give 0.
  // Kept as one bit per row: sixteen floats here were what pushed the kernel over 256 registers, and the reload of a spilled
  // register waits with vmcnt(0), which drained the next tile's prefetched weights in front of every tile.
  unsigned rmask = 0;
  {
    uint8_t mb[16];                                    // loaded unconditionally, as one batch (16 guarded loads were 16 round trips)
#pragma unroll
    for (int i = 0; i < 16; ++i) mb[i] = mask[(size_t)b * n + min(32 * jt + crow(i, h), n - 1)];
#pragma unroll
    for (int i = 0; i < 16; ++i) rmask |= ((32 * jt + crow(i, h) < n && mb[i]) ? 1u : 0u) << i;
  }
  const bool full = 32 * jt + 32 <= n;                 // wave-uniform: every row of the tile exists (no per-store branch)
  // the tile's bias travels with its weights, issued first: a load issued behind the prefetch returns behind it
  auto load_w = [&](uint4 (&wf)[KS], float& bv, int t) {
    const int tc = min(t, 3 * NCB - 1);
    bv = bias[32 * tc + r];
    const uint4* wp = W + ((size_t)tc * KS) * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[s] = wp[s * 64];
  };
  auto tile = [&](const uint4 (&wf)[KS], float bv, int t) {
    f32x16 acc = splat16(0.f);
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma32(af[s], frag_from_u4(wf[s]), acc);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ((rmask >> i) & 1u ? 1.f : 0.f) * (acc[i] + bv);
    if (t < NCB) {                                   // Q: fp32 rows
      float* qp = Q + ((size_t)b * n + 32 * jt) * DX + 32 * t + r;
      if (full) {
#pragma unroll
        for (int i = 0; i < 16; ++i) qp[(size_t)crow(i, h) * DX] = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (32 * jt + crow(i, h) < n) qp[(size_t)crow(i, h) * DX] = v[i];
      }
    } else {                                         // K, V: fragment words
      const int kv = (t - NCB) / NCB, cb = (t - NCB) - kv * NCB;
      uint4* o = kvf + (((size_t)(b * NT + jt) * NCB + cb) * 4 + kv * 2) * 64 + lane;
      const float lo8[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
      const float hi8[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
      o[0] = kv_words(lo8);
      o[64] = kv_words(hi8);
    }
  };
  uint4 wA[KS], wB[KS];
  float bA, bB;
  constexpr int TPW = 3 * NCB / QKV_PARTS;             // column tiles per wave
  static_assert(TPW % 2 == 0, "column tiles are walked in pairs");
  const int t0 = part * TPW, t1 = t0 + TPW;
  load_w(wA, bA, t0);
  for (int t = t0; t < t1; t += 2) {
    load_w(wB, bB, t + 1);
    __builtin_amdgcn_sched_barrier(0);
    tile(wA, bA, t);
    __builtin_amdgcn_sched_barrier(0);
    load_w(wA, bA, t + 2);                             // past the last pair: a clamped, unused fetch
    __builtin_amdgcn_sched_barrier(0);
    tile(wB, bB, t + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

extern "C" {

// Q (B*n, dx) fp32 = q(X)*mask;  kv_frag = K, V of the layer in da_kv_pack's fragment layout (da_kv_frag_bytes).
// W_bf16 = cat(q.weight, k.weight, v.weight) packed by diffalign_amd/pack.py bf16_weight; bias = cat of the three biases.
int da_qkv_frag(const float* X, const void* W_bf16, const float* bias, const uint8_t* node_mask, float* Q, void* kv_frag,
                int B, int n, int dx, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0, "da_qkv_frag: negative size");
  if (B == 0 || n == 0) return DA_OK;
  const int NT = (n + 31) / 32;
  const int grid = (QKV_PARTS * B * NT + 3) / 4;
  hipStream_t s = (hipStream_t)stream;
  if (dx == 256) k_qkv_frag<256><<<grid, 256, 0, s>>>(X, (const uint4*)W_bf16, bias, node_mask, Q, (uint4*)kv_frag, B, n);
  else if (dx == 64) k_qkv_frag<64><<<grid, 256, 0, s>>>(X, (const uint4*)W_bf16, bias, node_mask, Q, (uint4*)kv_frag, B, n);
  else { da_set_error("da_qkv_frag: dx=%d not built; built: 256, 64", dx); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_qkv_frag");
  return DA_OK;
}

}  // extern "C"
