// Edge input / output stages of the denoiser (neuralnet/transformer_model_with_y.py:657-661 and :672-704).
// Same machinery as the fused layer: transposed GEMM chain through accumulators, weights as pre-packed A fragments in
// LDS, one wave per query row (b,i), 32-pair tiles; both orientations (i,j) and (j,i) are evaluated by the same wave
// so the symmetrisation needs no second pass over the edge tensor.
#include "mfma_util.h"
#include "edge_in_layout.h"
#include "diffusion_dev.h"

template <int DE, int HE>
struct EdgeIoCfg {
  static constexpr int HT = HE / 32, KSH = HE / 16, MT = DE / 32, KS1 = DE / 16;
  // input stage:  W0 (HE x 16pad) natural-k fragments [HT], W1 (DE x HE) C-order fragments [MT*KSH]; params b0[HE] b1[DE]
  static constexpr int IN_FR = HT + MT * KSH;
  static constexpr int IN_PAR = HE + DE;
  static constexpr size_t IN_BYTES = (size_t)IN_FR * 1024 + IN_PAR * 4;
  // output stage: W0 (HE x DE) C-order fragments [HT*KS1], W1 (32pad x HE) C-order fragments [KSH]; params b0[HE] b1[32]
  static constexpr int OUT_FR = HT * KS1 + KSH;
  static constexpr int OUT_PAR = HE + 32;
  static constexpr size_t OUT_BYTES = (size_t)OUT_FR * 1024 + OUT_PAR * 4;
};

__device__ __forceinline__ void load_bias16(f32x16& acc, const float* p, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bv = *reinterpret_cast<const float4*>(p + 8 * g + 4 * h);
    acc[4 * g + 0] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
  }
}

// ---- input stage: E_h = encode_no_element(mask(sym(mlp_in_E(E)))) as bf16 ---------------------------------------
template <int DE, int HE>
__device__ __forceinline__ void mlp_in_chain(const float* __restrict__ src, int de_in, int h, int lane, const uint4* sW0,
                                             const uint4* sW1, const float* sP, f32x16 (&o)[DE / 32]) {
  using C = EdgeIoCfg<DE, HE>;
  float xv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) xv[j] = (8 * h + j < de_in) ? src[8 * h + j] : 0.f;
  const uint4 xu = {pack_bf16(xv[0], xv[1]), pack_bf16(xv[2], xv[3]), pack_bf16(xv[4], xv[5]), pack_bf16(xv[6], xv[7])};
  const bf16x8 xb = frag_from_u4(xu);
  bf16x8 hB[C::KSH];
#pragma unroll
  for (int ht = 0; ht < C::HT; ++ht) {
    f32x16 hacc;
    load_bias16(hacc, sP + 32 * ht, h);
    hacc = mfma32(lds_frag(sW0, ht, lane), xb, hacc);
#pragma unroll
    for (int i2 = 0; i2 < 16; ++i2) hacc[i2] = fmaxf(hacc[i2], 0.f);
    hB[2 * ht] = pack_acc(hacc, 0); hB[2 * ht + 1] = pack_acc(hacc, 1);
  }
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    load_bias16(o[mt], sP + HE + 32 * mt, h);
#pragma unroll
    for (int s = 0; s < C::KSH; ++s) o[mt] = mfma32(lds_frag(sW1, mt * C::KSH + s, lane), hB[s], o[mt]);
#pragma unroll
    for (int i2 = 0; i2 < 16; ++i2) o[mt][i2] = fmaxf(o[mt][i2], 0.f);
  }
}

template <int DE, int HE>
__global__ void __launch_bounds__(256) k_edge_in(const float* __restrict__ E_in, int de_in, const uint8_t* __restrict__ nm,
                                                 const uint4* __restrict__ blob, bf16_t* __restrict__ E_out, int B, int n,
                                                 const uint8_t* __restrict__ sym_flags) {
  using C = EdgeIoCfg<DE, HE>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* sW = reinterpret_cast<uint4*>(smem);
  if (sym_flags) {                                     // nothing to do when k_edge_in_tiles served every graph
    int slow = 0;
    for (int t = threadIdx.x; t < B; t += 256) slow |= !(sym_flags[t] && sym_flags[B + t]);
    if (!__syncthreads_or(slow)) return;
  }
  for (int t = threadIdx.x; t < (int)(C::IN_BYTES / 16); t += 256) sW[t] = blob[t];
  __syncthreads();
  const uint4* sW0 = sW; const uint4* sW1 = sW + C::HT * 64;
  const float* sP = reinterpret_cast<const float*>(sW + C::IN_FR * 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int NT = (n + 31) / 32, R = B * n;
  for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
    const int b = row / n, i = row - b * n;
    if (sym_flags && sym_flags[b] && sym_flags[B + b]) continue;   // symmetric one-hot graph: k_edge_in_tiles wrote it
    const bool mi = nm[row] != 0;
    for (int jt = 0; jt < NT; ++jt) {
      const int jl = jt * 32 + r; const bool jvalid = jl < n; const int jc = jvalid ? jl : n - 1;
      f32x16 o[C::MT];
      const float* pij = E_in + ((size_t)row * n + jc) * de_in;
      const float* pji = E_in + (((size_t)b * n + jc) * n + i) * de_in;
      // symmetric input tile (the diffusion state always is): mlp(E_ij) == mlp(E_ji) bit for bit, one evaluation is enough
      bool sym_tile;                                   // wave-uniform
      if (sym_flags) {
        sym_tile = sym_flags[b] != 0;                  // whole graph checked beforehand with coalesced tile reads (k_edge_in_tiles)
      } else {
        bool same = true;
#ifndef EXP_IN_NOMIRROR
        for (int c = 8 * h; c < min(de_in, 8 * h + 8); ++c) same = same && (pij[c] == pji[c]);
#endif
        sym_tile = __all(same);
      }
      mlp_in_chain<DE, HE>(pij, de_in, h, lane, sW0, sW1, sP, o);
      const float pm = (mi && jvalid && nm[b * n + jc] && jc != i) ? 1.f : 0.f;
      float sum = 0.f;
      if (sym_tile) {
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int i2 = 0; i2 < 16; ++i2) { const float v = (o[mt][i2] + o[mt][i2]) * 0.5f * pm; o[mt][i2] = v; sum += v; }
      } else {
        f32x16 om[C::MT];
        mlp_in_chain<DE, HE>(pji, de_in, h, lane, sW0, sW1, sP, om);
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
          for (int i2 = 0; i2 < 16; ++i2) { const float v = (o[mt][i2] + om[mt][i2]) * 0.5f * pm; o[mt][i2] = v; sum += v; }
      }
      sum = xhalf_sum(sum);
      if (sum == 0.f && h == 0) o[0][0] = 1.f;        // encode_no_element on the hidden features
#pragma unroll
      for (int u = 0; u < C::KS1; ++u) {
        const int mt = u >> 1, g0 = 2 * (u & 1);
        uint4 st = {pack_bf16(o[mt][4 * g0 + 0], o[mt][4 * g0 + 1]), pack_bf16(o[mt][4 * g0 + 2], o[mt][4 * g0 + 3]),
                    pack_bf16(o[mt][4 * g0 + 4], o[mt][4 * g0 + 5]), pack_bf16(o[mt][4 * g0 + 6], o[mt][4 * g0 + 7])};
        row16_to_corder(st);
#ifdef EXP_IN_NOSTORE
        if (jvalid && st.x == 0x12345u) *reinterpret_cast<uint4*>(E_out + ((size_t)row * n + jl) * DE + 16 * u + 8 * h) = st;
#else
        if (jvalid) *reinterpret_cast<uint4*>(E_out + ((size_t)row * n + jl) * DE + 16 * u + 8 * h) = st;
#endif
      }
    }
  }
}

// Symmetric one-hot graphs (every diffusion state is one; masked pairs may be all zero): de_in + 1 distinct results per pair.
// k_edge_in_table evaluates them with the very code of k_edge_in (same MFMA chain, same epilogue => same bits) into
// T[(de_in + 2)][DE] bf16 rows in the stored word order, row de_in the all-zero input, row de_in + 1 the masked pair;
// k_edge_in_tiles then writes E_h[b,i,j,:] = T[class(b,i,j)] with coalesced 16-byte stores: 28 B read + 2 DE B written
// per pair and no arithmetic.
template <int DE, int HE>
__global__ void __launch_bounds__(64) k_edge_in_table(int de_in, const uint4* __restrict__ blob, uint4* __restrict__ T,
                                                      uint8_t* __restrict__ flags, int n_flags, int* __restrict__ hist, int n_hist) {
  using C = EdgeIoCfg<DE, HE>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* sW = reinterpret_cast<uint4*>(smem);
  // the stage's flags start as set and its class counts as zero (two fill launches before): k_edge_in_tiles runs behind this kernel
  for (int t = threadIdx.x; t < n_flags; t += 64) flags[t] = 1;
  for (int t = threadIdx.x; t < n_hist; t += 64) hist[t] = 0;
  for (int t = threadIdx.x; t < (int)(C::IN_BYTES / 16); t += 64) sW[t] = blob[t];
  float* sI = reinterpret_cast<float*>(smem + C::IN_BYTES);       // [32][de_in]: row r = one-hot(r) (zero for r >= de_in)
  for (int t = threadIdx.x; t < 32 * de_in; t += 64) sI[t] = (t / de_in == t % de_in) ? 1.f : 0.f;
  __syncthreads();
  const uint4* sW0 = sW; const uint4* sW1 = sW + C::HT * 64;
  const float* sP = reinterpret_cast<const float*>(sW + C::IN_FR * 64);
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  for (int pass = 0; pass < 2; ++pass) {               // pass 0: free pairs (pm = 1), pass 1: the masked pair (pm = 0)
    const float pm = pass == 0 ? 1.f : 0.f;
    f32x16 o[C::MT];
    mlp_in_chain<DE, HE>(sI + r * de_in, de_in, h, lane, sW0, sW1, sP, o);
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int i2 = 0; i2 < 16; ++i2) { const float v = (o[mt][i2] + o[mt][i2]) * 0.5f * pm; o[mt][i2] = v; sum += v; }
    sum = xhalf_sum(sum);
    if (sum == 0.f && h == 0) o[0][0] = 1.f;
    const int trow = pass == 0 ? r : de_in + 1;
    const bool wr = pass == 0 ? r <= de_in : r == 0;
#pragma unroll
    for (int u = 0; u < C::KS1; ++u) {
      const int mt = u >> 1, g0 = 2 * (u & 1);
      uint4 st = {pack_bf16(o[mt][4 * g0 + 0], o[mt][4 * g0 + 1]), pack_bf16(o[mt][4 * g0 + 2], o[mt][4 * g0 + 3]),
                  pack_bf16(o[mt][4 * g0 + 4], o[mt][4 * g0 + 5]), pack_bf16(o[mt][4 * g0 + 6], o[mt][4 * g0 + 7])};
      row16_to_corder(st);
      if (wr) T[trow * (DE / 8) + 2 * u + h] = st;
    }
  }
}

// One workgroup per pair of mirrored 32x32 tiles (it <= jt) of one graph.  Both tiles are read row-wise (coalesced) and met
// in LDS; the workgroup (1) checks that they mirror each other bit for bit and that every pair is one-hot or all zero,
// clearing flags[b] / flags[B + b] otherwise (all 2B flags must be 1 on entry), (2) writes both output tiles from
// the table as if the graph passed: k_edge_in, launched afterwards, rewrites the graphs whose flags were cleared, and
// (3) counts the graph's pairs per table row in hist[b][.] (zero on entry): the pooled statistics of the hidden tensor
// follow from the counts and the table (k_e_stats_table) without another pass over it.
// CLS_ONLY: instead of the hidden rows, write the table row number of every pair into cls_map[b][i][j] (one byte per pair): the
// first edge layer then takes its input rows from the table itself (da_edge_layer_table) and the hidden tensor of the graphs that
// pass is never written or read.
template <int DE, bool CLS_ONLY>
__global__ void __launch_bounds__(256) k_edge_in_tiles(const float* __restrict__ E_in, int de_in, const uint8_t* __restrict__ nm,
                                                       const uint4* __restrict__ T, uint4* __restrict__ E_out, int n,
                                                       uint8_t* __restrict__ flags, int* __restrict__ hist,
                                                       uint8_t* __restrict__ cls_map) {
  constexpr int Q = DE / 8;                            // 16-byte words per pair
  int it, jt;
  tri_tile(blockIdx.x, (n + 31) / 32, it, jt);
  const int b = blockIdx.y, B = gridDim.y;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* sT = reinterpret_cast<uint4*>(smem);          // [(de_in + 2)][Q]
  float* sA = reinterpret_cast<float*>(sT + 18 * Q);   // tile (it, jt): [32 rows i][32 * de_in]
  __shared__ uint8_t cls[32][33];
  __shared__ int sh[IN_HIST];                          // pairs of this tile (and its mirror) per table row
  if (threadIdx.x < IN_HIST) sh[threadIdx.x] = 0;
  const int W = 32 * de_in, W4 = W / 4;
  const float* base = E_in + (size_t)b * n * n * de_in;
  const bool vec = ((n * de_in) & 3) == 0 && (reinterpret_cast<uintptr_t>(E_in) & 15) == 0;
  for (int t = threadIdx.x; t < (de_in + 2) * Q; t += 256) sT[t] = T[t];
  // Tile loads: all of a thread's (up to 8) 16-byte pieces are requested before the first is used, from clamped addresses; with one
  // guarded load per trip a tile cost seven dependent HBM round trips, and its mirror seven more (103 -> 60 us at B = 64, n = 300).
  constexpr int UN_T = 8;                              // 32 rows x 8 de_in float4 / 256 threads, de_in <= 8
  const bool batched = vec && 32 * W4 <= UN_T * 256;   // (vec: n de_in % 4 == 0, hence vw % 4 == 0 and every row start is 16-byte aligned)
  float4 qm[UN_T];                                     // the mirrored tile (jt, it), requested together with tile (it, jt): one HBM latency, not two
  {
    const int vw = min(32, n - jt * 32) * de_in;       // valid floats of a tile row
    if (batched) {
      float4 v[UN_T];
#pragma unroll
      for (int u = 0; u < UN_T; ++u) {
        const int t = min((int)threadIdx.x + u * 256, 32 * W4 - 1);
        const int ri = t / W4, o = (t - ri * W4) * 4;
        v[u] = *reinterpret_cast<const float4*>(base + ((size_t)min(it * 32 + ri, n - 1) * n + jt * 32) * de_in + max(min(o, vw - 4), 0));
      }
      {
        const int vwm = min(32, n - it * 32) * de_in;
#pragma unroll
        for (int u = 0; u < UN_T; ++u) {
          const int t = min((int)threadIdx.x + u * 256, 32 * W4 - 1);
          const int rj = t / W4, o = (t - rj * W4) * 4;
          qm[u] = *reinterpret_cast<const float4*>(base + ((size_t)min(jt * 32 + rj, n - 1) * n + it * 32) * de_in + max(min(o, vwm - 4), 0));
        }
      }
#pragma unroll
      for (int u = 0; u < UN_T; ++u) {
        const int t = threadIdx.x + u * 256;
        if (t < 32 * W4) {
          const int ri = t / W4, o = (t - ri * W4) * 4;
          const bool ok = it * 32 + ri < n && o < vw;
          *reinterpret_cast<float4*>(sA + ri * W + o) = ok ? v[u] : float4{0.f, 0.f, 0.f, 0.f};
        }
      }
    } else
    for (int t = threadIdx.x; t < 32 * W4; t += 256) {
      const int ri = t / W4, o = (t - ri * W4) * 4, i = it * 32 + ri;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < n) {
        const float* p = base + ((size_t)i * n + jt * 32) * de_in + o;
        if (vec && o + 4 <= vw) v = *reinterpret_cast<const float4*>(p);
        else { if (o < vw) v.x = p[0]; if (o + 1 < vw) v.y = p[1]; if (o + 2 < vw) v.z = p[2]; if (o + 3 < vw) v.w = p[3]; }
      }
      *reinterpret_cast<float4*>(sA + ri * W + o) = v;
    }
  }
  __syncthreads();
  bool hot = true;
  for (int t = threadIdx.x; t < 32 * 32; t += 256) {
    const int ri = t >> 5, cj = t & 31, i = it * 32 + ri, j = jt * 32 + cj;
    int ones = 0, zeros = 0, c = de_in;
    for (int k = 0; k < de_in; ++k) { const float v = sA[ri * W + cj * de_in + k]; ones += (v == 1.f); zeros += (v == 0.f); c = v != 0.f ? k : c; }
    const bool in = i < n && j < n;
    if (in) hot = hot && ((ones == 1 && zeros == de_in - 1) || zeros == de_in);
    const int k = (in && i != j && nm[b * n + i] && nm[b * n + j]) ? c : de_in + 1;
    cls[ri][cj] = k;
    // the tile's pairs per table row: one vote per row instead of a thousand LDS atomics on ten addresses
    for (int kk = 0; kk <= de_in + 1; ++kk) {
      const int cnt = __popcll(__ballot(in && k == kk));
      if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&sh[kk], cnt);
    }
  }
  bool bad = false;
  {
    const int vw = min(32, n - it * 32) * de_in;
    if (batched) {                                     // tile (jt, it): rows j, columns i (loaded above)
#pragma unroll
      for (int u = 0; u < UN_T; ++u) {
        const int t = threadIdx.x + u * 256;
        const int rj = t / W4, o = (t - rj * W4) * 4;
        if (t < 32 * W4 && jt * 32 + rj < n && o < vw) {
          const float v[4] = {qm[u].x, qm[u].y, qm[u].z, qm[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ci = (o + k) / de_in, c = (o + k) - ci * de_in;
            bad = bad || (v[k] != sA[ci * W + rj * de_in + c]);
          }
        }
      }
    } else
    for (int t = threadIdx.x; t < 32 * W4; t += 256) {  // tile (jt, it): rows j, columns i
      const int rj = t / W4, o = (t - rj * W4) * 4, j = jt * 32 + rj;
      if (j >= n) continue;
      const float* p = base + ((size_t)j * n + it * 32) * de_in + o;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec && o + 4 <= vw) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
      else for (int k = 0; k < 4; ++k) if (o + k < vw) v[k] = p[k];
      for (int k = 0; k < 4; ++k) {
        const int ci = (o + k) / de_in, c = (o + k) - ci * de_in;
        if (o + k < vw) bad = bad || (v[k] != sA[ci * W + rj * de_in + c]);
      }
    }
  }
  const bool wave_bad = __any(bad), wave_cold = __any(!hot);       // wave-wide votes, taken before the lane-0 branch
  if ((threadIdx.x & 63) == 0) {
    if (wave_bad) flags[b] = 0;
    if (wave_cold) flags[B + b] = 0;
  }
  __syncthreads();
  if (threadIdx.x < de_in + 2 && sh[threadIdx.x]) atomicAdd(&hist[b * IN_HIST + threadIdx.x], sh[threadIdx.x] * (jt != it ? 2 : 1));
  if (CLS_ONLY) {
    uint8_t* cb = cls_map + (size_t)b * n * n;
    for (int t = threadIdx.x; t < 32 * 32; t += 256) {
      const int ri = t >> 5, cj = t & 31;
      const int i = it * 32 + ri, j = jt * 32 + cj;
      if (i < n && j < n) cb[(size_t)i * n + j] = cls[ri][cj];
      if (jt != it) {                                  // tile (jt, it): row j = jt*32 + ri, column i = it*32 + cj
        const int j2 = jt * 32 + ri, i2 = it * 32 + cj;
        if (i2 < n && j2 < n) cb[(size_t)j2 * n + i2] = cls[cj][ri];
      }
    }
    return;
  }
  const int p = threadIdx.x / Q, q = threadIdx.x % Q;  // Q = 8: 32 pairs x 8 words per pass
  if (p < 32) {
    for (int ri = 0; ri < 32; ++ri) {
      const int i = it * 32 + ri, j = jt * 32 + p;
      if (i < n && j < n) E_out[(((size_t)b * n + i) * n + j) * Q + q] = sT[cls[ri][p] * Q + q];
    }
    if (jt != it)
      for (int rj = 0; rj < 32; ++rj) {
        const int j = jt * 32 + rj, i = it * 32 + p;
        if (i < n && j < n) E_out[(((size_t)b * n + j) * n + i) * Q + q] = sT[cls[p][rj] * Q + q];
      }
  }
}

// Pooled statistics (k_e_stats' partials: sum, sum of squares, min, max per channel) of the hidden tensor of the graphs
// that took the table path: out[b][0] from the class counts and the table, the other chunks neutral.  One block per graph.
template <int DE>
__global__ void __launch_bounds__(DE) k_e_stats_table(const uint8_t* __restrict__ flags, const bf16_t* __restrict__ T,
                                                      const int* __restrict__ hist, int de_in, float* __restrict__ out, int B,
                                                      int nchunk) {
  const int b = blockIdx.x, ch = threadIdx.x;
  if (!(flags[b] && flags[B + b])) return;             // k_e_stats reads this graph
  double s = 0.0, q = 0.0; float mn = INFINITY, mx = -INFINITY;
  for (int k = 0; k < de_in + 2; ++k) {
    const int cnt = hist[b * IN_HIST + k];
    if (!cnt) continue;
    const float v = bf16_lo((unsigned)T[k * DE + ch]);
    s += (double)cnt * v; q += (double)cnt * v * v; mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  float* o = out + (size_t)b * nchunk * 4 * DE;
  o[ch] = (float)s; o[DE + ch] = (float)q; o[2 * DE + ch] = mn; o[3 * DE + ch] = mx;
  for (int k = 1; k < nchunk; ++k) {
    float* p = o + (size_t)k * 4 * DE;
    p[ch] = 0.f; p[DE + ch] = 0.f; p[2 * DE + ch] = INFINITY; p[3 * DE + ch] = -INFINITY;
  }
}

// ---- output stage: logits E_o = sym((mlp_out_E(E_h) + E_in[..., :Ke]) * offdiag) (+ product->reactant skip) ------
template <int DE>
__device__ __forceinline__ void edge_row_load(const bf16_t* __restrict__ erow, int h, uint4 (&v)[DE / 16]) {
#pragma unroll
  for (int u = 0; u < DE / 16; ++u) v[u] = *reinterpret_cast<const uint4*>(erow + 16 * u + 8 * h);
}
template <int DE, int HE>
__device__ __forceinline__ void mlp_out_chain_regs(const uint4 (&raw)[DE / 16], int h, int lane, const uint4* sW0,
                                                   const uint4* sW1, const float* sP, f32x16& o) {
  using C = EdgeIoCfg<DE, HE>;
  bf16x8 eB[C::KS1];
#pragma unroll
  for (int u = 0; u < C::KS1; ++u) {
    uint4 v = raw[u];
    row16_to_corder(v);
    eB[u] = frag_from_u4(v);
  }
  bf16x8 hB[C::KSH];
#pragma unroll
  for (int ht = 0; ht < C::HT; ++ht) {
    f32x16 hacc;
    load_bias16(hacc, sP + 32 * ht, h);
#pragma unroll
    for (int s = 0; s < C::KS1; ++s) hacc = mfma32(lds_frag(sW0, ht * C::KS1 + s, lane), eB[s], hacc);
#pragma unroll
    for (int i2 = 0; i2 < 16; ++i2) hacc[i2] = fmaxf(hacc[i2], 0.f);
    hB[2 * ht] = pack_acc(hacc, 0); hB[2 * ht + 1] = pack_acc(hacc, 1);
  }
  load_bias16(o, sP + HE, h);
#pragma unroll
  for (int s = 0; s < C::KSH; ++s) o = mfma32(lds_frag(sW1, s, lane), hB[s], o);
}

template <int DE, int HE>
__device__ __forceinline__ void mlp_out_chain(const bf16_t* __restrict__ erow, int h, int lane, const uint4* sW0,
                                              const uint4* sW1, const float* sP, f32x16& o) {
  uint4 raw[DE / 16];
  edge_row_load<DE>(erow, h, raw);
  mlp_out_chain_regs<DE, HE>(raw, h, lane, sW0, sW1, sP, o);
}

template <int DE, int HE>
__global__ void __launch_bounds__(256) k_edge_out(const bf16_t* __restrict__ E_h, const float* __restrict__ E_in, int de_in,
                                                  const uint4* __restrict__ blob, const int* __restrict__ src,
                                                  const float* __restrict__ s1p, const float* __restrict__ s2p,
                                                  float* __restrict__ E_o, int B, int n, int Ke) {
  using C = EdgeIoCfg<DE, HE>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* sW = reinterpret_cast<uint4*>(smem);
  for (int t = threadIdx.x; t < (int)(C::OUT_BYTES / 16); t += 256) sW[t] = blob[t];
  __syncthreads();
  const uint4* sW0 = sW; const uint4* sW1 = sW + C::HT * C::KS1 * 64;
  const float* sP = reinterpret_cast<const float*>(sW + C::OUT_FR * 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int NT = (n + 31) / 32, R = B * n;
  const float s1 = src ? s1p[0] : 0.f, s2 = src ? s2p[0] : 0.f;
  for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
    const int b = row / n, i = row - b * n;
    const int si = src ? src[row] : -1;
    for (int jt = 0; jt < NT; ++jt) {
      const int jl = jt * 32 + r; const bool jvalid = jl < n; const int jc = jvalid ? jl : n - 1;
      f32x16 o, om;
      mlp_out_chain<DE, HE>(E_h + ((size_t)row * n + jc) * DE, h, lane, sW0, sW1, sP, o);
      mlp_out_chain<DE, HE>(E_h + (((size_t)b * n + jc) * n + i) * DE, h, lane, sW0, sW1, sP, om);
      if (!jvalid) continue;
      const float* ein = E_in + ((size_t)row * n + jl) * de_in;
      const float* einm = E_in + (((size_t)b * n + jl) * n + i) * de_in;
      const int sj = src ? src[b * n + jl] : -1;
      const bool aligned = si >= 0 && sj >= 0;
      const float* ealn = aligned ? E_in + (((size_t)b * n + si) * n + sj) * de_in : nullptr;
      float* dst = E_o + ((size_t)row * n + jl) * Ke;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * h + e;                      // class index = accumulator row crow(e,h), e<4
        if (c >= Ke) continue;
        float v = (jl == i) ? 0.f : 0.5f * ((o[e] + ein[c]) + (om[e] + einm[c]));
        if (src) v += aligned ? s1 * ealn[c] : (c == 0 ? s2 : 0.f);
        dst[c] = v;
      }
    }
  }
}

// ---- output stage in ONE pass over mirrored tiles: one workgroup per pair of 32x32 tiles (it <= jt) of one graph.
//   phase 1  tile (jt, it), rows j:  sB[j][i] = mlp_out_E(E_h[b,j,i]) + E_in[b,j,i]        (LDS, 32 KiB)
//   phase 2  tile (it, jt), rows i:  logits(i,j) = sym part + skip, written from registers; the symmetric part goes back
//            into sB[j][i] in place (each element is read and rewritten by the same lane)
//   phase 3  tile (jt, it) written row-wise from LDS (+ its own skip term); skipped on diagonal tiles
// Every hidden edge row is read once, row-wise, and no intermediate reaches HBM.  Same arithmetic per element as
// k_edge_out (a + b == b + a): bit-identical logits (tests).
// SAMPLE (da_edge_out_sample, Ke == 7): the reverse step's edge half replaces the logits output.  Phase 2 leaves the finished
// logits of tile (it, jt) (skip term included) in LDS instead of HBM; then every thread takes upper-triangle pairs (i < j) of the
// tile, evaluates the posterior over the 7x7 tables (posterior_elem_tab: k_posterior_E_tiles' arithmetic, same bits) and draws the
// class by the exponential race, from an explicit noise tensor or from the chain's counter-based noise (counter_exp: the values of
// da_counter_exp_noise, never materialised); the classes go through LDS to both mirrored tiles of z_s (one-hot fp32, or the
// condition's row where the pair is not free).  No logits, no phase 3.  (A first form that drew inside phase 2, on the half-wave
// that holds a row's pairs, needed 162 registers -- one workgroup per CU instead of two -- and ran 720 us against 339 + 237.)
constexpr int OT_NW = 8, OT_THR = 64 * OT_NW;   // waves / threads per workgroup of k_edge_out_tiles (4 and 16 measured: slower)
constexpr int OT_K = 7;                         // edge classes of the fused reverse step
struct OutSampleArgs {
  const float* zt; const uint8_t* nm; const float* Qt; const float* Qsb; const float* Qtb; float temp;
  const float* noise; const int64_t* seeds; unsigned long long stream;
  const float* orig; const uint8_t* me; float* zs;
  int keep_fixed;      // zs already holds the condition's values at the pairs that are not free to diffuse: they are not written
};
template <int DE, int HE, bool SAMPLE>
__global__ void __launch_bounds__(OT_THR, 4) k_edge_out_tiles(const bf16_t* __restrict__ E_h, const float* __restrict__ E_in, int de_in,
                                                        const uint4* __restrict__ blob, const int* __restrict__ src,
                                                        const float* __restrict__ s1p, const float* __restrict__ s2p,
                                                        float* __restrict__ E_o, int n, int Ke, OutSampleArgs sa) {
  using C = EdgeIoCfg<DE, HE>;
  int it, jt;
  tri_tile(blockIdx.x, (n + 31) / 32, it, jt);
  const int b = blockIdx.y;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* sW = reinterpret_cast<uint4*>(smem);
  constexpr size_t WB = (C::OUT_BYTES + 15) / 16 * 16;
  float* sB = reinterpret_cast<float*>(smem + WB);      // [32 rows j][32 columns i][8]
  float* sRow = sB + 32 * 32 * 8;                       // [wave][32 * 8]: one output row of tile (it, jt) per wave
  const bool vec4 = ((n * Ke) & 3) == 0 && (reinterpret_cast<uintptr_t>(E_o) & 15) == 0;   // rows of E_o start 16-byte aligned
  const uint4* sW0 = sW; const uint4* sW1 = sW + C::HT * C::KS1 * 64;
  const float* sP = reinterpret_cast<const float*>(sW + C::OUT_FR * 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const size_t gb = (size_t)b * n;
  const float s1 = src ? s1p[0] : 0.f, s2 = src ? s2p[0] : 0.f;
  __shared__ int sSrc[2][32];                           // alignment index of the tile's rows i ([0]) and j ([1]); -1: none
  if (threadIdx.x < 64) {
    const int w = threadIdx.x >> 5, q = (w ? jt : it) * 32 + (threadIdx.x & 31);
    sSrc[w][threadIdx.x & 31] = (src && q < n) ? src[gb + q] : -1;
  }
  __shared__ float sM[SAMPLE ? OT_K * OT_K * OT_K : 1];  // posterior factor table of this graph
  __shared__ uint8_t sCls[SAMPLE ? 32 : 1][33];          // drawn class of pair (row i, column j) of tile (it, jt), i < j
  __shared__ uint8_t sNm[2][32], sFree[2][32];           // node mask / free-to-diffuse flag of the tile's rows i ([0]) and j ([1])
  if constexpr (SAMPLE) {
    posterior_table_fill<OT_K>(sa.Qt + (size_t)b * OT_K * OT_K, sa.Qsb + (size_t)b * OT_K * OT_K, sa.Qtb + (size_t)b * OT_K * OT_K, sM,
                               threadIdx.x, OT_THR);
    if (threadIdx.x < 64) {
      const int w = threadIdx.x >> 5, q = (w ? jt : it) * 32 + (threadIdx.x & 31);
      sNm[w][threadIdx.x & 31] = q < n ? sa.nm[gb + q] : 0;
      sFree[w][threadIdx.x & 31] = (q < n && sa.me) ? sa.me[gb + q] : 1;
    }
  }
  if constexpr (SAMPLE) {
    if (sa.keep_fixed) {                                   // a tile pair without a free pair has nothing to do at all: leave before the weights are fetched
      __syncthreads();
      const unsigned long long fb0 = __ballot(sFree[lane >> 5][lane & 31] != 0);
      if ((fb0 & 0xffffffffull) == 0 || (fb0 >> 32) == 0) return;
    }
  }
  for (int t = threadIdx.x; t < (int)(C::OUT_BYTES / 16); t += OT_THR) sW[t] = blob[t];
  __syncthreads();                                         // weights, sSrc, sNm, sFree
  // SAMPLE: a pair whose endpoints are not both free to diffuse takes the condition's value (apply_mask), so its logits are never
  // looked at: rows i / j that are not free are neither loaded nor evaluated, and a tile pair without a free row on either side goes
  // straight to the copy of the condition.  With the reaction's product side fixed that is most of the tensor (73 % of the pairs
  // of the benchmark's graphs); the free pairs' logits, draws and noise indices are what they were.
  bool tile_live = true;
  if constexpr (SAMPLE) {
    const unsigned long long fb = __ballot(sFree[lane >> 5][lane & 31] != 0);
    tile_live = (fb & 0xffffffffull) != 0 && (fb >> 32) != 0;
  }
  auto live_i = [&](int ri) { return !SAMPLE || sFree[0][ri] != 0; };
  auto live_j = [&](int rj) { return !SAMPLE || sFree[1][rj] != 0; };
  uint4 cur[DE / 16], nxt[DE / 16];
  float ein_c[4], ein_n[4];                             // E_in[row, pair, 4h .. 4h+3] of the current / next row
  auto load_ein = [&](int row, int col, float (&d)[4]) {
    const float* p = E_in + ((gb + row) * n + col) * de_in + 4 * h;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = (4 * h + e < Ke) ? p[e] : 0.f;
  };
  {
    const int i = it * 32 + r, ic = i < n ? i : n - 1;
    if (tile_live && jt * 32 + wave < n && live_j(wave)) { edge_row_load<DE>(E_h + ((gb + jt * 32 + wave) * n + ic) * DE, h, cur); load_ein(jt * 32 + wave, ic, ein_c); }
  }
  for (int rj = wave; rj < 32 && tile_live; rj += OT_NW) {   // phase 1 (the next row's loads are issued before this row's MFMAs)
    const int j = jt * 32 + rj;
    if (j >= n) break;
    const int i = it * 32 + r; const bool iv = i < n; const int ic = iv ? i : n - 1;
    if (rj + OT_NW < 32 && j + OT_NW < n && live_j(rj + OT_NW)) { edge_row_load<DE>(E_h + ((gb + j + OT_NW) * n + ic) * DE, h, nxt); load_ein(j + OT_NW, ic, ein_n); }
    if (!live_j(rj)) {                                       // wave-uniform: nothing of this row is ever read
#pragma unroll
      for (int u = 0; u < DE / 16; ++u) cur[u] = nxt[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) ein_c[e] = ein_n[e];
      continue;
    }
    f32x16 o;
#ifdef EXP_OT_NOMFMA
    o = splat16(__builtin_bit_cast(float, cur[0].x));
#else
    mlp_out_chain_regs<DE, HE>(cur, h, lane, sW0, sW1, sP, o);
#endif
    if (iv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int c = 4 * h + e; if (c < Ke) sB[(rj * 32 + r) * 8 + c] = o[e] + ein_c[e]; }
    }
#pragma unroll
    for (int u = 0; u < DE / 16; ++u) cur[u] = nxt[u];
#pragma unroll
    for (int e = 0; e < 4; ++e) ein_c[e] = ein_n[e];
  }
  __syncthreads();                                         // sB complete
  const int jA = jt * 32 + r; const bool jv = jA < n; const int jc = jv ? jA : n - 1;
  const int sjA = sSrc[1][r];
  float aln_c[4], aln_n[4];                             // E_in[src i, src j, 4h .. 4h+3] where the pair is aligned, else 0
  auto load_aln = [&](int ri, float (&d)[4]) {
    const int si = sSrc[0][ri];
    const bool al = si >= 0 && sjA >= 0;
    const float* p = E_in + ((gb + (al ? si : 0)) * n + (al ? sjA : 0)) * de_in + 4 * h;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = (al && 4 * h + e < Ke) ? p[e] : 0.f;
  };
  if (tile_live && it * 32 + wave < n && live_i(wave)) {
    edge_row_load<DE>(E_h + ((gb + it * 32 + wave) * n + jc) * DE, h, cur); load_ein(it * 32 + wave, jc, ein_c);
    if (src) load_aln(wave, aln_c);
  }
  for (int ri = wave; ri < 32 && tile_live; ri += OT_NW) {   // phase 2
    const int i = it * 32 + ri;
    if (i >= n) break;
    if (ri + OT_NW < 32 && i + OT_NW < n && live_i(ri + OT_NW)) {
      edge_row_load<DE>(E_h + ((gb + i + OT_NW) * n + jc) * DE, h, nxt); load_ein(i + OT_NW, jc, ein_n);
      if (src) load_aln(ri + OT_NW, aln_n);
    }
    if (!live_i(ri)) {
#pragma unroll
      for (int u = 0; u < DE / 16; ++u) cur[u] = nxt[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) { ein_c[e] = ein_n[e]; aln_c[e] = aln_n[e]; }
      continue;
    }
    f32x16 o;
#ifdef EXP_OT_NOMFMA
    o = splat16(__builtin_bit_cast(float, cur[0].x));
#else
    mlp_out_chain_regs<DE, HE>(cur, h, lane, sW0, sW1, sP, o);
#endif
    float* srow = sRow + wave * 32 * 8;
    if (jv) {
      const bool aligned = sSrc[0][ri] >= 0 && sjA >= 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * h + e;
        if (c >= Ke) continue;
        float* m = sB + (r * 32 + ri) * 8 + c;          // (G + E_in)[b, j, i]
        float v = (jA == i) ? 0.f : 0.5f * ((o[e] + ein_c[e]) + *m);
        if constexpr (!SAMPLE) *m = v;
        if (src) v += aligned ? s1 * aln_c[e] : (c == 0 ? s2 : 0.f);
        if constexpr (SAMPLE) *m = v;                   // the finished logit of pair (i, j) stays in LDS for the draw
        else srow[r * Ke + c] = v;                      // the row's 32 * Ke logits, contiguous as in memory
      }
    }
#pragma unroll
    for (int u = 0; u < DE / 16; ++u) cur[u] = nxt[u];
#pragma unroll
    for (int e = 0; e < 4; ++e) { ein_c[e] = ein_n[e]; aln_c[e] = aln_n[e]; }
    // one coalesced store of the row from the wave's LDS staging (same-wave LDS traffic is ordered: no barrier)
    if constexpr (!SAMPLE) {
      const int cnt = min(32, n - jt * 32) * Ke;
      float* drow = E_o + ((gb + i) * n + jt * 32) * Ke;
      if (vec4 && (cnt & 3) == 0) {
        if (lane * 4 < cnt) *reinterpret_cast<float4*>(drow + lane * 4) = *reinterpret_cast<const float4*>(srow + lane * 4);
      } else {
        for (int t = lane; t < cnt; t += 64) drow[t] = srow[t];
      }
    }
  }
  if constexpr (SAMPLE) {
    __syncthreads();                                      // the tile's logits are complete in sB ([column j][row i][8])
    const size_t base = (size_t)b * n * n;
    for (int t = threadIdx.x; t < 32 * 32 && tile_live; t += OT_THR) { // pair (row ri, column cj): k_posterior_E_tiles' evaluation of it
      const int ri = t >> 5, cj = t & 31, i = it * 32 + ri, j = jt * 32 + cj;
      int c = 0;
#ifndef EXP_OT_NODRAW
      if (i < j && j < n && sNm[0][ri] && sNm[1][cj] && sFree[0][ri] && sFree[1][cj]) {
        const size_t pr = base + (size_t)i * n + j;
        float lg[OT_K], u[OT_K], q[OT_K];
        const float* lp = sB + (cj * 32 + ri) * 8;
#pragma unroll
        for (int k = 0; k < OT_K; ++k) lg[k] = lp[k];
        const int xt = onehot_argmax<OT_K>(sa.zt + pr * OT_K);
        posterior_elem_tab<OT_K>(lg, xt, sM, sa.temp, u);
        if (sa.noise) {
#pragma unroll
          for (int k = 0; k < OT_K; ++k) q[k] = sa.noise[pr * OT_K + k];
        } else {
          const unsigned long long seed = (unsigned long long)sa.seeds[b], e0 = ((unsigned long long)i * n + j) * OT_K;
#pragma unroll
          for (int k = 0; k < OT_K; ++k) q[k] = counter_exp(seed, sa.stream, e0 + k);
        }
        c = race<OT_K>(u, q);
      }
#endif
      sCls[ri][cj] = (uint8_t)c;
    }
    __syncthreads();
    // Both mirrored tiles of z_s, row-wise (a row's 32 * K floats are contiguous).  The condition's values for pairs that are not free
    // are gathered seven elements per thread at a time and the stores follow the batch: with one element per trip every trip waited
    // for its own load (14 dependent round trips per tile, about half of this kernel's time at B = 64, n = 300).
    constexpr int W = 32 * OT_K, UNW = 7;
    const float* og = sa.orig ? sa.orig : sa.zt;               // without a condition every pair is free and the value read is not used
    const bool keep = sa.keep_fixed != 0;                      // (wave-uniform) the holder already has the condition's values: free pairs only
    auto write_tile = [&](const bool mirrored) {
      for (int t0 = threadIdx.x; t0 < 32 * W; t0 += OT_THR * UNW) {
        float ov[UNW]; bool ok[UNW], fr[UNW], hot[UNW]; size_t qq[UNW];
#pragma unroll
        for (int u = 0; u < UNW; ++u) {
          const int t = min(t0 + u * OT_THR, 32 * W - 1);
          const int rr = t / W, o = t - rr * W, cx = o / OT_K, k = o - cx * OT_K;
          const int ri = mirrored ? cx : rr, cj = mirrored ? rr : cx;       // position inside tile (it, jt): row i, column j
          const int i = it * 32 + ri, j = jt * 32 + cj;
          ok[u] = t0 + u * OT_THR < 32 * W && i < n && j < n;
          const int c = mirrored ? sCls[ri][cj] : (i < j ? sCls[ri][cj] : (i > j ? sCls[cj][ri] : 0));   // i > j only inside a diagonal tile
          fr[u] = sFree[0][ri] && sFree[1][cj];
          hot[u] = k == c;
          qq[u] = (base + (mirrored ? (size_t)j * n + i : (size_t)i * n + j)) * OT_K + k;
          ov[u] = keep ? 0.f : og[ok[u] ? qq[u] : base * OT_K];   // unconditional per lane (a predicated load would make every later wait a vmcnt(0))
        }
#pragma unroll
        for (int u = 0; u < UNW; ++u)
          if (ok[u] && (fr[u] || !keep)) sa.zs[qq[u]] = fr[u] ? (hot[u] ? 1.f : 0.f) : ov[u];
      }
    };
#ifndef EXP_OT_NOWRITE
    write_tile(false);
    if (jt != it) write_tile(true);
#endif
    return;
  }
  if (jt == it) return;
  __syncthreads();
  // phase 3: rows j, columns i (contiguous floats of a tile row); the skip gathers of a batch are issued together
  const int cntB = min(32, n - it * 32) * Ke, tot = 32 * 32 * Ke;
  constexpr int UN = 7;
  for (int t0 = threadIdx.x; t0 < tot; t0 += OT_THR * UN) {
    float v[UN], g[UN]; bool ok[UN], al[UN]; size_t off[UN]; int cc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int t = t0 + u * OT_THR;
      const int rj = t / (32 * Ke), o = t - rj * 32 * Ke, j = jt * 32 + rj;
      ok[u] = t < tot && j < n && o < cntB;
      const int ci = ok[u] ? o / Ke : 0, c = ok[u] ? o - ci * Ke : 0;
      cc[u] = c;
      v[u] = ok[u] ? sB[(rj * 32 + ci) * 8 + c] : 0.f;
      const int sj = ok[u] ? sSrc[1][rj] : -1, si = sSrc[0][ci];
      al[u] = sj >= 0 && si >= 0;
      g[u] = al[u] ? E_in[((gb + sj) * n + si) * de_in + c] : 0.f;
      off[u] = ((gb + j) * n + it * 32) * Ke + o;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      float x = v[u];
      if (src) x += al[u] ? s1 * g[u] : (cc[u] == 0 ? s2 : 0.f);
      if (ok[u]) E_o[off[u]] = x;
    }
  }
}

// the part of k_edge_in_table every call needs when the table itself is already in the scratch (same weights as the call that wrote it)
__global__ void __launch_bounds__(256) k_edge_in_init(uint8_t* __restrict__ flags, int n_flags, int* __restrict__ hist, int n_hist) {
  for (int t = threadIdx.x; t < n_flags; t += 256) flags[t] = 1;
  for (int t = threadIdx.x; t < n_hist; t += 256) hist[t] = 0;
}

// ===================================================================================================
#ifndef EXP_IOGRID_CAP
#define EXP_IOGRID_CAP 2048
#endif
// Scratch of da_edge_in_sym_static: flags, table and class counts for up to DA_EDGE_IN_STATIC_MAX_B graphs (da_edge_in_scratch_bytes
// (4096, 16, 64) = 273,152 B).  Host code never uses this name as an address: hipGetSymbolAddress, per device.
#define DA_EDGE_IN_STATIC_MAX_B 4096
static __device__ __align__(256) uint8_t g_edge_in_scratch[288 * 1024];
static inline int io_grid(int rows) { int g = (rows + 3) / 4; return g < 1 ? 1 : (g > EXP_IOGRID_CAP ? EXP_IOGRID_CAP : g); }

template <int DE, int HE>
static int launch_in(const float* E_in, int de_in, const uint8_t* nm, const void* blob, void* E_out, int B, int n, hipStream_t s,
                     uint8_t* sym_flags = nullptr, uint8_t* cls_map = nullptr, bool table_valid = false) {
  using C = EdgeIoCfg<DE, HE>;
  if (sym_flags) {                                    // scratch: [B] symmetric, [B] one-hot, 16-byte aligned table
    const int nt = (n + 31) / 32;
    uint4* T = reinterpret_cast<uint4*>(sym_flags + in_table_offset(B));
    int* hist = reinterpret_cast<int*>(sym_flags + in_hist_offset(B, de_in, DE));
    if (de_in <= 8) {
      if (table_valid) k_edge_in_init<<<1, 256, 0, s>>>(sym_flags, 2 * B, hist, B * IN_HIST);
      else k_edge_in_table<DE, HE><<<1, 64, C::IN_BYTES + 32 * de_in * sizeof(float), s>>>(de_in, (const uint4*)blob, T, sym_flags, 2 * B, hist,
                                                                                           B * IN_HIST);
      const dim3 grid(nt * (nt + 1) / 2, B);
      const size_t lds = 18 * (DE / 8) * 16 + 32 * 32 * de_in * sizeof(float);
      if (cls_map) k_edge_in_tiles<DE, true><<<grid, 256, lds, s>>>(E_in, de_in, nm, T, (uint4*)E_out, n, sym_flags, hist, cls_map);
      else k_edge_in_tiles<DE, false><<<grid, 256, lds, s>>>(E_in, de_in, nm, T, (uint4*)E_out, n, sym_flags, hist, nullptr);
    } else if (hipMemsetAsync(sym_flags, 0, 2 * (size_t)B, s) != hipSuccess) { da_set_error("da_edge_in_sym: memset failed"); return DA_ERR_HIP; }
  }
  // behind the table path this kernel only rewrites the graphs that path gave up on (normally none: every workgroup reads the flags and
  // exits): a grid of 512 workgroups that walk the rows costs 5 us when there is nothing to do instead of 25 us for one workgroup per 4 rows
  const int grid_in = sym_flags ? (io_grid(B * n) < 512 ? io_grid(B * n) : 512) : io_grid(B * n);
  k_edge_in<DE, HE><<<grid_in, 256, C::IN_BYTES, s>>>(E_in, de_in, nm, (const uint4*)blob, (bf16_t*)E_out, B, n, sym_flags);
  return DA_OK;
}
template <int DE, int HE>
static int launch_out(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* s1,
                      const float* s2, float* E_o, int B, int n, int Ke, hipStream_t s) {
  using C = EdgeIoCfg<DE, HE>;
  k_edge_out<DE, HE><<<io_grid(B * n), 256, C::OUT_BYTES, s>>>((const bf16_t*)E_h, E_in, de_in, (const uint4*)blob, src, s1, s2, E_o, B, n, Ke);
  return DA_OK;
}

extern "C" {

long long da_edge_in_blob_bytes(int de, int he) {
  if (de == 64 && he == 128) return (long long)EdgeIoCfg<64, 128>::IN_BYTES;
  if (de == 32 && he == 32) return (long long)EdgeIoCfg<32, 32>::IN_BYTES;
  return -1;
}
long long da_edge_out_blob_bytes(int de, int he) {
  if (de == 64 && he == 128) return (long long)EdgeIoCfg<64, 128>::OUT_BYTES;
  if (de == 32 && he == 32) return (long long)EdgeIoCfg<32, 32>::OUT_BYTES;
  return -1;
}

// E_h(bf16) = encode_no_element(mask(sym(mlp_in_E(E))))     transformer_model_with_y.py:657-661
int da_edge_in(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, int B, int n, int de,
               int he, void* stream) {
  DA_CHECK_ARG(de_in >= 1 && de_in <= 16, "da_edge_in: input edge feature width %d not in 1..16", de_in);
  if (B == 0 || n == 0) return DA_OK;
  if (de == 64 && he == 128) launch_in<64, 128>(E_in, de_in, node_mask, blob, E_h, B, n, (hipStream_t)stream);
  else if (de == 32 && he == 32) launch_in<32, 32>(E_in, de_in, node_mask, blob, E_h, B, n, (hipStream_t)stream);
  else { da_set_error("da_edge_in: dims (de=%d, hidden_mlp E=%d) not built; built: (64,128), (32,32)", de, he); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_edge_in");
  return DA_OK;
}

// da_edge_in_sym for a caller whose first edge layer is da_edge_layer_table: graphs that are symmetric and one-hot get one byte per
// pair in cls_map (B,n,n) -- the row of the table in `scratch` that holds the pair's hidden features -- and their part of E_h is
// NOT written; the other graphs (flags cleared in `scratch`) get their hidden rows in E_h as da_edge_in writes them.
// de_in <= 8 (the table path); scratch: da_edge_in_scratch_bytes(B, de_in, de) bytes.
static int edge_in_classes_impl(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, uint8_t* cls_map,
                                uint8_t* scratch, int B, int n, int de, int he, bool table_valid, void* stream) {
  DA_CHECK_ARG(de_in >= 1 && de_in <= 8, "da_edge_in_classes: input edge feature width %d not in 1..8", de_in);
  DA_CHECK_ARG(scratch != nullptr && cls_map != nullptr, "da_edge_in_classes: scratch / cls_map is null");
  if (B == 0 || n == 0) return DA_OK;
  int rc;
  if (de == 64 && he == 128) rc = launch_in<64, 128>(E_in, de_in, node_mask, blob, E_h, B, n, (hipStream_t)stream, scratch, cls_map, table_valid);
  else if (de == 32 && he == 32) rc = launch_in<32, 32>(E_in, de_in, node_mask, blob, E_h, B, n, (hipStream_t)stream, scratch, cls_map, table_valid);
  else { da_set_error("da_edge_in_classes: dims (de=%d, hidden=%d) not built; built: (64,128), (32,32)", de, he); return DA_ERR_UNSUPPORTED; }
  if (rc) return rc;
  DA_CHECK_LAUNCH("da_edge_in_classes");
  return DA_OK;
}
int da_edge_in_classes(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, uint8_t* cls_map,
                       uint8_t* scratch, int B, int n, int de, int he, void* stream) {
  return edge_in_classes_impl(E_in, de_in, node_mask, blob, E_h, cls_map, scratch, B, n, de, he, false, stream);
}
// da_edge_in_classes for a caller that keeps `scratch` between calls: table_valid != 0 says that the scratch still holds the table an
// earlier call with the SAME blob, B and de_in left there (it depends on the weights alone), so only the flags and class counts are
// set up again (a 1-wave MLP evaluation of 15 us in front of every step's input stage otherwise).
int da_edge_in_classes_keep(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, uint8_t* cls_map,
                            uint8_t* scratch, int B, int n, int de, int he, int table_valid, void* stream) {
  return edge_in_classes_impl(E_in, de_in, node_mask, blob, E_h, cls_map, scratch, B, n, de, he, table_valid != 0, stream);
}

// Scratch da_edge_in_sym needs: 2B flag bytes (symmetric, one-hot), the (de_in + 2) x de bf16 table of the one-hot path and
// the per-graph class counts (16 ints).
long long da_edge_in_scratch_bytes(int B, int de_in, int de) {
  if (B < 0 || de_in < 1 || de_in > 16 || de < 8) return -1;
  return (long long)in_hist_offset(B, de_in, de) + (long long)B * IN_HIST * sizeof(int);
}

// Statistics partials (da_edge_stats' layout) of the graphs da_edge_in_sym served from its table, from the class counts it
// left in `scratch`; returns in *skip the flag array to hand to da_edge_stats_skip for the other graphs.
int da_edge_stats_table(const uint8_t* scratch, float* partials, int B, int de_in, int de, int nchunk, void* stream) {
  DA_CHECK_ARG(scratch != nullptr && nchunk >= 1 && de_in >= 1 && de_in <= 8, "da_edge_stats_table: bad arguments");
  if (B == 0) return DA_OK;
  const bf16_t* T = reinterpret_cast<const bf16_t*>(scratch + in_table_offset(B));
  const int* hist = reinterpret_cast<const int*>(scratch + in_hist_offset(B, de_in, de));
  if (de == 64) k_e_stats_table<64><<<B, 64, 0, (hipStream_t)stream>>>(scratch, T, hist, de_in, partials, B, nchunk);
  else if (de == 32) k_e_stats_table<32><<<B, 32, 0, (hipStream_t)stream>>>(scratch, T, hist, de_in, partials, B, nchunk);
  else { da_set_error("da_edge_stats_table: de=%d not built (32, 64)", de); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_edge_stats_table");
  return DA_OK;
}

// da_edge_in with the symmetry and one-hotness of every input graph established beforehand by a coalesced pass
// (sym_flags: da_edge_in_scratch_bytes() of scratch the caller owns, 256-byte aligned; on return its first B bytes say
// which graphs are symmetric, the next B which are one-hot) instead of a transposed, uncoalesced read per pair.
// Graphs that are both (every diffusion state) are written by a table gather.  Same result as da_edge_in, bit for bit.
int da_edge_in_sym(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, uint8_t* sym_flags, int B,
                   int n, int de, int he, void* stream) {
  DA_CHECK_ARG(de_in >= 1 && de_in <= 16, "da_edge_in_sym: input edge feature width %d not in 1..16", de_in);
  DA_CHECK_ARG(sym_flags != nullptr && ((uintptr_t)sym_flags & 15) == 0, "da_edge_in_sym: scratch is null or not 16-byte aligned");
  if (B == 0 || n == 0) return DA_OK;
  int rc;
  if (de == 64 && he == 128) rc = launch_in<64, 128>(E_in, de_in, node_mask, blob, E_h, B, n, (hipStream_t)stream, sym_flags);
  else if (de == 32 && he == 32) rc = launch_in<32, 32>(E_in, de_in, node_mask, blob, E_h, B, n, (hipStream_t)stream, sym_flags);
  else { da_set_error("da_edge_in_sym: dims (de=%d, hidden_mlp E=%d) not built; built: (64,128), (32,32)", de, he); return DA_ERR_UNSUPPORTED; }
  if (rc) return rc;
  DA_CHECK_LAUNCH("da_edge_in_sym");
  return DA_OK;
}

}  // extern "C"
// Device address of g_edge_in_scratch on the device that owns `stream` (the current device for the null stream), resolved once per
// device with hipGetSymbolAddress (each GPU holds its own copy of the array at its own address) and kept.  da_init() resolves it
// ahead of time, so that the first da_edge_in_sym_static call of a device may already be under stream capture.
static int edge_in_static_scratch(hipStream_t stream, uint8_t** out) {
  static uint8_t* ptr_of_device[DA_MAX_DEVICES];
  int dev = 0, sdev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DA_MAX_DEVICES) { da_set_error("da_edge_in_sym_static: no current HIP device"); return DA_ERR_HIP; }
  if (stream != nullptr && hipStreamGetDevice(stream, &sdev) == hipSuccess && sdev != dev) {
    da_set_error("da_edge_in_sym_static: the stream belongs to device %d but device %d is current", sdev, dev);
    return DA_ERR_ARG;
  }
  uint8_t* p = __atomic_load_n(&ptr_of_device[dev], __ATOMIC_ACQUIRE);
  if (p == nullptr) {
    hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_edge_in_scratch));
    if (e != hipSuccess || p == nullptr) { da_set_error("da_edge_in_sym_static: hipGetSymbolAddress: %s", hipGetErrorString(e)); return DA_ERR_HIP; }
    __atomic_store_n(&ptr_of_device[dev], p, __ATOMIC_RELEASE);
  }
  *out = p;
  return DA_OK;
}
extern "C" {
// Per-device set-up that must not happen under stream capture: resolves the static scratch of da_edge_in_sym_static on the CURRENT
// device.  Call once per device (diffalign_amd/_lib.py does, for the current device, when it loads the library on a GPU machine).
int da_init(void) {
  uint8_t* p = nullptr;
  return edge_in_static_scratch(nullptr, &p);
}

// da_edge_in_sym for callers that cannot allocate (HIP graph capture, serving loops): the scratch is a static device array of the
// library instead of a workspace argument.  No hipMalloc / hipFree; B <= DA_EDGE_IN_STATIC_MAX_B.  The array belongs to the
// device: calls on one GPU share it, so calls that overlap on one GPU (different streams) race; it is rewritten by every call.
int da_edge_in_sym_static(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, int B, int n, int de,
                          int he, void* stream) {
  DA_CHECK_ARG(B >= 0 && B <= DA_EDGE_IN_STATIC_MAX_B, "da_edge_in_sym_static: B = %d exceeds the static scratch (B <= %d); use da_edge_in_sym "
               "with a caller-owned scratch", B, DA_EDGE_IN_STATIC_MAX_B);
  DA_CHECK_ARG(de_in >= 1 && de_in <= 16 && de >= 8 && de <= 64, "da_edge_in_sym_static: widths (de_in=%d, de=%d) outside the static scratch's layout", de_in, de);
  if (B == 0 || n == 0) return DA_OK;
  DA_CHECK_ARG((size_t)da_edge_in_scratch_bytes(B, de_in, de) <= sizeof(g_edge_in_scratch), "da_edge_in_sym_static: scratch too small");
  uint8_t* scratch = nullptr;
  if (int rc = edge_in_static_scratch((hipStream_t)stream, &scratch)) return rc;
  return da_edge_in_sym(E_in, de_in, node_mask, blob, E_h, scratch, B, n, de, he, stream);
}

// E logits = sym((mlp_out_E(E_h)+E_in[..., :ke])*offdiag) [+ skip]   transformer_model_with_y.py:672-704
int da_edge_out(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1,
                const float* skip2, float* E_logits, int B, int n, int de, int he, int ke, void* stream) {
  DA_CHECK_ARG(ke >= 1 && ke <= 8 && ke <= de_in, "da_edge_out: ke=%d must be in 1..8 and <= input width", ke);
  DA_CHECK_ARG(!(src && (!skip1 || !skip2)), "da_edge_out: skip connection needs both scale pointers");
  if (B == 0 || n == 0) return DA_OK;
  if (de == 64 && he == 128) launch_out<64, 128>(E_h, E_in, de_in, blob, src, skip1, skip2, E_logits, B, n, ke, (hipStream_t)stream);
  else if (de == 32 && he == 32) launch_out<32, 32>(E_h, E_in, de_in, blob, src, skip1, skip2, E_logits, B, n, ke, (hipStream_t)stream);
  else { da_set_error("da_edge_out: dims (de=%d, hidden_mlp E=%d) not built; built: (64,128), (32,32)", de, he); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_edge_out");
  return DA_OK;
}

// Output stage in one pass over mirrored 32x32 tiles (see k_edge_out_tiles); same arguments and result as da_edge_out.
int da_edge_out_tiles(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1,
                      const float* skip2, float* E_logits, int B, int n, int de, int he, int ke, void* stream) {
  DA_CHECK_ARG(ke >= 1 && ke <= 8 && ke <= de_in, "da_edge_out_tiles: ke=%d must be in 1..8 and <= input width", ke);
  DA_CHECK_ARG(!(src && (!skip1 || !skip2)), "da_edge_out_tiles: skip connection needs both scale pointers");
  if (B == 0 || n == 0) return DA_OK;
  const int nt = (n + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  if (de == 64 && he == 128)
    k_edge_out_tiles<64, 128, false><<<dim3(nt * (nt + 1) / 2, B), OT_THR, (EdgeIoCfg<64, 128>::OUT_BYTES + 15) / 16 * 16 + (32 * 32 * 8 + OT_NW * 32 * 8) * sizeof(float), s>>>(
        (const bf16_t*)E_h, E_in, de_in, (const uint4*)blob, src, skip1, skip2, E_logits, n, ke, OutSampleArgs{});
  else if (de == 32 && he == 32)
    k_edge_out_tiles<32, 32, false><<<dim3(nt * (nt + 1) / 2, B), OT_THR, (EdgeIoCfg<32, 32>::OUT_BYTES + 15) / 16 * 16 + (32 * 32 * 8 + OT_NW * 32 * 8) * sizeof(float), s>>>(
        (const bf16_t*)E_h, E_in, de_in, (const uint4*)blob, src, skip1, skip2, E_logits, n, ke, OutSampleArgs{});
  else { da_set_error("da_edge_out_tiles: dims (de=%d, hidden_mlp E=%d) not built; built: (64,128), (32,32)", de, he); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_edge_out_tiles");
  return DA_OK;
}

// Output stage + the edge half of the reverse step in one kernel (sampling.py:82-86,165-178 behind transformer_model_with_y.py:
// 672-704): z_s E (B,n,n,7) one-hot fp32 drawn from p(z_s | z_t) of the logits this stage produces, with apply_mask folded in, exactly
// as da_edge_out_tiles followed by da_posterior_sample's edge half (draw-only form) gives -- same logits, same posterior arithmetic,
// same race: bit-identical z_s for the same noise.  Noise: noise_E (B,n,n,7) Exp(1) (upper triangle read), or NULL with chain_seeds (B)
// and `call`: the values da_counter_exp_noise(chain_seeds, call) would put there, generated in registers.  The logits are not written.
int da_edge_out_sample_ex(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1, const float* skip2,
                          const float* zt_E, const uint8_t* node_mask, const float* Qt_e, const float* Qsb_e, const float* Qtb_e,
                          float temp_e, const float* noise_E, const int64_t* chain_seeds, long long call, const float* orig_E, const uint8_t* mask_e,
                          float* zs_E, int B, int n, int de, int he, int ke, int flags, void* stream) {
  DA_CHECK_ARG((flags & ~1) == 0, "da_edge_out_sample_ex: unknown flags %d", flags);
  DA_CHECK_ARG(!(flags & 1) || (mask_e && orig_E), "da_edge_out_sample_ex: flag 1 (fixed pairs kept) needs a free mask and the condition");
  DA_CHECK_ARG(ke == OT_K && ke <= de_in, "da_edge_out_sample: built for %d edge classes (got %d, input width %d)", OT_K, ke, de_in);
  DA_CHECK_ARG(!(src && (!skip1 || !skip2)), "da_edge_out_sample: skip connection needs both scale pointers");
  DA_CHECK_ARG(zt_E && node_mask && Qt_e && Qsb_e && Qtb_e && zs_E, "da_edge_out_sample: null pointer");
  DA_CHECK_ARG(noise_E || (chain_seeds && call >= 0), "da_edge_out_sample: needs noise_E or chain_seeds + call");
  DA_CHECK_ARG(!mask_e || orig_E, "da_edge_out_sample: a free mask needs the condition's E");
  if (B == 0 || n == 0) return DA_OK;
  const int nt = (n + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  OutSampleArgs sa{zt_E, node_mask, Qt_e, Qsb_e, Qtb_e, temp_e, noise_E, chain_seeds, 2ull * (unsigned long long)call + 1ull, orig_E, mask_e, zs_E, flags & 1};
  if (de == 64 && he == 128)
    k_edge_out_tiles<64, 128, true><<<dim3(nt * (nt + 1) / 2, B), OT_THR, (EdgeIoCfg<64, 128>::OUT_BYTES + 15) / 16 * 16 + (32 * 32 * 8 + OT_NW * 32 * 8) * sizeof(float), s>>>(
        (const bf16_t*)E_h, E_in, de_in, (const uint4*)blob, src, skip1, skip2, nullptr, n, ke, sa);
  else if (de == 32 && he == 32)
    k_edge_out_tiles<32, 32, true><<<dim3(nt * (nt + 1) / 2, B), OT_THR, (EdgeIoCfg<32, 32>::OUT_BYTES + 15) / 16 * 16 + (32 * 32 * 8 + OT_NW * 32 * 8) * sizeof(float), s>>>(
        (const bf16_t*)E_h, E_in, de_in, (const uint4*)blob, src, skip1, skip2, nullptr, n, ke, sa);
  else { da_set_error("da_edge_out_sample: dims (de=%d, hidden_mlp E=%d) not built; built: (64,128), (32,32)", de, he); return DA_ERR_UNSUPPORTED; }
  DA_CHECK_LAUNCH("da_edge_out_sample");
  return DA_OK;
}
int da_edge_out_sample(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1, const float* skip2,
                       const float* zt_E, const uint8_t* node_mask, const float* Qt_e, const float* Qsb_e, const float* Qtb_e,
                       float temp_e, const float* noise_E, const int64_t* chain_seeds, long long call, const float* orig_E, const uint8_t* mask_e,
                       float* zs_E, int B, int n, int de, int he, int ke, void* stream) {
  return da_edge_out_sample_ex(E_h, E_in, de_in, blob, src, skip1, skip2, zt_E, node_mask, Qt_e, Qsb_e, Qtb_e, temp_e, noise_E, chain_seeds, call,
                               orig_E, mask_e, zs_E, B, n, de, he, ke, 0, stream);
}

}  // extern "C"
