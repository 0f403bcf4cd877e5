// Fused node-edge attention layer, edge side (the ~85 % of the reference's step time).
//
// Replaces, for one XEyTransformerLayer (neuralnet/layers.py:204-284 NodeEdgeBlock + :330-361 the E branch):
//   E1=e_mul(E)*m, E2=e_add(E)*m, Y=(Q_i*K_j/sqrt(df))*(E1+1)+E2, newE=e_out(ye1+(ye2+1)*Y)*m,
//   attn=softmax_j(Y) per channel WITHOUT key mask, wV=sum_j attn*V_j, newX_in = yx1+(yx2+1)*wV,
//   E' = LN2(Em + linE2(relu(linE1(Em)))),  Em = LN1(E + newE)
// in ONE pass over the (B,n,n,de) edge tensor: each 64-B..128-B edge row is read once and written once (bf16).
//
// Structure (MI355X): persistent workgroups, all layer weights resident in LDS as pre-packed MFMA fragments
// (132 KiB at full dims), every wave owns whole query rows (b,i) and walks the key axis in 32-pair tiles.
// Per tile and 32-channel block:  GEMM1 (pairs x channels, softmax-friendly: key index in registers, channel on the
// lane) -> elementwise Y, online softmax, FiLM -> in-register transpose by an identity MFMA -> GEMM2 accumulates
// newE^T (channels in registers, pair on the lane) -> LN1 -> FFN chained through accumulators -> LN2 -> store.
// No LDS traffic for activations, no barriers in the main loop.
#include "mfma_util.h"
#include "edge_in_layout.h"
#include <atomic>
#include <cstdlib>

// Timing-experiment hooks (tools/build_variant.sh): each removes one cost to measure what it hides.  Results are WRONG
// with any of them defined; the product build defines none.
#ifdef EXP_NOLDS
#define LDSF(base, frag, lane) lds_frag(base, 0, lane)
#else
#define LDSF(base, frag, lane) lds_frag(base, frag, lane)
#endif
#ifdef EXP_NOMFMA
#define mfma32(a, b, c) fake_mfma(a, b, c)
__device__ __forceinline__ f32x16 fake_mfma(bf16x8 a, bf16x8 b, f32x16 c) { c[0] += (float)(a[0] ^ b[0]); return c; }
#endif
#ifdef EXP_NOEXP
#define EXP2F(x) (x)
#else
#define EXP2F(x) __builtin_amdgcn_exp2f(x)
#endif

// Diagnostic build only (EXP_STAMP): per-wave cycle totals of the tile's segments, s_memtime around each, written to a
// buffer of their own that nothing else reads (tools/stamp_edge.py).  The product build has no stamp.
#ifdef EXP_STAMP
#define NSTAMP 12
__device__ long long g_stamps[4096 * NSTAMP];
struct Stamps { long long acc[NSTAMP]; long long prev; };
#define STAMP(i) { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_readcyclecounter(); S.acc[i] += t_ - S.prev; S.prev = t_; __builtin_amdgcn_sched_barrier(0); }
extern "C" int da_debug_read_stamps(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n); }
#else
struct Stamps {};
#define STAMP(i)
#endif

template <int DE, int DX, int DFF>
struct EdgeCfg {
  static constexpr int NCB = DX / 32;
  static constexpr int KS1 = DE / 16;
  static constexpr int MT = DE / 32;
  static constexpr int HT = DFF / 32;
  static constexpr int KSF = DFF / 16;
  static constexpr int W1_FR = NCB * 2 * KS1;
  static constexpr int WO_FR = NCB * MT * 2;
  static constexpr int F1_FR = HT * KS1;
  static constexpr int F2_FR = MT * KSF;
  static constexpr int N_FR = W1_FR + WO_FR + F1_FR + F2_FR;
  // fp32 params after the fragments: bm1[DX] ba[DX] ln1g[DE] ln1b[DE] bf1[DFF] bf2[DE] ln2g[DE] ln2b[DE]
  static constexpr int P_BM1 = 0, P_BA = DX, P_LN1G = 2 * DX, P_LN1B = 2 * DX + DE, P_BF1 = 2 * DX + 2 * DE,
                       P_BF2 = 2 * DX + 2 * DE + DFF, P_LN2G = 2 * DX + 3 * DE + DFF, P_LN2B = 2 * DX + 4 * DE + DFF,
                       N_PAR = 2 * DX + 5 * DE + DFF;
  static constexpr size_t BLOB_BYTES = (size_t)N_FR * 1024 + (size_t)N_PAR * 4;
#ifdef EXP_NOEPI
  static constexpr size_t LDS_SKIP = (size_t)(F1_FR + F2_FR) * 1024;      // timing experiment: the FFN fragments stay out of LDS
#else
  static constexpr size_t LDS_SKIP = 0;
#endif
  static constexpr size_t LDS_BLOB_BYTES = BLOB_BYTES - LDS_SKIP;
  static constexpr size_t STATE_BYTES_PER_WAVE = (size_t)NCB * 3 * 64 * 4;
};

struct EdgeLayerArgs {
  const bf16_t* E_in;     // (B,n,n,DE) bf16
  bf16_t* E_out;          // (B,n,n,DE) bf16
  const float* Q;         // (B*n, ldq) q(X)*mask
  int ldq;
  const uint4* KV;        // fragment layout, see k_kv_pack
  const uint8_t* node_mask;
  const float* ye2p1;     // (B,DX)  1 + y_e_mul(y)
  const float* ebias;     // (B,DE)  e_out.bias + e_out.W @ y_e_add(y)
  const float* yx1;       // (B,DX)  y_x_add(y)
  const float* yx2p1;     // (B,DX)  1 + y_x_mul(y)
  float* newX_in;         // (B,n,DX) = yx1 + yx2p1 * wV   (input of x_out)
  const uint4* blob;      // packed layer weights (EdgeCfg layout)
  float qscale;           // 1/sqrt(df)
  int B, n;
  float* split;           // tail-split scratch [8][slots_per_xcd][3][DX] (running max, sum, weighted sum) or null
  int slots_per_xcd;
  int stagger, prio;      // waves NW/2.. start `stagger`*64 cycles late / run at s_setprio 1 (breaks SIMD-partner lockstep)
  int skew;               // the older half of a workgroup's waves takes `skew` whole rows more, the younger half as many fewer
  int tail_old;           // the runs of the split rows all go to the older half
  int skip_dead;          // tiles without a single real pair (padded query row, or 32 padded keys) skip the attention part
  float* stats;           // Etoy statistics partials of E_in, k_e_stats' layout [B][stats_nchunk][4][DE], or null
  int stats_nchunk;
  int* ticket;            // work counter of the statistics tail (zero on entry)
  // table input (first layer, TAB instantiation): the input rows of graphs with tab_flags[b] && tab_flags[B + b] are not read from
  // E_in but taken from tab[cls[b][i][j]] (the input stage's (tab_rows x DE) bf16 table, da_edge_in_classes); other graphs: E_in
  const uint8_t* cls; const uint4* tab; const uint8_t* tab_flags; int tab_rows;
  // query rows to compute (da_edge_layer_ex): rowmap[0 .. n_rows) are the row numbers b n + i, ascending; the partition below works on
  // positions in this list.  null: every row.  (The last layer of a fused reverse step needs only the rows of nodes that are free to
  // diffuse: nobody reads E'[i][.] or newX_in[i] of the others.)
  const int* rowmap; int n_rows;
  // one word per query row b n + i: bit jt set = E_out[row][keys 32 jt .. 32 jt + 31] is wanted.  A tile whose bit is clear computes its
  // attention part only (the row's newX_in needs every key): no residual, LayerNorms, FFN or store for it.  null: every tile in full
  const unsigned* row_tiles;
  const int* n_rows_dev;  // the list's length read on the device (a launch captured in a HIP graph serves lists of any length); null: n_rows
};

// One 32-channel block of one 32-pair tile: GEMM1, Y, online softmax, FiLM, transpose, GEMM2 accumulate.
// MASKED=false is the fast path for tiles whose 32 key rows all exist and are real nodes (and a real query node).
// VALU diet (the kernel is VALU-issue bound, not MFMA bound):
//  * the e_mul bias (+1) enters GEMM1 as one extra MFMA (ones column x [hi,lo] bf16 split of the bias) instead of 16 v_mov;
//  * the e_add bias ba is dropped from Y: the per-channel softmax is shift invariant, and the FiLM product becomes
//    Z = f2c*y' + f2c*ba (one packed fma); masked keys carry y' = -ba so that their true score is 0;
//  * all elementwise work is written on float pairs so that it maps to v_pk_{mul,fma,add}_f32;
//  * the high halves of the packed K/V words are used as fp32 without masking (k_kv_pack compensates the rounding).
// Work partition of the persistent grid.  Rows are cut into 8 contiguous chunks, one per XCD (workgroups with equal
// blockIdx % 8 share an XCD and its L2).  Inside a chunk the W = (gridDim/8) * waves-per-workgroup waves take whole rows
// round-robin; with a split scratch the rows of the last, incomplete round are cut into runs of key tiles instead, so
// that every wave gets the same amount of work (at B*n = 19,200 rows and 2,048 waves the incomplete round would keep
// 37 % of the waves busy for a full row time).  A run leaves its online-softmax state in the scratch; k_edge_split_merge
// folds the runs of a row.  Both kernels (and the host sizing) use this one definition.
// The whole rows are not shared out evenly: the older half of a workgroup's waves (one of every SIMD pair) takes `skew` rows
// more and the younger half as many fewer, because the older wave of a pair wins the issue arbitration and would otherwise
// finish at ~80 % of the kernel's time (EdgeLayerArgs::skew and tail_old; 107.5 / 80 tiles instead of 93.75 / 93.75 at the
// benchmark size).
// (Handing the units out dynamically inside a workgroup, by a ticket counter in LDS, was measured and is 7 % SLOWER:
// with static shares the older wave of each SIMD pair wins the issue arbitration, finishes its share in 80 % of the time
// and leaves its partner a stretch alone; balanced pairs that stay in step get less out of the SIMD than that.)
struct EdgePart {
  int row_begin, row_end, W, nfull, rem, tpw;
  __host__ __device__ EdgePart(int R, int grid_blocks, int NW, int NT, int xcd, bool split, bool tail_old = false) {
    const int chunk = (R + 7) / 8;
    row_begin = xcd * chunk; row_end = row_begin + chunk < R ? row_begin + chunk : R;
    if (row_end < row_begin) row_end = row_begin;
    W = (grid_blocks >> 3) * NW;
    const int rows = row_end - row_begin;
    nfull = split ? rows / W : (rows + W - 1) / W;
    rem = split ? rows - nfull * W : 0;
    const int Wt = tail_old ? W / 2 : W;               // waves that take runs of the split rows (tail_old: the older half only)
    tpw = rem > 0 ? (rem * NT + Wt - 1) / Wt : 0;
  }
};

// GEMM1 of one 32-channel block: e1 = E W_mul^T + (1 + b_mul) (bias through the ones k-step), e2 = E W_add^T
template <int DE, int DX, int DFF>
__device__ __forceinline__ void gemm1_block(const int cb, const int lane, const uint4 (&eraw)[DE / 16], const uint4* sW1,
                                            const unsigned* sB1, const bf16x8 onesA, f32x16& e1, f32x16& e2) {
  using C = EdgeCfg<DE, DX, DFF>;
  const uint4 bfr = {sB1[cb * 64 + lane], 0u, 0u, 0u};
  e1 = mfma32(onesA, frag_from_u4(bfr), splat16(0.f));
  e2 = splat16(0.f);
#pragma unroll
  for (int s = 0; s < C::KS1; ++s) {
    e1 = mfma32(frag_from_u4(eraw[s]), LDSF(sW1, (cb * 2 + 0) * C::KS1 + s, lane), e1);
    e2 = mfma32(frag_from_u4(eraw[s]), LDSF(sW1, (cb * 2 + 1) * C::KS1 + s, lane), e2);
  }
}

// transpose Z (pairs x 32 channels, in e1) -> Z^T in registers by an identity MFMA, then GEMM2: acc2 += W_out Z^T
template <int DE, int DX, int DFF>
__device__ __forceinline__ void gemm2_block(const int cb, const int lane, const f32x16& z, const uint4* sWO, const bf16x8 id0,
                                            const bf16x8 id1, f32x16 (&acc2)[DE / 32]) {
  using C = EdgeCfg<DE, DX, DFF>;
#ifdef EXP_NOTRANSPOSE
  const bf16x8 zb0 = pack_acc(z, 0), zb1 = pack_acc(z, 1);          // timing experiment: what the identity-MFMA transpose costs
#else
  f32x16 zt = splat16(0.f);
  zt = mfma32(pack_acc(z, 0), id0, zt);
  zt = mfma32(pack_acc(z, 1), id1, zt);
  const bf16x8 zb0 = pack_acc(zt, 0), zb1 = pack_acc(zt, 1);
#endif
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    acc2[mt] = mfma32(LDSF(sWO, (cb * C::MT + mt) * 2 + 0, lane), zb0, acc2[mt]);
    acc2[mt] = mfma32(LDSF(sWO, (cb * C::MT + mt) * 2 + 1, lane), zb1, acc2[mt]);
  }
}

// Opaque use+def: the value must be computed here (keeps LLVM's IR-level sinking from moving a chunk's arithmetic
// past the scheduling barrier that follows it; emits no instruction).
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

// GEMM1 of the NEXT channel block, one MFMA per call, fed into the elementwise work of the current block (below).
// Each step issues its MFMA with the LDS fragment fetched by the previous step and fetches the next fragment.
template <int DE, int DX, int DFF>
struct Gemm1Feed {
  using C = EdgeCfg<DE, DX, DFF>;
  static constexpr int N = 1 + 2 * C::KS1;
  const bool on;                 // false: no next block, every step is a no-op
  const int cbn, lane;
  const uint4 (&eraw)[DE / 16];
  const uint4* sW1;
  const bf16x8 onesA;
  f32x16& f1;
  f32x16& f2;
  uint4 frag;                    // operand of the next step
  __device__ __forceinline__ Gemm1Feed(bool on_, int cbn_, int lane_, const uint4 (&eraw_)[DE / 16], const uint4* sW1_,
                                       const unsigned* sB1, bf16x8 onesA_, f32x16& f1_, f32x16& f2_)
      : on(on_), cbn(cbn_), lane(lane_), eraw(eraw_), sW1(sW1_), onesA(onesA_), f1(f1_), f2(f2_) {
    if (on) frag = uint4{sB1[cbn * 64 + lane], 0u, 0u, 0u};
  }
  __device__ __forceinline__ void step(const int i) {
    if (!on || i >= N) return;
    const bf16x8 w = frag_from_u4(frag);
    if (i + 1 < N) frag = sW1[((cbn * 2 + (i & 1)) * C::KS1 + (i >> 1)) * 64 + lane];   // fragment of step i+1: (s, which) = (i>>1, i&1)
    if (i == 0) { f1 = mfma32(onesA, w, splat16(0.f)); return; }
    const int s2 = (i - 1) >> 1;
    if ((i - 1) & 1) f2 = mfma32(frag_from_u4(eraw[s2]), w, i == 2 ? splat16(0.f) : f2);
    else f1 = mfma32(frag_from_u4(eraw[s2]), w, f1);
  }
};

// Elementwise middle of one channel block: Y, online softmax, FiLM.  In: e1, e2 (GEMM1).  Out: e1 = Z.
// The work is cut into chunks of four keys; after each chunk one MFMA of the next block's GEMM1 is issued (feed) and a
// scheduling barrier pins that order, so that the wave's own matrix and vector work overlap (left alone, the compiler
// issues the nine MFMAs back to back and the ~110 VALU instructions as one run).
template <int DE, int DX, int DFF, bool MASKED>
__device__ __forceinline__ void softmax_film_block(const int cb, const int lane, f32x16& e1, const f32x16& e2, const float* sP,
                                                   uint4 (&kv)[4], const uint4* kv_next, const float qc, const float f2c,
                                                   const bool mi, const unsigned nmbits, const unsigned vbits, float& sm_m,
                                                   float& sm_l, float& sm_a, Gemm1Feed<DE, DX, DFF>& feed, Stamps& S) {
  using C = EdgeCfg<DE, DX, DFF>;
  constexpr float LOG2E = 1.4426950408889634f;
  const int r = lane & 31;
  const float bac = sP[C::P_BA + 32 * cb + r];
  const float baf = bac * f2c;
  const unsigned kk[8] = {kv[0].x, kv[0].y, kv[0].z, kv[0].w, kv[1].x, kv[1].y, kv[1].z, kv[1].w};
  const unsigned vv[8] = {kv[2].x, kv[2].y, kv[2].z, kv[2].w, kv[3].x, kv[3].y, kv[3].z, kv[3].w};
  float tmax = -INFINITY;
  int fs = 0;                                       // next feed step
#ifdef EXP_NOINTERLEAVE
#pragma unroll
  for (; fs < Gemm1Feed<DE, DX, DFF>::N; ++fs) feed.step(fs);
#else
  feed.step(fs++);
#endif
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // K/V word order (k_kv_pack): words (2m, 2m+1) carry keys (4m, 4m+1) in their low halves and (4m+2, 4m+3) in their
    // high halves, so both float pairs below are adjacent registers: two shifts per four keys, no other unpacking
    const int m = k >> 1;
    f32x2 kq;
    if (k & 1) kq = f32x2{__builtin_bit_cast(float, kk[2 * m]), __builtin_bit_cast(float, kk[2 * m + 1])};
    else kq = f32x2{bf16_lo(kk[2 * m]), bf16_lo(kk[2 * m + 1])};
    kq *= qc;
    const f32x2 a1 = {e1[2 * k], e1[2 * k + 1]}, a2 = {e2[2 * k], e2[2 * k + 1]};
    f32x2 y = kq * a1 + a2;
    if (MASKED) {
      const int b0 = ((2 * k) & 3) + 8 * ((2 * k) >> 2), b1 = b0 + 1;
      y.x = (mi && ((nmbits >> b0) & 1u)) ? y.x : -bac;
      y.y = (mi && ((nmbits >> b1) & 1u)) ? y.y : -bac;
      tmax = fmaxf(tmax, ((vbits >> b0) & 1u) ? y.x : -INFINITY);
      tmax = fmaxf(tmax, ((vbits >> b1) & 1u) ? y.y : -INFINITY);
    } else {
#ifndef EXP_NOSHIFT
      tmax = fmaxf(tmax, fmaxf(y.x, y.y));
#endif
    }
    pin(y);
    e1[2 * k] = y.x; e1[2 * k + 1] = y.y;
#if !defined(EXP_LATEKV) && !defined(EXP_NOKV)
    if (k == 3) kv[0] = kv_next[0];                  // first K word group is dead: fetch the next block's now
#endif
#ifdef EXP_NOSHIFT
    if (k & 1) { feed.step(fs++); __builtin_amdgcn_sched_barrier(0); }
#else
    if (k & 1) { pin(tmax); feed.step(fs++); __builtin_amdgcn_sched_barrier(0); }
#endif
  }
#ifndef EXP_NOKV
#ifndef EXP_LATEKV
  kv[1] = kv_next[64];                             // second K word group (the first was refilled after key 7 of 16)
#else
  kv[0] = kv_next[0]; kv[1] = kv_next[64];          // K words are dead: fetch the next block's in place
#endif
#endif
  STAMP(7)   // loop 1: Y, running max (+ 5 feed MFMAs)
  // online softmax in the log2 domain: p = 2^(y*log2e - m*log2e)
  const float mnew = fmaxf(sm_m, tmax);
  const float mref2 = ((MASKED && mnew == -INFINITY) ? 0.f : mnew) * LOG2E;
  const float sc = __builtin_amdgcn_exp2f(fmaf(sm_m, LOG2E, -mref2));
  f32x2 psum = {0.f, 0.f}, pv = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int m = k >> 1;
    f32x2 v2;
    if (k & 1) v2 = f32x2{__builtin_bit_cast(float, vv[2 * m]), __builtin_bit_cast(float, vv[2 * m + 1])};
    else v2 = f32x2{bf16_lo(vv[2 * m]), bf16_lo(vv[2 * m + 1])};
    const f32x2 y = {e1[2 * k], e1[2 * k + 1]};
#ifdef EXP_NOSHIFT
    f32x2 p = {EXP2F(y.x), EXP2F(y.y)};
#else
    const f32x2 arg = y * LOG2E - mref2;
    f32x2 p = {EXP2F(arg.x), EXP2F(arg.y)};
#endif
    if (MASKED) {
      const int b0 = ((2 * k) & 3) + 8 * ((2 * k) >> 2), b1 = b0 + 1;
      p.x = ((vbits >> b0) & 1u) ? p.x : 0.f;
      p.y = ((vbits >> b1) & 1u) ? p.y : 0.f;
    }
    psum += p;
    pv = p * v2 + pv;
    f32x2 z = y * f2c + baf;                                     // FiLM multiplicative part (+ the deferred e_add bias)
    pin(z);
    e1[2 * k] = z.x; e1[2 * k + 1] = z.y;
#if !defined(EXP_LATEKV) && !defined(EXP_NOKV)
    if (k == 3) kv[2] = kv_next[128];
#endif
    if (k & 1) { pin(psum); pin(pv); feed.step(fs++); __builtin_amdgcn_sched_barrier(0); }
  }
#pragma unroll
  for (; fs < Gemm1Feed<DE, DX, DFF>::N; ++fs) feed.step(fs);      // (DE = 64: none left)
#ifndef EXP_NOKV
#ifndef EXP_LATEKV
  kv[3] = kv_next[192];
#else
  kv[2] = kv_next[128]; kv[3] = kv_next[192];       // V words likewise
#endif
#endif
  STAMP(8)   // loop 2: exp, sums, FiLM (+ 4 feed MFMAs)
#ifdef EXP_NOSHIFT
  if (!MASKED) { sm_l += psum.x + psum.y; sm_a += pv.x + pv.y; sm_m = 0.f; return; }
#endif
  sm_l = fmaf(sm_l, sc, psum.x + psum.y);
  sm_a = fmaf(sm_a, sc, pv.x + pv.y);
  sm_m = mnew;
}

template <int DE, int DX, int DFF, int NTHREADS, bool TAB = false>
__global__ void __launch_bounds__(NTHREADS) k_edge_layer(EdgeLayerArgs a) {
  using C = EdgeCfg<DE, DX, DFF>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* sW = reinterpret_cast<uint4*>(smem);
  {
    const int n16 = (int)(C::LDS_BLOB_BYTES / 16), cut = (C::W1_FR + C::WO_FR) * 64;
    for (int t = threadIdx.x; t < n16; t += NTHREADS) sW[t] = a.blob[(C::LDS_SKIP && t >= cut) ? t + (int)(C::LDS_SKIP / 16) : t];
  }
  __syncthreads();
  const uint4* sW1 = sW;
  const uint4* sWO = sW1 + C::W1_FR * 64;
  const uint4* sF1 = sWO + C::WO_FR * 64;
  const uint4* sF2 = sF1 + C::F1_FR * 64;
  const float* sP = reinterpret_cast<const float*>(C::LDS_SKIP ? sF1 : sF2 + C::F2_FR * 64);
  unsigned* sB1 = reinterpret_cast<unsigned*>(smem + C::LDS_BLOB_BYTES);          // [NCB][64] e_mul bias as a B-operand dword
  float* sQF = reinterpret_cast<float*>(smem + C::LDS_BLOB_BYTES + C::NCB * 256) + (threadIdx.x >> 6) * (2 * DX + DE);
  const uint4* sTab = reinterpret_cast<const uint4*>(smem + C::LDS_BLOB_BYTES + C::NCB * 256 + (NTHREADS / 64) * (2 * DX + DE) * 4);   // [tab_rows][DE / 8]
  if (TAB) {
    uint4* wt = const_cast<uint4*>(sTab);
    for (int t = threadIdx.x; t < a.tab_rows * (DE / 8); t += NTHREADS) wt[t] = a.tab[t];
  }
  // bias operand of the extra GEMM1 k-step: lanes 0-31 hold (hi, lo) bf16 split of bm1[32cb + r] at k = 0, 1; lanes 32-63 zero
  for (int t = threadIdx.x; t < C::NCB * 64; t += NTHREADS) {
    const int l = t & 63, cbb = t >> 6;
    unsigned w = 0u;
    if (l < 32) {
      const float v = sP[C::P_BM1 + 32 * cbb + l];
      const unsigned hi = pack_bf16(v, 0.f) & 0xffffu;
      const float rest = v - __builtin_bit_cast(float, hi << 16);
      w = hi | (pack_bf16(rest, 0.f) << 16);
    }
    sB1[t] = w;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  constexpr int NW = NTHREADS / 64;
  const int n = a.n;
  const int NT = (n + 31) / 32;
  const int R = a.rowmap ? (a.n_rows_dev ? *a.n_rows_dev : a.n_rows) : a.B * n;

  const bf16x8 id0 = ident_frag(0, r, h), id1 = ident_frag(1, r, h);
  bf16x8 onesA;                                         // A operand with ones in the k = 0, 1 columns
#pragma unroll
  for (int j = 0; j < 8; ++j) onesA[j] = (h == 0 && j < 2) ? (short)0x3F80 : (short)0;

  // XCD-aware persistent walk over work units (EdgePart): whole rows, then runs of key tiles of the split rows
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const EdgePart part(R, gridDim.x, NW, NT, xcd, a.split != nullptr, a.tail_old != 0);
  const int wid = bi * NW + wave;
  int kfull = 0, kextra = 0;
  // uneven static shares: the wave that was launched first on a SIMD wins the issue arbitration against its partner, so it is
  // given more rows -- `nfc` common rounds, then 2 * skew half-rounds for the older half of the waves only
  const int skew = min(a.skew, part.nfull), nfc = part.nfull - skew;
  const bool older = wave < NW / 2;
  const int oidx = bi * (NW / 2) + wave;
  const int tw = a.tail_old ? (older ? oidx : -1) : wid;   // index among the waves that take runs of the split rows
  int u0 = tw < 0 ? 0 : tw * part.tpw, u1 = tw < 0 ? 0 : min(part.rem * NT, u0 + part.tpw);

  if (wave >= NW / 2) {
    if (a.prio) __builtin_amdgcn_s_setprio(1);
    for (int t = 0; t < a.stagger; ++t) __builtin_amdgcn_s_sleep(1);
  }
#ifdef EXP_STAMP
  Stamps S;
  for (int q = 0; q < NSTAMP; ++q) S.acc[q] = 0;
  const long long rt0 = __builtin_amdgcn_s_memrealtime();
  S.prev = __builtin_readcyclecounter();
#else
  Stamps S;
#endif
  for (;;) {
    int row, jt0, jt1, slot = -1;
    if (kfull < nfc) {
      row = part.row_begin + kfull * part.W + wid; jt0 = 0; jt1 = NT; ++kfull;
      if (row >= part.row_end) continue;
    } else if (older && kextra < 2 * skew) {
      row = part.row_begin + nfc * part.W + kextra * (part.W / 2) + oidx; jt0 = 0; jt1 = NT; ++kextra;
      if (row >= part.row_end) continue;
    } else if (u0 < u1) {
      const int rl = u0 / NT;
      jt0 = u0 - rl * NT; jt1 = min(NT, jt0 + (u1 - u0));
      row = part.row_begin + part.nfull * part.W + rl; slot = u0; u0 += jt1 - jt0;
    } else {
      break;
    }
    if (a.rowmap) row = a.rowmap[row];
    const int b = row / n, i = row - b * n;
    const bool mi = a.node_mask[row] != 0;
    // online-softmax state per channel block (running max, sum, weighted sum): registers (the cb loop is unrolled)
    float sm_m[C::NCB], sm_l[C::NCB], sm_a[C::NCB];
#pragma unroll
    for (int cb = 0; cb < C::NCB; ++cb) { sm_m[cb] = -INFINITY; sm_l[cb] = 0.f; sm_a[cb] = 0.f; }
    // per-row lane constants (scaled query, FiLM scale, e_out bias) staged in this wave's LDS slice: ds_read per block
    // instead of 2*NCB live registers (keeps the 2-waves/SIMD build free of scratch spills)
    for (int c = lane; c < DX; c += 64) {
      sQF[c] = a.Q[(size_t)row * a.ldq + c] * a.qscale;
      sQF[DX + c] = a.ye2p1[(size_t)b * DX + c];
    }
    if (lane < DE) sQF[2 * DX + lane] = a.ebias[(size_t)b * DE + lane];
    const bf16_t* Erow = a.E_in + (size_t)row * n * DE;
    bf16_t* Orow = a.E_out + (size_t)row * n * DE;

    // Software-pipelined loads, no extra registers: the E tile of pair block jt+1 is fetched into eraw as soon as the
    // residual of block jt has consumed it (the FFN hides the HBM latency), the K (V) words of the next channel block
    // replace the current ones as soon as they are dead, and the key-mask byte is fetched one tile ahead.
    uint4 eraw[C::KS1];
    // TAB: this graph's input rows come from the LDS table, addressed by one class byte per pair (fetched one tile ahead like the mask)
    const bool tabrow = TAB && a.tab_flags[b] && a.tab_flags[a.B + b];     // wave-uniform
    const uint8_t* crow = TAB ? a.cls + (size_t)row * n : nullptr;
    unsigned ecls = 0;
    {
      const int j0c = min(jt0 * 32 + r, n - 1);
      if (tabrow) {
        const unsigned c0 = crow[j0c];
#pragma unroll
        for (int u = 0; u < C::KS1; ++u) eraw[u] = sTab[c0 * (DE / 8) + 2 * u + h];
        ecls = crow[min(j0c + 32, n - 1)];
      } else {
#pragma unroll
        for (int u = 0; u < C::KS1; ++u) eraw[u] = *reinterpret_cast<const uint4*>(Erow + (size_t)j0c * DE + 16 * u + 8 * h);
      }
    }
    const uint4* kvrow = a.KV + ((size_t)b * NT * C::NCB) * 4 * 64 + lane;
    uint4 kvA[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) kvA[e] = kvrow[((size_t)jt0 * C::NCB * 4 + e) * 64];
    uint8_t nmb = a.node_mask[b * n + min(jt0 * 32 + r, n - 1)];
    const unsigned ktm = a.row_tiles ? (unsigned)__builtin_amdgcn_readfirstlane((int)a.row_tiles[row]) : 0xffffffffu;
    STAMP(0)   // row setup
    for (int jt = jt0; jt < jt1; ++jt) {
      const int jl = jt * 32 + r;
      const bool jvalid = jl < n;
      const bool mj = jvalid && (nmb != 0);
      nmb = a.node_mask[b * n + min(jl + 32, n - 1)];
      // ---- E tile: this lane's pair row, 16 B per k-step, to C-order
#pragma unroll
      for (int u = 0; u < C::KS1; ++u) row16_to_corder(eraw[u]);
      const unsigned nm_all = (unsigned)__ballot(mj), v_all = (unsigned)__ballot(jvalid);
      const unsigned nmbits = nm_all >> (4 * h);       // bit p: key row p of this tile is a real node
      const unsigned vbits = v_all >> (4 * h);         // bit p: key row p exists (j < n)
      const float pm = (mi && mj) ? 1.f : 0.f;         // pair mask of this lane's column
      const bool fast = mi && (nm_all == 0xffffffffu); // wave-uniform

      f32x16 acc2[C::MT];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) acc2[mt] = splat16(0.f);

      const uint4* kvt = kvrow + (size_t)jt * C::NCB * 4 * 64;
      const uint4* kvn = (jt + 1 < NT) ? kvt + C::NCB * 4 * 64 : kvt;   // last tile: harmless reload
      // A tile without one real pair (padded query row, or 32 padded / missing keys: most of a USPTO-sized graph padded to
      // MAX_ATOMS_RXN) needs no attention work: newE is masked to zero there, so E' = LN2(FFN(LN1(E))) comes from the tile's
      // epilogue alone, and its keys enter the per-channel softmax with score 0 and V = 0 (layers.py:254: no key mask), i.e. as
      // count * exp(0 - m) in the denominator: a closed form per channel block instead of GEMM1 / exp / GEMM2 for 32 x 256 values.
      const bool dead = a.skip_dead && (!mi || nm_all == 0u);          // wave-uniform
      if (dead) {
        const float cnt = (float)__popc(vbits & 0x0f0f0f0fu);          // valid key rows of this lane half
        if (cnt > 0.f) {
          constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
          for (int cb = 0; cb < C::NCB; ++cb) {
            const float yv = -sP[C::P_BA + 32 * cb + r];               // score 0 in the kernel's shifted domain (y' = y - b_add)
            const float mnew = fmaxf(sm_m[cb], yv), mref2 = mnew * LOG2E;
            const float sc = __builtin_amdgcn_exp2f(fmaf(sm_m[cb], LOG2E, -mref2));
            const float pz = __builtin_amdgcn_exp2f(fmaf(yv, LOG2E, -mref2));
            sm_l[cb] = fmaf(sm_l[cb], sc, cnt * pz);
            sm_a[cb] *= sc;
            sm_m[cb] = mnew;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) kvA[e] = kvn[e * 64];               // the K/V pipeline expects block 0 of the next tile
      } else {
      // Channel blocks, software pipelined: GEMM1 of block cb+1 is issued beside the elementwise work of block cb, so the
      // wave's own MFMAs and VALU overlap (the two waves of a SIMD do not hide each other's dependency stalls).
      f32x16 e1, e2;
      gemm1_block<DE, DX, DFF>(0, lane, eraw, sW1, sB1, onesA, e1, e2);
      STAMP(1)   // tile head: E tile to C-order, masks, GEMM1 of block 0
      if (fast) {
#pragma unroll
        for (int cb = 0; cb < C::NCB; ++cb) {
          f32x16 f1, f2;
          Gemm1Feed<DE, DX, DFF> feed(cb + 1 < C::NCB, cb + 1, lane, eraw, sW1, sB1, onesA, f1, f2);
          softmax_film_block<DE, DX, DFF, false>(cb, lane, e1, e2, sP, kvA, (cb + 1 < C::NCB) ? kvt + (cb + 1) * 4 * 64 : kvn, sQF[32 * cb + r], sQF[DX + 32 * cb + r], mi,
                                                nmbits, vbits, sm_m[cb], sm_l[cb], sm_a[cb], feed, S);
          gemm2_block<DE, DX, DFF>(cb, lane, e1, sWO, id0, id1, acc2);
          STAMP(9)   // transpose + GEMM2
          if (cb + 1 < C::NCB) { e1 = f1; e2 = f2; }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int cb = 0; cb < C::NCB; ++cb) {
          f32x16 f1, f2;
          Gemm1Feed<DE, DX, DFF> feed(cb + 1 < C::NCB, cb + 1, lane, eraw, sW1, sB1, onesA, f1, f2);
          softmax_film_block<DE, DX, DFF, true>(cb, lane, e1, e2, sP, kvA, (cb + 1 < C::NCB) ? kvt + (cb + 1) * 4 * 64 : kvn, sQF[32 * cb + r], sQF[DX + 32 * cb + r], mi,
                                                nmbits, vbits, sm_m[cb], sm_l[cb], sm_a[cb], feed, S);
          gemm2_block<DE, DX, DFF>(cb, lane, e1, sWO, id0, id1, acc2);
          STAMP(9)   // transpose + GEMM2
          if (cb + 1 < C::NCB) { e1 = f1; e2 = f2; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      }   // !dead
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2)   // channel-block loop
      // ---- epilogue of the tile: newE^T + bias, pair mask, residual, LN1
      float sum = 0.f;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bias = *reinterpret_cast<const float4*>(sQF + 2 * DX + 32 * mt + 8 * g + 4 * h);
          const uint4& er = eraw[2 * mt + (g >> 1)];
          const unsigned d0 = (g & 1) ? er.z : er.x, d1 = (g & 1) ? er.w : er.y;
          const float ev[4] = {bf16_lo(d0), bf16_hi(d0), bf16_lo(d1), bf16_hi(d1)};
          const float bb[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = ev[e] + pm * (acc2[mt][4 * g + e] + bb[e]);
            acc2[mt][4 * g + e] = x;
            sum += x;
          }
        }
      }
      {
        const int jn = min(jl + 32, n - 1);             // next tile of this row (clamped; unused after the last tile)
#ifndef EXP_NOELOAD
        if (tabrow) {
#pragma unroll
          for (int u = 0; u < C::KS1; ++u) eraw[u] = sTab[ecls * (DE / 8) + 2 * u + h];
          ecls = crow[min(jl + 64, n - 1)];             // class byte of the tile after the next
        } else {
#pragma unroll
          for (int u = 0; u < C::KS1; ++u) eraw[u] = *reinterpret_cast<const uint4*>(Erow + (size_t)jn * DE + 16 * u + 8 * h);
        }
#endif
      }
#ifdef EXP_NOEPI
      // timing experiment (split-epilogue proposal): kernel 1 would end here, storing E + newE as bf16; LN1 / FFN / LN2 move to a second kernel
#pragma unroll
      for (int u = 0; u < C::KS1; ++u) {
        const int mt = u >> 1;
        unsigned d[4];
#pragma unroll
        for (int gg2 = 0; gg2 < 2; ++gg2) {
          const int g = 2 * (u & 1) + gg2;
          d[2 * gg2 + 0] = pack_bf16(acc2[mt][4 * g + 0], acc2[mt][4 * g + 1]);
          d[2 * gg2 + 1] = pack_bf16(acc2[mt][4 * g + 2], acc2[mt][4 * g + 3]);
        }
        uint4 st = {d[0], d[1], d[2], d[3]};
        row16_to_corder(st);
        if (jvalid) *reinterpret_cast<uint4*>(Orow + (size_t)jl * DE + 16 * u + 8 * h) = st;
      }
      if (sum == 123.456f) Orow[0] = 0;
#else
      if (!((ktm >> jt) & 1u)) continue;                 // nobody reads E_out[row][this tile] (the next tile's rows are on their way)
      const float mean1 = xhalf_sum(sum) * (1.f / DE);
      float vs = 0.f;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int i2 = 0; i2 < 16; ++i2) { const float d = acc2[mt][i2] - mean1; vs = fmaf(d, d, vs); }
      const float rstd1 = rsqrtf(xhalf_sum(vs) * (1.f / DE) + 1e-5f);
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 gg = *reinterpret_cast<const float4*>(sP + C::P_LN1G + 32 * mt + 8 * g + 4 * h);
          const float4 be = *reinterpret_cast<const float4*>(sP + C::P_LN1B + 32 * mt + 8 * g + 4 * h);
          const float g4[4] = {gg.x, gg.y, gg.z, gg.w}, b4[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2[mt][4 * g + e] = fmaf((acc2[mt][4 * g + e] - mean1) * rstd1, g4[e], b4[e]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      STAMP(3)   // residual + LN1 (+ next E tile load issue)
      // ---- FFN chained through accumulators:  H^T = relu(W1 Em^T + b1),  O^T = W2 H^T + b2
      bf16x8 emB[C::KS1];
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) emB[s] = pack_acc(acc2[s >> 1], s & 1);
      bf16x8 hB[C::KSF];
#pragma unroll
      for (int ht = 0; ht < C::HT; ++ht) {
        f32x16 hacc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(sP + C::P_BF1 + 32 * ht + 8 * g + 4 * h);
          hacc[4 * g + 0] = bv.x; hacc[4 * g + 1] = bv.y; hacc[4 * g + 2] = bv.z; hacc[4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int s = 0; s < C::KS1; ++s) hacc = mfma32(LDSF(sF1, ht * C::KS1 + s, lane), emB[s], hacc);
        // relu on the packed bf16 words: as 16-bit integers, negative floats (and -0) are negative -> v_pk_max_i16 with 0
        hB[2 * ht + 0] = __builtin_elementwise_max(pack_acc(hacc, 0), (bf16x8)(short)0);
        hB[2 * ht + 1] = __builtin_elementwise_max(pack_acc(hacc, 1), (bf16x8)(short)0);
      }
      __builtin_amdgcn_sched_barrier(0);
      float sum2 = 0.f;
      f32x16 o[C::MT];
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(sP + C::P_BF2 + 32 * mt + 8 * g + 4 * h);
          o[mt][4 * g + 0] = bv.x; o[mt][4 * g + 1] = bv.y; o[mt][4 * g + 2] = bv.z; o[mt][4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int s = 0; s < C::KSF; ++s) o[mt] = mfma32(LDSF(sF2, mt * C::KSF + s, lane), hB[s], o[mt]);
#pragma unroll
        for (int i2 = 0; i2 < 16; ++i2) { o[mt][i2] += acc2[mt][i2]; sum2 += o[mt][i2]; }
      }
      STAMP(4)   // FFN
      const float mean2 = xhalf_sum(sum2) * (1.f / DE);
      float vs2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int i2 = 0; i2 < 16; ++i2) { const float d = o[mt][i2] - mean2; vs2 = fmaf(d, d, vs2); }
      const float rstd2 = rsqrtf(xhalf_sum(vs2) * (1.f / DE) + 1e-5f);
      // ---- LN2, pack to bf16 (C-order), back to row order, 16-byte stores
#pragma unroll
      for (int u = 0; u < C::KS1; ++u) {
        const int mt = u >> 1;
        unsigned d[4];
#pragma unroll
        for (int gg2 = 0; gg2 < 2; ++gg2) {
          const int g = 2 * (u & 1) + gg2;
          const float4 gv = *reinterpret_cast<const float4*>(sP + C::P_LN2G + 32 * mt + 8 * g + 4 * h);
          const float4 bv = *reinterpret_cast<const float4*>(sP + C::P_LN2B + 32 * mt + 8 * g + 4 * h);
          const float y0 = fmaf((o[mt][4 * g + 0] - mean2) * rstd2, gv.x, bv.x);
          const float y1 = fmaf((o[mt][4 * g + 1] - mean2) * rstd2, gv.y, bv.y);
          const float y2 = fmaf((o[mt][4 * g + 2] - mean2) * rstd2, gv.z, bv.z);
          const float y3 = fmaf((o[mt][4 * g + 3] - mean2) * rstd2, gv.w, bv.w);
          d[2 * gg2 + 0] = pack_bf16(y0, y1);
          d[2 * gg2 + 1] = pack_bf16(y2, y3);
        }
        uint4 st = {d[0], d[1], d[2], d[3]};
        row16_to_corder(st);   // involution: C-order -> natural 16 bytes at [16u + 8h]
#ifdef EXP_NOSTORE
        if (st.x == 0x12345678u && jvalid) *reinterpret_cast<uint4*>(Orow + (size_t)jl * DE + 16 * u + 8 * h) = st;
#else
        if (jvalid) *reinterpret_cast<uint4*>(Orow + (size_t)jl * DE + 16 * u + 8 * h) = st;
#endif
      }
#endif
      STAMP(5)   // LN2 + store
    }
    // ---- end of the unit: merge the two lane halves' softmax states; a whole row writes the x_out input, a run of a
    // split row leaves (max, sum, weighted sum) for k_edge_split_merge
#pragma unroll
    for (int cb = 0; cb < C::NCB; ++cb) {
      const float m_ = sm_m[cb], l_ = sm_l[cb], a_ = sm_a[cb];
      const float mo = __shfl_xor(m_, 32), lo = __shfl_xor(l_, 32), ao = __shfl_xor(a_, 32);
      const float M = fmaxf(m_, mo);
      const float s0 = __expf(m_ - M), s1 = __expf(mo - M);
      const float l = l_ * s0 + lo * s1, av = a_ * s0 + ao * s1;
      if (h == 0) {
        const int c = 32 * cb + r;
        if (slot < 0) {
          a.newX_in[(size_t)row * DX + c] = fmaf(a.yx2p1[(size_t)b * DX + c], av / l, a.yx1[(size_t)b * DX + c]);
        } else {
          float* o = a.split + ((size_t)(xcd * a.slots_per_xcd + slot) * 3) * DX + c;
          o[0] = M; o[DX] = l; o[2 * DX] = av;
        }
      }
    }
    STAMP(6)     // row end
  }
  // ---- tail: pooled statistics of the INPUT tensor (Etoy of this layer, layers.py:31-47; k_e_stats' partials).  With static
  // shares the older wave of every SIMD pair is done after ~80 % of the kernel's time; instead of leaving the SIMD to its
  // partner alone it streams E_in once more, one (graph, chunk) ticket at a time, until the tickets run out -- the
  // separate 737 MB statistics pass disappears behind the partner waves' latency-bound tail.
  if (a.stats != nullptr) {
    constexpr int G = DE / 8, RL = 64 / G, U = 8;      // lanes per pair row, rows per wave instruction, loads in flight
    const size_t Np = (size_t)n * n;
    const int nchunk = a.stats_nchunk, ntick = a.B * nchunk;
    const size_t per = (Np + nchunk - 1) / nchunk;
    const int g = lane % G, rl = lane / G;
    for (;;) {
      int t = 0;
      if (lane == 0) {
        t = atomicAdd(a.ticket, 1);
        // every wave of the grid draws exactly one ticket >= ntick: whoever draws the last of them puts the counter back to zero
        if (t == ntick + (int)gridDim.x * NW - 1) atomicExch(a.ticket, 0);
      }
      t = __shfl(t, 0);
      if (t >= ntick) break;
      const int b = t / nchunk, ch = t - b * nchunk;
      const size_t p0 = ch * per, p1 = min(Np, p0 + per);
      float s[8], ss[8], mn[8], mx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; mn[e] = INFINITY; mx[e] = -INFINITY; }
      auto fold = [&](const uint4& u) {
        const unsigned d[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (e & 1) ? bf16_hi(d[e >> 1]) : bf16_lo(d[e >> 1]);
          s[e] += v; ss[e] = fmaf(v, v, ss[e]); mn[e] = fminf(mn[e], v); mx[e] = fmaxf(mx[e], v);
        }
      };
      const bf16_t* base = a.E_in + (size_t)b * Np * DE + 8 * g;
      size_t p = p0 + rl;
      for (; p + (U - 1) * RL < p1; p += U * RL) {
        uint4 u[U];
#pragma unroll
        for (int k = 0; k < U; ++k) u[k] = *reinterpret_cast<const uint4*>(base + (p + k * RL) * DE);
#pragma unroll
        for (int k = 0; k < U; ++k) fold(u[k]);
      }
      for (; p < p1; p += RL) fold(*reinterpret_cast<const uint4*>(base + p * DE));
#pragma unroll
      for (int off = G; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s[e] += __shfl_xor(s[e], off); ss[e] += __shfl_xor(ss[e], off);
          mn[e] = fminf(mn[e], __shfl_xor(mn[e], off)); mx[e] = fmaxf(mx[e], __shfl_xor(mx[e], off));
        }
      }
      if (lane < G) {
        float* o = a.stats + ((size_t)(b * nchunk + ch) * 4) * DE + 8 * lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = s[e]; o[DE + e] = ss[e]; o[2 * DE + e] = mn[e]; o[3 * DE + e] = mx[e]; }
      }
    }
  }
#ifdef EXP_STAMP
  if (lane == 0) {
    const int gw = blockIdx.x * NW + wave;
    S.acc[11] = __builtin_amdgcn_s_memrealtime() - rt0;
    for (int q = 0; q < NSTAMP; ++q) g_stamps[gw * NSTAMP + q] = S.acc[q];
  }
#endif
}


// folds the runs of every split row (EdgePart) into the x_out input: newX_in = yx1 + yx2p1 * (sum_r a_r e^{m_r-M}) / (sum_r l_r e^{m_r-M})
__global__ void __launch_bounds__(256) k_edge_split_merge(const float* split, int slots_per_xcd, const float* yx1, const float* yx2p1,
                                                          float* newX_in, int R, int n, int DX, int grid_blocks, int NW, int max_rem,
                                                          int tail_old, const int* rowmap, const int* n_rows_dev) {
  if (n_rows_dev) R = *n_rows_dev;
  const int NT = (n + 31) / 32;
  const int xcd = blockIdx.x / max_rem, rl = blockIdx.x % max_rem;
  const EdgePart part(R, grid_blocks, NW, NT, xcd, true, tail_old != 0);
  if (rl >= part.rem) return;
  int row = part.row_begin + part.nfull * part.W + rl;
  if (rowmap) row = rowmap[row];
  const int b = row / n;
  for (int c = threadIdx.x; c < DX; c += blockDim.x) {
    float M = -INFINITY, L = 0.f, A = 0.f;
    for (int u = rl * NT, end = (rl + 1) * NT; u < end;) {
      const float* p = split + ((size_t)(xcd * slots_per_xcd + u) * 3) * DX + c;
      const float m = p[0], l = p[DX], av = p[2 * DX];
      const float Mn = fmaxf(M, m);
      const float s0 = __expf(M - Mn), s1 = __expf(m - Mn);
      L = L * s0 + l * s1; A = A * s0 + av * s1; M = Mn;
      const int seg_end = (u / part.tpw + 1) * part.tpw;
      u = seg_end < end ? seg_end : end;
    }
    newX_in[(size_t)row * DX + c] = fmaf(yx2p1[(size_t)b * DX + c], A / L, yx1[(size_t)b * DX + c]);
  }
}

// K,V (fp32, natural (B,n,DX), already masked) -> bf16 fragment layout
//   [b][jt][cb][kv][half2][lane][8]: element e of (half2,lane=(r,h)) is row 32jt + crow(8*half2+e, h), channel 32cb + r
//   (the 8 elements are stored in the word order described in the kernel body)
__global__ void k_kv_pack(const float* K, const float* V, int ldkv, uint4* out, int B, int n, int DX) {
  const int NT = (n + 31) / 32, NCB = DX / 32;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t tot = (size_t)B * NT * NCB * 4 * 64;
  if (t >= tot) return;
  const int lane = t & 63; const int half2 = (t >> 6) & 1; const int kv = (t >> 7) & 1;
  size_t rest = t >> 8;
  const int cb = rest % NCB; rest /= NCB;
  const int jt = rest % NT; const int b = rest / NT;
  const int r = lane & 31, h = lane >> 5;
  const float* src = kv ? V : K;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int j = 32 * jt + crow(8 * half2 + e, h);
    v[e] = j < n ? src[((size_t)b * n + j) * ldkv + 32 * cb + r] : 0.f;
  }
  const uint4 o = kv_words(v);   // word order and rounding: mfma_util.h
  out[t] = o;
}

// ===================================================================================================
template <int DE, int DX, int DFF, int NT_, bool TAB = false>
static int launch_edge_nt(const EdgeLayerArgs& a, int grid_blocks, hipStream_t s) {
  using C = EdgeCfg<DE, DX, DFF>;
  auto kern = k_edge_layer<DE, DX, DFF, NT_, TAB>;
  constexpr size_t TAB_BYTES = TAB ? (size_t)18 * DE * 2 : 0;       // up to 18 table rows (de_in <= 16)
  static std::atomic<bool> attr_set[DA_MAX_DEVICES];     // the attribute is per device (and per template instance)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DA_MAX_DEVICES) { da_set_error("edge layer: no current HIP device"); return DA_ERR_HIP; }
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C::LDS_BLOB_BYTES + C::NCB * 256 + (NT_ / 64) * (2 * DX + DE) * 4 + TAB_BYTES));
    if (e != hipSuccess) { da_set_error("edge layer: cannot reserve %zu B of LDS: %s", C::BLOB_BYTES, hipGetErrorString(e)); return DA_ERR_HIP; }
    attr_set[dev].store(true, std::memory_order_release);
  }
  kern<<<grid_blocks, NT_, C::LDS_BLOB_BYTES + C::NCB * 256 + (NT_ / 64) * (2 * DX + DE) * 4 + TAB_BYTES, s>>>(a);
  return DA_OK;
}
// Tuning knobs of the launch (-1 = choose automatically).  Read from the environment once (DA_EDGE_THREADS, DA_EDGE_SKEW,
// DA_EDGE_TAIL_OLD, DA_EDGE_STAGGER, DA_EDGE_PRIO) and settable through da_edge_layer_tuning (tests, A/B timing).  Process-wide:
// change them only while no other thread is inside da_edge_layer.
struct EdgeTuning { int threads, skew, tail_old, stagger, prio, skip_dead; };
static EdgeTuning& edge_tuning() {
  static EdgeTuning t = [] {
    auto env = [](const char* k, int dflt) { const char* e = getenv(k); return e ? atoi(e) : dflt; };
    EdgeTuning u;
    u.threads = env("DA_EDGE_THREADS", 512) == 256 ? 256 : 512;   // default: 8 waves/workgroup = 2 waves per SIMD
    u.skew = env("DA_EDGE_SKEW", -1); u.tail_old = env("DA_EDGE_TAIL_OLD", -1);
    u.stagger = env("DA_EDGE_STAGGER", 0); u.prio = env("DA_EDGE_PRIO", 0);
    u.skip_dead = env("DA_EDGE_SKIP_DEAD", 1) != 0;
    return u;
  }();
  return t;
}
#ifdef EXP_NT768
static int edge_threads() { return 768; }
#else
static int edge_threads() { return edge_tuning().threads; }
#endif
template <int DE, int DX, int DFF>
static int launch_edge(const EdgeLayerArgs& a, int grid_blocks, hipStream_t s) {
  if (a.cls != nullptr) {
    if (edge_threads() == 512) return launch_edge_nt<DE, DX, DFF, 512, true>(a, grid_blocks, s);
    return launch_edge_nt<DE, DX, DFF, 256, true>(a, grid_blocks, s);
  }
#ifdef EXP_NT768
  return launch_edge_nt<DE, DX, DFF, 768>(a, grid_blocks, s);   // timing experiment: three waves per SIMD (needs EXP_NOEPI for the LDS)
#endif
  if (edge_threads() == 512) return launch_edge_nt<DE, DX, DFF, 512>(a, grid_blocks, s);
  return launch_edge_nt<DE, DX, DFF, 256>(a, grid_blocks, s);
}
static int edge_grid(int rows) {
  int grid = 256;                                   // one persistent workgroup per CU
#ifdef EXP_GRID
  { const char* e = getenv("DA_EDGE_GRID"); if (e) grid = atoi(e); }   // timing experiment: fewer CUs (clock / power probe)
#endif
  int need = ((rows + 3) / 4 + 7) / 8 * 8;          // enough workgroups for one row per wave
  if (need < grid) grid = need < 8 ? 8 : need;
  return grid;
}
// largest number of split rows of any XCD chunk (0: the rows divide evenly, nothing to split)
static int edge_max_rem(int rows, int n) {
  const int NT = (n + 31) / 32, grid = edge_grid(rows), NW = edge_threads() / 64;
  int mx = 0;
  for (int x = 0; x < 8; ++x) { const EdgePart p(rows, grid, NW, NT, x, true); if (p.rem > mx) mx = p.rem; }
  return mx;
}

extern "C" {

// Launch tuning of da_edge_layer (see EdgeTuning): threads 256 | 512 (0 = keep), skew / tail_old -1 = automatic, stagger, prio,
// skip_dead (1: tiles without a real pair take the closed form, 0: every tile runs the attention part).
// da_edge_split_scratch_bytes depends on `threads`: size the scratch after this call.
int da_edge_layer_tuning(int threads, int skew, int tail_old, int stagger, int prio, int skip_dead) {
  DA_CHECK_ARG(threads == 0 || threads == 256 || threads == 512, "da_edge_layer_tuning: threads must be 0 (keep), 256 or 512");
  DA_CHECK_ARG(skew >= -1 && tail_old >= -1 && tail_old <= 1 && stagger >= 0 && prio >= 0 && prio <= 3, "da_edge_layer_tuning: bad value");
  EdgeTuning& t = edge_tuning();
  if (threads) t.threads = threads;
  t.skew = skew; t.tail_old = tail_old; t.stagger = stagger; t.prio = prio; t.skip_dead = skip_dead != 0;
  return DA_OK;
}

// size in bytes of the packed per-layer weight blob the edge kernel expects (layout: EdgeCfg / diffalign_amd/pack.py)
long long da_edge_blob_bytes(int de, int dx, int dff) {
  if (de == 64 && dx == 256 && dff == 128) return (long long)EdgeCfg<64, 256, 128>::BLOB_BYTES;
  if (de == 32 && dx == 64 && dff == 64) return (long long)EdgeCfg<32, 64, 64>::BLOB_BYTES;
  return -1;
}
long long da_kv_frag_bytes(int B, int n, int dx) { return (long long)B * ((n + 31) / 32) * (dx / 32) * 4 * 64 * 16; }

int da_kv_pack(const float* K, const float* V, int ldkv, void* out, int B, int n, int dx, void* stream) {
  DA_CHECK_ARG(dx % 32 == 0 && ldkv >= dx, "da_kv_pack: dx must be a multiple of 32");
  if (B == 0 || n == 0) return DA_OK;
  size_t tot = (size_t)B * ((n + 31) / 32) * (dx / 32) * 4 * 64;
  k_kv_pack<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(K, V, ldkv, (uint4*)out, B, n, dx);
  DA_CHECK_LAUNCH("da_kv_pack");
  return DA_OK;
}

// Fused edge side of one XEyTransformerLayer (see file header).  All pointers device; E bf16.
// bytes of the tail-split scratch da_edge_layer can use for this problem size (0: no split needed)
long long da_edge_split_scratch_bytes_rows(long long rows, int n, int dx) {
  if (rows == 0 || n <= 0 || rows > 0x7fffffffLL || rows < -0x7fffffffLL) return 0;
  const int NT = (n + 31) / 32;
  if (rows < 0)          // a row list whose length is read on the device (estimate -rows): room for any number of split rows
    return (long long)8 * ((edge_grid((int)-rows) >> 3) * (edge_threads() / 64) - 1) * NT * 3 * dx * (long long)sizeof(float);
  return (long long)8 * edge_max_rem((int)rows, n) * NT * 3 * dx * (long long)sizeof(float);
}
long long da_edge_split_scratch_bytes(int B, int n, int dx) { return B <= 0 ? 0 : da_edge_split_scratch_bytes_rows((long long)B * n, n, dx); }

static int edge_layer_impl(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                           const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                           const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                           long long split_scratch_bytes, float* stats, int stats_nchunk, int* ticket, void* stream,
                           const uint8_t* cls = nullptr, const uint8_t* in_scratch = nullptr, int de_in = 0,
                           const int* rowmap = nullptr, int n_rows = 0, const int* n_rows_dev = nullptr,
                           const unsigned* row_tiles = nullptr) {
  DA_CHECK_ARG(B >= 0 && n >= 0, "da_edge_layer: negative size");
  DA_CHECK_ARG(E_in != E_out, "da_edge_layer: in-place not supported");
  DA_CHECK_ARG(!rowmap || (n_rows >= 0 && (long long)n_rows <= (long long)B * n), "da_edge_layer_ex: n_rows = %d outside 0 .. B n", n_rows);
  if (B == 0 || n == 0 || (rowmap && n_rows == 0)) return DA_OK;
  const int rows = rowmap ? n_rows : B * n, NT = (n + 31) / 32;
  const int grid = edge_grid(rows);
  // (list length on the device: `rows` is the caller's estimate -- it fixes the grid and the shares --, and the split rows are sized for
  // any length: fewer than the waves of an XCD chunk)
  const int max_rem = (rowmap && n_rows_dev) ? (grid >> 3) * (edge_threads() / 64) - 1 : edge_max_rem(rows, n);
  float* split = nullptr;
  if (split_scratch && max_rem > 0) {
    DA_CHECK_ARG(split_scratch_bytes >= (long long)8 * max_rem * NT * 3 * dx * (long long)sizeof(float),
                 "da_edge_layer: split scratch of %lld B is smaller than da_edge_split_scratch_bytes", split_scratch_bytes);
    split = (float*)split_scratch;
  }
  EdgeLayerArgs a{(const bf16_t*)E_in, (bf16_t*)E_out, Q, ldq, (const uint4*)KV_frag, node_mask, ye2p1, ebias, yx1, yx2p1,
                  newX_in, (const uint4*)blob, qscale, B, n, split, max_rem * NT, 0, 0, 0, 0, 1, stats, stats_nchunk, ticket,
                  cls, cls ? reinterpret_cast<const uint4*>(in_scratch + in_table_offset(B)) : nullptr, in_scratch, de_in + 2,
                  rowmap, n_rows, n <= 1024 ? row_tiles : nullptr, rowmap ? n_rows_dev : nullptr};
  { const EdgeTuning tune = edge_tuning();
    a.stagger = tune.stagger; a.prio = tune.prio; a.skip_dead = tune.skip_dead;
    // default skew: ~60 % of a SIMD pair's rows to the older wave (measured optimum at 9 rows per wave: 11 / 7, -3 % per launch)
    // default shares: ~57 % of a SIMD pair's tiles to the older wave (the per-slot run times of the EXP_STAMP build cross there:
    // older 2067 + 209 s, younger 2595 - 162 s k-cycles for s rows of skew at 9 rows per wave), from whole rows of skew plus,
    // optionally, all runs of the split rows
    if (edge_threads() == 512) {
      const EdgePart p0(rows, grid, 8, NT, 0, split != nullptr);
      const float tail_rows = p0.W > 0 ? (float)p0.rem / p0.W : 0.f, pair = 2.f * (p0.nfull + tail_rows);
      float best = 1e9f;
      for (int to = 0; to <= (split ? 1 : 0); ++to)
        for (int sk = 0; sk <= p0.nfull; ++sk) {
          const float d = fabsf((p0.nfull + sk + (to ? 2.f : 1.f) * tail_rows) / pair - 0.575f);
          if (d < best) { best = d; a.skew = sk; a.tail_old = to; }
        }
    }
    if (tune.skew >= 0) a.skew = tune.skew;
    if (tune.tail_old >= 0) a.tail_old = tune.tail_old; }
  int rc;
  if (de == 64 && dx == 256 && dff == 128) rc = launch_edge<64, 256, 128>(a, grid, (hipStream_t)stream);
  else if (de == 32 && dx == 64 && dff == 64) rc = launch_edge<32, 64, 64>(a, grid, (hipStream_t)stream);
  else { da_set_error("da_edge_layer: dims (de=%d, dx=%d, dffE=%d) not built; built: (64,256,128), (32,64,64)", de, dx, dff); return DA_ERR_UNSUPPORTED; }
  if (rc) return rc;
  DA_CHECK_LAUNCH("da_edge_layer");
  if (split) {
    k_edge_split_merge<<<8 * max_rem, 256, 0, (hipStream_t)stream>>>(split, max_rem * NT, yx1, yx2p1, newX_in, rows, n, dx, grid,
                                                                    edge_threads() / 64, max_rem, a.tail_old, rowmap, rowmap ? n_rows_dev : nullptr);
    DA_CHECK_LAUNCH("da_edge_layer (split merge)");
  }
  return DA_OK;
}

int da_edge_layer(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                  const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                  const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                  long long split_scratch_bytes, void* stream) {
  return edge_layer_impl(E_in, E_out, Q, ldq, KV_frag, node_mask, ye2p1, ebias, yx1, yx2p1, newX_in, blob, qscale, B, n, de, dx, dff,
                         split_scratch, split_scratch_bytes, nullptr, 0, nullptr, stream);
}

// da_edge_layer that also leaves the Etoy statistics partials of E_in (da_edge_stats' layout, stats_nchunk chunks per graph)
// in stats_partials: the waves that finish their share of the layer early compute them (see the kernel's tail).
// ticket: one int of device memory that is ZERO when the kernel starts; the kernel's last wave puts it back to zero, so a
// caller clears it once, when it allocates it, and may hand the same word to every later launch on the same stream.
int da_edge_layer_stats(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                        const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                        const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                        long long split_scratch_bytes, float* stats_partials, int stats_nchunk, int* ticket, void* stream) {
  DA_CHECK_ARG(stats_partials != nullptr && ticket != nullptr && stats_nchunk >= 1, "da_edge_layer_stats: statistics buffers are null");
  return edge_layer_impl(E_in, E_out, Q, ldq, KV_frag, node_mask, ye2p1, ebias, yx1, yx2p1, newX_in, blob, qscale, B, n, de, dx, dff,
                         split_scratch, split_scratch_bytes, stats_partials, stats_nchunk, ticket, stream);
}

// da_edge_layer for the FIRST layer behind da_edge_in_classes: the hidden input rows of the graphs that took the table path come
// from the table in `in_scratch` through cls_map (one byte per pair) instead of E_in -- 737 MB of the benchmark batch that the input
// stage no longer writes and this layer no longer reads; the other graphs (flags cleared) read E_in as usual.  Same bits as
// da_edge_layer on the hidden tensor da_edge_in_sym would have written.
int da_edge_layer_table(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                        const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                        const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                        long long split_scratch_bytes, const uint8_t* cls_map, const uint8_t* in_scratch, int de_in, void* stream) {
  DA_CHECK_ARG(cls_map != nullptr && in_scratch != nullptr && de_in >= 1 && de_in <= 8, "da_edge_layer_table: class map / scratch / width");
  return edge_layer_impl(E_in, E_out, Q, ldq, KV_frag, node_mask, ye2p1, ebias, yx1, yx2p1, newX_in, blob, qscale, B, n, de, dx, dff,
                         split_scratch, split_scratch_bytes, nullptr, 0, nullptr, stream, cls_map, in_scratch, de_in);
}

// The general form: the three entry points above with every optional part as an argument (NULL / 0: absent), and
//   rowmap[0 .. n_rows): the query rows b n + i to compute, ascending (NULL: all).  The rows left out get neither their row of E_out nor
//   of newX_in (both stay as they were); the split scratch is sized by da_edge_split_scratch_bytes_rows(n_rows, n, dx).
//   n_rows_dev (optional): the list's length as an int in device memory, read by the kernels -- n_rows is then the caller's estimate
//   (grid and shares are chosen for it) and the scratch is sized by da_edge_split_scratch_bytes_rows(-n_rows, n, dx) (any length).
//   row_tiles (optional, B n words, n <= 1024; ignored beyond): bit jt of word b n + i set = E_out[b][i][32 jt .. 32 jt + 31] is wanted.
//   Tiles whose bit is clear take part in the row's attention (newX_in is complete) but their E_out entries stay as they were.
int da_edge_layer_ex(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                     const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                     const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                     long long split_scratch_bytes, float* stats_partials, int stats_nchunk, int* ticket, const uint8_t* cls_map,
                     const uint8_t* in_scratch, int de_in, const int* rowmap, int n_rows, const int* n_rows_dev, const unsigned* row_tiles,
                     void* stream) {
  DA_CHECK_ARG((stats_partials == nullptr) == (ticket == nullptr) && (stats_partials == nullptr || stats_nchunk >= 1),
               "da_edge_layer_ex: statistics partials and ticket go together");
  DA_CHECK_ARG((cls_map == nullptr) == (in_scratch == nullptr) && (cls_map == nullptr || (de_in >= 1 && de_in <= 8)),
               "da_edge_layer_ex: class map and input scratch go together (1 <= de_in <= 8)");
  return edge_layer_impl(E_in, E_out, Q, ldq, KV_frag, node_mask, ye2p1, ebias, yx1, yx2p1, newX_in, blob, qscale, B, n, de, dx, dff,
                         split_scratch, split_scratch_bytes, stats_partials, stats_nchunk, ticket, stream, cls_map, in_scratch, de_in,
                         rowmap, n_rows, n_rows_dev, row_tiles);
}

}  // extern "C"
