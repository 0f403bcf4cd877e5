// Batched partial symmetric eigensolver for the extra features: one workgroup per graph.
//
// The reference takes a full fp64 torch.linalg.eigh of every (n,n) Laplacian at every step and then uses only: the number
// of eigenvalues below 1e-5 * n_real, the 5 eigenvalues after those, eigenvector 0 and the 2 eigenvectors after the
// (near-)zero ones (neuralnet/extra_features.py:117-128, :158-204).  The library routine (rocSOLVER syevd, batched) needs
// 11.7 ms for 64 matrices of order 300.  Here, per graph and in ONE launch:
//   1. Householder tridiagonalisation in place (fp64, reflectors kept in the rows they annihilate), skipping rows that
//      are already zero (isolated nodes, padding: most of a noised reaction graph);
//   2. Sturm counts / bisection on the tridiagonal matrix for exactly the eigenvalues that are used;
//   3. inverse iteration for the three eigenvectors, transformed back through the reflectors;
//   4. the features themselves (the same code as k_extra_eigen in k_extra.hip).
// Inside a degenerate eigenspace (the zero eigenvalues of a graph with several components; symmetric molecules) the
// vectors are one basis among many, as the reference's LAPACK vectors are: eigenvalue features are comparable, the
// eigenvector features only through basis-independent properties (tests/golden/common.py:check_extra).
#include <atomic>
#include <type_traits>
#include <cstdlib>
#include "common.h"

// LDS of k_extra_eig_solve in doubles, without the triangle: u, p, d, e, h (5n), eigenvectors (3n), solve work (9n), reductions (64),
// eigenvalues (8), partial sums (1024), and the int area (index list n, block starts n + 1, counts 64, misc 320)
__host__ __device__ inline int eig_fixed_doubles(int n) { return 17 * n + 64 + 8 + 1024 + (2 * n + 1 + 64 + 320 + 1) / 2; }
constexpr int EIG_MAX_THREADS = 1024;
constexpr int EIG_NP = 4;            // parts of a Sturm count (cut at exact zeros of the sub-diagonal)
constexpr int EIG_PTS = 10;          // multisection points per eigenvalue and round
#ifndef EXP_EIG_ROUNDS
#define EXP_EIG_ROUNDS 16
#endif
constexpr int EIG_ROUNDS = EXP_EIG_ROUNDS;       // 11^16 = 4.6e16: the bracket shrinks to rounding level

// wave-wide sum, the same value on every lane: four DPP steps inside each row of 16 lanes (quad swaps, half-row and row mirrors:
// a few cycles each, where a ds_bpermute round trip per butterfly step costs ~100), then the four row sums in a fixed order
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int2 w = __builtin_bit_cast(int2, v);
  int2 r;
  r.x = __builtin_amdgcn_update_dpp(0, w.x, CTRL, 0xf, 0xf, false);
  r.y = __builtin_amdgcn_update_dpp(0, w.y, CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int2 w = __builtin_bit_cast(int2, v);
  int2 r;
  r.x = __builtin_amdgcn_readlane(w.x, l); r.y = __builtin_amdgcn_readlane(w.y, l);
  return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double wsum(double v) {
  v += dpp_f64<0xB1>(v);            // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);            // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);           // row_half_mirror
  v += dpp_f64<0x140>(v);           // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ double wmin(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
  return v;
}

// The symmetric matrix being reduced: only its LOWER triangle (i >= j) is kept and updated.  Two homes with the same arithmetic:
// the caller's (n,n) array in global memory, and a packed triangle in LDS for the leading block once it fits (every step of the
// reduction is a handful of dependent phases over a k x k block: from LDS a phase costs an LDS round trip instead of an L2 one).
struct MatGlobal {
  double* A; int n;
  static constexpr bool kLds = false;
  __device__ __forceinline__ double& at(int i, int j) const { return A[(size_t)i * n + j]; }
  __device__ __forceinline__ double sym(int i, int j) const { return i >= j ? A[(size_t)i * n + j] : A[(size_t)j * n + i]; }
};
struct MatLds {
  double* t;
  static constexpr bool kLds = true;
  __device__ __forceinline__ double& at(int i, int j) const { return t[((i * (i + 1)) >> 1) + j]; }
  __device__ __forceinline__ double sym(int i, int j) const { return i >= j ? t[((i * (i + 1)) >> 1) + j] : t[((j * (j + 1)) >> 1) + i]; }
};

// number of eigenvalues < x of the tridiagonal block rows [lo, hi) (d, e2 = e^2; e[i] couples i-1 and i; e2[lo] is not read): sign changes
// of the leading principal minors p_i = (d_i - x) p_{i-1} - e_i^2 p_{i-2}, kept in range by rescaling (two dependent FMAs per row
// instead of the fp64 division of the ratio form; a vanishing minor counts as a change and is nudged, as LAPACK's pivmin).
// Where e[i] is exactly zero the recurrence restarts by itself (p_i = (d_i - x) p_{i-1}: the sign changes iff d_i < x), so the
// count of the whole matrix is the sum of the counts of its blocks: the parts run on different threads.
__device__ int sturm_count(const double* d, const double* e2, int lo, int hi, double x, double pivmin) {
  if (lo >= hi) return 0;
  double pm = 1.0, p = d[lo] - x;
  if (fabs(p) < pivmin) p = -pivmin;
  int cnt = p < 0.0;
  // four rows per trip, branch-free: rows behind the block's end are the neutral row (d - x = 1, e = 0: p stays, no sign change);
  // the dependent chain of a row is e^2 p_{i-2} -> fma -> the pivmin select; the rescaling (magnitudes grow or shrink by at most
  // ~1e5 per row of a Laplacian of order <= 1024) is applied once per trip
  double dn[4], en[4];                                 // the next trip's rows, in flight while the current ones are evaluated
  auto load = [&](int i0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int ic = min(i0 + q, hi - 1); dn[q] = d[ic]; en[q] = e2[ic]; }   // clamped: no branch around the loads
  };
  load(lo + 1);
  for (int i0 = lo + 1; i0 < hi; i0 += 4) {
    double dx[4], ee[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const bool in = i0 + q < hi; dx[q] = in ? dn[q] - x : 1.0; ee[q] = in ? en[q] : 0.0; }
    load(i0 + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double pn = fma(dx[q], p, -(ee[q] * pm));
      pn = fabs(pn) < pivmin * fabs(p) ? -pivmin * p : pn;            // |p_i / p_{i-1}| < pivmin
      cnt += (pn < 0.0) != (p < 0.0);
      pm = p; p = pn;
    }
    const double ap = fabs(p);
    const double sc = ap > 1e100 ? 1e-100 : (ap < 1e-100 ? 1e100 : 1.0);
    pm *= sc; p *= sc;
  }
  return cnt;
}

// 1 / x from the hardware's seed and two Newton steps (about one ulp; the compiler's exact division is three times the dependent chain)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// one inverse-iteration solve (T - lam I) y = y in place on the block rows [lo, hi) of the tridiagonal matrix, e[lo] == 0 and
// e[hi] == 0 (or hi == n): Gaussian elimination with partial pivoting (U keeps two super-diagonals b, c; work arrays a, b, c
// indexed by row), sequential, one thread.  The blocks of a block-diagonal T are independent systems -- the elimination of the
// whole matrix does exactly these operations row by row -- so every block gets a thread of its own.
__device__ void tri_solve(const double* d, const double* e, int lo, int hi, double lam, double* y, double* a, double* b, double* c, double eps) {
  double ai = d[lo] - lam, bi = lo + 1 < hi ? e[lo + 1] : 0.0, yi = y[lo];   // current row i: diagonal, first super-diagonal, rhs
#pragma unroll 4
  for (int i = lo; i + 1 < hi; ++i) {
    const double sub = e[i + 1];                                       // row i+1: sub, diagonal, super
    const double dn = d[i + 1] - lam, sn = (i + 2 < hi) ? e[i + 2] : 0.0, yn = y[i + 1];
    if (fabs(sub) > fabs(ai)) {                                        // swap rows i and i+1
      const double ip = fast_rcp(sub), m = ai * ip;                        // a[] keeps the pivot's reciprocal: no division in the back substitution
      a[i] = ip; b[i] = dn; c[i] = sn; y[i] = yn;
      ai = bi - m * dn; bi = -m * sn; yi = yi - m * yn;
    } else {
      const double piv = ai == 0.0 ? eps : ai;
      const double ip = fast_rcp(piv), m = sub * ip;
      a[i] = ip; b[i] = bi; c[i] = 0.0; y[i] = yi;
      ai = dn - m * bi; bi = sn; yi = yn - m * yi;
    }
  }
  if (ai == 0.0) ai = eps;
  double y1 = yi * fast_rcp(ai), y2 = 0.0;                             // y[i+1], y[i+2] of the back substitution
  y[hi - 1] = y1;
#pragma unroll 4
  for (int i = hi - 2; i >= lo; --i) {
    const double v = (y[i] - b[i] * y1 - c[i] * y2) * a[i];
    y[i] = v; y2 = y1; y1 = v;
  }
}

// Diagnostic build only (EXP_EIG_STAMP): cycles of the Householder step's six phases, summed over the steps of graph 0 by its thread 0
#ifdef EXP_EIG_STAMP
__device__ long long g_eig_phase[8];
extern "C" int da_debug_read_eig_phases(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_eig_phase), sizeof(long long) * 8); }
#define EIG_PH(i) { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long t_ = __builtin_readcyclecounter(); g_eig_phase[i] += t_ - ph_t; ph_t = t_; } }
#else
#define EIG_PH(i)
#endif

struct EigLds {
  double *u, *p, *d, *e, *h, *y, *w, *red, *lam, *pp, *tri;
  int *iu, *seg, *cnt, *misc;
};

// One Householder step of the tridiagonalisation: row k (its entries j < k) is annihilated left of the sub-diagonal, the leading
// k x k block gets the rank-2 update (lower triangle only), the reflector stays in row k, h_k = u^T u / 2 in s_h, the new
// sub-diagonal entry in s_e.  Six barriers; every reduction is summed in a fixed order (bit-reproducible run to run, and the same
// bits whichever home the matrix has).  The reflector lives inside the connected component of node k (a reaction graph is many
// components: the product, the reactant fragments, isolated atoms), everything outside is exactly zero and is skipped:
// U = {j: u_j != 0} bounds the rows of the product A u, T = {j: u_j != 0 or (A u)_j != 0} the rank-2 update.
// Returns with nothing pending on row k; `first_u` receives the smallest index of U (the caller's coupling check), or k.
template <int EIG_THREADS, class M>
__device__ __forceinline__ void house_step(const M& Mx, double* __restrict__ Aglob, int n, int k, int pb, const EigLds& S, int& first_u) {
  constexpr int EIG_NWV = EIG_THREADS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* s_part = S.red + pb * 32;                   // [2][16] per-wave partials of this step (double buffered over steps)
  int* s_cnt = S.cnt + pb * 32;                       // [2][16] per-wave counts of the two compactions
  first_u = k;
#ifdef EXP_EIG_STAMP
  long long ph_t = __builtin_readcyclecounter();
#endif
  // (a) row k: u = A[k][0:k], sum of squares left of the sub-diagonal, the index set U (u_{k-1} = f - g is never zero)
  const double x0 = tid < k ? Mx.at(k, tid) : 0.0;
  const bool inU = tid < k && (x0 != 0.0 || tid == k - 1);
  double part = tid < k - 1 ? x0 * x0 : 0.0;
  if (tid < k) S.u[tid] = x0;
  part = wsum(part);
  const unsigned long long mU = __ballot(inU);
  if (lane == 0) { s_part[wave] = part; s_cnt[wave] = __popcll(mU); }
  if (tid == 0) S.d[k] = Mx.at(k, k);                 // final: no later step touches row k
  if (tid == k - 1) S.lam[pb] = x0;                   // f, for everyone (its owner overwrites the entry itself with f - g below)
  __syncthreads();                                                                // (1)
  EIG_PH(0)
  double off2 = 0.0;
#pragma unroll
  for (int w = 0; w < EIG_NWV; ++w) off2 += s_part[w];
  const double f = S.lam[pb];
  if (off2 == 0.0) {                                  // nothing to annihilate: the row is already tridiagonal
    if (tid == 0) { S.e[k] = f; S.h[k] = 0.0; if (M::kLds) Aglob[(size_t)k * n + k] = S.d[k]; }
    return;
  }
  const double hh = off2 + f * f;
  const double g = f >= 0.0 ? -sqrt(hh) : sqrt(hh);
  const double h = hh - f * g;                        // = u^T u / 2
  const double xk = tid == k - 1 ? f - g : x0;        // this thread's entry of u
  if (tid == k - 1) S.u[tid] = xk;
  if (tid == 0) { S.e[k] = g; S.h[k] = h; if (M::kLds) Aglob[(size_t)k * n + k] = S.d[k]; }
  int offU = 0, nU = 0;
#pragma unroll
  for (int w = 0; w < EIG_NWV; ++w) { const int c = s_cnt[w]; offU += w < wave ? c : 0; nU += c; }
  if (inU) S.iu[offU + __popcll(mU & ((1ull << lane) - 1ull))] = tid;
  if (tid < k) {                                      // the reflector is final: row k of the caller's array (and of the LDS copy)
    Aglob[(size_t)k * n + tid] = xk;
    if (M::kLds) Mx.at(k, tid) = xk;
  }
  __syncthreads();                                                                // (2)
  EIG_PH(1)
  first_u = S.iu[0];
  // (b) h p = A u:  (A u)_j = sum_{i in U} A(i,j) u_i, one thread per column j and row class (as many classes as the threads allow: a
  // thread's share is a chain of dependent LDS round trips); partial sums meet in s_pp[class][column].  U = all rows (a dense
  // component, e.g. what a random-weight denoiser leaves): no index list in the way.
  const int CW = k;                                   // k <= EIG_THREADS
  const int RC = min(16, EIG_THREADS / CW);
  {
    const int j = tid % CW, rc = tid / CW;
    if (rc < RC) {
      double acc[4] = {0, 0, 0, 0};
      auto walk = [&](auto dense) {
        auto row = [&](int ii) { return decltype(dense)::value ? ii : S.iu[ii]; };
        int ii = rc;
        for (; ii + 7 * RC < nU; ii += 8 * RC) {
          double v[8]; int ix[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) { ix[q] = row(ii + q * RC); v[q] = Mx.sym(ix[q], j); }
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[q & 3] = fma(v[q], S.u[ix[q]], acc[q & 3]);
        }
        for (; ii < nU; ii += RC) { const int i = row(ii); acc[0] = fma(Mx.sym(i, j), S.u[i], acc[0]); }
      };
      if (nU == k) walk(std::true_type{}); else walk(std::false_type{});
      S.pp[rc * CW + j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
  }
  __syncthreads();                                                                // (3)
  EIG_PH(2)
  double pj = 0.0;
  if (tid < k) {
    for (int rc = 0; rc < RC; ++rc) pj += S.pp[rc * CW + tid];
    S.p[tid] = pj;
  }
  double kp = tid < k ? xk * pj : 0.0;
  const bool inT = tid < k && (xk != 0.0 || pj != 0.0);
  kp = wsum(kp);
  const unsigned long long mT = __ballot(inT);
  if (lane == 0) { s_part[16 + wave] = kp; s_cnt[16 + wave] = __popcll(mT); }
  __syncthreads();                                                                // (4)
  EIG_PH(3)
  double upu = 0.0;
  int offT = 0, nT = 0;
#pragma unroll
  for (int w = 0; w < EIG_NWV; ++w) { upu += s_part[16 + w]; const int c = s_cnt[16 + w]; offT += w < wave ? c : 0; nT += c; }
  if (inT) S.iu[offT + __popcll(mT & ((1ull << lane) - 1ull))] = tid;     // (the list of U is dead: every reader passed barrier 3)
  const double invh = 1.0 / h;
  const double K = upu * invh * invh * 0.5;           // u^T p / (2h) with p = (A u) / h
  if (tid < k) S.p[tid] = fma(pj, invh, -(K * xk));   // q = p / h - K u, once per entry (s_p's only reader before this was its owner)
  __syncthreads();                                                                // (5)
  EIG_PH(4)
  // (c) A -= u q^T + q u^T on the lower triangle of T x T (q in s_p).  A thread owns a column and a class of rows; the lanes of
  // a wave walk the SAME rows (row i of the triangle is contiguous in j: conflict-free, u_i and q_i are broadcasts) and sit out
  // the rows above their column's diagonal.
  const int CT = nT;
  const int RT = nT > 0 ? min(16, EIG_THREADS / CT) : 1;
  {
    const int jj = nT > 0 ? tid % CT : 0, rc = nT > 0 ? tid / CT : RT;
    if (rc < RT) {
      auto walk = [&](auto dense) {
        auto row = [&](int ii) { return decltype(dense)::value ? ii : S.iu[ii]; };
        const int j = row(jj);
        const double uj = S.u[j], qj = S.p[j];
        int ii = rc + ((jj - rc + RT - 1) / RT) * RT;            // first row of this class at or below the diagonal (jj >= rc: else rc)
        if (jj < rc) ii = rc;
        for (; ii + 3 * RT < nT; ii += 4 * RT) {
          double v[4]; int ix[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { ix[q] = row(ii + q * RT); v[q] = Mx.at(ix[q], j); }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double ui = S.u[ix[q]], qi = S.p[ix[q]];
            Mx.at(ix[q], j) = v[q] - fma(qj, ui, qi * uj);
          }
        }
        for (; ii < nT; ii += RT) {
          const int i = row(ii);
          const double ui = S.u[i], qi = S.p[i];
          Mx.at(i, j) = Mx.at(i, j) - fma(qj, ui, qi * uj);
        }
      };
      if (nT == k) walk(std::true_type{}); else walk(std::false_type{});
    }
  }
  __syncthreads();                                                                // (6) this step's stores before the next row is read
  EIG_PH(5)
}

// Applies the reflectors of rows s_rows[r0 .. r1) (ascending) to the vector y, one wave: lane l holds entries l, l + 64, ... in
// registers; a reflector row comes from the LDS triangle (k <= k_top) or from the caller's array, D rows at a time, the next D
// already in flight while the current ones are applied (an L2 round trip is ~2,000 cycles, a reflector ~250).  Writes back the
// entries [own_lo, own_hi).
template <int NQ, int D>
__device__ __forceinline__ void back_transform(const double* __restrict__ A, int n, const double* tri, int k_top, const int* s_rows,
                                               const double* s_invh, int r0, int r1, double* y, int own_lo, int own_hi) {
  const int lane = threadIdx.x & 63;
  double yq[NQ], un[D][NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { const int j = lane + 64 * q; yq[q] = j < n ? y[j] : 0.0; }
  auto fetch = [&](int r) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int k = r + d < r1 ? s_rows[r + d] : 0;            // 0: no row (all entries zero)
      if (k <= k_top) {
        const double* u = tri + ((k * (k + 1)) >> 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int j = lane + 64 * q; un[d][q] = j < k ? u[j] : 0.0; }
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) { const int j = lane + 64 * q; un[d][q] = j < k ? A[(size_t)k * n + j] : 0.0; }
      }
    }
  };
  if (r0 < r1) fetch(r0);
  for (int r = r0; r < r1; r += D) {
    double uc[D][NQ];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int q = 0; q < NQ; ++q) uc[d][q] = un[d][q];
    if (r + D < r1) fetch(r + D);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (r + d < r1) {                                        // (uniform)
        double dot = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) dot = fma(uc[d][q], yq[q], dot);
        dot = wsum(dot) * s_invh[s_rows[r + d]];
#pragma unroll
        for (int q = 0; q < NQ; ++q) yq[q] = fma(-dot, uc[d][q], yq[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) { const int j = lane + 64 * q; if (j < n && j >= own_lo && j < own_hi) y[j] = yq[q]; }
}

// Where the features go: n_comp[b * ld_nc], ev5[b * ld_ev + 0..4], nonlcc[(b n + i) * ld_nl], vec2[(b n + i) * ld_v2 + 0..1] --
// dense tensors of their own (1, 5, 1, 2) or columns of the denoiser's input rows (da_extra_eigen_solve_into)
struct EigOut { float *n_comp, *ev5, *nonlcc, *vec2; int ld_nc, ld_ev, ld_nl, ld_v2; };

template <int EIG_THREADS>
__global__ void __launch_bounds__(EIG_THREADS) k_extra_eig_solve(double* __restrict__ Lmat, const uint8_t* __restrict__ node_mask,
                                                                 const float* __restrict__ pad_noise, const EigOut O,
                                                                 double* __restrict__ evals_out, int n,
                                                                 const int* __restrict__ fixed_from, int* __restrict__ cache_state,
                                                                 double* __restrict__ cache, int mmax, int k_lds) {
  constexpr int EIG_NWV = EIG_THREADS / 64;
  extern __shared__ double sm[];
  EigLds S;
  S.u = sm;                         // n   Householder vector of the current step
  S.p = sm + n;                     // n
  S.d = sm + 2 * n;                 // n   diagonal of T
  S.e = sm + 3 * n;                 // n   sub-diagonal of T (e[i] couples i-1 and i)
  S.h = sm + 4 * n;                 // n   u^T u / 2 of every reflector (0: skipped)
  S.y = sm + 5 * n;                 // 3n  eigenvectors
  S.w = sm + 8 * n;                 // 9n  work of the three tridiagonal solves (a, b, c each)
  S.red = sm + 17 * n;              // 64
  S.lam = S.red + 64;               // 8
  S.pp = S.lam + 8;                 // EIG_THREADS  partial sums of A u per (row class, column): summed in a fixed order
  S.iu = reinterpret_cast<int*>(S.pp + EIG_MAX_THREADS);   // n      compacted index list
  S.seg = S.iu + n;                 // n + 1  starts of the blocks of T
  S.cnt = S.seg + n + 1;            // 64     per-wave counts
  S.misc = S.cnt + 64;              // 320    Sturm counts per (eigenvalue, point), cuts, brackets' indices
  S.tri = sm + eig_fixed_doubles(n);                   // packed lower triangle of the leading block (k_lds + 1 rows)
  __shared__ int s_ncomp, s_nreal, s_coupled, s_nseg;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* A = Lmat + (size_t)b * n * n;
  const uint8_t* m = node_mask + (size_t)b * n;

#ifdef EXP_EIG_STAMP
  const long long st0 = __builtin_readcyclecounter();
#endif
  // ---- 1. tridiagonalisation (rows k = n-1 .. 1; reflector u_k stored in A[k][0:k], h_k = u^T u / 2 in s_h).
  // Fixed trailing block (da_extra_eigen_solve_cached): the nodes p0 .. n-1 keep all their edges from call to call (the product
  // side of a reaction along a reverse chain, graph.apply_mask) and share no edge with the nodes before them.  Rows k >= p0 are
  // reduced first (k runs downwards) and their reflectors never leave the block, so what those steps leave behind -- the rows'
  // lower triangles with the updated diagonal, e_k and h_k -- is the same bits at every call: kept per graph after the first
  // call (state 0 -> 1) and copied back afterwards, and the reduction starts at row p0.  A block that does share an edge with
  // the rest is noticed while it is reduced (state -1: always reduced in full).
  const int p0 = cache ? min(max(fixed_from[b], 0), n) : n;
  const int cstate = (cache && p0 < n && n - p0 <= mmax) ? cache_state[b] : -1;
  double* cA = cache ? cache + (size_t)b * (mmax + 2) * mmax : nullptr;       // [mmax][mmax] rows, then e, then h
  int kstart = n - 1;
  if (tid == 0) s_coupled = 0;
  if (cstate == 1) {
    const int mm = n - p0;
    for (int t = tid; t < mm * mm; t += EIG_THREADS) {
      const int rk = t / mm, cj = t - rk * mm;
      if (cj <= rk) A[(size_t)(p0 + rk) * n + p0 + cj] = cA[(size_t)rk * mmax + cj];
    }
    for (int t = tid; t < mm; t += EIG_THREADS) { S.e[p0 + t] = cA[(size_t)mmax * mmax + t]; S.h[p0 + t] = cA[(size_t)(mmax + 1) * mmax + t]; }
    kstart = p0;                                        // row p0 itself has nothing before it: the loop sets e = 0, h = 0 for it
  }
  __syncthreads();
  const MatGlobal Mg{A, n};
  const MatLds Ml{S.tri};
  int k_top = -1;                                       // rows 0 .. k_top live in LDS from the step that reaches k_lds on
  int step = 0;
  for (int k = kstart; k >= 1; --k, ++step) {
    int first_u;
    if (k_top < 0 && k <= k_lds) {                      // the leading block fits: move its lower triangle into LDS
      for (int t = tid; t < ((k + 1) * (k + 2)) >> 1; t += EIG_THREADS) {
        int i = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while (((i + 1) * (i + 2)) >> 1 <= t) ++i;
        while ((i * (i + 1)) >> 1 > t) --i;
        S.tri[t] = A[(size_t)i * n + (t - ((i * (i + 1)) >> 1))];
      }
      k_top = k;
      __syncthreads();
    }
    if (k_top >= 0) house_step<EIG_THREADS>(Ml, A, n, k, step & 1, S, first_u);
    else house_step<EIG_THREADS>(Mg, A, n, k, step & 1, S, first_u);
    if (cstate == 0 && k >= p0 && tid == 0 && first_u < p0) s_coupled = 1;   // the block reaches outside itself: not cacheable
  }
  __syncthreads();
#ifdef EXP_EIG_STAMP
  const long long st1 = __builtin_readcyclecounter();
#endif
  if (cstate == 0) {                                    // first call: keep what the block's steps left (nothing later touched it)
    const int mm = n - p0;
    if (!s_coupled) {
      for (int t = tid; t < mm * mm; t += EIG_THREADS) {
        const int rk = t / mm, cj = t - rk * mm;
        if (cj <= rk) cA[(size_t)rk * mmax + cj] = (cj == rk && p0 + rk <= kstart && p0 + rk >= 1) ? S.d[p0 + rk] : A[(size_t)(p0 + rk) * n + p0 + cj];
      }
      for (int t = tid; t < mm; t += EIG_THREADS) {
        cA[(size_t)mmax * mmax + t] = p0 + t >= 1 ? S.e[p0 + t] : 0.0; cA[(size_t)(mmax + 1) * mmax + t] = p0 + t >= 1 ? S.h[p0 + t] : 0.0;
      }
    }
    if (tid == 0) cache_state[b] = s_coupled ? -1 : 1;
  }
  // diagonal of T: rows the loop visited left theirs in s_d; row 0 and (cached call) the rows above kstart come from the matrix
  for (int i = tid; i < n; i += EIG_THREADS)
    if (i == 0 || i > kstart) S.d[i] = (i == 0 && k_top >= 0) ? Ml.at(0, 0) : A[(size_t)i * n + i];
  if (tid == 0) { S.e[0] = 0.0; S.h[0] = 0.0; s_nreal = 0; }
  __syncthreads();
  int c = 0;
  for (int i = tid; i < n; i += EIG_THREADS) c += m[i] ? 1 : 0;
  if (c) atomicAdd(&s_nreal, c);
  // Gershgorin bounds and the largest coupling (min / max: exact in any order)
  double gl = INFINITY, gu = -INFINITY, emax = 0.0;
  for (int i = tid; i < n; i += EIG_THREADS) {
    const double r = fabs(S.e[i]) + (i + 1 < n ? fabs(S.e[i + 1]) : 0.0);
    gl = fmin(gl, S.d[i] - r); gu = fmax(gu, S.d[i] + r); emax = fmax(emax, fabs(S.e[i]));
  }
  gl = wmin(gl); gu = wmax(gu); emax = wmax(emax);
  if (lane == 0) { S.red[wave] = gl; S.red[16 + wave] = gu; S.red[32 + wave] = emax; }
  // blocks of T: a new block starts at every row whose coupling to the previous row is exactly zero (ordered list), and the cut
  // rows of the Sturm counts' EIG_NP parts: the block starts nearest to the quarter points
  int* s_cut = S.misc + 256;                            // [EIG_NP + 1]
  if (tid <= EIG_NP) s_cut[tid] = tid == 0 ? 0 : 0x7fffffff;
  __syncthreads();
  {
    int base = 0;                                       // ordered compaction over chunks of EIG_THREADS rows
    for (int i0 = 0; i0 < n; i0 += EIG_THREADS) {
      const int i = i0 + tid;
      const bool st = i < n && (i == 0 || S.e[i] == 0.0);
      const unsigned long long mb = __ballot(st);
      if (lane == 0) S.cnt[wave] = __popcll(mb);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < EIG_NWV; ++w) { const int cc = S.cnt[w]; off += w < wave ? cc : 0; tot += cc; }
      if (st) {
        S.seg[base + off + __popcll(mb & ((1ull << lane) - 1ull))] = i;
        if (i > 0)
          for (int q = 1; q < EIG_NP; ++q) atomicMin(&s_cut[q], abs(i - q * n / EIG_NP) * 2048 + i);
      }
      base += tot;
      __syncthreads();
    }
    if (tid == 0) { s_nseg = base; S.seg[base] = n; }
  }
  __syncthreads();
  gl = S.red[0]; gu = S.red[16]; emax = S.red[32];
#pragma unroll
  for (int w = 1; w < EIG_NWV; ++w) { gl = fmin(gl, S.red[w]); gu = fmax(gu, S.red[16 + w]); emax = fmax(emax, S.red[32 + w]); }
  const double nreal = (double)s_nreal;
  const int nseg = s_nseg;
  int cut_lo = 0, cut_hi = 0;                           // this thread's part of a Sturm count
  {
    const int part = tid / (6 * EIG_PTS);
    if (part < EIG_NP) {
      const int a0 = part == 0 ? 0 : s_cut[part], a1 = part + 1 == EIG_NP ? n : s_cut[part + 1];
      cut_lo = a0 == 0x7fffffff ? n : (part == 0 ? 0 : a0 & 2047);
      cut_hi = a1 == 0x7fffffff ? n : (part + 1 == EIG_NP ? n : a1 & 2047);
    }
  }

  // ---- 2. the eigenvalues that are used: count below the threshold, then indices 0 and ncomp .. ncomp+4
  const double tnorm = fmax(fabs(gl), fabs(gu));
  gl -= 2.2e-16 * tnorm * n + 1e-290; gu += 2.2e-16 * tnorm * n + 1e-290;
  const double pivmin = fmax(2.3e-308 * fmax(1.0, emax * emax), 1e-300);
  double* s_e2 = S.u;                                   // e^2 (the Householder vector's array is free now)
  for (int i = tid; i < n; i += EIG_THREADS) s_e2[i] = S.e[i] * S.e[i];
  int* s_c = S.misc;                                    // [6 * EIG_PTS] counts per (eigenvalue, point), summed over the parts
  int* s_lo = S.misc + 264;                             // [6] index of the last point with count <= idx, per round
  double* s_br = S.w;                                   // [6][2] brackets
  if (tid == 0) s_ncomp = 0;
  if (tid < 6) { s_br[2 * tid] = gl; s_br[2 * tid + 1] = gu; }
  __syncthreads();
  if (tid % (6 * EIG_PTS) == 0 && tid / (6 * EIG_PTS) < EIG_NP) {                   // eigenvalue / n_real < 1e-5  (:119, :166)
    const int cc = sturm_count(S.d, s_e2, cut_lo, cut_hi, 1e-5 * nreal, pivmin);
    if (cc) atomicAdd(&s_ncomp, cc);
  }
  __syncthreads();
  const int ncomp = s_ncomp;
  // six eigenvalues (indices 0, ncomp .. ncomp+4) by multisection: EIG_PTS interior points of the current bracket per eigenvalue and
  // round (3.5 bits per round instead of 1), each count cut into EIG_NP parts: 6 x 10 x 4 threads.  The count is bound by fp64
  // vector throughput (about twenty dependent instructions per row): 42 points per round, as in round 2, cost 2.9 x the row evaluations
  {
    const int r = tid % (6 * EIG_PTS), part = tid / (6 * EIG_PTS);
    const int ev = r / EIG_PTS, pt = r - ev * EIG_PTS;
    const int idx = ev == 0 ? 0 : ncomp + ev - 1;
    for (int round = 0; round < EIG_ROUNDS; ++round) {
      if (tid < 6 * EIG_PTS) s_c[tid] = 0;
      if (tid < 6) s_lo[tid] = -1;
      __syncthreads();
      if (part < EIG_NP && idx < n && cut_lo < cut_hi) {
        const double lo = s_br[2 * ev], hi = s_br[2 * ev + 1];
        const double x = lo + (hi - lo) * ((double)(pt + 1) / (double)(EIG_PTS + 1));
        const int cc = sturm_count(S.d, s_e2, cut_lo, cut_hi, x, pivmin);
        if (cc) atomicAdd(&s_c[r], cc);
      }
      __syncthreads();
      if (part == 0 && idx < n && s_c[r] <= idx) atomicMax(&s_lo[ev], pt);          // eigenvalue idx is >= point pt
      __syncthreads();
      if (tid < 6) {
        const double lo = s_br[2 * tid], hi = s_br[2 * tid + 1];
        const int q = s_lo[tid];                           // bracket becomes [point q, point q+1]  (point -1 = lo, point PTS = hi)
        const double nlo = lo + (hi - lo) * ((double)(q + 1) / (double)(EIG_PTS + 1));
        const double nhi = lo + (hi - lo) * ((double)(q + 2) / (double)(EIG_PTS + 1));
        s_br[2 * tid] = q < 0 ? lo : nlo; s_br[2 * tid + 1] = q + 1 >= EIG_PTS ? hi : nhi;
      }
      __syncthreads();
    }
    if (tid < 6) {
      const int id2 = tid == 0 ? 0 : ncomp + tid - 1;
      S.lam[tid] = id2 < n ? 0.5 * (s_br[2 * tid] + s_br[2 * tid + 1]) : 2.0 * nreal;
    }
    __syncthreads();
  }
  if (tid == 0) O.n_comp[(size_t)b * O.ld_nc] = (float)ncomp;
  if (tid < 5) O.ev5[(size_t)b * O.ld_ev + tid] = (ncomp + tid < n) ? (float)(S.lam[1 + tid] / nreal) : 2.f;   // :169-173
#ifndef EXP_EIG_STAMP
  if (evals_out && tid < 6) evals_out[(size_t)b * 6 + tid] = S.lam[tid];
#endif

#ifdef EXP_EIG_STAMP
  const long long st2 = __builtin_readcyclecounter();
#endif
  // ---- 3. eigenvectors 0, ncomp, ncomp+1 of T by inverse iteration: one thread per (vector, block of T); the three iterations'
  // max-norm scalings are block-wide reductions
  const double eps3 = 2.2e-16 * fmax(tnorm, 1e-300);
  __syncthreads();                                      // (s_br lives in s_w: the brackets are dead from here on)
  for (int v = 0; v < 3; ++v) {
    const int idx = v == 0 ? 0 : ncomp + v - 1;
    double* y = S.y + v * n;
    for (int i = tid; i < n; i += EIG_THREADS) {
      // deterministic pseudo-random start (a vector of ones is orthogonal to many eigenvectors of a Laplacian)
      unsigned hsh = (unsigned)(i * 2654435761u) ^ (unsigned)((v + 1) * 40503u) ^ (unsigned)(b * 97u);
      hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
      y[i] = idx < n ? 0.5 + (double)(hsh & 0xffffu) / 65536.0 : 0.0;
    }
  }
  __syncthreads();
  {
    double lamv[3];
    bool live[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int idx = v == 0 ? 0 : ncomp + v - 1;
      live[v] = idx < n;
      double lam = S.lam[v == 0 ? 0 : v];
      // separate (nearly) equal eigenvalues slightly, as LAPACK's stein does, so that the two solves do not return one vector
      if (v == 2 && fabs(lam - S.lam[1]) <= 10.0 * eps3) lam = S.lam[1] + 10.0 * eps3;
      lamv[v] = lam + (v + 1) * eps3;                   // never exactly singular
    }
    const int units = 3 * nseg;
    const bool wave_per_unit = units <= EIG_NWV;        // few blocks: one per wave (no divergence between lanes)
    for (int it = 0; it < 3; ++it) {
      for (int u = wave_per_unit ? (lane == 0 ? wave : units) : tid; u < units; u += EIG_THREADS) {
        const int v = u / nseg, sg = u - v * nseg;
        if (live[v])
          tri_solve(S.d, S.e, S.seg[sg], S.seg[sg + 1], lamv[v], S.y + v * n, S.w + v * 3 * n, S.w + v * 3 * n + n, S.w + v * 3 * n + 2 * n, eps3);
      }
      __syncthreads();
      double nm3[3] = {0.0, 0.0, 0.0};
      for (int i = tid; i < n; i += EIG_THREADS)
#pragma unroll
        for (int v = 0; v < 3; ++v) nm3[v] = fmax(nm3[v], fabs(S.y[v * n + i]));
#pragma unroll
      for (int v = 0; v < 3; ++v) { nm3[v] = wmax(nm3[v]); if (lane == 0) S.red[v * 16 + wave] = nm3[v]; }
      __syncthreads();
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        double nrm = 0.0;
#pragma unroll
        for (int w = 0; w < EIG_NWV; ++w) nrm = fmax(nrm, S.red[v * 16 + w]);
        if (live[v] && !(nrm > 0.0)) live[v] = false;   // (uniform) the vector vanished: left as it is
        if (live[v]) {
          const double sc = 1.0 / nrm;
          for (int i = tid; i < n; i += EIG_THREADS) S.y[v * n + i] *= sc;
        }
      }
      __syncthreads();
    }
  }
  if (ncomp + 1 < n) {                                  // the two vectors after the zero ones: orthonormal pair (degenerate pairs)
    double* y1 = S.y + n; double* y2 = S.y + 2 * n;
    if (fabs(S.lam[2] - S.lam[1]) <= 1e-8 * fmax(1.0, tnorm)) {                      // (uniform)
      if (wave == 0) {
        double d11 = 0.0, d12 = 0.0;
        for (int i = lane; i < n; i += 64) { d11 = fma(y1[i], y1[i], d11); d12 = fma(y1[i], y2[i], d12); }
        d11 = wsum(d11); d12 = wsum(d12);
        if (d11 > 0.0) for (int i = lane; i < n; i += 64) y2[i] -= d12 / d11 * y1[i];
      }
    }
  }
  __syncthreads();
#ifdef EXP_EIG_STAMP
  const long long st3 = __builtin_readcyclecounter();
#endif
  // back-transformation x = H_{n-1} ... H_1 y: reflector k acts on entries 0..k-1.  One wave per vector, the vector's entries in
  // registers (lane l holds entries l, l + 64, ...), reflector rows from the LDS triangle where they live there, else from the
  // caller's array, always one reflector ahead of the one being applied.  With a closed fixed block (the cached calls) its
  // reflectors touch the entries from p0 on only and the others the entries before p0: two waves per vector, one per range
  // (the entries outside a reflector's range are exact zeros of u: leaving them out does not change a bit).
  int* s_rows = S.iu;                                   // rows with a reflector, ascending
  double* s_invh = S.w;                                 // 1 / h_k (the solves' work arrays are dead)
  __shared__ int s_nrows, s_nrows_lo;
  {
    int base = 0, base_lo = 0;
    for (int i0 = 0; i0 < n; i0 += EIG_THREADS) {
      const int i = i0 + tid;
      const bool has = i >= 1 && i < n && S.h[i] != 0.0;
      if (has) s_invh[i] = 1.0 / S.h[i];
      const unsigned long long mb = __ballot(has), ml = __ballot(has && i <= p0);
      if (lane == 0) { S.cnt[wave] = __popcll(mb); S.cnt[16 + wave] = __popcll(ml); }
      __syncthreads();
      int off = 0, tot = 0, tot_lo = 0;
#pragma unroll
      for (int w = 0; w < EIG_NWV; ++w) { const int cc = S.cnt[w]; off += w < wave ? cc : 0; tot += cc; tot_lo += S.cnt[16 + w]; }
      if (has) s_rows[base + off + __popcll(mb & ((1ull << lane) - 1ull))] = i;
      base += tot; base_lo += tot_lo;
      __syncthreads();
    }
    if (tid == 0) { s_nrows = base; s_nrows_lo = base_lo; }
  }
  __syncthreads();
  {
    const bool split = cache != nullptr && p0 < n && (cstate == 1 || (cstate == 0 && !s_coupled));
    const int vec = wave % 3, strm = wave / 3;
    if (wave < (split ? 6 : 3)) {
      double* y = S.y + vec * n;
      const int r0 = (split && strm == 1) ? s_nrows_lo : 0;
      const int r1 = (split && strm == 0) ? s_nrows_lo : s_nrows;
      const int own_lo = (split && strm == 1) ? p0 : 0, own_hi = (split && strm == 0) ? p0 : n;
      // NQ 64-entry pieces per row, D rows in flight (registers: 2 D NQ + NQ doubles)
      if (EIG_THREADS <= 512 && n <= 320) back_transform<5, 4>(A, n, S.tri, k_top, s_rows, s_invh, r0, r1, y, own_lo, own_hi);
      else if (EIG_THREADS <= 512) back_transform<8, 4>(A, n, S.tri, k_top, s_rows, s_invh, r0, r1, y, own_lo, own_hi);
      else back_transform<16, 1>(A, n, S.tri, k_top, s_rows, s_invh, r0, r1, y, own_lo, own_hi);
    }
  }
  __syncthreads();
  if (wave < 3) {
    double* y = S.y + wave * n;
    double nn = 0.0;
    for (int j = lane; j < n; j += 64) nn = fma(y[j], y[j], nn);
    nn = wsum(nn);
    const double sc = nn > 0.0 ? 1.0 / sqrt(nn) : 0.0;
    for (int j = lane; j < n; j += 64) y[j] *= sc;
  }
  __syncthreads();

#ifdef EXP_EIG_STAMP
  const long long st4 = __builtin_readcyclecounter();
  if (evals_out && tid == 0) { double* o = evals_out + (size_t)b * 6; o[0] = (double)(st1 - st0); o[1] = (double)(st2 - st1); o[2] = (double)(st3 - st2); o[3] = (double)(st4 - st3); }
#endif
  // ---- 4. features (:120, :187-202), as k_extra_eigen
  double* s_first = S.p;
  const double m0 = m[0] ? 1.0 : 0.0;
  for (int i = tid; i < n; i += EIG_THREADS) {
    const double mi = m[i] ? 1.0 : 0.0;
    const double v = S.y[i] * mi * m0;
    s_first[i] = nearbyint(v * 1000.0) / 1000.0 * mi + (double)pad_noise[(size_t)b * n + i] * (1.0 - mi);
  }
  __syncthreads();
  // the mode of s_first (most frequent value; ties: the smallest value): every thread counts its value's occurrences, then a
  // block-wide arg-max over (count, -value)
  int bc = -1; double bv = 0.0;
  for (int i = tid; i < n; i += EIG_THREADS) {
    const double v = s_first[i];
    int kk = 0;
    for (int j = 0; j < n; ++j) kk += (s_first[j] == v) ? 1 : 0;
    if (kk > bc || (kk == bc && v < bv)) { bc = kk; bv = v; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int oc = __shfl_xor(bc, off); const double ov = __shfl_xor(bv, off);
    if (oc > bc || (oc == bc && ov < bv)) { bc = oc; bv = ov; }
  }
  if (lane == 0) { S.cnt[wave] = bc; S.red[wave] = bv; }
  __syncthreads();
  bc = S.cnt[0]; bv = S.red[0];
#pragma unroll
  for (int w = 1; w < EIG_NWV; ++w) { const int oc = S.cnt[w]; const double ov = S.red[w]; if (oc > bc || (oc == bc && ov < bv)) { bc = oc; bv = ov; } }
  const double mode = bv;
  for (int i = tid; i < n; i += EIG_THREADS) {
    const bool mi = m[i] != 0;
    O.nonlcc[((size_t)b * n + i) * O.ld_nl] = (mi && s_first[i] != mode) ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = ncomp + k;
      double v = 0.0;
      if (col < n && mi && m[col]) v = S.y[(1 + k) * n + i];
      O.vec2[((size_t)b * n + i) * O.ld_v2 + k] = (float)v;
    }
  }
}

extern "C" {

static int eig_solve_launch(double* L, const uint8_t* node_mask, const float* pad_noise, const EigOut O, double* evals6, int B, int n,
                            const int* fixed_from, int* cache_state, double* cache, int mmax, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 1 && n <= 1024, "da_extra_eigen_solve: n = %d outside 1..1024 (LDS work arrays)", n);
  DA_CHECK_ARG(cache == nullptr || (fixed_from != nullptr && cache_state != nullptr && mmax >= 1 && mmax <= n),
               "da_extra_eigen_solve_cached: cache given without fixed_from / cache_state, or mmax = %d outside 1..n", mmax);
  if (B == 0) return DA_OK;
  // the packed lower triangle of the leading block goes into what is left of the CU's LDS: rows 0 .. k_lds
  const size_t lds_max = 160 * 1024 - 64;
  const size_t fixed = (size_t)eig_fixed_doubles(n) * sizeof(double);
  DA_CHECK_ARG(fixed <= lds_max, "da_extra_eigen_solve: n = %d needs %zu B of LDS", n, fixed);
  int k_lds = n - 1;
  while (k_lds >= 1 && fixed + (size_t)((k_lds + 1) * (k_lds + 2) / 2) * sizeof(double) > lds_max) --k_lds;
  if (k_lds < 8) k_lds = 0;                               // not worth it: everything stays in the caller's array
  const size_t lds = fixed + (k_lds ? (size_t)((k_lds + 1) * (k_lds + 2) / 2) * sizeof(double) : 0);
  static std::atomic<bool> attr_set[DA_MAX_DEVICES];     // the attribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DA_MAX_DEVICES) { da_set_error("da_extra_eigen_solve: no current HIP device"); return DA_ERR_HIP; }
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_extra_eig_solve<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_extra_eig_solve<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_extra_eig_solve<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) { da_set_error("da_extra_eigen_solve: cannot reserve LDS: %s", hipGetErrorString(e)); return DA_ERR_HIP; }
    attr_set[dev].store(true, std::memory_order_release);
  }
  // a step of the reduction is a chain of barrier-separated phases over at most n entries: 8 waves serve n <= 512 with cheaper barriers
  static const int force_threads = [] { const char* e = getenv("DA_EIG_THREADS"); return e ? atoi(e) : 0; }();   // A/B hook
  if (force_threads == 256 && n <= 256)
    k_extra_eig_solve<256><<<B, 256, lds, (hipStream_t)stream>>>(L, node_mask, pad_noise, O, evals6, n, fixed_from, cache_state, cache, mmax, k_lds);
  else if (n <= 512 && force_threads != 1024)
    k_extra_eig_solve<512><<<B, 512, lds, (hipStream_t)stream>>>(L, node_mask, pad_noise, O, evals6, n, fixed_from, cache_state, cache, mmax, k_lds);
  else
    k_extra_eig_solve<1024><<<B, 1024, lds, (hipStream_t)stream>>>(L, node_mask, pad_noise, O, evals6, n, fixed_from, cache_state, cache, mmax, k_lds);
  DA_CHECK_LAUNCH("da_extra_eigen_solve");
  return DA_OK;
}

/* see include/diffalign_hip.h */
int da_extra_eigen_solve_cached(double* L, const uint8_t* node_mask, const float* pad_noise, float* n_comp, float* ev5, float* nonlcc,
                                float* vec2, double* evals6 /*may be NULL*/, int B, int n, const int* fixed_from, int* cache_state,
                                double* cache, int mmax, void* stream) {
  const EigOut O = {n_comp, ev5, nonlcc, vec2, 1, 5, 1, 2};
  return eig_solve_launch(L, node_mask, pad_noise, O, evals6, B, n, fixed_from, cache_state, cache, mmax, stream);
}

int da_extra_eigen_solve(double* L, const uint8_t* node_mask, const float* pad_noise, float* n_comp, float* ev5, float* nonlcc,
                         float* vec2, double* evals6 /*may be NULL*/, int B, int n, void* stream) {
  return da_extra_eigen_solve_cached(L, node_mask, pad_noise, n_comp, ev5, nonlcc, vec2, evals6, B, n, nullptr, nullptr, nullptr, 0, stream);
}

int da_extra_eigen_solve_into(double* L, const uint8_t* node_mask, const float* pad_noise, float* x_feat, int ldx, float* y_feat, int ldy,
                              int B, int n, const int* fixed_from, int* cache_state, double* cache, int mmax, void* stream) {
  DA_CHECK_ARG(x_feat != nullptr && y_feat != nullptr && ldx >= 3 && ldy >= 6, "da_extra_eigen_solve_into: need x_feat, y_feat, ldx >= 3 and ldy >= 6");
  const EigOut O = {y_feat, y_feat + 1, x_feat, x_feat + 1, ldy, ldy, ldx, ldx};
  return eig_solve_launch(L, node_mask, pad_noise, O, nullptr, B, n, fixed_from, cache_state, cache, mmax, stream);
}

}  // extern "C"
