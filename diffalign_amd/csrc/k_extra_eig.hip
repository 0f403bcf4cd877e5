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
#include "common.h"

#define EIG_THREADS 1024

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// block-wide sum; s_red: 16 doubles of LDS; every thread gets the result
__device__ __forceinline__ double bsum(double v, double* s_red) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < EIG_THREADS / 64; ++w) t += s_red[w];
  return t;
}

// ordered compaction of {j < k : pred(j)} into list[] (LDS); returns the count to every thread.  One index per thread
// (k <= EIG_THREADS); cnt: 16 ints of LDS.  Two barriers.
template <class Pred>
__device__ __forceinline__ int compact(int* list, int* cnt, int k, Pred pred) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool f = tid < k && pred(tid);
  const unsigned long long m = __ballot(f);
  if (lane == 0) cnt[wave] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < EIG_THREADS / 64; ++w) { const int c = cnt[w]; off += w < wave ? c : 0; tot += c; }
  if (f) list[off + __popcll(m & ((1ull << lane) - 1ull))] = tid;
  __syncthreads();
  return tot;
}

// number of eigenvalues of the tridiagonal (d, e) [e[i] couples i-1 and i, e[0] unused] that are < x: sign changes of the
// leading principal minors p_i = (d_i - x) p_{i-1} - e_i^2 p_{i-2}, kept in range by rescaling (two dependent FMAs per row
// instead of the fp64 division of the ratio form; a vanishing minor counts as a change and is nudged, as LAPACK's pivmin)
__device__ int sturm_count(const double* d, const double* e, int n, double x, double pivmin) {
  double pm = 1.0, p = d[0] - x;
  if (fabs(p) < pivmin) p = -pivmin;
  int cnt = p < 0.0;
  for (int i0 = 1; i0 < n; i0 += 8) {                 // eight rows' LDS reads in flight together, branch-free arithmetic
    double dd[8], ee[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int i = min(i0 + q, n - 1); dd[q] = d[i]; ee[q] = e[i]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (i0 + q < n) {                                // uniform
        double pn = fma(dd[q] - x, p, -(ee[q] * ee[q]) * pm);
        pn = fabs(pn) < pivmin * fabs(p) ? -pivmin * p : pn;          // |p_i / p_{i-1}| < pivmin
        cnt += (pn < 0.0) != (p < 0.0);
        const double ap = fabs(pn);
        const double sc = ap > 1e100 ? 1e-100 : (ap < 1e-100 ? 1e100 : 1.0);
        pm = p * sc; p = pn * sc;
      }
    }
  }
  return cnt;
}

// k-th smallest eigenvalue (0-based) by bisection on [lo, hi]
__device__ double bisect(const double* d, const double* e, int n, int k, double lo, double hi, double pivmin) {
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (sturm_count(d, e, n, mid, pivmin) > k) hi = mid; else lo = mid;
    if (hi - lo <= 4e-16 * fmax(fabs(lo), fabs(hi)) + 1e-300) break;
  }
  return 0.5 * (lo + hi);
}

// one inverse-iteration solve (T - lam I) y = y in place: Gaussian elimination with partial pivoting on the tridiagonal
// (U keeps two super-diagonals b, c; work arrays a, b, c of length n in LDS), sequential, one thread.  The row being
// eliminated is carried in registers, so the only loop-carried chain is the arithmetic, not LDS round trips.
__device__ void tri_solve(const double* d, const double* e, int n, double lam, double* y, double* a, double* b, double* c, double eps) {
  double ai = d[0] - lam, bi = n > 1 ? e[1] : 0.0, yi = y[0];          // current row i: diagonal, first super-diagonal, rhs
#pragma unroll 4
  for (int i = 0; i + 1 < n; ++i) {
    const double sub = e[i + 1];                                       // row i+1: sub, diagonal, super
    const double dn = d[i + 1] - lam, sn = (i + 2 < n) ? e[i + 2] : 0.0, yn = y[i + 1];
    if (fabs(sub) > fabs(ai)) {                                        // swap rows i and i+1
      const double m = ai / sub;
      a[i] = sub; b[i] = dn; c[i] = sn; y[i] = yn;
      ai = bi - m * dn; bi = -m * sn; yi = yi - m * yn;
    } else {
      const double piv = ai == 0.0 ? eps : ai;
      const double m = sub / piv;
      a[i] = piv; b[i] = bi; c[i] = 0.0; y[i] = yi;
      ai = dn - m * bi; bi = sn; yi = yn - m * yi;
    }
  }
  if (ai == 0.0) ai = eps;
  double y1 = yi / ai, y2 = 0.0;                                       // y[i+1], y[i+2] of the back substitution
  y[n - 1] = y1;
#pragma unroll 4
  for (int i = n - 2; i >= 0; --i) {
    const double v = (y[i] - b[i] * y1 - c[i] * y2) / a[i];
    y[i] = v; y2 = y1; y1 = v;
  }
}

__global__ void __launch_bounds__(EIG_THREADS) k_extra_eig_solve(double* __restrict__ Lmat, const uint8_t* __restrict__ node_mask,
                                                                 const float* __restrict__ pad_noise, float* __restrict__ n_comp,
                                                                 float* __restrict__ ev5, float* __restrict__ nonlcc,
                                                                 float* __restrict__ vec2, double* __restrict__ evals_out, int n,
                                                                 const int* __restrict__ fixed_from, int* __restrict__ cache_state,
                                                                 double* __restrict__ cache, int mmax) {
  extern __shared__ double sm[];
  double* s_u = sm;                 // n   Householder vector of the current step
  double* s_p = sm + n;             // n
  double* s_d = sm + 2 * n;         // n   diagonal of T
  double* s_e = sm + 3 * n;         // n   sub-diagonal of T (s_e[i] couples i-1 and i)
  double* s_h = sm + 4 * n;         // n   u^T u / 2 of every reflector (0: skipped)
  double* s_y = sm + 5 * n;         // 3n  eigenvectors
  double* s_w = sm + 8 * n;         // 9n  work of the three tridiagonal solves (a, b, c each)
  double* s_red = sm + 17 * n;      // 32
  double* s_lam = s_red + 32;       // 8
  double* s_pp = s_lam + 8;         // EIG_THREADS  partial sums of A u per (row class, column): summed in a fixed order
  __shared__ int s_ncomp, s_nreal, s_best, s_coupled;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NWV = EIG_THREADS / 64;
  double* A = Lmat + (size_t)b * n * n;
  const uint8_t* m = node_mask + (size_t)b * n;

#ifdef EXP_EIG_STAMP
  const long long st0 = __builtin_readcyclecounter();
#endif
  // ---- 1. tridiagonalisation (rows k = n-1 .. 1; reflector u_k stored in A[k][0:k], h_k = u^T u / 2 in s_h).
  // Five barriers per step: the step is latency bound (one workgroup per matrix), so phases are merged wherever the data
  // allows: reductions through per-wave partials that every thread sums itself, p from per-row-class partials in LDS that the
  // column's own thread adds up in a fixed order (bit-reproducible: the sampled chains downstream flip draws on 1-ulp changes of
  // the eigen-features), q never materialised.
  double* s_part = s_red;          // [2][16] per-wave partials, double buffered
  int* s_iu = reinterpret_cast<int*>(s_w);             // [n] compacted index list
  int* s_wc = reinterpret_cast<int*>(s_w + n);         // [16] per-wave counts of the compaction
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
    for (int t = tid; t < mm; t += EIG_THREADS) { s_e[p0 + t] = cA[(size_t)mmax * mmax + t]; s_h[p0 + t] = cA[(size_t)(mmax + 1) * mmax + t]; }
    kstart = p0;                                        // row p0 itself has nothing before it: the loop sets e = 0, h = 0 for it
    __syncthreads();
  }
  for (int k = kstart; k >= 1; --k) {
    double part = 0.0;
    for (int j = tid; j < k; j += EIG_THREADS) { const double x = A[(size_t)k * n + j]; s_u[j] = x; s_p[j] = 0.0; if (j < k - 1) part += x * x; }
    part = wsum(part);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();                                                              // (1)
    double off2 = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) off2 += s_part[w];
    const double f = s_u[k - 1];
    if (off2 == 0.0) {                                // nothing to annihilate: the row is already tridiagonal
      if (tid == 0) { s_e[k] = f; s_h[k] = 0.0; }
      __syncthreads();
      continue;
    }
    const double hh = off2 + f * f;
    const double g = f >= 0.0 ? -sqrt(hh) : sqrt(hh);
    const double h = hh - f * g;                      // = u^T u / 2
    __syncthreads();                                                              // (2) everyone has read f
    if (tid == 0) { s_e[k] = g; s_h[k] = h; s_u[k - 1] = f - g; }
    __syncthreads();                                                              // (3)
    // The reflector lives inside the connected component of node k (a reaction graph is many components: the product, the
    // reactant fragments, isolated atoms), everything outside is exactly zero and is skipped: U = {j: u_j != 0} bounds the
    // rows of the product A u, T = {j: u_j != 0 or (A u)_j != 0} the rank-2 update.  Ordered compaction (deterministic sums).
    const int nU = compact(s_iu, s_wc, k, [&](int j) { return s_u[j] != 0.0; });
    if (cstate == 0 && k >= p0 && tid == 0 && s_iu[0] < p0) s_coupled = 1;   // the block reaches outside itself: not cacheable
    // h p = A u.  A is symmetric: (A u)_j = sum_{i in U} A[i][j] u_i, one thread per column j and row class, the rows walked
    // with 16 independent loads in flight; coalesced in j; partial sums meet in s_pp[row class][column]
    const int CW = (k + 63) & ~63;                      // columns rounded up to whole waves
    const int RC = EIG_THREADS / CW > 0 ? EIG_THREADS / CW : 1;
    for (int t = tid; t < RC * CW; t += EIG_THREADS) {
      const int j = t % CW, rc = t / CW;
      if (j >= k) continue;
      double acc[4] = {0, 0, 0, 0};
      int ii = rc;
      for (; ii + 15 * RC < nU; ii += 16 * RC) {
        double v[16]; int ix[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { ix[q] = s_iu[ii + q * RC]; v[q] = A[(size_t)ix[q] * n + j]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q & 3] = fma(v[q], s_u[ix[q]], acc[q & 3]);
      }
      for (; ii < nU; ii += RC) { const int i = s_iu[ii]; acc[0] = fma(A[(size_t)i * n + j], s_u[i], acc[0]); }
      const double tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      s_pp[rc * CW + j] = tot;
    }
    __syncthreads();                                                              // (4)
    double kp = 0.0;
    for (int j = tid; j < k; j += EIG_THREADS) {        // k <= EIG_THREADS: thread j owns column j here and in compact() below
      double pj = 0.0;
      for (int rc = 0; rc < RC; ++rc) pj += s_pp[rc * CW + j];
      s_p[j] = pj;
      kp = fma(s_u[j], pj, kp);
    }
    kp = wsum(kp);
    if (lane == 0) s_part[16 + wave] = kp;
    const int nT = compact(s_iu, s_wc, k, [&](int j) { return s_u[j] != 0.0 || s_p[j] != 0.0; });   // (barriers inside)
    double upu = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) upu += s_part[16 + w];
    const double invh = 1.0 / h;
    const double K = upu * invh * invh * 0.5;           // u^T p / (2h) with p = (A u) / h
    // A -= u q^T + q u^T on T x T, q = p - K u formed on the fly; row k keeps the reflector
    const int CT = (nT + 63) & ~63;
    const int RT = EIG_THREADS / CT > 0 ? EIG_THREADS / CT : 1;
    for (int t = tid; t < RT * CT; t += EIG_THREADS) {
      const int jj = t % CT, rc = t / CT;
      if (jj >= nT) continue;
      const int j = s_iu[jj];
      const double uj = s_u[j], qj = s_p[j] * invh - K * uj;
      int ii = rc;
      for (; ii + 15 * RT < nT; ii += 16 * RT) {
        double v[16]; int ix[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { ix[q] = s_iu[ii + q * RT]; v[q] = A[(size_t)ix[q] * n + j]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const double ui = s_u[ix[q]], qi = s_p[ix[q]] * invh - K * ui;
          A[(size_t)ix[q] * n + j] = v[q] - (ui * qj + qi * uj);
        }
      }
      for (; ii < nT; ii += RT) {
        const int i = s_iu[ii];
        const double ui = s_u[i], qi = s_p[i] * invh - K * ui;
        A[(size_t)i * n + j] -= ui * qj + qi * uj;
      }
    }
    for (int j = tid; j < k; j += EIG_THREADS) A[(size_t)k * n + j] = s_u[j];
    __syncthreads();                                                              // (6) stores of this step before the next row load
  }
#ifdef EXP_EIG_STAMP
  const long long st1 = __builtin_readcyclecounter();
#endif
  if (cstate == 0) {                                    // first call: keep what the block's steps left (nothing later touched it)
    const int mm = n - p0;
    if (!s_coupled) {
      for (int t = tid; t < mm * mm; t += EIG_THREADS) {
        const int rk = t / mm, cj = t - rk * mm;
        if (cj <= rk) cA[(size_t)rk * mmax + cj] = A[(size_t)(p0 + rk) * n + p0 + cj];
      }
      for (int t = tid; t < mm; t += EIG_THREADS) {
        cA[(size_t)mmax * mmax + t] = p0 + t >= 1 ? s_e[p0 + t] : 0.0; cA[(size_t)(mmax + 1) * mmax + t] = p0 + t >= 1 ? s_h[p0 + t] : 0.0;
      }
    }
    if (tid == 0) cache_state[b] = s_coupled ? -1 : 1;
  }
  for (int i = tid; i < n; i += EIG_THREADS) s_d[i] = A[(size_t)i * n + i];
  if (tid == 0) { s_e[0] = 0.0; s_h[0] = 0.0; s_nreal = 0; }
  __syncthreads();
  int c = 0;
  for (int i = tid; i < n; i += EIG_THREADS) c += m[i] ? 1 : 0;
  if (c) atomicAdd(&s_nreal, c);
  __syncthreads();
  const double nreal = (double)s_nreal;

  // ---- 2. the eigenvalues that are used: count below the threshold, then indices 0 and ncomp .. ncomp+4
  double gl = 0.0, gu = 0.0, emax = 0.0;               // Gershgorin bounds (every thread computes them: n is small)
  for (int i = 0; i < n; ++i) {
    const double r = fabs(s_e[i]) + (i + 1 < n ? fabs(s_e[i + 1]) : 0.0);
    gl = i == 0 ? s_d[i] - r : fmin(gl, s_d[i] - r);
    gu = i == 0 ? s_d[i] + r : fmax(gu, s_d[i] + r);
    emax = fmax(emax, fabs(s_e[i]));
  }
  const double tnorm = fmax(fabs(gl), fabs(gu));
  gl -= 2.2e-16 * tnorm * n + 1e-290; gu += 2.2e-16 * tnorm * n + 1e-290;
  const double pivmin = fmax(2.3e-308 * fmax(1.0, emax * emax), 1e-300);
  if (tid == 0) s_ncomp = sturm_count(s_d, s_e, n, 1e-5 * nreal, pivmin);          // eigenvalue / n_real < 1e-5  (:119, :166)
  __syncthreads();
  const int ncomp = s_ncomp;
  // six eigenvalues (indices 0, ncomp .. ncomp+4) by multisection: 42 threads per eigenvalue evaluate the Sturm count at
  // 42 interior points of the current bracket at once (5.4 bits per round instead of 1)
  {
    constexpr int PTS = 42;                             // 6 x 42 threads = 4 waves, one per SIMD: the count is fp64-VALU bound
    const int ev = tid / PTS, pt = tid - ev * PTS;
    int* s_lo = reinterpret_cast<int*>(s_w);            // [6] index of the last point with count <= idx, per round
    double* s_br = s_w + 8;                             // [6][2] brackets
    if (tid < 6) { s_br[2 * tid] = gl; s_br[2 * tid + 1] = gu; }
    __syncthreads();
    for (int round = 0; round < 11; ++round) {         // 43^11 = 9e17: the bracket shrinks to rounding level
      if (tid < 6) s_lo[tid] = -1;
      __syncthreads();
      if (ev < 6) {
        const int idx = ev == 0 ? 0 : ncomp + ev - 1;
        if (idx < n) {
          const double lo = s_br[2 * ev], hi = s_br[2 * ev + 1];
          const double x = lo + (hi - lo) * ((double)(pt + 1) / (double)(PTS + 1));
          if (sturm_count(s_d, s_e, n, x, pivmin) <= idx) atomicMax(&s_lo[ev], pt);   // eigenvalue idx is >= x
        }
      }
      __syncthreads();
      if (tid < 6) {
        const double lo = s_br[2 * tid], hi = s_br[2 * tid + 1];
        const int q = s_lo[tid];                           // bracket becomes [point q, point q+1]  (point -1 = lo, point PTS = hi)
        const double nlo = lo + (hi - lo) * ((double)(q + 1) / (double)(PTS + 1));
        const double nhi = lo + (hi - lo) * ((double)(q + 2) / (double)(PTS + 1));
        s_br[2 * tid] = q < 0 ? lo : nlo; s_br[2 * tid + 1] = q + 1 >= PTS ? hi : nhi;
      }
      __syncthreads();
    }
    if (tid < 6) {
      const int idx = tid == 0 ? 0 : ncomp + tid - 1;
      s_lam[tid] = idx < n ? 0.5 * (s_br[2 * tid] + s_br[2 * tid + 1]) : 2.0 * nreal;
    }
    __syncthreads();
  }
  if (tid == 0) n_comp[b] = (float)ncomp;
  if (tid < 5) ev5[(size_t)b * 5 + tid] = (ncomp + tid < n) ? (float)(s_lam[1 + tid] / nreal) : 2.f;   // :169-173
#ifndef EXP_EIG_STAMP
  if (evals_out && tid < 6) evals_out[(size_t)b * 6 + tid] = s_lam[tid];
#endif

#ifdef EXP_EIG_STAMP
  const long long st2 = __builtin_readcyclecounter();
#endif
  // ---- 3. eigenvectors 0, ncomp, ncomp+1 of T by inverse iteration (threads 0, 64, 128: one per vector)
  const double eps3 = 2.2e-16 * fmax(tnorm, 1e-300);
  for (int v = 0; v < 3; ++v) {
    const int idx = v == 0 ? 0 : ncomp + v - 1;
    double* y = s_y + v * n;
    for (int i = tid; i < n; i += EIG_THREADS) {
      // deterministic pseudo-random start (a vector of ones is orthogonal to many eigenvectors of a Laplacian)
      unsigned hsh = (unsigned)(i * 2654435761u) ^ (unsigned)((v + 1) * 40503u) ^ (unsigned)(b * 97u);
      hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
      y[i] = idx < n ? 0.5 + (double)(hsh & 0xffffu) / 65536.0 : 0.0;
    }
  }
  __syncthreads();
  if ((tid & 63) == 0 && wave < 3) {
    const int v = wave;
    const int idx = v == 0 ? 0 : ncomp + v - 1;
    if (idx < n) {
      double* y = s_y + v * n;
      double lam = s_lam[v == 0 ? 0 : v];
      // separate (nearly) equal eigenvalues slightly, as LAPACK's stein does, so that the two solves do not return one vector
      if (v == 2 && fabs(lam - s_lam[1]) <= 10.0 * eps3) lam = s_lam[1] + 10.0 * eps3;
      lam += (v + 1) * eps3;                           // never exactly singular
      for (int it = 0; it < 3; ++it) {
        tri_solve(s_d, s_e, n, lam, y, s_w + v * 3 * n, s_w + v * 3 * n + n, s_w + v * 3 * n + 2 * n, eps3);
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm = fmax(nrm, fabs(y[i]));
        if (!(nrm > 0.0)) break;
        const double sc = 1.0 / nrm;
        for (int i = 0; i < n; ++i) y[i] *= sc;
      }
    }
  }
  __syncthreads();
  if (tid == 0 && ncomp + 1 < n) {                     // the two vectors after the zero ones: orthonormal pair (degenerate pairs)
    double* y1 = s_y + n; double* y2 = s_y + 2 * n;
    double d11 = 0.0, d12 = 0.0;
    for (int i = 0; i < n; ++i) { d11 += y1[i] * y1[i]; d12 += y1[i] * y2[i]; }
    if (fabs(s_lam[2] - s_lam[1]) <= 1e-8 * fmax(1.0, tnorm) && d11 > 0.0) for (int i = 0; i < n; ++i) y2[i] -= d12 / d11 * y1[i];
  }
  __syncthreads();
#ifdef EXP_EIG_STAMP
  const long long st3 = __builtin_readcyclecounter();
#endif
  // back-transformation x = H_{n-1} ... H_1 y: reflector k acts on entries 0..k-1; one wave per vector
  if (wave < 3) {
    double* y = s_y + wave * n;
    for (int k = 1; k < n; ++k) {
      const double h = s_h[k];
      if (h == 0.0) continue;
      const double* u = A + (size_t)k * n;
      double dot = 0.0;
      for (int j = lane; j < k; j += 64) dot = fma(u[j], y[j], dot);
      dot = wsum(dot) / h;
      for (int j = lane; j < k; j += 64) y[j] -= dot * u[j];
    }
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
  double* s_first = s_p;
  int* s_cnt = reinterpret_cast<int*>(s_u);
  const double m0 = m[0] ? 1.0 : 0.0;
  for (int i = tid; i < n; i += EIG_THREADS) {
    const double mi = m[i] ? 1.0 : 0.0;
    const double v = s_y[i] * mi * m0;
    s_first[i] = nearbyint(v * 1000.0) / 1000.0 * mi + (double)pad_noise[(size_t)b * n + i] * (1.0 - mi);
  }
  __syncthreads();
  for (int i = tid; i < n; i += EIG_THREADS) {
    const double v = s_first[i];
    int k = 0;
    for (int j = 0; j < n; ++j) k += (s_first[j] == v) ? 1 : 0;
    s_cnt[i] = k;
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    for (int i = 1; i < n; ++i)
      if (s_cnt[i] > s_cnt[best] || (s_cnt[i] == s_cnt[best] && s_first[i] < s_first[best])) best = i;
    s_best = best;
  }
  __syncthreads();
  const double mode = s_first[s_best];
  for (int i = tid; i < n; i += EIG_THREADS) {
    const bool mi = m[i] != 0;
    nonlcc[(size_t)b * n + i] = (mi && s_first[i] != mode) ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int col = ncomp + k;
      double v = 0.0;
      if (col < n && mi && m[col]) v = s_y[(1 + k) * n + i];
      vec2[((size_t)b * n + i) * 2 + k] = (float)v;
    }
  }
}

extern "C" {

/* see include/diffalign_hip.h */
int da_extra_eigen_solve_cached(double* L, const uint8_t* node_mask, const float* pad_noise, float* n_comp, float* ev5, float* nonlcc,
                                float* vec2, double* evals6 /*may be NULL*/, int B, int n, const int* fixed_from, int* cache_state,
                                double* cache, int mmax, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 1 && n <= 1024, "da_extra_eigen_solve: n = %d outside 1..1024 (LDS work arrays)", n);
  DA_CHECK_ARG(cache == nullptr || (fixed_from != nullptr && cache_state != nullptr && mmax >= 1 && mmax <= n),
               "da_extra_eigen_solve_cached: cache given without fixed_from / cache_state, or mmax = %d outside 1..n", mmax);
  if (B == 0) return DA_OK;
  const size_t lds = ((size_t)17 * n + 32 + 8 + EIG_THREADS) * sizeof(double);
  static std::atomic<bool> attr_set[DA_MAX_DEVICES];     // the attribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DA_MAX_DEVICES) { da_set_error("da_extra_eigen_solve: no current HIP device"); return DA_ERR_HIP; }
  if (!attr_set[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_extra_eig_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) { da_set_error("da_extra_eigen_solve: cannot reserve LDS: %s", hipGetErrorString(e)); return DA_ERR_HIP; }
    attr_set[dev].store(true, std::memory_order_release);
  }
  DA_CHECK_ARG(lds <= 160 * 1024 - 64, "da_extra_eigen_solve: n = %d needs %zu B of LDS", n, lds);
  k_extra_eig_solve<<<B, EIG_THREADS, lds, (hipStream_t)stream>>>(L, node_mask, pad_noise, n_comp, ev5, nonlcc, vec2, evals6, n,
                                                                  fixed_from, cache_state, cache, mmax);
  DA_CHECK_LAUNCH("da_extra_eigen_solve");
  return DA_OK;
}

int da_extra_eigen_solve(double* L, const uint8_t* node_mask, const float* pad_noise, float* n_comp, float* ev5, float* nonlcc,
                         float* vec2, double* evals6 /*may be NULL*/, int B, int n, void* stream) {
  return da_extra_eigen_solve_cached(L, node_mask, pad_noise, n_comp, ev5, nonlcc, vec2, evals6, B, n, nullptr, nullptr, nullptr, 0, stream);
}

}  // extern "C"
