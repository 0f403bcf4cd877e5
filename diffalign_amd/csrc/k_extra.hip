// Extra input features of the denoiser, computed on the device.
//
// Replaces neuralnet/extra_features.py:22-298 (cycle counts, Laplacian eigen-features) and
// neuralnet/extra_features_molecular.py:6-72 (charge, valency, weight), which the reference evaluates on the CPU at
// every step after moving E, X and node_mask to the host (diffusion/diffusion_abstract.py:350-372).
//
// Division of work: this file holds the passes over the (B,n,n,ke) edge state and the small per-graph reductions; the
// two adjacency powers A^2, A^3 are plain library GEMMs (torch.bmm) and the eigen-decomposition is k_extra_eig.hip (or,
// optionally, torch.linalg.eigh on the device followed by k_extra_eigen below) -- see diffalign_amd/extra_features.py.
// All kernels here are HBM/latency bound: E is read once (28 B per ordered pair), the three n x n fp32 matrices once.
//
// The cycle formulas use A = A^T (the diffusion state is symmetric by construction, helpers.py:743-745):
//   diag(A^4)_i = sum_j (A^2)_ij^2,  diag(A^5)_i = sum_j (A^2)_ij (A^3)_ij,  tr(A^6) = sum_ij (A^3)_ij^2,
// so that only A^2 and A^3 are formed (the reference forms A^2..A^6, extra_features.py:234-242).
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- pass over the edge state: adjacency, Laplacian, valency sums ---------------------------------------------------
// one wave per row (b,i).  A_ij = sum_{c>=1} E_ijc (extra_features.py:81), the diagonal of L = sym(D - A_m) on real nodes and 2n
// on padded ones (:105-109, :133-146), valency sums of extra_features_molecular.py:40-42 and :57-59.  The off-diagonal part
// of L needs the transposed entry (the reference symmetrises L): k_extra_lap takes it from A through LDS tiles -- read here
// from E it was one cache line per lane and pair, 200 us of this pass's 250 at B = 64, n = 300.
// KE > 0: the class count as a compile-time constant (every load of a pair issued up front, two pairs per lane in flight: with
// the run-time bound each class sat behind its own `c < ke` branch and was its own round trip); KE = 0: any ke <= 16.
template <int KE>
__global__ void __launch_bounds__(256) k_extra_adj(const float* __restrict__ E, const uint8_t* __restrict__ node_mask,
                                                   const float* __restrict__ bond_orders, float* __restrict__ A,
                                                   double* __restrict__ L, float* __restrict__ val_weighted,
                                                   float* __restrict__ val_plain, int B, int n, int ke_rt) {
  constexpr int KMAX = KE > 0 ? KE : 16;
  const int ke = KE > 0 ? KE : ke_rt;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * n) return;
  const int b = row / n, i = row - b * n;
  const bool mi = node_mask[row] != 0;
  float ord[KMAX];
#pragma unroll
  for (int c = 0; c < KMAX; ++c) ord[c] = bond_orders[min(c, ke - 1)];
  const float* Er = E + (size_t)row * n * ke;
  float vw = 0.f, vp = 0.f;
  double deg = 0.0;
  auto pair = [&](int j, const float (&v)[KMAX], uint8_t mjb) {      // v: the pair's classes (entries past ke unused)
    float a = 0.f, bw = -INFINITY, bp = -INFINITY;
    int iw = 0, ip = 0;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) {
      if (c < ke) {
        if (c >= 1) a += v[c];
        const float w = v[c] * ord[c];
        if (w > bw) { bw = w; iw = c; }       // first maximum, as torch.argmax
        if (v[c] > bp) { bp = v[c]; ip = c; }
      }
    }
    A[(size_t)row * n + j] = a;
    vw += (float)iw; vp += (float)ip;
    deg += (mi && mjb != 0) ? (double)a : 0.0;
  };
  auto load = [&](int j, float (&v)[KMAX], uint8_t& mjb) {            // unconditional, from clamped addresses
    const float* e = Er + (size_t)min(j, n - 1) * ke;
#pragma unroll
    for (int c = 0; c < KMAX; ++c) v[c] = e[KE > 0 ? c : min(c, ke - 1)];
    mjb = node_mask[b * n + min(j, n - 1)];
  };
  for (int j = lane; j < n; j += 128) {
    float v0[KMAX], v1[KMAX];
    uint8_t m0, m1;
    load(j, v0, m0); load(j + 64, v1, m1);
    pair(j, v0, m0);
    if (j + 64 < n) pair(j + 64, v1, m1);
  }
  vw = wave_sum(vw); vp = wave_sum(vp);
  if (lane == 0) { val_weighted[row] = vw; val_plain[row] = vp; }
  if (L) {
    deg = wave_sum(deg);
    if (lane == 0) {
      // (D - A)_ii on real nodes (the masked self entry is part of the row sum and of A), 2n on padded ones
      const float* e = Er + (size_t)i * ke;
      float aii = 0.f;
      for (int c = 1; c < ke; ++c) aii += e[c];
      L[(size_t)row * n + i] = mi ? deg - (double)aii : 2.0 * (double)n;   // (aii is 0 in every diffusion state: no self loops)
    }
  }
}

// off-diagonal entries of the Laplacian: L_ij = -(A_m[i][j] + A_m[j][i]) / 2 with A_m = A on pairs of real nodes, else 0.  One
// workgroup per 32 x 32 tile; the mirrored tile of A comes through LDS so that both reads and the write are row-wise.
__global__ void __launch_bounds__(256) k_extra_lap(const float* __restrict__ A, const uint8_t* __restrict__ node_mask,
                                                   double* __restrict__ L, int n) {
  __shared__ float s0[32][33], s1[32][33];
  const int nt = (n + 31) / 32, b = blockIdx.y, it = blockIdx.x / nt, jt = blockIdx.x - it * nt;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  const float* Ab = A + (size_t)b * n * n;
  float v0[4], v1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                          // unconditional loads from clamped addresses; entries outside are never stored
    const int rr = r0 + 8 * q;
    v0[q] = Ab[(size_t)min(32 * it + rr, n - 1) * n + min(32 * jt + c, n - 1)];
    v1[q] = Ab[(size_t)min(32 * jt + rr, n - 1) * n + min(32 * it + c, n - 1)];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { s0[r0 + 8 * q][c] = v0[q]; s1[r0 + 8 * q][c] = v1[q]; }
  __syncthreads();
  const int j = 32 * jt + c;
  const bool mj = j < n && node_mask[b * n + min(j, n - 1)] != 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = r0 + 8 * q, i = 32 * it + rr;
    if (i < n && j < n && i != j) {
      const bool m = mj && node_mask[b * n + i] != 0;
      const double am = m ? (double)s0[rr][c] : 0.0, atm = m ? (double)s1[c][rr] : 0.0;
      L[((size_t)b * n + i) * n + j] = -0.5 * (am + atm);
    }
  }
}

// ---- cycle counts (extra_features.py:227-298 KNodeCycles, :75-90 NodeCycleFeatures) ----------------------------------
// one workgroup per graph; phase 1: per-row sums over A, k2 = A^2, k3 = A^3 into LDS; phase 2: A d, A tri, the per-node
// counts and the graph totals.
__global__ void __launch_bounds__(1024) k_extra_cycles(const float* __restrict__ A, const float* __restrict__ K2,
                                                       const float* __restrict__ K3, const uint8_t* __restrict__ node_mask,
                                                       float* __restrict__ x_cycles, float* __restrict__ y_cycles, int n) {
  extern __shared__ float sm[];
  float* s_d = sm;            // degree
  float* s_k2 = sm + n;       // (A^2)_ii
  float* s_tri = sm + 2 * n;  // (A^3)_ii
  float* s_a4 = sm + 3 * n;   // (A^4)_ii
  float* s_a5 = sm + 4 * n;   // (A^5)_ii
  float* s_red = sm + 5 * n;  // 16 waves x 8 partial graph sums
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NWV = blockDim.x >> 6;
  const size_t base = (size_t)b * n * n;
  float g_s3 = 0.f, g_t3 = 0.f, g_tr6 = 0.f;          // sum k3, sum A*k2^2, sum k3^2   (lane-partial, this wave's rows)
  for (int i = wave; i < n; i += NWV) {
    const float* a = A + base + (size_t)i * n;
    const float* p = K2 + base + (size_t)i * n;
    const float* q = K3 + base + (size_t)i * n;
    float d = 0.f, a4 = 0.f, a5 = 0.f;
    for (int j = lane; j < n; j += 64) {
      const float av = a[j], pv = p[j], qv = q[j];
      d += av; a4 = fmaf(pv, pv, a4); a5 = fmaf(pv, qv, a5);
      g_s3 += qv; g_t3 = fmaf(av * pv, pv, g_t3); g_tr6 = fmaf(qv, qv, g_tr6);
    }
    d = wave_sum(d); a4 = wave_sum(a4); a5 = wave_sum(a5);
    if (lane == 0) { s_d[i] = d; s_k2[i] = p[i]; s_tri[i] = q[i]; s_a4[i] = a4; s_a5[i] = a5; }
  }
  __syncthreads();
  // phase 2
  float g_c3 = 0.f, g_c4 = 0.f, g_c5 = 0.f, g_t2 = 0.f, g_t4 = 0.f, g_t5 = 0.f, g_t7 = 0.f, g_t9 = 0.f, g_t10 = 0.f;   // lane 0 only
  for (int i = wave; i < n; i += NWV) {
    const float* a = A + base + (size_t)i * n;
    float ad = 0.f, at = 0.f;
    for (int j = lane; j < n; j += 64) { const float av = a[j]; ad = fmaf(av, s_d[j], ad); at = fmaf(av, s_tri[j], at); }
    ad = wave_sum(ad); at = wave_sum(at);
    if (lane == 0) {
      const float d = s_d[i], tri = s_tri[i], k2 = s_k2[i], a4 = s_a4[i], a5 = s_a5[i];
      const float c3 = tri;                                             // :244-247
      const float c4 = a4 - d * (d - 1.f) - ad;                         // :249-252
      const float c5 = a5 - 2.f * tri * d - at + tri;                   // :254-259
      const float m = node_mask[b * n + i] ? 1.f : 0.f;
      float* xo = x_cycles + ((size_t)b * n + i) * 3;
      const float x3 = (c3 / 2.f) * m / 10.f, x4 = (c4 / 2.f) * m / 10.f, x5 = (c5 / 2.f) * m / 10.f;   // :84-88
      xo[0] = x3 > 1.f ? 1.f : x3; xo[1] = x4 > 1.f ? 1.f : x4; xo[2] = x5 > 1.f ? 1.f : x5;
      g_c3 += c3; g_c4 += c4; g_c5 += c5;
      g_t2 = fmaf(tri, tri, g_t2); g_t4 = fmaf(k2, a4, g_t4); g_t5 += a4; g_t7 = fmaf(k2 * k2, k2, g_t7); g_t9 = fmaf(k2, k2, g_t9); g_t10 += k2;
    }
  }
  g_s3 = wave_sum(g_s3); g_t3 = wave_sum(g_t3); g_tr6 = wave_sum(g_tr6);
  if (lane == 0) {
    float* r = s_red + wave * 12;
    r[0] = g_c3; r[1] = g_c4; r[2] = g_c5; r[3] = g_tr6; r[4] = g_t2; r[5] = g_t3; r[6] = g_t4; r[7] = g_t5; r[8] = g_s3; r[9] = g_t7; r[10] = g_t9; r[11] = g_t10;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[12];
    for (int q = 0; q < 12; ++q) { float s = 0.f; for (int w = 0; w < NWV; ++w) s += s_red[w * 12 + q]; t[q] = s; }
    const float k3y = t[0] / 6.f, k4y = t[1] / 8.f, k5y = t[2] / 10.f;
    // :261-277: term1 tr A^6, term2 tr (A^3 o A^3), term3 sum A o (A^2)^2, term4 sum diag(A^2) diag(A^4), term5 tr A^4, term6 tr A^3,
    //           term7 sum diag(A^2)^3, term8 sum A^3, term9 sum diag(A^2)^2, term10 tr A^2
    const float c6 = t[3] - 3.f * t[4] + 9.f * t[5] - 6.f * t[6] + 6.f * t[7] - 4.f * t[0] + 4.f * t[9] + 3.f * t[8] - 12.f * t[10] + 4.f * t[11];
    const float k6y = c6 / 12.f;
    float* yo = y_cycles + (size_t)b * 4;
    const float v[4] = {k3y / 10.f, k4y / 10.f, k5y / 10.f, k6y / 10.f};
    for (int q = 0; q < 4; ++q) yo[q] = v[q] > 1.f ? 1.f : v[q];
  }
}

// ---- eigen-features from the eigendecomposition (extra_features.py:117-128, :158-204) --------------------------------
// one workgroup per graph.  evals (B,n) ascending, evecs (B,n,n) with element (i,k) at i*ld_row + k*ld_col.
__global__ void __launch_bounds__(256) k_extra_eigen(const double* __restrict__ evals, const double* __restrict__ evecs,
                                                     long long ld_row, long long ld_col, const uint8_t* __restrict__ node_mask,
                                                     const float* __restrict__ pad_noise, float* __restrict__ n_comp,
                                                     float* __restrict__ ev5, float* __restrict__ nonlcc,
                                                     float* __restrict__ vec2, int n) {
  extern __shared__ double sd[];
  double* s_first = sd;                                   // n
  int* s_cnt = reinterpret_cast<int*>(sd + n);            // n
  __shared__ int s_nreal, s_ncomp, s_best;
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint8_t* m = node_mask + (size_t)b * n;
  const double* ev = evals + (size_t)b * n;
  const double* V = evecs + (size_t)b * n * n;
  if (tid == 0) { s_nreal = 0; s_ncomp = 0; }
  __syncthreads();
  int c = 0;
  for (int i = tid; i < n; i += blockDim.x) c += m[i] ? 1 : 0;
  atomicAdd(&s_nreal, c);
  __syncthreads();
  const double nreal = (double)s_nreal;
  c = 0;
  for (int k = tid; k < n; k += blockDim.x) c += (ev[k] / nreal < 1e-5) ? 1 : 0;            // :119, :166
  atomicAdd(&s_ncomp, c);
  __syncthreads();
  const int ncomp = s_ncomp;
  if (tid == 0) n_comp[b] = (float)ncomp;
  if (tid < 5) ev5[(size_t)b * 5 + tid] = (ncomp + tid < n) ? (float)(ev[ncomp + tid] / nreal) : 2.f;   // :169-173
  // first eigenvector, rounded to 3 decimals, random values on the padded slots (:187-190); the column mask of :120
  const double m0 = m[0] ? 1.0 : 0.0;
  for (int i = tid; i < n; i += blockDim.x) {
    const double mi = m[i] ? 1.0 : 0.0;
    const double v = V[(size_t)i * ld_row] * mi * m0;
    s_first[i] = nearbyint(v * 1000.0) / 1000.0 * mi + (double)pad_noise[(size_t)b * n + i] * (1.0 - mi);
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const double v = s_first[i];
    int k = 0;
    for (int j = 0; j < n; ++j) k += (s_first[j] == v) ? 1 : 0;
    s_cnt[i] = k;
  }
  __syncthreads();
  if (tid == 0) {                       // mode: highest count, smallest value among ties (torch.mode)
    int best = 0;
    for (int i = 1; i < n; ++i)
      if (s_cnt[i] > s_cnt[best] || (s_cnt[i] == s_cnt[best] && s_first[i] < s_first[best])) best = i;
    s_best = best;
  }
  __syncthreads();
  const double mode = s_first[s_best];
  for (int i = tid; i < n; i += blockDim.x) {
    const bool mi = m[i] != 0;
    nonlcc[(size_t)b * n + i] = (mi && s_first[i] != mode) ? 1.f : 0.f;                       // :192-193
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                                            // :196-202
      const int col = ncomp + k;
      double v = 0.0;
      if (col < n && mi && m[col]) v = V[(size_t)i * ld_row + (size_t)col * ld_col];
      vec2[((size_t)b * n + i) * 2 + k] = (float)v;
    }
  }
}

// ---- molecular features (extra_features_molecular.py:29-72) ----------------------------------------------------------
__global__ void __launch_bounds__(256) k_extra_molecular(const float* __restrict__ X, const float* __restrict__ valencies,
                                                         const float* __restrict__ atom_weights,
                                                         const float* __restrict__ val_weighted, const float* __restrict__ val_plain,
                                                         float max_weight, float* __restrict__ charge_val, float* __restrict__ weight,
                                                         int n, int kx) {
  __shared__ float s_w[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float wsum = 0.f;
  for (int i = tid; i < n; i += blockDim.x) {
    const float* x = X + ((size_t)b * n + i) * kx;
    float bx = -INFINITY, bv = -INFINITY;
    int ix = 0, iv = 0;
    for (int c = 0; c < kx; ++c) {
      const float v = x[c], w = v * valencies[c];
      if (v > bx) { bx = v; ix = c; }
      if (w > bv) { bv = w; iv = c; }
    }
    charge_val[((size_t)b * n + i) * 2 + 0] = (float)iv - val_weighted[(size_t)b * n + i];   // :46-48 (argmax index, as the reference)
    charge_val[((size_t)b * n + i) * 2 + 1] = val_plain[(size_t)b * n + i];                  // :59-60
    wsum += atom_weights[ix];                                                                // :70-72 (padded rows: class 0)
  }
  wsum = wave_sum(wsum);
  if ((tid & 63) == 0) s_w[tid >> 6] = wsum;
  __syncthreads();
  if (tid == 0) weight[b] = (s_w[0] + s_w[1] + s_w[2] + s_w[3]) / max_weight;
}

extern "C" {

int da_extra_adjacency(const float* E, const uint8_t* node_mask, const float* bond_orders, float* A, double* L,
                       float* val_weighted, float* val_plain, int B, int n, int ke, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && ke >= 1 && ke <= 16, "da_extra_adjacency: need 1 <= ke <= 16 (got %d)", ke);
  if (B == 0 || n == 0) return DA_OK;
  const dim3 grid(cdiv(B * n, 4));
  hipStream_t s = (hipStream_t)stream;
  switch (ke) {
    case 5: k_extra_adj<5><<<grid, 256, 0, s>>>(E, node_mask, bond_orders, A, L, val_weighted, val_plain, B, n, ke); break;
    case 7: k_extra_adj<7><<<grid, 256, 0, s>>>(E, node_mask, bond_orders, A, L, val_weighted, val_plain, B, n, ke); break;
    case 4: k_extra_adj<4><<<grid, 256, 0, s>>>(E, node_mask, bond_orders, A, L, val_weighted, val_plain, B, n, ke); break;
    default: k_extra_adj<0><<<grid, 256, 0, s>>>(E, node_mask, bond_orders, A, L, val_weighted, val_plain, B, n, ke); break;
  }
  if (L) {
    const int nt = (n + 31) / 32;
    k_extra_lap<<<dim3(nt * nt, B), 256, 0, (hipStream_t)stream>>>(A, node_mask, L, n);
  }
  DA_CHECK_LAUNCH("da_extra_adjacency");
  return DA_OK;
}

int da_extra_cycles(const float* A, const float* K2, const float* K3, const uint8_t* node_mask, float* x_cycles,
                    float* y_cycles, int B, int n, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && n <= 4096, "da_extra_cycles: n = %d exceeds the LDS row buffers (4096)", n);
  if (B == 0 || n == 0) return DA_OK;
  k_extra_cycles<<<B, 1024, (5 * n + 16 * 12) * sizeof(float), (hipStream_t)stream>>>(A, K2, K3, node_mask, x_cycles, y_cycles, n);
  DA_CHECK_LAUNCH("da_extra_cycles");
  return DA_OK;
}

int da_extra_eigen(const double* evals, const double* evecs, long long ld_row, long long ld_col, const uint8_t* node_mask,
                   const float* pad_noise, float* n_comp, float* ev5, float* nonlcc, float* vec2, int B, int n, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && n <= 4096, "da_extra_eigen: n = %d exceeds the LDS buffers (4096)", n);
  if (B == 0 || n == 0) return DA_OK;
  k_extra_eigen<<<B, 256, n * (sizeof(double) + sizeof(int)), (hipStream_t)stream>>>(evals, evecs, ld_row, ld_col, node_mask, pad_noise,
                                                                                   n_comp, ev5, nonlcc, vec2, n);
  DA_CHECK_LAUNCH("da_extra_eigen");
  return DA_OK;
}

int da_extra_molecular(const float* X, const float* valencies, const float* atom_weights, const float* val_weighted,
                       const float* val_plain, float max_weight, float* charge_val, float* weight, int B, int n, int kx,
                       void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && kx >= 1, "da_extra_molecular: bad sizes");
  if (B == 0) return DA_OK;
  k_extra_molecular<<<B, 256, 0, (hipStream_t)stream>>>(X, valencies, atom_weights, val_weighted, val_plain, max_weight, charge_val, weight, n, kx);
  DA_CHECK_LAUNCH("da_extra_molecular");
  return DA_OK;
}

}  // extern "C"
