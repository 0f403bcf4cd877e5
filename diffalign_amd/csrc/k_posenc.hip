// Positional encodings shared between mapped reactant / product atoms, on the device.
//
// Replaces the per-graph Python loops of neuralnet/transformer_model_with_y.py:213-503 (PositionalEmbedding), which move
// atom_map_numbers and mol_assignments to the CPU: product index per node (:400-412, :259-275), the sinusoidal table
// (:114-131), the gather of per-atom-map rows (:492-503) and the Laplacian of the product block (:133-171, :173-211, :240-256).
// The eigendecomposition between da_pos_laplacian and da_pos_lap_gather is torch.linalg.eigh on the device (fp32, as
// laplacian_eigenvectors_gpu :197-201).  All kernels are tiny and latency bound (once per batch, not per step).
#include "common.h"

// position[b,i] = 1-based rank (in node order) of the product node that carries atom-map number am[b,i]; 0 if none.
// The product is the molecule with the largest assignment index (:241, :400).  Duplicate numbers: the last product node wins,
// as the sequential scatter_ of :267 / :410 does; prod_node[b,k] = node index of the k-th product node (k >= 1).
__global__ void __launch_bounds__(256) k_pos_product_index(const int64_t* __restrict__ am, const int64_t* __restrict__ mol,
                                                           int* __restrict__ position, int* __restrict__ prod_node,
                                                           int* __restrict__ n_prod, int n) {
  extern __shared__ int s_map[];          // [n + 1] atom-map number -> rank   (numbers > n are rejected on the host)
  __shared__ int s_pmax;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* a = am + (size_t)b * n;
  const int64_t* m = mol + (size_t)b * n;
  if (tid == 0) s_pmax = 0;
  for (int i = tid; i <= n; i += blockDim.x) s_map[i] = 0;
  __syncthreads();
  int mx = 0;
  for (int i = tid; i < n; i += blockDim.x) mx = max(mx, (int)m[i]);
  atomicMax(&s_pmax, mx);
  __syncthreads();
  const int pidx = s_pmax;
  // rank of node i among the product nodes = 1 + #product nodes before it
  for (int i = tid; i < n; i += blockDim.x) {
    if ((int)m[i] != pidx) continue;
    int rank = 1;
    for (int j = 0; j < i; ++j) rank += ((int)m[j] == pidx) ? 1 : 0;
    prod_node[(size_t)b * (n + 1) + rank] = i;
    const int64_t num = a[i];
    if (num >= 0 && num <= n) atomicMax(&s_map[num], rank);
  }
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < n; i += blockDim.x) {
    const int64_t num = a[i];
    position[(size_t)b * n + i] = (num >= 0 && num <= n) ? s_map[num] : 0;
    cnt += ((int)m[i] == pidx) ? 1 : 0;
  }
  atomicAdd(&n_prod[b], cnt);
}

// SinusoidalPosEmb (:114-131) of value[b,i] * 1000: out[..., :half] = sin(x * f_j), out[..., half:] = cos(x * f_j),
// f_j = exp(-j * ln(10000) / (half - 1)) in fp32 (freq is computed on the host exactly as torch does)
__global__ void k_pos_sinusoidal(const float* __restrict__ value, const float* __restrict__ freq, float* __restrict__ out,
                                 long long rows, int dim) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (t >= rows * half) return;
  const long long r = t / half;
  const int j = (int)(t - r * half);
  const float x = value[r] * 1000.f;
  const float e = x * freq[j];
  out[r * dim + j] = sinf(e);
  out[r * dim + half + j] = cosf(e);
}

// out[b,i,:] = table[b, idx[b,i], :]  (per-graph row tables; the Gaussian variant :492-503 with idx = atom-map number)
__global__ void k_pos_gather_rows(const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out,
                                  int n, int rows_per_graph, int dim) {
  const int b = blockIdx.y, i = blockIdx.x;
  const int64_t k = idx[(size_t)b * n + i];
  const float* src = table + ((size_t)b * rows_per_graph + (size_t)k) * dim;
  float* dst = out + ((size_t)b * n + i) * dim;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) dst[c] = src[c];
}

// symmetric-normalised Laplacian of the product block inside the n x n frame (:147-154, :192-198: unweighted edges where the
// bond class is non-zero and both ends are product nodes; L = I - D^-1/2 A D^-1/2, isolated nodes keep L_ii = 1), fp32.
// One wave per row.
__global__ void __launch_bounds__(256) k_pos_laplacian(const int64_t* __restrict__ Ecls, const int64_t* __restrict__ mol,
                                                       float* __restrict__ Lout, int B, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * n) return;
  const int b = row / n, i = row - b * n;
  const int64_t* m = mol + (size_t)b * n;
  int pidx = 0;
  for (int j = lane; j < n; j += 64) pidx = max(pidx, (int)m[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pidx = max(pidx, __shfl_xor(pidx, off));
  const bool pi = (int)m[i] == pidx;
  // degrees: deg_i from this row; deg_j needs row j: recomputed per lane below (n small, once per batch)
  const int64_t* Ei = Ecls + (size_t)row * n;
  int deg_i = 0;
  for (int j = lane; j < n; j += 64) deg_i += (pi && (int)m[j] == pidx && j != i && Ei[j] != 0) ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) deg_i += __shfl_xor(deg_i, off);
  const float di = deg_i > 0 ? 1.f / sqrtf((float)deg_i) : 0.f;
  for (int j = lane; j < n; j += 64) {
    float v = 0.f;
    if (j == i) v = 1.f;
    else if (pi && (int)m[j] == pidx && Ei[j] != 0) {
      const int64_t* Ej = Ecls + ((size_t)b * n + j) * n;
      int deg_j = 0;
      for (int q = 0; q < n; ++q) deg_j += ((int)m[q] == pidx && q != j && Ej[q] != 0) ? 1 : 0;
      const float dj = deg_j > 0 ? 1.f / sqrtf((float)deg_j) : 0.f;
      v = -di * dj;
    }
    Lout[(size_t)row * n + j] = v;
  }
}

// out[b,i,c] = sign[b,c] * evecs[b, prod_node[b, position[b,i]], 1 + c]  for c < k_b = min(max_eig, n_prod[b]) and
// position > 0, else 0 (:254-259, :275): the k lowest non-trivial eigenvectors, rows of the product shared through the
// atom-map numbers, zero padded to dim.
__global__ void k_pos_lap_gather(const float* __restrict__ evecs, long long ld_row, long long ld_col, const int* __restrict__ position,
                                 const int* __restrict__ prod_node, const int* __restrict__ n_prod, const float* __restrict__ sign,
                                 float* __restrict__ out, int n, int dim, int max_eig) {
  const int b = blockIdx.y, i = blockIdx.x;
  const int pos = position[(size_t)b * n + i];
  const int k = min(max_eig, n_prod[b]);
  const int node = pos > 0 ? prod_node[(size_t)b * (n + 1) + pos] : 0;
  const float* V = evecs + (size_t)b * n * n;
  float* dst = out + ((size_t)b * n + i) * dim;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) {
    float v = 0.f;
    if (pos > 0 && c < k && 1 + c < n) v = sign[(size_t)b * max_eig + c] * V[(size_t)node * ld_row + (size_t)(1 + c) * ld_col];
    dst[c] = v;
  }
}

extern "C" {

int da_pos_product_index(const int64_t* atom_map_numbers, const int64_t* mol_assignments, int* position, int* prod_node,
                         int* n_prod, int B, int n, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && n <= 16000, "da_pos_product_index: n = %d exceeds the LDS table", n);
  if (B == 0 || n == 0) return DA_OK;
  hipError_t e = hipMemsetAsync(n_prod, 0, sizeof(int) * B, (hipStream_t)stream);
  if (e == hipSuccess) e = hipMemsetAsync(prod_node, 0, sizeof(int) * (size_t)B * (n + 1), (hipStream_t)stream);
  if (e != hipSuccess) { da_set_error("da_pos_product_index: memset failed: %s", hipGetErrorString(e)); return DA_ERR_HIP; }
  k_pos_product_index<<<B, 256, (n + 1) * sizeof(int), (hipStream_t)stream>>>(atom_map_numbers, mol_assignments, position, prod_node, n_prod, n);
  DA_CHECK_LAUNCH("da_pos_product_index");
  return DA_OK;
}

int da_pos_sinusoidal(const float* value, const float* freq, float* out, long long rows, int dim, void* stream) {
  DA_CHECK_ARG(rows >= 0 && dim >= 2 && dim % 2 == 0, "da_pos_sinusoidal: dim must be even");
  if (rows == 0) return DA_OK;
  const long long tot = rows * (dim / 2);
  k_pos_sinusoidal<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(value, freq, out, rows, dim);
  DA_CHECK_LAUNCH("da_pos_sinusoidal");
  return DA_OK;
}

int da_pos_gather_rows(const float* table, const int64_t* idx, float* out, int B, int n, int rows_per_graph, int dim, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && rows_per_graph >= 1 && dim >= 1, "da_pos_gather_rows: bad sizes");
  if (B == 0 || n == 0) return DA_OK;
  k_pos_gather_rows<<<dim3(n, B), 64, 0, (hipStream_t)stream>>>(table, idx, out, n, rows_per_graph, dim);
  DA_CHECK_LAUNCH("da_pos_gather_rows");
  return DA_OK;
}

int da_pos_laplacian(const int64_t* E_classes, const int64_t* mol_assignments, float* L, int B, int n, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0, "da_pos_laplacian: negative size");
  if (B == 0 || n == 0) return DA_OK;
  k_pos_laplacian<<<cdiv(B * n, 4), 256, 0, (hipStream_t)stream>>>(E_classes, mol_assignments, L, B, n);
  DA_CHECK_LAUNCH("da_pos_laplacian");
  return DA_OK;
}

int da_pos_lap_gather(const float* evecs, long long ld_row, long long ld_col, const int* position, const int* prod_node,
                      const int* n_prod, const float* sign, float* out, int B, int n, int dim, int max_eig, void* stream) {
  DA_CHECK_ARG(B >= 0 && n >= 0 && dim >= 1 && max_eig >= 0, "da_pos_lap_gather: bad sizes");
  if (B == 0 || n == 0) return DA_OK;
  k_pos_lap_gather<<<dim3(n, B), 64, 0, (hipStream_t)stream>>>(evecs, ld_row, ld_col, position, prod_node, n_prod, sign, out, n, dim, max_eig);
  DA_CHECK_LAUNCH("da_pos_lap_gather");
  return DA_OK;
}

}  // extern "C"
