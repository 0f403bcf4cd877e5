// Dense conversion, inpainting and per-chain noise of the sampling drivers.
//   da_to_dense      utils/graph.py:189-210  to_dense (PyG to_dense_batch / to_dense_adj + encode_no_element) on the device
//   da_inpaint       utils/graph.py:621-656  fix_nodes_and_edges_by_idx, and the write-back of diffusion/sampling.py:42-45,175-178
//   da_counter_exp_noise   Exp(1) race noise as a pure function of (chain seed, draw call, element): what makes a sharded
//                    sampling run independent of the rank layout (diffalign_amd/distributed.py)
// All HBM-bound scatter / fill kernels; nothing here is on the per-layer path.
#include "common.h"
#include "diffusion_dev.h"

// ptr[b] = first row of graph b in the sorted batch vector, ptr[B] = N
__global__ void k_batch_ptr(const int64_t* __restrict__ batch, long long N, int B, int64_t* __restrict__ ptr) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > N) return;
  const int64_t lo = i == 0 ? -1 : batch[i - 1];
  const int64_t hi = i == N ? (int64_t)B : batch[i];
  for (int64_t b = lo + 1; b <= hi && b <= B; ++b) ptr[b] = i;
}

__global__ void k_dense_nodes(const float* __restrict__ x, const int64_t* __restrict__ batch, const int64_t* __restrict__ ptr,
                              const int64_t* __restrict__ am, const int64_t* __restrict__ mol, long long N, int n, int kx,
                              float* __restrict__ X, uint8_t* __restrict__ node_mask, int64_t* __restrict__ am_d, int64_t* __restrict__ mol_d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t b = batch[i];
  const long long slot = b * n + (i - ptr[b]);
  for (int c = 0; c < kx; ++c) X[slot * kx + c] = x[i * kx + c];
  node_mask[slot] = 1;
  if (am_d) am_d[slot] = am[i];
  if (mol_d) mol_d[slot] = mol[i];
}

// PyG to_dense_adj: adj[b, i - ptr_b, j - ptr_b, :] += edge_attr (duplicates add up); self loops dropped (remove_self_loops)
__global__ void k_dense_edges(const int64_t* __restrict__ edge_index, const float* __restrict__ edge_attr, const int64_t* __restrict__ batch,
                              const int64_t* __restrict__ ptr, long long M, int n, int ke, float* __restrict__ E) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M) return;
  const int64_t s = edge_index[e], d = edge_index[M + e];
  if (s == d) return;
  const int64_t b = batch[s];
  const long long i = s - ptr[b], j = d - ptr[b];
  if (i < 0 || j < 0 || i >= n || j >= n) return;          // an edge across graphs: PyG would raise; dropped here
  float* o = E + (((long long)b * n + i) * n + j) * ke;
  for (int c = 0; c < ke; ++c) {
    const float v = edge_attr[e * ke + c];
    if (v != 0.f) atomicAdd(o + c, v);
  }
}

// encode_no_element (utils/placeholder.py:9-18) on rows of width k: a row whose sum is 0 gets channel 0 := 1
__global__ void k_encode_rows(float* __restrict__ A, long long rows, int k) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int c = 0; c < k; ++c) s += A[r * k + c];
  if (s == 0.f) A[r * k] = 1.f;
}

__global__ void k_inpaint(const int* __restrict__ nodes, int Nn, const int* __restrict__ edges, int Ne, uint8_t* __restrict__ mask_X,
                          uint8_t* __restrict__ mask_E, float* __restrict__ zX, float* __restrict__ zE, const float* __restrict__ oX,
                          const float* __restrict__ oE, int B, int n, int kx, int ke) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < Nn) {
    const int b = nodes[2 * t], i = nodes[2 * t + 1];
    if (b < 0 || b >= B || i < 0 || i >= n) return;
    const long long r = (long long)b * n + i;
    if (mask_X) mask_X[r] = 1;
    if (zX) for (int c = 0; c < kx; ++c) zX[r * kx + c] = oX[r * kx + c];
  } else if (t - Nn < Ne) {
    const int* q = edges + 3 * (t - Nn);
    const int b = q[0], i = q[1], j = q[2];
    if (b < 0 || b >= B || i < 0 || i >= n || j < 0 || j >= n) return;
    const long long p0 = ((long long)b * n + i) * n + j, p1 = ((long long)b * n + j) * n + i;
    if (mask_E) { mask_E[p0] = 1; mask_E[p1] = 1; }
    if (zE) for (int c = 0; c < ke; ++c) { zE[p0 * ke + c] = oE[p0 * ke + c]; zE[p1 * ke + c] = oE[p1 * ke + c]; }
  }
}

__global__ void k_counter_exp_noise(const int64_t* __restrict__ seeds, unsigned long long stream, float* __restrict__ out, long long per_chain,
                                    long long total) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long long b = t / per_chain;
  out[t] = counter_exp((unsigned long long)seeds[b], stream, (unsigned long long)(t - b * per_chain));
}

static inline unsigned grid_for(long long n, int threads) { return (unsigned)((n + threads - 1) / threads); }

extern "C" {

int da_to_dense(const float* x, const int64_t* batch, const int64_t* edge_index, const float* edge_attr, const int64_t* atom_map,
                const int64_t* mol_assignment, long long N, long long M, int B, int n, int kx, int ke, float* X, float* E,
                uint8_t* node_mask, int64_t* atom_map_dense, int64_t* mol_dense, int64_t* ptr, void* stream) {
  DA_CHECK_ARG(N >= 0 && M >= 0 && B >= 0 && n >= 0 && kx >= 1 && ke >= 1, "da_to_dense: negative size");
  DA_CHECK_ARG(x && batch && X && E && node_mask && ptr && (M == 0 || (edge_index && edge_attr)), "da_to_dense: null pointer");
  DA_CHECK_ARG((atom_map_dense == nullptr || atom_map) && (mol_dense == nullptr || mol_assignment), "da_to_dense: dense output without its input");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  const long long R = (long long)B * n;
  bool ok = hipMemsetAsync(X, 0, sizeof(float) * R * kx, s) == hipSuccess && hipMemsetAsync(E, 0, sizeof(float) * R * n * ke, s) == hipSuccess &&
            hipMemsetAsync(node_mask, 0, R, s) == hipSuccess;
  if (atom_map_dense) ok = ok && hipMemsetAsync(atom_map_dense, 0, sizeof(int64_t) * R, s) == hipSuccess;
  if (mol_dense) ok = ok && hipMemsetAsync(mol_dense, 0, sizeof(int64_t) * R, s) == hipSuccess;
  if (!ok) { da_set_error("da_to_dense: memset failed"); return DA_ERR_HIP; }
  k_batch_ptr<<<grid_for(N + 1, 256), 256, 0, s>>>(batch, N, B, ptr);
  if (N > 0) k_dense_nodes<<<grid_for(N, 256), 256, 0, s>>>(x, batch, ptr, atom_map, mol_assignment, N, n, kx, X, node_mask, atom_map_dense, mol_dense);
  if (M > 0) k_dense_edges<<<grid_for(M, 256), 256, 0, s>>>(edge_index, edge_attr, batch, ptr, M, n, ke, E);
  k_encode_rows<<<grid_for(R, 256), 256, 0, s>>>(X, R, kx);
  k_encode_rows<<<grid_for(R * n, 256), 256, 0, s>>>(E, R * n, ke);
  DA_CHECK_LAUNCH("da_to_dense");
  return DA_OK;
}

int da_inpaint(const int* nodes, int n_nodes, const int* edges, int n_edges, uint8_t* mask_X, uint8_t* mask_E, float* zX, float* zE,
               const float* orig_X, const float* orig_E, int B, int n, int kx, int ke, void* stream) {
  DA_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && (n_nodes == 0 || nodes) && (n_edges == 0 || edges), "da_inpaint: index lists");
  DA_CHECK_ARG((zX == nullptr || orig_X) && (zE == nullptr || orig_E), "da_inpaint: write-back without the original tensors");
  if (n_nodes + n_edges == 0) return DA_OK;
  k_inpaint<<<grid_for(n_nodes + n_edges, 256), 256, 0, (hipStream_t)stream>>>(nodes, n_nodes, edges, n_edges, mask_X, mask_E, zX, zE, orig_X,
                                                                            orig_E, B, n, kx, ke);
  DA_CHECK_LAUNCH("da_inpaint");
  return DA_OK;
}

int da_counter_exp_noise(const int64_t* chain_seeds, long long call, float* noise_X, float* noise_E, int B, int n, int kx, int ke, void* stream) {
  DA_CHECK_ARG(chain_seeds && call >= 0 && B >= 0 && n >= 0, "da_counter_exp_noise: bad arguments");
  if (B == 0 || n == 0) return DA_OK;
  hipStream_t s = (hipStream_t)stream;
  if (noise_X) k_counter_exp_noise<<<grid_for((long long)B * n * kx, 256), 256, 0, s>>>(chain_seeds, 2ull * call, noise_X, (long long)n * kx, (long long)B * n * kx);
  if (noise_E) k_counter_exp_noise<<<grid_for((long long)B * n * n * ke, 256), 256, 0, s>>>(chain_seeds, 2ull * call + 1ull, noise_E, (long long)n * n * ke,
                                                                                          (long long)B * n * n * ke);
  DA_CHECK_LAUNCH("da_counter_exp_noise");
  return DA_OK;
}

}  // extern "C"
