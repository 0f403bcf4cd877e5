/* diffalign_hip.h -- C-ABI of the MI355X (gfx950) DiffAlign hot-path library  (libdiffalign_hip.so)
 *
 * The reference (Aalto-QuML/DiffAlign) is pure Python/PyTorch and has no FFI layer; the drop-in boundary is the
 * Python nn.Module / mixin contract.  This header is the C boundary underneath it: plain pointers
 * and sizes, no torch types.  Each entry point names the reference code it replaces (paths relative to the
 * reference checkout).  Conventions:
 *   - every pointer is a DEVICE pointer unless marked "host"; tensors are contiguous, batch-first, channels-last;
 *   - float = fp32, node_mask / mask_x / mask_e = uint8 (0/1, torch.bool storage), indices = int64;
 *   - `stream` is a hipStream_t (0 = default stream); calls only enqueue work, they never synchronise or allocate;
 *   - return 0 on success, <0 on error; da_last_error() gives the message (thread-local).
 */
#ifndef DIFFALIGN_HIP_H
#define DIFFALIGN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Version of this header's contracts; a binding built against another version must refuse the library (diffalign_amd/_lib.py does).
 * 3: da_init, the bf16-storage training kernels (da_attn_core_*_bf16, da_linear_wgrad_mixed) and the fused residual blocks (da_res_ln_*, da_relu_dropout*) added, da_edge_out_g / da_edge_out_finish / da_skip_grad removed.  2: da_y_film / da_y_update / da_y_update_film / da_y_pre expect the
 * y-chain weights k-major, i.e. transposed to (in, out); da_edge_layer_stats puts its ticket word back to zero itself. */
#define DA_ABI_VERSION 3
int da_abi_version(void);
const char* da_last_error(void);
/* Per-device set-up that must not run under stream capture (today: the device address of da_edge_in_sym_static's scratch).  Call once
 * per device, with that device current, before the first captured call.  No counterpart in the reference. */
int da_init(void);

/* ---- L0 container + masks ------------------------------------------------------------------------------------ */

/* PlaceHolder.mask(collapse=False), in place: X*=m_i, E*=m_i*m_j, E[diag]=0, then encode_no_element (a feature vector
 * whose sum is 0 gets channel 0 := 1).  Replaces diffalign/utils/placeholder.py:9-18,81-102.  X or E may be NULL. */
int da_mask_dense(float* X, float* E, const uint8_t* node_mask, int B, int n, int dx, int de, void* stream);

/* PlaceHolder.mask(collapse=True): argmax over the class axis, masked entries -> 0.
 * Replaces diffalign/utils/placeholder.py:90-94. */
int da_collapse(const float* X, const float* E, const uint8_t* node_mask, int64_t* X_idx, int64_t* E_idx, int B, int n,
                int dx, int de, void* stream);

/* Free-to-diffuse node masks (1 = diffuse, 0 = keep the condition) in closed form; mask_E[b,i,j] = mask_e[b,i] &
 * mask_e[b,j].  mode: 0 product, 1 sn, 2 product_and_sn, 3 reactant, 4 reactant_and_sn, 5 atom_mapping.
 * `states` = class indices whose nodes stay fixed (the reference always adds 'none' = 0).
 * Replaces diffalign/utils/graph.py:462-534 (get_mask), :536-563 (get_mask_sn, one SuNo per graph), :565-594
 * (get_mask_product), :607-619 (from_mask_to_maskX_maskE: no (B*n)^2 outer product here), :678-699. */
int da_free_masks(const float* X_orig, const uint8_t* node_mask, const int64_t* atom_map_numbers, int B, int n, int kx,
                  int mode, int suno_idx, int include_supernode, const int* states /*device*/, int n_states,
                  uint8_t* mask_x, uint8_t* mask_e, void* stream);

/* apply_mask, in place on z: where not free take orig (x100 when as_logits).  zX or zE may be NULL.
 * Replaces diffalign/utils/graph.py:411-460. */
int da_apply_mask(float* zX, float* zE, const float* origX, const float* origE, const uint8_t* mask_x,
                  const uint8_t* mask_e, int B, int n, int kx, int ke, int as_logits, void* stream);

/* Dense conversion of a PyG-style sparse batch on the device: to_dense_batch / to_dense_adj + encode_no_element.
 * Replaces diffalign/utils/graph.py:189-210 (to_dense).  x (N,kx) node features of all graphs, batch (N) sorted graph index per
 * node, edge_index (2,M) global node indices, edge_attr (M,ke), atom_map / mol_assignment (N) int64 (nullable).  Outputs, n =
 * largest graph: X (B,n,kx), E (B,n,n,ke) (self loops dropped, duplicate edges add up, then any all-zero row / pair gets class 0),
 * node_mask (B,n), atom_map_dense / mol_dense (B,n) (nullable).  ptr = (B+1) int64 of caller-owned scratch (row offsets). */
int da_to_dense(const float* x, const int64_t* batch, const int64_t* edge_index, const float* edge_attr, const int64_t* atom_map,
                const int64_t* mol_assignment, long long N, long long M, int B, int n, int kx, int ke, float* X, float* E,
                uint8_t* node_mask, int64_t* atom_map_dense, int64_t* mol_dense, int64_t* ptr, void* stream);

/* Inpainting from index lists: nodes (n_nodes,2) int32 rows (b,i), edges (n_edges,3) int32 rows (b,i,j) (both orientations are
 * touched).  mask_X (B,n) / mask_E (B,n,n) uint8, nullable: set to 1 at the listed entries (the masks of
 * diffalign/utils/graph.py:621-656 fix_nodes_and_edges_by_idx; the caller zero-fills them).  zX / zE, nullable: the listed rows /
 * pairs take orig's values (diffusion/sampling.py:42-45,175-178).  Work is O(list length), not O(B n^2). */
int da_inpaint(const int* nodes, int n_nodes, const int* edges, int n_edges, uint8_t* mask_X, uint8_t* mask_E, float* zX, float* zE,
               const float* orig_X, const float* orig_E, int B, int n, int kx, int ke, void* stream);

/* Exp(1) race noise as a pure function of (chain seed, draw call, element index within the chain): noise_X (B,n,kx) for stream
 * 2*call, noise_E (B,n,n,ke) for stream 2*call+1, chain_seeds (B) int64.  No counterpart in the reference (it draws from torch's
 * global generator); this is what makes sharded sampling independent of the rank layout (diffalign_amd/distributed.py). */
int da_counter_exp_noise(const int64_t* chain_seeds, long long call, float* noise_X, float* noise_E, int B, int n, int kx, int ke,
                         void* stream);

/* ---- L1 diffusion math ---------------------------------------------------------------------------------------- */

/* One fused reverse step:  pred = log_softmax(logits)*temperature  (diffusion/sampling.py:82)
 *   -> p(z_s|z_t) = get_p_zs_given_zt(return_prob=True) incl. .mask()   (utils/diffusion/helpers.py:193-229,785-807)
 *   -> categorical draw per node / per unordered pair, upper triangle mirrored  (helpers.py:706-752)
 *   -> apply_mask against the condition                                 (utils/graph.py:411-460, sampling.py:165-178)
 * Q* are the per-graph transition tables as get_Qt/get_Qt_bar return them, (B,K,K) (noise_schedule.py:147-165):
 * Qt = Q_t, Qsb = Qbar_{t-1}, Qtb = Qbar_t.  noise_* ~ Exp(1), same shape as the logits: the draw is
 * argmax_k p_k/q_k, which is torch.multinomial's CPU algorithm.  prob_*, zs_* (one-hot fp32) are optional outputs;
 * mask_x/mask_e NULL = no conditioning. */
int da_posterior_sample(const float* logits_X, const float* logits_E, const float* zt_X, const float* zt_E,
                        const uint8_t* node_mask, const float* Qt_x, const float* Qsb_x, const float* Qtb_x,
                        const float* Qt_e, const float* Qsb_e, const float* Qtb_e, float temp_x, float temp_e,
                        const float* noise_X, const float* noise_E, const float* orig_X, const float* orig_E,
                        const uint8_t* mask_x, const uint8_t* mask_e, float* prob_X, float* prob_E, float* zs_X,
                        float* zs_E, int B, int n, int kx, int ke, void* stream);

/* sample_discrete_features from explicit probabilities and Exp(1) noise; bit-exact w.r.t. the reference for identical
 * inputs.  Outputs class indices (B,n), (B,n,n) and optionally one-hot fp32.
 * Replaces diffalign/utils/diffusion/helpers.py:706-752. */
int da_sample_discrete(const float* prob_X, const float* prob_E, const uint8_t* node_mask, const float* noise_X,
                       const float* noise_E, int64_t* X_idx, int64_t* E_idx, float* X_onehot, float* E_onehot, int B,
                       int n, int kx, int ke, void* stream);

/* Forward-noising probabilities probX = X @ Qbar_t.X, probE = E @ Qbar_t.E followed by .mask().
 * Replaces diffalign/diffusion/diffusion_abstract.py:253-257 (apply_noise; the draw is da_sample_discrete). */
int da_noise_probs(const float* X, const float* E, const uint8_t* node_mask, const float* Qtb_x, const float* Qtb_e,
                   float* prob_X, float* prob_E, int B, int n, int kx, int ke, void* stream);

/* ---- ELBO / CE terms ----------------------------------------------------------------------------------------------
 * Per-node (B,n) and per-ordered-pair (B,n,n) rows of one term of the objective; true / zt one-hot fp32, logits fp32.
 *  mode 0  L_t, probability space: KL(q(z_s|z_t,x_0) || p(z_s|z_t)), both apply_mask-ed, eps inside the log
 *          (diffalign/diffusion/elbo.py:85-162 with log=False; helpers.py:754-771)
 *  mode 1  L_t, log space (training): same with the log-domain posterior, eps 1e-30 (elbo.py:127-157, helpers.py:772-781)
 *  mode 2  KL prior rows: KL(limit || x_0 Qbar_T), both apply_mask-ed (elbo.py:197-238, helpers.py:115-137)
 *  mode 3  reconstruction rows: sum true * log_softmax(apply_mask(logits, as_logits)) (elbo.py:182-193, helpers.py:139-151)
 *  mode 4  sum true * log_softmax(logits), unmasked: the negated CE rows of helpers.ce (helpers.py:153-191)
 * Tables: modes 0/1 Qa = Q_t, Qb = Qbar_{t-1} per graph (B,K,K), q_per_graph = 1; mode 2 Qa = Qbar_T as one (K,K) table
 * (q_per_graph = 0) and lim_* the limit distributions (K).  mask_x / mask_e: per-node free masks (da_free_masks);
 * pair (i,j) is free when mask_e[i] & mask_e[j].  Unused pointers may be NULL. */
int da_elbo_rows(int mode, const float* true_X, const float* true_E, const float* zt_X, const float* zt_E,
                 const float* logits_X, const float* logits_E, const float* Qa_x, const float* Qb_x, const float* Qa_e,
                 const float* Qb_e, int q_per_graph, const float* lim_x, const float* lim_e, const uint8_t* mask_x,
                 const uint8_t* mask_e, float eps, float* rows_X, float* rows_E, int B, int n, int kx, int ke, void* stream);

/* helpers.mean_without_masked (utils/diffusion/helpers.py:59-106), device part: out (B,4) = per graph the sum of rows_X
 * over free nodes, their count, the sum of rows_E over free pairs, their count (the divisions are the caller's). */
int da_masked_sums(const float* rows_X, const float* rows_E, const uint8_t* mask_x, const uint8_t* mask_e, float* out, int B,
                   int n, void* stream);

/* ---- training step, first stage of the backward (diffusion_abstract.py:134-225) ----------------------
 * helpers.ce (utils/diffusion/helpers.py:153-191) backward: F.cross_entropy(mean) over the free rows gives
 *   grad_X[r] = weight_X / #free nodes * (softmax(logits_X[r]) - onehot(argmax true_X[r]))  on free nodes, 0 elsewhere,
 *   grad_E[p] = weight_E / #free pairs * (softmax(logits_E[p]) - onehot(argmax true_E[p]))  on free pairs, 0 elsewhere
 * (weight_X = diffuse_nodes, weight_E = diffuse_edges * lambda_E), i.e. what loss.backward() leaves in pred.X.grad and
 * pred.E.grad.  sums (B,4) is da_masked_sums' output (its counts are used); grad_X / grad_E may be NULL. */
int da_ce_grad(const float* true_X, const float* true_E, const float* logits_X, const float* logits_E, const uint8_t* mask_x,
               const uint8_t* mask_e, const float* sums, float weight_X, float weight_E, float* grad_X, float* grad_E, int B, int n,
               int kx, int ke, void* stream);

/* ---- training path: the non-GEMM pieces of the fp32 forward / backward (diffusion_abstract.py:134-225, training/helpers.py:139-142).
 * Every nn.Linear of the training forward and both of its backward products are library GEMMs under torch.autograd; these entry
 * points are the bodies of the torch.autograd.Functions of diffalign_amd/autograd_ops.py.  All tensors fp32. ---- */

/* Core of NodeEdgeBlock.forward (layers.py:225-265) per pair and channel: Y = Q_i K_j qscale (E1 m_i m_j + 1) + E2 m_i m_j,
 * Z = ye1 + (ye2 + 1) Y (the input of e_out), attn = softmax over keys j per channel (no key mask), wV = sum_j attn V_j.
 * Q, K, V (B,n,dx) masked; E1 = e_mul(E), E2 = e_add(E) (B,n,n,dx) unmasked; ye1 = y_e_add(y), ye2 = y_e_mul(y) (B,dx).
 * smax / ssum (B,n,dx): the softmax statistics the backward reuses. */
int da_attn_core_fwd(const float* Q, const float* K, const float* V, const float* E1, const float* E2, const uint8_t* node_mask,
                     const float* ye1, const float* ye2, float qscale, float* Z, float* wV, float* smax, float* ssum, int B, int n, int dx,
                     void* stream);
/* Its backward: from dZ (B,n,n,dx) and dwV (B,n,dx) to dQ, dK, dV (B,n,dx), dE1, dE2 (B,n,n,dx; zero on masked pairs), dye1, dye2
 * (B,dx).  With a 16-byte aligned workspace of da_attn_core_bwd_workspace bytes and dx % 4 == 0: four channels per thread, dK / dV as
 * per-row-block partial results folded in block order (deterministic); without it: fp32 atomics on dK / dV (order-dependent in the last
 * bits).  dye1, dye2: one atomic per row block and channel in both forms. */
int da_attn_core_bwd(const float* Q, const float* K, const float* V, const float* E1, const float* E2, const uint8_t* node_mask,
                     const float* ye2, float qscale, const float* wV, const float* smax, const float* ssum, const float* dZ, const float* dwV,
                     float* dQ, float* dK, float* dV, float* dE1, float* dE2, float* dye1, float* dye2, void* workspace, long long workspace_bytes,
                     int B, int n, int dx, void* stream);
long long da_attn_core_bwd_workspace(int B, int n, int dx);
/* Weight and bias gradient of an nn.Linear applied to R rows (training/helpers.py:139 `loss.backward()` for the per-pair Linears e_mul,
 * e_add, e_out, linE1, linE2, mlp_in_E.2, mlp_out_E.0, where R = B n n): dW (out,in) = dY^T X, db (out) = column sums of dY (NULL: skipped);
 * X (R,in), dY (R,out) fp32 row-major.  Split over R, fp32 MFMA, partial results in `workspace` (da_linear_wgrad_workspace bytes), summed in
 * slice order.  Built for (in,out) in {(64,256), (256,64), (64,128), (128,64), (64,64), (128,128)}; others: DA_ERR_UNSUPPORTED. */
long long da_linear_wgrad_workspace(long long R, int in_features, int out_features);
int da_linear_wgrad(const float* X, const float* dY, float* dW, float* db, void* workspace, long long workspace_bytes, long long R,
                    int in_features, int out_features, void* stream);
/* Mixed-precision training (the reference trains under fp16 autocast: diffusion/diffusion_abstract.py:181-183, training/helpers.py:139-141):
 * the same three kernels with the per-pair streams in bf16 STORAGE -- E1, E2, Z / dZ, dE1, dE2 of the attention core, X and / or dY of the
 * weight gradient (x_bf16 / dy_bf16 != 0) -- and fp32 arithmetic; per-node and per-graph tensors stay fp32.  The attention pair: dx % 4 == 0
 * with dx / 4 dividing 256, workspace as da_attn_core_bwd_workspace. */
int da_attn_core_fwd_bf16(const float* Q, const float* K, const float* V, const void* E1, const void* E2, const uint8_t* node_mask,
                          const float* ye1, const float* ye2, float qscale, void* Z, float* wV, float* smax, float* ssum, int B, int n, int dx,
                          void* stream);
int da_attn_core_bwd_bf16(const float* Q, const float* K, const float* V, const void* E1, const void* E2, const uint8_t* node_mask,
                          const float* ye2, float qscale, const float* wV, const float* smax, const float* ssum, const void* dZ, const float* dwV,
                          float* dQ, float* dK, float* dV, void* dE1, void* dE2, float* dye1, float* dye2, void* workspace,
                          long long workspace_bytes, int B, int n, int dx, void* stream);
int da_linear_wgrad_mixed(const void* X, int x_bf16, const void* dY, int dy_bf16, float* dW, float* db, void* workspace,
                          long long workspace_bytes, long long R, int in_features, int out_features, void* stream);
/* Fused residual block of XEyTransformerLayer (layers.py:341-359: norm(res + dropout(branch))) for the training step, and its
 * backward.  s = res + dropout_p(mask(u)); out = LayerNorm(s) * gamma + beta (eps 1e-5); out_bf16 (may be NULL): out again in bf16, the
 * operand of the next bf16 GEMM.  u / d_u are fp32 or bf16 (u_bf16).  mask_mode 0: none; 1: row r is zeroed unless node_mask[r]
 * (rows = B n); 2: row (b n + i) n + j is zeroed unless node_mask[b,i] and node_mask[b,j] (rows = B n n; `newE * e_mask`, layers.py:262).
 * Dropout stores no mask: the keep bits are a hash of (element, *seed, salt) -- seed is read on the device, so a captured graph draws
 * fresh masks per replay -- and the backward regenerates them from the same (seed, salt, p); p = 0 is the reference's eval arithmetic
 * exactly.  The backward takes the gradient of out and / or out_bf16 (either may be NULL), returns d_res = d s, d_u (may be NULL),
 * dgamma, dbeta; workspace: da_res_ln_bwd_workspace(D) bytes.  D in {16, 32, 64, 128, 256}; fp32 operands 16-byte, bf16 8-byte aligned. */
int da_res_ln_fwd(const float* res, const void* u, int u_bf16, const uint8_t* node_mask, int mask_mode, int n, const float* gamma,
                  const float* beta, const long long* seed, int salt, float p, float* s_out, float* out, void* out_bf16, long long rows,
                  int D, void* stream);
long long da_res_ln_bwd_workspace(int D);
int da_res_ln_bwd(const float* s, const float* gamma, const float* d_out, const void* d_out_bf16, const uint8_t* node_mask, int mask_mode, int n,
                  const long long* seed, int salt, float p, float* d_res, void* d_u, int u_bf16, float* dgamma, float* dbeta,
                  void* workspace, long long workspace_bytes, long long rows, int D, void* stream);
/* dropout_p(relu(h)) in place between the two Linears of a feed-forward branch (layers.py:349-357), fp32 or bf16 storage, numel % 4 == 0;
 * backward: dx = dy / (1 - p) where the stored result is non-zero, else 0 (no mask kept; dx may alias dy). */
int da_relu_dropout(void* h, int is_bf16, long long numel, const long long* seed, int salt, float p, void* stream);
int da_relu_dropout_bwd(const void* h, const void* dy, void* dx, int is_bf16, long long numel, float p, void* stream);
/* LayerNorm backward (eps 1e-5; layers.py:341-359): s (rows,D) = the normalised input, dy the gradient of gamma * xhat + beta ->
 * ds (rows,D), dgamma (D), dbeta (D).  D in {32, 64, 128, 256}; s, dy, ds 16-byte aligned; workspace: 8 * 1024 * D bytes for the
 * per-workgroup partial sums of dgamma / dbeta (folded in workgroup order: deterministic). */
int da_layernorm_bwd(const float* s, const float* gamma, const float* dy, float* ds, float* dgamma, float* dbeta, void* workspace,
                     long long workspace_bytes, long long rows, int D, void* stream);
/* Xtoy / Etoy pooled statistics (layers.py:15-47) of T (B,N,d): z (B,4,d) = [mean, min, max, unbiased std] over the N rows, arg
 * (B,2,d) int32 = row index of the min / max (first occurrence); and the backward dT from dz.  workspace: room for the per-slice
 * partial results, 32 * B * 64 * d bytes covers every N (NULL / too small: one workgroup column per graph, slow for N = n*n). */
int da_pool_stats_fwd(const float* T, float* z, int* arg, void* workspace, long long workspace_bytes, int B, int N, int d, void* stream);
int da_pool_stats_bwd(const float* T, const float* z, const int* arg, const float* dz, float* dT, int B, int N, int d, void* stream);
/* d elbo_batch_quick / d logits (elbo.py:273-334, avg_over_batch=True): free rows of graphs with t > 1 carry the log-space
 * KL(q(z_s|z_t,x_0) || p(z_s|z_t)) of compute_Lt (helpers.py:772-781), graphs with t == 1 (t_is_1[b] != 0) carry -log p(x_0).
 * Qt = Q_t, Qsb = Qbar_{t-1} per graph (B,K,K); sums = da_masked_sums' output (its counts); weight_X = diffusion_steps *
 * diffuse_nodes, weight_E = diffusion_steps * lambda_E * diffuse_edges. */
int da_vb_grad(const float* true_X, const float* true_E, const float* zt_X, const float* zt_E, const float* logits_X, const float* logits_E,
               const float* Qt_x, const float* Qsb_x, const float* Qt_e, const float* Qsb_e, const uint8_t* t_is_1, const uint8_t* mask_x,
               const uint8_t* mask_e, const float* sums, float weight_X, float weight_E, float* grad_X, float* grad_E, int B, int n, int kx,
               int ke, void* stream);
/* One AdamW step (torch.optim.AdamW semantics, decoupled weight decay; amsgrad when max_exp_avg_sq != NULL: utils/setup.py:356) on
 * a flat buffer of N parameters; grad is multiplied by grad_scale first (1 / world size, or the inverse loss scale).
 * slot_active (ceil(N / 64) bytes, or NULL = all): 64-element slots whose byte is 0 are left untouched -- parameters without a
 * gradient in this step, which torch.optim skips (no decay, moments frozen). */
int da_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long long N, float lr, float beta1,
             float beta2, float eps, float weight_decay, int step, float grad_scale, const uint8_t* slot_active, void* stream);
/* da_adamw with the number of updates done so far in device memory (*step_counter, int, starts at 0; incremented by the call): no
 * per-step host value in the launch, so a whole training step can be replayed from a HIP graph. */
int da_adamw_devstep(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long long N, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int* step_counter, float grad_scale, const uint8_t* slot_active,
                     void* stream);

/* ---- L2 denoiser (graph transformer) -------------------------------------------------------------------------------
 * Hidden edge features travel between these calls as bf16 (B,n,n,de); node features X and globals y stay fp32.
 * Weights arrive pre-packed by diffalign_amd/pack.py: "blob" = MFMA fragments (1 KiB each, lane-linear, bf16) followed
 * by fp32 bias / LayerNorm vectors, in the layout documented in csrc/k_edge_layer.hip (EdgeCfg) and csrc/k_edge_io.hip
 * (EdgeIoCfg); node-side nn.Linear weights as bf16 [N padded to 32][K padded to 16].
 * Built shapes: (de,dx,dim_ffE,hidden_mlp E) = (64,256,128,128) [align_absorbing / align_marginal] and (32,64,64,32)
 * [tiny test config]; anything else returns DA_ERR_UNSUPPORTED (-3) with a message. */

long long da_edge_blob_bytes(int de, int dx, int dff);      /* -1 when the shape is not built */
long long da_edge_in_blob_bytes(int de, int he);
long long da_edge_out_blob_bytes(int de, int he);
long long da_kv_frag_bytes(int B, int n, int dx);
long long da_x_post_blob_bytes(int dx, int dffx);           /* -1 when the shape is not built */
long long da_x_in_blob_bytes(int dx_in, int hx, int dx);    /* -1 when the shape is not built */

/* out[M,N] = act(A[M,K] @ W[N,K]^T + bias) (* rowmask): every nn.Linear of the node path -- mlp_in_X / mlp_out_X
 * (transformer_model_with_y.py:584-585,602-603), q/k/v/x_out (layers.py:217-218,256,273), linX1/linX2 (layers.py:349).
 * bf16 MFMA, fp32 accumulate.  W_bf16 holds the (N,K) weight zero-padded to (Npad32, Kpad16) in MFMA B-fragment order
 * (diffalign_amd/pack.py bf16_weight): 1 KiB fragments [nt][ks], lane (r,h) element j = W[32nt + r][16ks + 8h + j]; K <= 256. */
int da_linear(const float* A, int lda, const void* W_bf16, int Kpad, const float* bias, const uint8_t* rowmask, float* out,
              int ldc, int M, int N, int K, int relu, void* stream);

/* Node side of one XEyTransformerLayer after the attention, fused: newX = x_out(newX_in) * mask (layers.py:268-273);
 * X1 = normX1(X + newX); X' = normX2(X1 + linX2(relu(linX1(X1)))) (layers.py:343-351).  fp32 (rows, dx) row-major in/out,
 * blob = diffalign_amd/pack.py x_post_blob (XPostCfg layout).  Built: (dx, dim_ffX) = (256,256), (64,64). */
int da_x_post(const float* newX_in, const float* X, const uint8_t* node_mask, const void* blob, float* X_out, int rows, int dx,
              int dffx, void* stream);
/* The same stage on 16-row tiles (v_mfma_f32_16x16x32_bf16, two waves per SIMD): blob = pack.py x_post16_blob (XPost16Cfg layout,
 * da_x_post16_blob_bytes; -1: dims not built).  Same arithmetic, sums grouped by the tile shape. */
long long da_x_post16_blob_bytes(int dx, int dffx);
int da_x_post16(const float* newX_in, const float* X, const uint8_t* node_mask, const void* blob, float* X_out, int rows, int dx,
                int dffx, void* stream);

/* Node input stage, fused: X_h = encode_no_element(relu(mlp_in_X.2(relu(mlp_in_X.0(X)))) * mask) + pos_encodings
 * (transformer_model_with_y.py:584-585, 661, 666).  X (rows, dx_in) fp32 row-major with dx_in <= 96, pos (rows, dx) or NULL,
 * blob = diffalign_amd/pack.py x_in_blob (XInCfg layout).  Built: (hidden_mlp X, dx) = (256,256), (64,64). */
int da_x_in(const float* X, int dx_in, const uint8_t* node_mask, const float* pos, const void* blob, float* X_h, int rows, int hx, int dx,
            void* stream);

/* out = LayerNorm(a + b) * gamma + beta, eps 1e-5: normX1 / normX2 (layers.py:341,351). b may be NULL. */
int da_add_layernorm(const float* a, const float* b, const float* gamma, const float* beta, float* out, int rows, int D,
                     void* stream);

/* X_h = encode_no_element(mlp_in_X(X) * mask) + pos_encodings, in place (transformer_model_with_y.py:661,666). */
int da_x_in_finish(float* X_h, const uint8_t* node_mask, const float* pos_encodings /*may be NULL*/, int rows, int D,
                   void* stream);

/* E_h(bf16) = encode_no_element(mask(sym(mlp_in_E(E))))  (transformer_model_with_y.py:657-661; 1 <= de_in <= 16). */
int da_edge_in(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, int B, int n, int de,
               int he, void* stream);
/* da_edge_in with a per-graph pre-pass (coalesced 32x32 tile reads) that establishes which graphs are symmetric and which
 * are one-hot, instead of one transposed read per pair; graphs that are both (every diffusion state: helpers.py:743-745)
 * (all-zero pairs allowed) are written by a gather from the (de_in + 2)-row table of the stage's possible results.  sym_flags = caller-owned scratch
 * of da_edge_in_scratch_bytes(B, de_in, de) bytes, 16-byte aligned; on return bytes [0,B) = symmetric, [B,2B) = one-hot or zero.
 * Bit-identical to da_edge_in. */
long long da_edge_in_scratch_bytes(int B, int de_in, int de);
int da_edge_in_sym(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, uint8_t* sym_flags, int B,
                   int n, int de, int he, void* stream);
/* da_edge_in_sym without a workspace argument, for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The scratch is a 288 KiB static
 * __device__ array of the library, per GPU; B <= 4096 (larger batches: DA_ERR_ARG; use da_edge_in_sym).  Allocates nothing.
 * Calls on one GPU share the array: calls that overlap on one GPU race.  Same E_h as da_edge_in_sym, bit for bit. */
int da_edge_in_sym_static(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, int B, int n, int de,
                          int he, void* stream);
/* Etoy statistics of the stage's output without another pass over it: da_edge_stats_table writes da_edge_stats' partials
 * for the graphs da_edge_in_sym served from its table (from the class counts it left in `scratch`), da_edge_stats_skip
 * runs da_edge_stats on the other graphs only (skip = the same scratch).  Sums are exact products count x value instead of
 * fp32 running sums: equal to da_edge_stats within fp32 rounding, not bit for bit. */
int da_edge_stats_table(const uint8_t* scratch, float* partials, int B, int de_in, int de, int nchunk, void* stream);
int da_edge_stats_skip(const void* E_bf16, float* partials, const uint8_t* skip, int B, int n, int de, int nchunk, void* stream);

/* y_h = mlp_in_y(y)  (transformer_model_with_y.py:590-591), fp32. */
int da_y_in(const float* y, const float* W0, const float* b0, const float* W1, const float* b1, float* out, int B, int din,
            int dh, int dy, void* stream);

/* Per-layer FiLM vectors from y (layers.py:245-247,268-270): ye2p1 = 1+y_e_mul(y), ebias = e_out.b + e_out.W y_e_add(y),
 * yx1 = y_x_add(y), yx2p1 = 1+y_x_mul(y).  w = host array of 10 device pointers: y_e_add.{W^T,b}, y_e_mul.{W^T,b},
 * y_x_add.{W^T,b}, y_x_mul.{W^T,b} (^T = stored (in,out) = (dy,dx)), e_out.{W TRANSPOSED (dx,de), b}. */
int da_y_film(const float* y, const float* const* w /*host*/, float* ye2p1, float* ebias, float* yx1, float* yx2p1, int B,
              int dy, int dx, int de, void* stream);

/* q, k, v projections of one layer fused with the K/V fragment packing (layers.py:217-223): Q (B*n, dx) fp32 = q(X)*mask,
 * kv_frag = K, V in da_kv_pack's layout (da_kv_frag_bytes).  W_bf16 = cat(q,k,v).weight packed like da_linear's operand,
 * bias = cat of the three biases (3*dx).  Built: dx = 256, 64. */
int da_qkv_frag(const float* X, const void* W_bf16, const float* bias, const uint8_t* node_mask, float* Q, void* kv_frag,
                int B, int n, int dx, void* stream);

/* K,V (fp32 (B*n, ldkv), masked) -> bf16 fragment layout read by da_edge_layer. */
int da_kv_pack(const float* K, const float* V, int ldkv, void* kv_frag, int B, int n, int dx, void* stream);

/* The fused edge side of one XEyTransformerLayer: NodeEdgeBlock (layers.py:204-284) + the E branch of
 * XEyTransformerLayer.forward (layers.py:343-355) in one pass over E: E1/E2 projections, Y, per-channel softmax over
 * keys (no key mask, as the reference), weighted V, FiLM, e_out, LN1, FFN, LN2.  Also writes newX_in = yx1 + yx2p1*wV,
 * the input of x_out, for the rows of real query nodes; rows of padded query nodes hold finite don't-care values (x_out's
 * result is masked to zero there, layers.py:273).  E_in != E_out. */
int da_edge_layer(const void* E_in, void* E_out, const float* Q, int ldq, const void* kv_frag, const uint8_t* node_mask,
                  const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                  const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch /* nullable */,
                  long long split_scratch_bytes, void* stream);
/* da_edge_layer that also produces the Etoy statistics of its INPUT tensor (what the y branch of the same layer needs,
 * layers.py:31-47) in da_edge_stats' partials layout [B][stats_nchunk][4][de]: the waves that finish their share of the
 * layer early stream E_in once more instead of idling, so the separate statistics pass disappears.  ticket: one int
 * of device memory that is zero when the kernel starts; the kernel's last wave puts it back to zero, so the caller clears it once.  Same E_out /
 * newX_in as da_edge_layer, bit for bit. */
int da_edge_layer_stats(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                        const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                        const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                        long long split_scratch_bytes, float* stats_partials, int stats_nchunk, int* ticket, void* stream);

/* First layer without a materialised hidden input (round 2).  da_edge_in_classes = da_edge_in_sym that writes, for the graphs
 * that are symmetric and one-hot (every diffusion state), one byte per pair -- the row of the table in `scratch` holding
 * mlp_in_E of that pair (transformer_model_with_y.py:657-662) -- into cls_map (B,n,n) instead of their hidden rows in E_h; the
 * other graphs get their E_h rows as da_edge_in writes them.  da_edge_layer_table = da_edge_layer whose input rows of the
 * table-path graphs come from that table (in_scratch = the same scratch) instead of E_h.  de_in <= 8. */
int da_edge_in_classes(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, uint8_t* cls_map,
                       uint8_t* scratch, int B, int n, int de, int he, void* stream);
/* da_edge_in_classes for a caller that keeps `scratch` between calls (no counterpart in the reference: its mlp_in_E runs on every pair,
 * transformer_model_with_y.py:657-661): table_valid != 0 = the scratch still holds the table an earlier call with the same blob, B and
 * de_in left there; only the flags and class counts are then set up again. */
int da_edge_in_classes_keep(const float* E_in, int de_in, const uint8_t* node_mask, const void* blob, void* E_h, uint8_t* cls_map,
                            uint8_t* scratch, int B, int n, int de, int he, int table_valid, void* stream);
int da_edge_layer_table(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                        const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                        const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                        long long split_scratch_bytes, const uint8_t* cls_map, const uint8_t* in_scratch, int de_in, void* stream);

/* The general form of the three entry points above: every optional part as an argument (NULL / 0: absent), plus a list of the query
 * rows to compute: rowmap[0 .. n_rows) int32, the row numbers b n + i in ascending order (NULL: every row).  The rows left out get
 * neither their row of E_out nor of newX_in (both stay as they were).  For the LAST layer of a fused reverse step
 * (diffusion/sampling.py:76-86 + utils/graph.py:411-470): a pair with an endpoint that is not free to diffuse takes the condition's
 * value whatever its logits are, so only the rows of free nodes are ever read again.  The split scratch is then sized by
 * da_edge_split_scratch_bytes_rows(n_rows, n, dx).  n_rows_dev (optional): the list's length as an int32 in device memory, read by
 * the kernels, so that a launch captured in a HIP graph serves lists of any length; n_rows is then the caller's estimate (grid and
 * shares are chosen for it) and the scratch is sized by da_edge_split_scratch_bytes_rows(-n_rows, n, dx).
 * row_tiles (optional; B n words, used for n <= 1024): bit jt of word b n + i set = E_out[b][i][32 jt .. 32 jt + 31] is wanted; a tile
 * whose bit is clear takes part in the row's attention (newX_in is complete) and its E_out entries stay as they were.  For the last
 * two layers of a fused reverse step: the last reads E[i][.] of free rows only, so the one before forms no E_out for the other rows,
 * and of the last layer's E_out the output stage reads the pairs with both ends free. */
long long da_edge_split_scratch_bytes_rows(long long rows, int n, int dx);
int da_edge_layer_ex(const void* E_in, void* E_out, const float* Q, int ldq, const void* KV_frag, const uint8_t* node_mask,
                     const float* ye2p1, const float* ebias, const float* yx1, const float* yx2p1, float* newX_in,
                     const void* blob, float qscale, int B, int n, int de, int dx, int dff, void* split_scratch,
                     long long split_scratch_bytes, float* stats_partials, int stats_nchunk, int* ticket, const uint8_t* cls_map,
                     const uint8_t* in_scratch, int de_in, const int* rowmap, int n_rows, const int* n_rows_dev, const unsigned* row_tiles,
                     void* stream);
/* Launch tuning of da_edge_layer / da_edge_layer_stats, process-wide (no counterpart in the reference; for tests and A/B timing):
 * threads per workgroup 256 | 512 (0 = keep), skew = whole rows the older wave of every SIMD pair takes more than the younger one,
 * tail_old = the split rows' runs go to the older waves only (both: -1 = chosen from the problem size), stagger, prio = start
 * delay / s_setprio level of the younger half, skip_dead = tiles without one real pair (padded query row, 32 padded keys) skip
 * the attention arithmetic and enter the softmax in closed form (default 1).  Results do not depend on any of them beyond the fp32
 * order of the softmax sums (split rows, dead tiles); E_out is bit-identical under all of them.  da_edge_split_scratch_bytes depends on `threads`.  Defaults come from DA_EDGE_THREADS / _SKEW / _TAIL_OLD /
 * _STAGGER / _PRIO / _SKIP_DEAD, read once. */
int da_edge_layer_tuning(int threads, int skew, int tail_old, int stagger, int prio, int skip_dead);
/* Load balance of da_edge_layer's persistent grid: with a scratch of this many bytes (0: not needed for this size) the
 * query rows of the last, incomplete round are cut into runs of key tiles over all waves and folded by a small second
 * kernel; without it (NULL) whole rows only.  Results are identical up to the fp32 order of the softmax merge. */
long long da_edge_split_scratch_bytes(int B, int n, int dx);

/* Pooled-statistics partials (sum, sum of squares, min, max per channel) for Etoy / Xtoy (layers.py:15-47):
 * partials are (B, nchunk, 4, d) fp32 and are combined (fp64, unbiased std) inside da_y_update. */
int da_edge_stats(const void* E_bf16, float* partials, int B, int n, int de, int nchunk, void* stream);
int da_x_stats(const float* X, float* partials, int B, int n, int D, int nchunk, void* stream);

/* y' of one layer: new_y = y_out(y + Xtoy(X) + Etoy(E)); y = norm_y1(y + new_y); y = norm_y2(y + lin_y2(relu(lin_y1(y))))
 * (layers.py:277-281,346-359).  w = host array of 16 device pointers: x_y.lin.W^T, x_y.lin.b, e_y.lin.W^T, e_y.lin.b,
 * y_out.0.{W^T,b}, y_out.2.{W^T,b}, norm_y1.{w,b}, lin_y1.W^T, lin_y1.b, lin_y2.W^T, lin_y2.b, norm_y2.{w,b} (^T = stored (in,out)). */
int da_y_update(const float* y, const float* xstats, const float* estats, const float* const* w /*host*/, float* out,
                int B, int n, int dx, int de, int dy, int dff, int nchunk_x, int nchunk_e, void* stream);
/* The part of da_y_update that does not depend on the layer's edge kernel: xy (B,dy) = x_y.lin(Xtoy statistics of the layer's input X)
 * (layers.py:15-29, 277) from da_x_stats' partials; w_upd = da_y_update's pointer array. */
int da_y_pre(const float* xstats, const float* const* w_upd, float* xy, int B, int n, int dx, int dy, int nchunk_x, void* stream);
/* da_y_update with that term given and, when film_w is not NULL, da_y_film of the NEXT layer in the same launch (layers.py:241-262 of
 * layer l+1 behind :277-281,346-359 of layer l): two dependent launches less on the chain between two edge kernels. */
int da_y_update_film(const float* y, const float* xy, const float* estats, const float* const* w_upd, float* out,
                     const float* const* film_w, float* ye2p1, float* ebias, float* yx1, float* yx2p1, int B, int n, int dx, int de,
                     int dy, int dff, int nchunk_e, void* stream);

/* src[b,r] = product node carrying the same atom-map number as reactant-side node r, else -1: the gather form of
 * get_X_prods_and_E_prods_aligned + create_permutation_matrix_torch (transformer_model_with_y.py:506-555,
 * utils/math_utils.py:3-44) for alignment_type='correct'. */
int da_align_index(const int64_t* atom_map_numbers, const int64_t* mol_assignments, int* src, int B, int n, void* stream);

/* X logits = mlp_out_X(X_h) + X_in[..., :kx] (+ skip_scaling * X_in[src])  (transformer_model_with_y.py:676,699-701). */
int da_x_out_finish(const float* lin, int ldl, const float* X_in, int dx_in, const int* src /*may be NULL*/,
                    const float* skip_scale, float* out, int B, int n, int kx, void* stream);

/* E logits = sym((mlp_out_E(E_h) + E_in[..., :ke]) * offdiag) and, when src != NULL, the product->reactant skip:
 * mapped pairs += skip1 * E_in[src_i, src_j], every other pair += skip2 on class 0
 * (transformer_model_with_y.py:672-704). */
int da_edge_out(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1,
                const float* skip2, float* E_logits, int B, int n, int de, int he, int ke, void* stream);

/* da_edge_out in one pass over mirrored 32x32 tiles (every hidden edge row read once, row-wise; the mirrored tile meets
 * its partner in LDS; no scratch).  Same arguments, bit-identical logits. */
int da_edge_out_tiles(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1,
                      const float* skip2, float* E_logits, int B, int n, int de, int he, int ke, void* stream);
/* Output stage + the edge half of the reverse step in one kernel (diffusion/sampling.py:82-86,165-178 behind
 * neuralnet/transformer_model_with_y.py:672-704; utils/diffusion/helpers.py:193-245,706-752; utils/graph.py:411-534 apply_mask):
 * zs_E (B,n,n,7) one-hot fp32 drawn from p(z_s | z_t) of the logits this stage computes, apply_mask folded in -- bit-identical to
 * da_edge_out_tiles followed by da_posterior_sample's edge half for the same noise, without the (B,n,n,7) logits round trip.
 * Noise: noise_E (B,n,n,7) Exp(1), or NULL with chain_seeds (B) + call: the values da_counter_exp_noise(chain_seeds, call) would
 * write, generated in registers.  The logits are not written (callers that want them use the separate form).  ke must be 7. */
int da_edge_out_sample(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1, const float* skip2,
                       const float* zt_E, const uint8_t* node_mask, const float* Qt_e, const float* Qsb_e, const float* Qtb_e,
                       float temp_e, const float* noise_E, const int64_t* chain_seeds, long long call, const float* orig_E, const uint8_t* mask_e,
                       float* zs_E, int B, int n, int de, int he, int ke, void* stream);
/* da_edge_out_sample with flags.  flags & 1: zs_E already holds the condition's values at every pair that is not free to diffuse
 * (apply_mask, utils/graph.py:411-470, gives those pairs the same value at every step of a chain, so a caller that alternates
 * between two state buffers -- diffalign_amd/graphs.py, sampling.py -- has them in place after the first two steps): only the
 * free pairs are written, and tile pairs without a free pair are not touched at all.  Needs mask_e and orig_E. */
int da_edge_out_sample_ex(const void* E_h, const float* E_in, int de_in, const void* blob, const int* src, const float* skip1,
                          const float* skip2, const float* zt_E, const uint8_t* node_mask, const float* Qt_e, const float* Qsb_e,
                          const float* Qtb_e, float temp_e, const float* noise_E, const int64_t* chain_seeds, long long call,
                          const float* orig_E, const uint8_t* mask_e, float* zs_E, int B, int n, int de, int he, int ke, int flags,
                          void* stream);
/* ---- extra input features of the denoiser (reference: CPU round trip at every step,
 * diffusion/diffusion_abstract.py:350-372).  The two adjacency powers and the batched symmetric eigendecomposition between
 * these calls are library routines (torch.bmm, torch.linalg.eigh on the device; diffalign_amd/extra_features.py). ---- */

/* One pass over the one-hot edge state E (B,n,n,ke) fp32: A = sum of the non-'none' channels (neuralnet/
 * extra_features.py:81), the fp64 combinatorial Laplacian of the masked graph with 2n on the padded diagonal (:103-109,
 * :133-146; L may be NULL), and per node the sums over j of argmax_c(E_ijc * bond_orders_c) and of argmax_c(E_ijc)
 * (neuralnet/extra_features_molecular.py:40-42, :57-59). */
int da_extra_adjacency(const float* E, const uint8_t* node_mask, const float* bond_orders, float* A, double* L,
                       float* val_weighted, float* val_plain, int B, int n, int ke, void* stream);

/* k-cycle counts per node (k = 3,4,5; /10, clipped at 1, masked) and per graph (k = 3..6) from A, K2 = A@A, K3 = K2@A
 * (extra_features.py:75-90, :227-298).  A must be symmetric (every diffusion state is): diag(A^4), diag(A^5) and
 * tr(A^6) are formed from K2 and K3. */
int da_extra_cycles(const float* A, const float* K2, const float* K3, const uint8_t* node_mask, float* x_cycles /*(B,n,3)*/,
                    float* y_cycles /*(B,4)*/, int B, int n, void* stream);

/* Eigen-features from the ascending eigenvalues (B,n) and eigenvectors (element (i,k) of graph b at
 * b*n*n + i*ld_row + k*ld_col) of the Laplacian (extra_features.py:117-128, :158-204): number of (near-)zero
 * eigenvalues of L / n_real, the 5 next eigenvalues (2 beyond the spectrum), the not-in-largest-component indicator
 * (mode of the first eigenvector rounded to 3 decimals; pad_noise (B,n) stands for the reference's randn on padded
 * slots) and the 2 eigenvectors after the zero ones. */
int da_extra_eigen(const double* evals, const double* evecs, long long ld_row, long long ld_col, const uint8_t* node_mask,
                   const float* pad_noise, float* n_comp /*(B)*/, float* ev5 /*(B,5)*/, float* nonlcc /*(B,n)*/,
                   float* vec2 /*(B,n,2)*/, int B, int n, void* stream);

/* The same features as torch.linalg.eigh + da_extra_eigen, in one launch, one workgroup per graph: in-place Householder
 * tridiagonalisation of L (B,n,n) fp64 (DESTROYED), Sturm count / bisection for the eigenvalues that are used (the count
 * below 1e-5 * n_real, then 5), inverse iteration + back-transformation for eigenvectors 0, n_comp, n_comp + 1
 * (extra_features.py:117-128, :158-204; no counterpart in the reference: it calls LAPACK for the full decomposition).
 * evals6 (B,6), optional: eigenvalues 0 and n_comp .. n_comp+4 of L.  n <= 1024. */
int da_extra_eigen_solve(double* L, const uint8_t* node_mask, const float* pad_noise, float* n_comp, float* ev5, float* nonlcc,
                         float* vec2, double* evals6, int B, int n, void* stream);

/* da_extra_eigen_solve for a caller that evaluates the same graphs again and again with a trailing block of nodes whose edges
 * never change -- the product side of a reaction along a reverse chain (sampling.py:76-178 calls the extra features at every
 * step; utils/graph.py:411-470 apply_mask keeps every edge with a fixed endpoint).  fixed_from (B) int32: first node of graph
 * b's fixed block (n: none).  cache (B, mmax + 2, mmax) fp64 and cache_state (B) int32, zero before the first call, belong to
 * the caller and to THIS sequence of graphs: the first call reduces everything and keeps the block's reflectors (state 1, or -1
 * when the block shares an edge with the nodes before it), later calls copy them back and reduce only the rows before the block.
 * Same bits as da_extra_eigen_solve as long as the block really is unchanged (the caller's promise; not checked).
 * cache == NULL: da_extra_eigen_solve.  Blocks larger than mmax are reduced in full every time. */
int da_extra_eigen_solve_cached(double* L, const uint8_t* node_mask, const float* pad_noise, float* n_comp, float* ev5,
                                float* nonlcc, float* vec2, double* evals6, int B, int n, const int* fixed_from, int* cache_state,
                                double* cache, int mmax, void* stream);

/* da_extra_eigen_solve_cached writing straight into the denoiser's input rows (diffusion_abstract.py:350-372 concatenates the
 * features behind X and y: extra_features.py:66-72 orders them [cycles(3), nonlcc, vec2(2)] per node and [n, cycles(4), n_comp,
 * ev5(5)] per graph): x_feat[(b n + i) * ldx + 0..2] = nonlcc, vec2[0], vec2[1];  y_feat[b * ldy + 0..5] = n_comp, ev5[0..4].
 * Same values as the dense form; ldx >= 3, ldy >= 6; cache == NULL: uncached. */
int da_extra_eigen_solve_into(double* L, const uint8_t* node_mask, const float* pad_noise, float* x_feat, int ldx, float* y_feat,
                              int ldy, int B, int n, const int* fixed_from, int* cache_state, double* cache, int mmax, void* stream);

/* charge = argmax_c(X_c * valencies_c) - val_weighted, valency = val_plain, weight = sum_i atom_weights[argmax X_i] /
 * max_weight (extra_features_molecular.py:29-72; the argmax INDEX is used, as in the reference). */
int da_extra_molecular(const float* X, const float* valencies, const float* atom_weights, const float* val_weighted,
                       const float* val_plain, float max_weight, float* charge_val /*(B,n,2)*/, float* weight /*(B)*/,
                       int B, int n, int kx, void* stream);

/* ---- positional encodings shared between mapped reactant / product atoms (reference: per-graph Python loops on
 * the CPU, neuralnet/transformer_model_with_y.py:213-503 PositionalEmbedding) ---- */

/* position (B,n): 1-based rank, inside the product (the molecule with the largest assignment), of the product node that
 * carries the node's atom-map number, 0 if none (:400-412, :259-275; last product node wins on duplicates, as the
 * reference's sequential scatter_); prod_node (B,n+1): node index of the k-th product node; n_prod (B). */
int da_pos_product_index(const int64_t* atom_map_numbers, const int64_t* mol_assignments, int* position, int* prod_node,
                         int* n_prod, int B, int n, void* stream);
/* SinusoidalPosEmb (:114-131): out (rows,dim) = [sin(1000 v f_j), cos(1000 v f_j)], freq (dim/2) as torch computes it. */
int da_pos_sinusoidal(const float* value, const float* freq, float* out, long long rows, int dim, void* stream);
/* out[b,i,:] = table[b, idx[b,i], :]  (Gaussian encodings, :492-503, idx = atom-map number). */
int da_pos_gather_rows(const float* table, const int64_t* idx, float* out, int B, int n, int rows_per_graph, int dim, void* stream);
/* fp32 symmetric-normalised Laplacian (B,n,n) of the product block: unweighted edges where the bond class is non-zero and
 * both ends are product nodes, L_ii = 1 everywhere (:147-154, :192-198). */
int da_pos_laplacian(const int64_t* E_classes, const int64_t* mol_assignments, float* L, int B, int n, void* stream);
/* out (B,n,dim): columns 1..k (k = min(max_eig, n_prod)) of the ascending eigenvectors, row of the product node at
 * `position`, times sign (B,max_eig); zero elsewhere (:254-259, :275). */
int da_pos_lap_gather(const float* evecs, long long ld_row, long long ld_col, const int* position, const int* prod_node,
                      const int* n_prod, const float* sign, float* out, int B, int n, int dim, int max_eig, void* stream);

#ifdef __cplusplus
}
#endif
#endif
