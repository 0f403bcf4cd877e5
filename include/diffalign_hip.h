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

int da_abi_version(void);
const char* da_last_error(void);

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

#ifdef __cplusplus
}
#endif
#endif
