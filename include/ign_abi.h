/* ign_abi.h -- C ABI of libign_hip.so: the MI355X (gfx950) hot path of the Interpretability-Gated-Network.
 *
 * The reference (001camellia/Speech-Imagery-EEG, IGN/ = InterpretGatedNetwork/) has no FFI: its hot path is a
 * chain of PyTorch eager ops inside Python modules.  Each entry point below replaces one such chain; the
 * reference lines it stands in for are cited per function.  Binding side: INTEGRATION.md (ctypes).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (contiguous fp32 unless noted); the library never
 *    allocates or frees device memory; its only mutable state is the optional timing registry (ign_timing_*);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are enqueued and the call
 *    returns immediately;
 *  - return 0 on success; IGN_E_* (<0) on argument errors (nothing was launched); -(hipError_t) on a launch error;
 *    ign_last_error() gives a human-readable reason for the calling thread;
 *  - `workspace` buffers are caller-provided, sized by the matching *_workspace_bytes().
 */
#ifndef IGN_ABI_H
#define IGN_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGN_ABI_VERSION 1

#define IGN_E_ARG      (-1001)  /* bad pointer / dimension                                */
#define IGN_E_UNSUP    (-1002)  /* combination not implemented (see message)              */
#define IGN_E_TOOBIG   (-1003)  /* a row does not fit the LDS staging budget              */

/* distance / gate selectors: `mode` = IGN_DIST_* | IGN_GATE_*                                           */
#define IGN_DIST_L1    0   /* mean |x-w|      IGN/model/Shapelet.py:74 ("euclidean", the default)        */
#define IGN_DIST_MSE   1   /* mean (x-w)^2    IGN/model/Shapelet.py:24-40 (memory_efficient branch)      */
#define IGN_DIST_COS   2   /* 1 - cos         IGN/model/Shapelet.py:64-66                                */
#define IGN_DIST_PEARS 3   /* 1 - pearson     IGN/model/Shapelet.py:67-69,11-19                          */
#define IGN_GATE_RBF   0x00 /* exp(-(eps d)^2) + straight-through max   IGN/model/Shapelet.py:77-84      */
#define IGN_GATE_LTS   0x10 /* straight-through soft-min + sigmoid(thr - min_d)  Shapelet.py:105-111     */

int         ign_abi_version(void);
const char* ign_last_error(void);

/* Instance normalisation fused with the (B,T,C) -> (B,C,T) transpose.
 * Replaces IGN/model/Shapelet.py:186-187  (rearrange 'b t c -> b c t'; (x-mean_T)/(std_T(unbiased)+eps)).
 * xt_bct (nullable) additionally receives the un-normalised transpose (input of the FCN expert,
 * IGN/model/FullyConvNet.py:53).                                                                        */
int ign_instnorm_fwd(const float* x_btc, float* xn_bct, float* xt_bct, int B, int T, int C, float eps,
                     void* stream);
/* Same pass, additionally max |x| of the raw input as an atomic maximum into *amax_slot (caller zeroes it): the magnitude bound
 * of the FCN expert's first fp16 GEMM operand (ign_clconv_fwd_h3), taken where the batch is staged anyway instead of by a
 * separate ign_absmax pass when both experts run on one stream.                                                              */
int ign_instnorm_fwd_amax(const float* x_btc, float* xn_bct, float* xt_bct, int B, int T, int C, float eps, float* amax_slot,
                          void* stream);

/* Plain (B,T,C) -> (B,C,T) transpose of a batch: what `x.permute(0, 2, 1)` means for a consumer that needs electrodes-first rows
 * (the EEG-CNN baseline, IGN/model/eegcnn.py:134, fed from the time-first item contract of IGN/data_factory/uea.py).           */
int ign_transpose_btc_to_bct(const float* x_btc, float* out_bct, int B, int T, int C, void* stream);

/* On-GPU input pipeline of the CHISCO loader: raw (B,C,T) recordings -> standardised (B,T,C) batches.
 * Replaces, per batch, Normalizer('per_sample_std') of IGN/data_factory/eeg.py:332-367 (per sample and channel over time:
 * (x - mean) / (std(ddof=1) + eps)) and the (C,T) -> (T,C) item transpose that IGN/data_factory/uea.py:7-42 batches.
 * stats_ws: B*C*2 floats of workspace (mean, 1/(std+eps) per row).                                                    */
int ign_standardise_nct_to_btc(const float* x_nct, float* out_btc, float* stats_ws, int B, int C, int T, float eps,
                               void* stream);

/* Shapelet transform of ONE length group: sliding-window distance + gate, never materialising (B,Tw,K,C,L).
 * Replaces IGN/model/Shapelet.py:60-84 (GATE_RBF) / :96-111 (GATE_LTS).   Tw = (T-L)/stride + 1.
 *   xn_bct   (B,C,T)    normalised input
 *   w_kcl    (K,C,L)    shapelets            thr_kc (K,C) LTS thresholds (NULL for RBF)
 *   p_out    row-major (B, ld) ; this group writes columns [col0 + k*C + c]     (max_p / sigmoid gate)
 *   dmin_out same indexing                                                       (min_t d)
 *   tstar    (B,K,C) int32  arg-max_t p  (RBF) / arg-min_t d (LTS); first index on ties (torch.argmax)
 *   zmu      (B,K,C,2)      {Z, mu}: RBF Z=sum_t exp(p_t), mu=sum_t softmax_t(p) p_t ;
 *                           LTS Z=sum_t exp(-(d_t-dmin)), mu=sum_t softmin_t(d) d_t      (for backward)
 *   d_save   (B,C,K,Tw)     every window distance, kept for the backward (NULL: not saved)
 *   xstat_save (B,C,Tw)     cosine / pearson only (else NULL): |x_win| resp. sqrt(sum (x_win-mean)^2), for the backward
 * IGN_DIST_PEARS expects w_kcl already CENTRED over its last axis (w - mean_j w; the caller's autograd projects the
 * gradient back), so that <x_win, w_c> equals the reference's centred numerator (Shapelet.py:11-19).               */
int ign_shapelet_fwd(const float* xn_bct, const float* w_kcl, const float* thr_kc,
                     float* p_out, float* dmin_out, int ld, int col0,
                     int32_t* tstar, float* zmu, float* d_save, float* xstat_save,
                     int B, int C, int T, int K, int L, int stride, float eps, int mode, void* stream);

/* All G (<= 8) length groups of a bank -- the loop over `self.shapelets` of IGN/model/Shapelet.py:190-196 -- in one call: the
 * per-group arguments of ign_shapelet_fwd as tables of G entries (host arrays; the outputs share p_out / dmin_out / ld).  Every
 * group is validated before the first launch; results are bitwise those of G ign_shapelet_fwd calls.                         */
int ign_shapelet_fwd_bank(const float* xn_bct, int G, const float* const* w_kcl, const float* const* thr_kc, float* p_out,
                          float* dmin_out, int ld, const int* col0, int32_t* const* tstar, float* const* zmu,
                          float* const* d_save, float* const* xstat_save, int B, int C, int T, const int* K, const int* L,
                          const int* stride, float eps, int mode, void* stream);

/* Backward of the above w.r.t. the shapelets (autograd of Shapelet.py:60-84 / :96-111; closed form in
 * SURVEY.md App. A).  g_out is dloss/dp_out with the same (ld, col0) indexing as p_out.
 *   gw_kcl (K,C,L) receives dloss/dw (overwritten, deterministic: fixed-order two-stage reduction over B)
 *   workspace: ign_shapelet_bwd_workspace_bytes() bytes.
 * Gradients w.r.t. the input are not produced (inputs are data: IGN/exp/experiment_classification.py:315).
 * LTS: dloss/dthr is a (B,KC) elementwise reduction the caller forms from g_out and p_out.
 * sign(0) convention (IGN_DIST_L1 only).  The reference differentiates |x - w| with aten::sgn, sign(0) = 0
 * (IGN/model/Shapelet.py:74).  The kernel accumulates P_j = sum_{t: x > w} A_t and returns 2 P_j - sum_t A_t, which counts an
 * element with x[b,c,t+j] == w[k,c,j] (bit-equal floats) as sign = -1.  Hence, exactly,
 *     gw_kcl[k,c,j] = reference[k,c,j] - sum_{(b,t): x[b,c,t*stride+j] == w[k,c,j]} A[b,k,c,t],   A = -(dloss/dd[b,t,k,c]) / L,
 * and the two agree wherever no sample is bit-equal to the weight it is compared with (always, for weights that an
 * optimiser has moved; a shapelet initialised as a copy of an input window is the reachable exception -- and with the RBF gate
 * a perfect-match window has d = 0, dp/dd = 0, so A = 0 there).  Forward outputs are unaffected.  Pinned by the reference
 * fixtures tests/golden/shapelet_tie_{l1,lts}.npz: tests/test_gpu_shapelet.py::test_exact_ties_differ_from_sgn0_by_exactly_
 * the_documented_term asserts this identity (and nothing else) at 1e-4.  Other distances have no kink (MSE: the factor x - w
 * is 0 at a tie; cosine / pearson: smooth).
 * stride > 1 (IGN/model/Shapelet.py:162: int(log2 L) once seq_len >= 3000 -- run_uea.sh's MotorImagery, EigenWorms) is
 * served by a generic-step kernel; shapelets longer than 2048 positions are split over several blocks.  Rows up to
 * T ~ 40 000 fit the forward's LDS staging (160 KB per CU); beyond that both calls return IGN_E_TOOBIG.             */
size_t ign_shapelet_bwd_workspace_bytes(int B, int C, int T, int K, int L, int stride, int mode);
int ign_shapelet_bwd(const float* xn_bct, const float* w_kcl, const float* g_out, const float* p_out,
                     const float* dmin_out, int ld, int col0,
                     const int32_t* tstar, const float* zmu, const float* d_save,
                     const float* xstat_save, const float* wnorm_kc /* (K,C) sqrt(sum_j w^2): cosine / pearson, else NULL */,
                     float* gw_kcl, void* workspace,
                     int B, int C, int T, int K, int L, int stride, float eps, int mode, void* stream);

/* Backward of every group of a bank in one call (the per-group arguments of ign_shapelet_bwd as tables of G <= 8 host entries):
 * G backward launches, then ONE reduction launch over all groups that also adds add_scale_dev[0] * gw_add[g] into group g's
 * result (gw_add / add_scale_dev nullable) -- the batch-independent gradient of the diversity regulariser
 * (ign_sbm_reg_fwd_bwd), so that a parameter with two gradient sources needs no accumulate kernel.  Every group is validated
 * before the first launch; with nothing added the results are bitwise those of G ign_shapelet_bwd calls (B <= 256).
 * workspace: ign_shapelet_bwd_bank_workspace_bytes() bytes.                                                              */
size_t ign_shapelet_bwd_bank_workspace_bytes(int G, int B, int C, int T, const int* K, const int* L, const int* stride, int mode);
int ign_shapelet_bwd_bank(const float* xn_bct, int G, const float* const* w_kcl, const float* g_out, const float* p_out,
                          const float* dmin_out, int ld, const int* col0, const int32_t* const* tstar, const float* const* zmu,
                          const float* const* d_save, const float* const* xstat_save, const float* const* wnorm_kc,
                          float* const* gw_kcl, const float* const* gw_add, const float* add_scale_dev, void* workspace,
                          int B, int C, int T, const int* K, const int* L, const int* stride, float eps, int mode, void* stream);

/* Fused attention core softmax(scale * Q K^T) V, exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32), scores never
 * materialised.  Replaces IGN/layers/SelfAttention_Family.py:56-75 (FullAttention: no mask, dropout 0) and the
 * attention inside nn.TransformerEncoderLayer of IGN/model/eegcnn.py:219-228.
 *   q (B,L,H,E), k/v (B,S,H,E): unit stride over E, stride E over H; *_sb / *_sl are the ELEMENT strides of the
 *   batch and sequence axes (multiples of 4), so packed qkv projections can be passed without a copy.
 *   out (B,L,H,E) contiguous; lse (B,H,L) log-sum-exp of the scaled scores (saved for the backward).
 *   E in {16, 32, 64, 128}; all pointers 16-byte aligned.                                                          */
int ign_attn_fwd(const float* q, const float* k, const float* v, float* out, float* lse,
                 int B, int L, int S, int H, int E,
                 long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                 float scale, void* stream);
/* Backward: gq (B,L,H,E), gk/gv (B,S,H,E) contiguous outputs (overwritten); gout (B,L,H,E) contiguous;
 * delta_ws: B*H*L floats of workspace.  dQ recomputes the scores instead of using float atomics: deterministic.  */
/* The forward on the bf16 matrix cores at fp32 accuracy (split-bf16 products as in ign_clconv_*_x6; K / V tiles are split while
 * they are staged, Q and P in registers).  Same arguments, outputs and saved statistics as ign_attn_fwd, so ign_attn_bwd follows
 * either.                                                                                                                   */
int ign_attn_fwd_x6(const float* q, const float* k, const float* v, float* out, float* lse, int B, int L, int S, int H, int E,
                    long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                    float scale, void* stream);
int ign_attn_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* gout,
                 float* gq, float* gk, float* gv, float* delta_ws,
                 int B, int L, int S, int H, int E,
                 long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                 float scale, void* stream);
/* The backward in split-bf16 form: delta, then dV, dK (key blocks, each recomputing S) and dQ (query blocks) as three kernels;
 * same arguments and results as ign_attn_bwd (gradients bitwise reproducible, no atomics).  E <= 64 runs without register
 * spills; E = 128 is accepted but slower than ign_attn_bwd.                                                                */
int ign_attn_bwd_x6(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* gout,
                 float* gq, float* gk, float* gv, float* delta_ws,
                 int B, int L, int S, int H, int E,
                 long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                 float scale, void* stream);
/* Both with ONE product per MFMA step on operands rounded to bf16 (Q, K, V, P, dO, dS; fp32 accumulation and softmax): what the
 * reference's default bf16-autocast mode computes for the two attention matmuls.  Same arguments and outputs.             */
int ign_attn_fwd_bf16(const float* q, const float* k, const float* v, float* out, float* lse, int B, int L, int S, int H, int E,
                    long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                    float scale, void* stream);
int ign_attn_bwd_bf16(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* gout,
                 float* gq, float* gk, float* gv, float* delta_ws,
                 int B, int L, int S, int H, int E,
                 long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                 float scale, void* stream);
/* ign_attn_bwd_x6 (bf16 = 0) / ign_attn_bwd_bf16 (bf16 = 1) writing gq / gk / gv with the caller's batch and sequence strides
 * (elements, multiples of 4; head stride E): the three gradients can land in ONE packed (B, L, 3, H, E) buffer -- the gradient of
 * a fused q/k/v projection -- without a gather pass.                                                                      */
int ign_attn_bwd_x6_strided(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* gout,
                 float* gq, float* gk, float* gv, float* delta_ws,
                 int B, int L, int S, int H, int E,
                 long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                 float scale, void* stream,
                            long long g_sb, long long g_sl, int bf16);

/* The attention core on two fp16 planes / three products per element pair (the "h3" arithmetic described at ign_clconv_fwd_h3;
 * E <= 64): Q, K, V, dO are scaled by powers of two from the device-side bounds bq / bk / bv / bgo (upper bounds of their maxima),
 * scores are un-scaled inside the exp2, probabilities (<= 1) are split after a multiplication by 2^14, the score gradient by its
 * hard bound 2 E max|dO| max|V|.  Same results as ign_attn_fwd_x6 / ign_attn_bwd_x6 at the fp32 rounding level.  g_sb = g_sl = 0:
 * contiguous gradients, else the strided outputs of ign_attn_bwd_x6_strided.                                                  */
int ign_attn_fwd_h3(const float* q, const float* k, const float* v, float* out, float* lse,
                    int B, int L, int S, int H, int E,
                    long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                    float scale, void* stream, const float* bq, const float* bk, const float* bv);
int ign_attn_bwd_h3(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* gout,
                    float* gq, float* gk, float* gv, float* delta_ws,
                    int B, int L, int S, int H, int E,
                    long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl,
                    float scale, void* stream, long long g_sb, long long g_sl,
                    const float* bq, const float* bk, const float* bv, const float* bgo,
                    float* g_amax /* nullable: max over |gq|, |gk|, |gv| as an atomic maximum (caller zeroes) -- the bound of
                                     the packed gradient for the projection's backward GEMMs */);

/* Skinny expert-head GEMM  out[b,n] = sum_f X[b,f] W[n,f] (+ bias[n]),  N <= 16 classes, F % 4 == 0, row pitch ldx.
 * Replaces nn.Linear at IGN/model/Shapelet.py:171,200 (SBM head), IGN/model/Transformer.py:72,109,
 * IGN/model/FullyConvNet.py:50,58.  Backward: gX (B,ldx) and/or gW (N,F), gbias (N) (any may be NULL); sums over the
 * batch run in ascending order (deterministic).                                                                    */
int ign_head_fwd(const float* X, const float* W, const float* bias, float* out, int B, int F, int N, long long ldx,
                 void* stream);
int ign_head_bwd(const float* g_out, const float* X, const float* W, float* gX, float* gW, float* gbias,
                 int B, int F, int N, long long ldx, void* stream);
/* The same with gW += add_scale_dev[0] * gW_add (both nullable; add_scale_dev NULL = 1): a batch-independent gradient of W --
 * the L1 regulariser mean|W| of IGN/model/Shapelet.py:219 -- lands in the same store instead of an accumulate kernel.   */
int ign_head_bwd_acc(const float* g_out, const float* X, const float* W, float* gX, float* gW, float* gbias,
                     const float* gW_add, const float* add_scale_dev, int B, int F, int N, long long ldx, void* stream);

/* Gini-index gate of the two experts, forward and backward.  Replaces IGN/model/InterpGN.py:44-52:
 * eta = (N*sum softmax(sbm)^2 - 1)/(N-1); if use_gating_value and eta > gating_value: eta = 1 (test time);
 * out = eta*sbm + (1-eta)*dnn.   out (B,N), eta (B).  Backward: geta (B) may be NULL.                              */
int ign_gate_fwd(const float* sbm, const float* dnn, float* out, float* eta, int B, int N, float gating_value,
                 int use_gating_value, void* stream);
int ign_gate_bwd(const float* sbm, const float* dnn, const float* gout, const float* geta, float* gsbm, float* gdnn,
                 int B, int N, float gating_value, int use_gating_value, void* stream);

/* IGN's training-loss tail in one launch: gini gate + CE(mixture, y) + beta*CE(sbm, y) (batch means) and the gradients of
 * that sum w.r.t. both experts' logits.  Replaces IGN/exp/experiment_classification.py:320-329 (the two F.cross_entropy
 * terms) together with IGN/model/InterpGN.py:44-52 and their autograd.  labels: int64 (B), values in [0, N).
 * out (B,N), eta (B), loss2 = {CE(out,y), CE(sbm,y), their beta-weighted sum}, gsbm / gdnn (B,N) = d(CE(out,y) + beta*CE(sbm,y)) / d logits.  N <= 16. */
int ign_loss_fwd_bwd(const float* sbm, const float* dnn, const long long* labels, float* out, float* eta, float* loss2,
                     float* gsbm, float* gdnn, int B, int N, float beta, void* stream);
/* The same with the model's regulariser value added to loss2[2] on the device (`reg`: one float, nullable): the whole training
 * loss CE(out,y) + info.loss.mean() + beta*CE(sbm,y) of IGN/exp/experiment_classification.py:325-329 in one launch.        */
int ign_loss_fwd_bwd_reg(const float* sbm, const float* dnn, const long long* labels, const float* reg, float* out, float* eta,
                         float* loss2, float* gsbm, float* gdnn, int B, int N, float beta, void* stream);

/* Shapelet diversity regulariser of one length group, forward and gradient in one launch.
 * Replaces IGN/model/Shapelet.py:223-230:  mean_{c,i,j} exp(-||w[i,c,:] - w[j,c,:] + eps||_2) (1 - delta_ij), eps = 1e-6.
 * loss_part_c (C): per-channel partial sums (their sum is the group's loss); gw_kcl (K,C,L): d loss / d w.  K <= 16.  */
int ign_diversity_fwd_bwd(const float* w_kcl, float* loss_part_c, float* gw_kcl, int K, int C, int L, float eps,
                          void* stream);

/* Both regularisers of the shapelet bottleneck model -- ShapeBottleneckModel.loss(), IGN/model/Shapelet.py:217-230 -- value and
 * gradients in ONE launch:  loss_out[0] = lambda_reg * mean|W| + lambda_div * sum_g diversity_g  (G = 0 skips the diversity
 * term, as the reference does for lambda_div <= 0);  gW_reg (nW) = lambda_reg * sign(W) / nW (sign(0) = 0);  gw_kcl[g] (K,C,L) =
 * lambda_div * d diversity_g / d w.  Per-block partials are combined in a fixed order by the block that finishes last (bitwise
 * reproducible).  workspace: ign_sbm_reg_workspace_bytes() bytes, ZERO-FILLED ONCE by the caller and then owned by one model on
 * one stream (the kernel re-arms its ticket counter itself).  G <= 8, K <= 16.                                             */
size_t ign_sbm_reg_workspace_bytes(int G, int C, long long nW);
int ign_sbm_reg_fwd_bwd(const float* W, float* gW_reg, long long nW, float lambda_reg, int G, const float* const* w_kcl,
                        float* const* gw_kcl, const int* K, const int* L, int C, float lambda_div, float eps, float* loss_out,
                        void* workspace, void* stream);

/* One Adam step over flat buffers (torch.optim.Adam semantics, no weight decay / amsgrad): replaces the per-tensor
 * optimizer.step() of IGN/exp/experiment_classification.py:338.  `step` is the 1-based step count.               */
int ign_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                  float beta1, float beta2, float eps, int step, void* stream);

/* Copy `count` gradient tensors (src[i], n[i] floats) to flat + off[i] in one launch: the flat gradient bucket of the
 * data-parallel step is filled by one kernel instead of one accumulate kernel per parameter.  src / off / n are HOST arrays. */
int ign_gather_flat(const void* const* src, const long long* off, const long long* n, int count, float* flat, void* stream);

/* The same step with the step count kept on the device (*step_dev is incremented, bc_dev[2] receives the bias
 * corrections): nothing host-side changes between steps, so the launch sequence can be captured into a hipGraph.      */
int ign_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                      float beta1, float beta2, float eps, int* step_dev, float* bc_dev, void* stream);

/* EEG-CNN block 1 without its (B,F1,C,T) intermediate (1 GB at B=256) -- see csrc/ign_eegcnn.hip for the algebra.
 * Replaces the BatchNorm-1 batch statistics of IGN/model/eegcnn.py:90-91 (block1_conv1 -> block1_bn1):
 *   y1[r,f,t] = sum_j w1[f,j] xpad[r, t+j]   (rows r = (b,c); 'same' zero padding, pad_left on the left)
 *   fwd:  m2[f]   = sum_{r,t} (y1 - mu[f])^2                     (mu = the batch mean, supplied by the caller)
 *   bwd:  g[f,j]  = sum_{r,t} (y1 - mu[f]) xpad[r, t+j]          (= 1/2 d m2[f] / d w1[f,j])
 * x (rows,T), w1 (F1,k1), mu (F1); workspace: ign_conv1_sumsq_workspace_bytes().  Deterministic reductions.       */
size_t ign_conv1_sumsq_workspace_bytes(int rows, int F1, int k1);
int ign_conv1_sumsq_fwd(const float* x, const float* w1, const float* mu, float* m2, void* workspace,
                        int rows, int T, int F1, int k1, int pad_left, void* stream);
int ign_conv1_sumsq_bwd(const float* x, const float* w1, const float* mu, float* g_fj, void* workspace,
                        int rows, int T, int F1, int k1, int pad_left, void* stream);

/* LayerNorm over the last dimension of an (R, D) matrix (nn.LayerNorm semantics: biased variance, eps inside the sqrt), forward
 * and backward; D % 4 == 0, D <= 4096 (backward: D <= 2048).  fwd saves mean / rstd (R each).  bwd: gx and, when not NULL,
 * dgamma / dbeta (fixed-order two-pass reduction: reproducible); part = ign_layernorm_parts(R, D) * 2 * D floats of workspace.
 * Replaces IGN/layers/Transformer_EncDec.py:36-37,48,76-77, the norms of nn.TransformerEncoderLayer in IGN/model/eegcnn.py:219-228
 * and IGN/model/TimesNet.py:197.                                                                                          */
long long ign_layernorm_parts(long long R, int D);
int ign_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long long R, int D,
                      float eps, void* stream);
/* y = LayerNorm(x + res) in one pass: the residual connection of a post-norm encoder layer (IGN/layers/Transformer_EncDec.py:
 * 36-37,48; nn.TransformerEncoderLayer of IGN/model/eegcnn.py:219-228) in front of its LayerNorm.  `sum` (R, D) receives x + res,
 * the tensor ign_layernorm_bwd takes as `x`; its gx is the gradient of both addends.                                          */
int ign_layernorm_res_fwd(const float* x, const float* res, float* sum, const float* gamma, const float* beta, float* y,
                          float* mean, float* rstd, long long R, int D, float eps, void* stream);
int ign_layernorm_bwd(const float* x, const float* gy, const float* gamma, const float* mean, const float* rstd, float* gx,
                      float* dgamma, float* dbeta, float* part, long long R, int D, void* stream);
/* ign_layernorm_bwd that also takes max |gx| as an atomic maximum into *amax_slot (caller zeroes): the magnitude bound of dL/dx
 * for the fp16 GEMMs of the dense layer behind (ign_clconv_fwd_h3 / ign_linear_wgrad_h3), where gx is produced anyway.           */
int ign_layernorm_bwd_amax(const float* x, const float* gy, const float* gamma, const float* mean, const float* rstd, float* gx,
                           float* dgamma, float* dbeta, float* part, float* amax_slot, long long R, int D, void* stream);
/* Lag sums C[d] = sum_rows sum_u x[row][u] x[row][u+d], d < K <= 128 -- with edge terms from the first / last k-1 samples of
 * each row they give the window Gram matrix G[j,j'] = sum x_pad[t+j] x_pad[t+j'] and BatchNorm-1's batch variance of
 * IGN/model/eegcnn.py:90-91 as the quadratic form w1^T G w1 (gradient 2 G w1): K*T FMA per row for all filters, no backward
 * pass over the data.  part: (ign_autocorr_parts(rows), K) partial sums, added up by the caller in double.                   */
long long ign_autocorr_parts(int rows);
int ign_autocorr_fwd(const float* x_rows, float* part, int rows, int T, int K, void* stream);
/* The edge terms of that Gram matrix: per row only the first / last k-1 samples of the zero-padded row xp (left pad pad_left)
 * enter.  part: (ign_edge_lagprod_parts(rows), 2, 124, 128) floats, [b][0][s][d] = sum over the block's rows of xp[s] xp[s+d]
 * (head, s + d < k-1), [b][1][s][d] = the same at xp[T+s] (tail); summed over b by the caller in double.  k <= 125.  Column
 * d = 127 (never a lag) carries the plain column sums of the same samples: [b][0][s][127] = sum xp[s], [b][1][s][127] = sum
 * xp[T+s] -- what the zero padding removes from the per-tap sums behind BatchNorm-1's batch MEAN (ign_bn1_gram).                */
long long ign_edge_lagprod_parts(int rows);
int ign_edge_lagprod_fwd(const float* x_rows, float* part, int rows, int T, int k, int pad_left, void* stream);
/* BatchNorm-1 of the EEG-CNN block (IGN/model/eegcnn.py:90-91: batch statistics of the (B, F1, C, T) temporal convolution that is
 * never formed) from data statistics, in three small launches instead of ~45 float64 library kernels (cumsum / flip / cat /
 * gather / einsum and their autograd):
 *   ign_bn1_data_stats  x (rows, T) -> the window Gram matrix G (k, k) and the per-tap sums S (k), float64, in four launches and no
 *                     library call: lag sums + row sums (ign_autocorr_sum_fwd), edge terms + column sums of the first / last k-1
 *                     samples (ign_edge_lagprod_fwd), their per-block partials summed in float64, and the assembly
 *                        G[j][j+d] = C[d] - sum_{s<j} Dh[s][d] - sum_{s>=j} Dt[s][d],  S[j] = total - (samples tap j never sees).
 *                     workspace: ign_bn1_data_stats_workspace_bytes.  IGN_E_UNSUP for rows longer than 1024 samples.
 *   ign_bn1_fold_fwd  per filter f: mu = w_f . S / n, var = w_f^T G w_f / n - mu^2, a = gamma_f / sqrt(var + eps), b = beta_f - a mu;
 *                     outputs alpha[f D + i] = a, cshift[f D + i] = b * rs[f D + i] (rs = row sums of the depthwise weights: the
 *                     affine map of BatchNorm-1 as it enters the fused BatchNorm-2 op), updates the running statistics
 *                     (nullable) with `momentum`, saves G w_f, mu, var, 1/sqrt for the backward
 *   ign_bn1_fold_bwd  gradients of w1 (F1, k), gamma, beta and rs from those of alpha / cshift (closed form: d mu = S / n,
 *                     d var = 2 G w / n - 2 mu S / n).
 * k <= 125, F1 * D <= 4096.                                                                                                    */
size_t ign_bn1_data_stats_workspace_bytes(int rows, int T, int k);
int ign_bn1_data_stats(const float* x_rows, int rows, int T, int k, int pad_left, void* workspace, double* G, double* S, void* stream);
/* ign_autocorr_fwd restricted to its register-tiled kernel (T <= 1024), which then also returns the plain sum of each block's rows:
 * rowsum_part[p], p < *used_parts = partial rows actually written (later rows of `part` are not touched).                         */
int ign_autocorr_sum_fwd(const float* x_rows, float* part, float* rowsum_part, int rows, int T, int K, int* used_parts, void* stream);
int ign_bn1_fold_fwd(const float* w1, const float* gamma, const float* beta, const float* rs, const double* G, const double* S,
                     double n, float eps, float momentum, float* running_mean, float* running_var, float* alpha, float* cshift,
                     double* saved /* (F1, k + 4) */, int F1, int k, int Dm, void* stream);
int ign_bn1_fold_bwd(const float* g_alpha, const float* g_cshift, const float* w1, const float* gamma, const float* rs,
                     const double* S, const double* saved, double n, float* g_w1, float* g_gamma, float* g_beta, float* g_rs,
                     int F1, int k, int Dm, void* stream);
/* Depthwise (per-channel) 1-D convolution over time, zero 'same' padding: y[b,c,t] = sum_j w[c,j] xpad[b,c,t+j].
 * Replaces the temporal convolutions of IGN/model/eegcnn.py:67 (after the channel contraction) and :78 (block2_conv1).
 * flip=1 correlates with the reversed filter (gradient w.r.t. x: call with dy and pad_left = k-1-pad_left).
 * bwd_weight: dw[c,j] = sum_{b,t} dy[b,c,t] xpad[b,c,t+j]  (workspace: ign_dwconv1d_bwd_weight_workspace_bytes()). */
int ign_dwconv1d_fwd(const float* x, const float* w, float* y, int B, int C, int T, int k, int pad_left, int flip,
                     void* stream);
size_t ign_dwconv1d_bwd_weight_workspace_bytes(int B, int C, int k);
int ign_dwconv1d_bwd_weight(const float* x, const float* dy, float* dw, void* workspace, int B, int C, int T, int k,
                            int pad_left, void* stream);

/* EEG-CNN block, the ops around the depthwise temporal convolutions (IGN/model/eegcnn.py:67-108), (B, channels, T) layout.
 * Channel contraction u[b,o,t] = sum_c W[o,c] x[b,c,t]: the depthwise SPATIAL conv (eegcnn.py:71,92; Conv2d(F1, D*F1, (C,1),
 * groups=F1) applied after the temporal conv commutes with it, see csrc/ign_eegcnn.hip) and the POINTWISE conv (:79,100).
 *   wt_ci64: W^T zero-padded to (Ci, 64) floats (Co <= 64, Ci <= 128).  The input gradient is the same call with W in place of
 *   W^T; bwd_weight: dW[o,c] = sum_{b,t} du[b,o,t] x[b,c,t] (workspace: ign_chan_contract_bwd_weight_workspace_bytes()).     */
int ign_chan_contract_fwd(const float* x_bct, const float* wt_ci64, float* u_bot, int B, int Ci, int Co, int T, void* stream);
size_t ign_chan_contract_bwd_weight_workspace_bytes(int B, int Ci, int Co, int T);
int ign_chan_contract_bwd_weight(const float* du_bot, const float* x_bct, float* dw_oc, void* workspace, int B, int Ci, int Co,
                                 int T, void* stream);

/* BatchNorm2d with batch statistics + ELU + AvgPool2d((1,P)) as one op (eegcnn.py:72-74,93-95 / :80-82,101-103).
 *   ign_chan_stats: sums_c2 (C,2) DOUBLES = per-channel sum v, sum v^2 over (b,t)  (workspace: ign_chan_stats_workspace_bytes()).
 *   ign_affine_elu_pool_fwd: out[b,c,tp] = mean_{i<P} ELU(scale[c] v[b,c,tp P+i] + shift[c]); the caller folds BatchNorm (and the
 *     affine map in front of it) into scale / shift; Tp = T / P.
 *   backward: ign_bn_elu_pool_bwd_sums -> (C,2) doubles S1 = sum dz, S2 = sum dz (v - mean[c]) with dz = ELU'(.) dout / P;
 *     ign_bn_elu_pool_bwd_apply: dv = ka[c] dz + kb[c] + kc[c] v (the BatchNorm backward as three per-channel coefficients).
 *   ign_bn_fold_fwd / _bwd: the per-channel algebra between the two passes in one launch each -- forward: moments of v, the
 *     affine map y = alpha v + c in front of the BatchNorm (NULL: identity), gamma, beta -> scale / shift of the apply pass,
 *     fold_c2 (C,2) doubles {mean_v, r = rsqrt(alpha^2 var_v + eps)} kept for the backward, and the nn.BatchNorm running-statistics
 *     update (momentum, unbiased variance; NULL pointers: skipped); backward: (S1, S2) -> ka / kb / kc and dgamma, dbeta, dalpha. */
size_t ign_chan_stats_workspace_bytes(int B, int C);
int ign_chan_stats(const float* v_bct, double* sums_c2, void* workspace, int B, int C, int T, void* stream);
int ign_affine_elu_pool_fwd(const float* v_bct, const float* scale_c, const float* shift_c, float* out, int B, int C, int T, int P,
                            void* stream);
int ign_bn_elu_pool_bwd_sums(const float* v_bct, const float* dout, const float* scale_c, const float* shift_c, const float* mean_c,
                             double* sums_c2, void* workspace, int B, int C, int T, int P, void* stream);
int ign_bn_elu_pool_bwd_apply(const float* v_bct, const float* dout, const float* scale_c, const float* shift_c, const float* ka_c,
                              const float* kb_c, const float* kc_c, float* dv_bct, int B, int C, int T, int P, void* stream);
int ign_bn_fold_fwd(const double* sums_c2, const float* alpha_c, const float* cshift_c, const float* gamma_c, const float* beta_c,
                    float* scale_c, float* shift_c, double* fold_c2, float* running_mean, float* running_var, int C, long long n,
                    float eps, float momentum, void* stream);
int ign_bn_fold_bwd(const double* sums_c2, const double* fold_c2, const float* alpha_c, const float* gamma_c, float* ka_c, float* kb_c,
                    float* kc_c, float* dgamma_c, float* dbeta_c, float* dalpha_c, int C, long long n, float eps, void* stream);

/* ---- FCN expert: channels-last 1-D convolution as an implicit GEMM on the fp32 matrix cores, BatchNorm + ReLU
 * folded into the GEMM prologues / epilogues (csrc/ign_clconv_{f32,x6}.hip, ign_bn.hip).  Replaces IGN/model/FullyConvNet.py:31-59
 * (3 x [Conv1d -> BatchNorm1d -> ReLU] -> AdaptiveAvgPool1d) and its autograd.  Activations are (B, T, C) row-major
 * (the loader's layout); a 'valid' convolution: Tout = Tin - k + 1.  All sums are fixed-order (deterministic).
 *
 * ign_clconv_pack_weights: w (Co,Ci,k) [torch Conv1d layout] -> wt_fwd (Co, k*Ci) with kk = j*Ci + ci, and (if not
 *   NULL) wt_dgrad (Ci, k*Co) with kk = jj*Co + co holding w[co][ci][k-1-jj].
 * ign_clconv_fwd: y[b,t,co] = bias[co] + sum_{j,ci} in[b,t+j,ci] w[co,ci,j], where in = x, or relu(pro_a[ci]*x + pro_b[ci])
 *   when pro_a/pro_b are given (the BatchNorm affine + ReLU of the block below, applied while staging).
 *   stat_part (ign_clconv_mtiles(B*Tout), 2, Co), nullable: per-tile sum_y and sum_y^2 for this block's BatchNorm.
 * ign_clconv_dgrad: given dyp (B, Tout + 2(k-1), Co) = dL/dy zero-padded by k-1 rows on both sides of every sample,
 *   computes dL/dz[b,t,ci] (z = this conv's input = relu(a_in*y_in + b_in)) and, in the epilogue, the ReLU-masked
 *   g_in = dL/dz * [a_in*y_in + b_in > 0] (B,Tin,Ci) plus stat_part (ign_clconv_mtiles(B*Tin), 2, Ci) = per-tile
 *   sum g_in and sum g_in*(y_in - mean_in)*invstd_in   (BatchNorm backward sums of the block below).
 * ign_clconv_wgrad: dw[co,ci,j] = sum_{b,t} dy[b,t,co] in[b,t+j,ci]  (in as for fwd; dy rows are read from a buffer
 *   padded by dy_pad rows per side).  Co % 4 == 0.  workspace: ign_clconv_wgrad_workspace_bytes().                  */
long long ign_clconv_mtiles(long long M);
int ign_clconv_pack_weights(const float* w_oik, float* wt_fwd, float* wt_dgrad, int Co, int Ci, int k, void* stream);
int ign_clconv_fwd(const float* x, const float* wt_fwd, const float* bias, const float* pro_a, const float* pro_b,
                   float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream);
int ign_clconv_dgrad(const float* dyp, const float* wt_dgrad, const float* y_in, const float* a_in, const float* b_in,
                     const float* mean_in, const float* invstd_in, float* g_in, float* stat_part,
                     int B, int Tin, int Ci, int Co, int k, void* stream);
/* Split-bf16 ("x6") variants of fwd / dgrad: the same maths at fp32 rounding-level accuracy on the bf16 matrix cores.
 * gfx950 executes fp32-input MFMA at the fp32 vector rate; here every fp32 operand is split exactly into three bf16 terms
 * and the six partial products of weight >= 2^-16 are accumulated in fp32 (v_mfma_f32_32x32x16_bf16): 2.7x the fp32-MFMA
 * throughput, error vs float64 at the level of fp32 accumulation (tests/test_gpu_fcn.py).  The weights arrive pre-split:
 * ign_clconv_pack_weights_x3 writes wt3_fwd (ign_clconv_x3_elems(Co, Ci, k) bf16) and, if not NULL, wt3_dgrad
 * (ign_clconv_x3_elems(Ci, Co, k) bf16) in step-block-major order: the 3 planes x 128 rows x 16 channels that one (n-tile,
 * 16-channel chunk, tap) step of the kernel consumes are one contiguous 12 KB block (zero in padded rows / channels).  Activations stay fp32 in HBM
 * and are split while they are staged into LDS -- once per (128 + k - 1)-row span and 16-channel chunk, shared by all k taps.
 * The x6 kernels tile every sample separately: stat_part has ign_clconv_x6_mtiles(B, Tout) [fwd] resp. (B, Tin) [dgrad]
 * rows.  k <= 16.                                                                                                       */
int ign_clconv_kpad(int C);                          /* channels rounded up to 16 */
long long ign_clconv_x3_elems(int rows, int chans, int k);   /* bf16 elements of a packed set: fwd (Co, Ci, k), dgrad (Ci, Co, k) */
long long ign_clconv_x6_mtiles(int B, int rows);     /* rows of stat_part for the x6 kernels: B * ceil(rows / 128) */
int ign_clconv_pack_weights_x3(const float* w_oik, void* wt3_fwd, void* wt3_dgrad, int Co, int Ci, int k, void* stream);
int ign_clconv_fwd_x6(const float* x, const void* wt3_fwd, const float* bias, const float* pro_a, const float* pro_b,
                      float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream);
int ign_clconv_dgrad_x6(const float* dyp, const void* wt3_dgrad, const float* y_in, const float* a_in, const float* b_in,
                        const float* mean_in, const float* invstd_in, float* g_in, float* stat_part,
                        int B, int Tin, int Ci, int Co, int k, void* stream);
/* Weight gradient on the bf16 matrix cores (same split): operands staged row-major and read with the transposing LDS read
 * ds_read_b64_tr_b16; one staged input span serves all k taps.  k in {2, 3, 5, 8} (the FCN expert's kernels), Co % 4 == 0;
 * k = 1 (a Linear layer over the B*Tin rows; needs Ci % 4 == 0, dy_pad == 0, no prologue) runs 128 x 128 tiles instead.
 * Same arguments and result as ign_clconv_wgrad.                                                                        */
size_t ign_clconv_wgrad_x6_workspace_bytes(int B, int Tin, int Ci, int Co, int k);
int ign_clconv_wgrad_x6(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                        float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream);
/* Deferred reduction: ign_clconv_wgrad_x6 / _bf16 called with dw_oik == NULL (k > 1) leave their ign_clconv_wgrad_x6_nsplit()
 * partial slabs in `workspace`; ign_clconv_wgrad_reduce_multi then reduces up to 8 layers in ONE launch (same ascending-order
 * sums as the per-layer reduction: identical bits).  Tables are host arrays.                                                  */
int ign_clconv_wgrad_x6_nsplit(int B, int Tin, int Ci, int Co, int k);
int ign_clconv_wgrad_reduce_multi(int n, const void* const* part, float* const* dw_oik, const int* nsplit, const int* Co,
                                  const int* Ci, const int* k, void* stream);
/* ign_clconv_pack_weights_x3 for up to 8 layers in ONE launch; counters[l] (nullable table / entries): an int64 device counter
 * incremented by one -- nn.BatchNorm1d.num_batches_tracked of the BatchNorm behind layer l in training mode
 * (IGN/model/FullyConvNet.py:31-50 builds Conv1d + BatchNorm1d pairs).                                                       */
int ign_clconv_pack_weights_x3_multi(int n, const float* const* w_oik, void* const* wt3_fwd, void* const* wt3_dgrad,
                                     const int* Co, const int* Ci, const int* k, long long* const* counters, void* stream);

/* ---- Two-plane fp16 GEMMs ("h3"): fp32 accuracy from THREE products per element pair instead of six.
 * Every fp32 operand x of a GEMM is first multiplied by a power of two s (exact) chosen from an upper bound of the tensor's
 * magnitude so that 2^13 <= bound * s < 2^14, then split into two fp16 terms h0 = fp16(s x), h1 = fp16(s x - h0) (the residual
 * is an exact fp32 subtraction), and the products a0 b0 + a0 b1 + a1 b0 are accumulated in fp32 by v_mfma_f32_32x32x16_f16; the
 * epilogue multiplies by 1 / (s_a s_b) (exact).  What is dropped is O(2^-22 |a b|) per product -- at the level of the fp32
 * accumulation error of a K ~ 1000 reduction; measured against float64 the results are as close as the six-product bf16 kernels
 * and the fp32-MFMA kernels (tests/test_gpu_fcn.py runs every case on all three).  Without the scaling fp16's narrow exponent
 * range would lose the second term of small operands (gradients of 1e-5: 5e-4 relative error); with it the absolute error of
 * an element is <= 2^-22 of the tensor's bound.  Bounds are device scalars (`slots`: 4 floats per layer, see ign_fcn_scan):
 *   weights        max |W|                       -- ign_fcn_scan, from the parameters
 *   layer input    l = 0: max |x| (ign_absmax);  l > 0: max_c(|gamma_c| sqrt(R - 1) + |beta_c|) of the BatchNorm in front, a HARD
 *                  bound of relu(gamma yhat + beta): a value standardised with the batch's own statistics over R rows cannot
 *                  exceed sqrt(R - 1) (training mode only; eval mode keeps the bf16 kernels)
 *   dL/dy          max |dL/dy|, taken by ign_bn_bwd_apply_amax as it writes the tensor (integer atomicMax on the bit patterns of
 *                  non-negative floats: exact and order-independent, so results stay bitwise reproducible)
 * Supported magnitudes: the scale exponent is clamped to +-60, i.e. bounds from 2^-46 (1.4e-14; below it the operand loses
 * precision gradually and finally reads as zero -- gradients of that size no longer move an fp32 weight) to 2^74 (1.9e22; a
 * tensor beyond it overflows fp16 and the result is NaN, as loudly as the diverged run that produced it).  A bound of 0 or a
 * non-finite bound selects scale 1.  A bound should BE an upper bound; the scaled bound lies in [2^13, 2^14) and fp16 reaches
 * 65504, so an element up to 3.9 times the bound is still split exactly like any other (the host side relies on a factor 1.13 of
 * this headroom where dL/du of a GELU inherits the bound of dL/dy), an element beyond 4 times the bound overflows to infinity.
 * Replaces the same reference lines as ign_clconv_fwd / _dgrad / _wgrad (IGN/model/FullyConvNet.py:31-59 and its autograd).   */
int ign_absmax(const float* x, long long n, float* slot /* max'ed into, caller zeroes */, void* stream);
/* ... of up to 16 tensors in one launch (a model's dense-layer weights once per step): slots[i] = max |x[i][0..count[i])|.     */
int ign_absmax_multi(int n, const float* const* x, const long long* count, float* const* slots, void* stream);
/* ign_fcn_scan also clears `zero[0..nzero)` (nullable / 0): the identically-zero gradients of the convolution biases in front of
 * a batch-statistics BatchNorm are views of that buffer, so the backward needs no fill launch.                               */
int ign_fcn_scan(int nl, const float* const* w, const long long* nw, const float* const* gamma_prev, const float* const* beta_prev,
                 const int* C_prev, const long long* R_prev, float* slots /* (nl, 4): written */, float* zero, long long nzero,
                 void* stream);
int ign_clconv_pack_weights_h2_multi(int n, const float* const* w_oik, void* const* wt_fwd, void* const* wt_dgrad, const int* Co,
                                     const int* Ci, const int* k, long long* const* counters, const float* const* w_bounds,
                                     void* stream);
int ign_clconv_fwd_h3(const float* x, const void* wt_h2, const float* bias, const float* pro_a, const float* pro_b, float* y,
                      float* stat_part, const float* bound_in, const float* bound_w, int B, int Tin, int Ci, int Co, int k,
                      void* stream);
/* the same, additionally max'ing max |y| into the device float amax_out (nullable; the caller zeroes it): the magnitude bound of
 * the output for the next GEMM that consumes it -- taken in the epilogue instead of by a pass over the tensor (ign_absmax)     */
int ign_clconv_fwd_h3_amax(const float* x, const void* wt_h2, const float* bias, const float* pro_a, const float* pro_b, float* y,
                           float* stat_part, const float* bound_in, const float* bound_w, float* amax_out, int B, int Tin, int Ci,
                           int Co, int k, void* stream);
/* du = (g W) * gelu'(u) in ONE GEMM: the input gradient of the feed-forward's second dense layer z = gelu(u) W^T + b
 * (IGN/layers/Transformer_EncDec.py:46-47) THROUGH the activation -- the exact-GELU derivative Phi(u) + u phi(u) is applied in the
 * epilogue, so dL/dy is never written and aten::gelu_backward's three passes over the (rows, d_ff) tensors disappear.  g (M, Co),
 * wd_h2 = the packed TRANSPOSED weight (as for the input gradient via ign_clconv_fwd_h3), u / du (M, Ci); Ci % 256 == 0,
 * Co % 4 == 0; amax_out nullable (max |du|).  f16x3 arithmetic only (IGN_E_UNSUP otherwise: callers keep the two-kernel route).  */
int ign_linear_dgrad_gelu_h3(const float* g, const void* wd_h2, const float* u, float* du, const float* bound_g, const float* bound_w,
                             float* amax_out, long long M, int Co, int Ci, void* stream);
int ign_clconv_dgrad_h3(const float* dyp, const void* wt_h2_dgrad, const float* y_in, const float* a_in, const float* b_in,
                        const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, const float* bound_dy,
                        const float* bound_w, int B, int Tin, int Ci, int Co, int k, void* stream);
int ign_clconv_wgrad_h3(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b, float* dw_oik,
                        void* workspace, const float* bound_dy, const float* bound_x, int B, int Tin, int Ci, int Co, int k,
                        void* stream);
int ign_linear_wgrad_h3(const float* dy, const float* x, float* dw, float* db, void* workspace, const float* bound_dy,
                        const float* bound_x, long long M, int Ci, int Co, void* stream);
int ign_bn_bwd_apply_amax(const float* g, const float* y, const float* a, const float* mean, const float* invstd,
                          const float* dbeta, const float* dgamma, float* dyp, float* amax_slot, int B, int T, int C, int pad,
                          int training, void* stream);

/* The three split-bf16 GEMMs with ONE product per step: operands rounded to bf16 (round-to-nearest-even), products and sums in
 * fp32 -- the arithmetic of the reference's default bf16-autocast mode (IGN/exp/experiment_classification.py:319; `--amp`
 * switches it OFF).  Same packed weights (plane 0 is read), same arguments, same workspace as the *_x6 entry points.        */
int ign_clconv_fwd_bf16(const float* x, const void* wt3_fwd, const float* bias, const float* pro_a, const float* pro_b,
                        float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream);
int ign_clconv_dgrad_bf16(const float* dyp, const void* wt3_dgrad, const float* y_in, const float* a_in, const float* b_in,
                          const float* mean_in, const float* invstd_in, float* g_in, float* stat_part,
                          int B, int Tin, int Ci, int Co, int k, void* stream);
int ign_clconv_wgrad_bf16(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                          float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream);
/* A Linear layer's weight AND bias gradient in one pass over dy (M, Co) and x (M, Ci): dW = dy^T x on the split-bf16 k = 1 kernel,
 * db = column sums of dy accumulated by the tiles that stage those values anyway (db may be NULL).  Workspace:
 * ign_clconv_wgrad_x6_workspace_bytes(1, M, Ci, Co, 1).  _bf16: the single-product (autocast) form.  Co % 4 == Ci % 4 == 0. */
int ign_linear_wgrad_x6(const float* dy, const float* x, float* dw, float* db, void* workspace, long long M, int Ci, int Co,
                        void* stream);
int ign_linear_wgrad_bf16(const float* dy, const float* x, float* dw, float* db, void* workspace, long long M, int Ci, int Co,
                          void* stream);
size_t ign_clconv_wgrad_workspace_bytes(int B, int Tin, int Ci, int Co, int k);
int ign_clconv_wgrad(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                     float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream);

/* BatchNorm1d glue of the same expert (IGN/model/FullyConvNet.py:33,40,47; torch.nn.BatchNorm1d semantics: biased batch
 * variance for normalisation, unbiased for running_var, momentum update).  C % 4 == 0, C <= 1024.
 * finalize_fwd: partial sums (nparts, 2, C) over R rows -> a = gamma*invstd, b = beta - a*mean (so BN(y) = a*y + b),
 *   mean, invstd; updates running_mean / running_var in place when they are not NULL.
 * affine_eval: the same a, b, mean, invstd from the running statistics (module.eval()).
 * relu_pool_fwd: pooled[b,c] = mean_t relu(a_c*y[b,t,c] + b_c)            (BatchNorm + ReLU + AdaptiveAvgPool1d(1)).
 * relu_pool_bwd: g[b,t,c] = gpool[b,c]/T * [a_c*y + b_c > 0] and per-block partials (ign_bn_relu_pool_bwd_parts(B,T), 2, C)
 *   of sum g and sum g*yhat.
 * finalize_bwd: partials -> dbeta = sum g, dgamma = sum g*yhat.
 * bwd_apply: dy = a*(g - (dbeta + yhat*dgamma)/R) [training] or a*g [eval], written into (B, pad + T + pad, C) with the
 *   pad rows zeroed (the layout ign_clconv_dgrad / ign_clconv_wgrad read).                                          */
int ign_bn_finalize_fwd(const float* part, int nparts, long long R, int C, const float* gamma, const float* beta,
                        float eps, float momentum, float* running_mean, float* running_var,
                        float* a, float* b, float* mean, float* invstd, void* stream);
int ign_bn_affine_eval(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                       float eps, int C, float* a, float* b, float* mean, float* invstd, void* stream);
int ign_bn_relu_pool_fwd(const float* y, const float* a, const float* b, float* pooled, int B, int T, int C, void* stream);
/* The same with the FCN expert's class head in the same launch: logits[b][n] = bias[n] + sum_c pooled[b][c] W[n][c]  (W (N,C),
 * bias nullable; replaces nn.Linear at IGN/model/FullyConvNet.py:50,58 on top of the pooling): a sample's pooled row is complete
 * inside its block.  `pooled` is still written (the head's weight gradient needs it: ign_head_bwd).                          */
int ign_bn_relu_pool_head_fwd(const float* y, const float* a, const float* b, float* pooled, const float* W, const float* bias,
                              float* logits, int B, int T, int C, int N, void* stream);
long long ign_bn_relu_pool_bwd_parts(int B, int T);
int ign_bn_relu_pool_bwd(const float* y, const float* gpool, const float* a, const float* b, const float* mean,
                         const float* invstd, float* g, float* part, int B, int T, int C, void* stream);
int ign_bn_finalize_bwd(const float* part, int nparts, int C, float* dbeta, float* dgamma, void* stream);
int ign_bn_bwd_apply(const float* g, const float* y, const float* a, const float* mean, const float* invstd,
                     const float* dbeta, const float* dgamma, float* dyp, int B, int T, int C, int pad, int training,
                     void* stream);

/* Per-kernel HIP-event timing (measurement only; off by default).  When enabled every kernel launch made by
 * this library is bracketed by hipEventRecord on the caller's stream.  ign_timing_read() waits for the recorded
 * events of `label` ("shp_fwd", "shp_bwd", "reduce_parts", "instnorm", "attn_fwd", "attn_bwd_dkdv", "attn_bwd_dq",
 * "attn_delta", "head_fwd", "head_bwd_x", "head_bwd_w", "adam",
 * "conv1_sumsq_fwd", "conv1_sumsq_bwd", "dwconv1d", "dwconv1d_bwd_w"), and returns the accumulated
 * device milliseconds and launch count since the last enable.  Not for use under graph capture.             */
int ign_timing_enable(int on);
int ign_timing_read(const char* label, double* total_ms, long long* launches);

#ifdef __cplusplus
}
#endif
#endif /* IGN_ABI_H */
