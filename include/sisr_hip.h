/* libsisr_hip.so -- C ABI of the MI355X (gfx950) SISR forward/backward kernels.
 *
 * Drop-in boundary.  The reference has no FFI: its hot path is the nn.Module tree built by the
 * model handlers (Code/SISR/models/advanced/architectures.py, attention_manipulators/architectures.py)
 * and executed by BaseModel.run_train / run_eval (Code/SISR/models/__init__.py:466-522).  Each entry
 * point below replaces the ATen kernels one of those modules dispatches to; the citation on each
 * function names the reference module.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - The caller owns every buffer (device pointers unless stated); nothing here allocates, frees,
 *    synchronises or throws.  Workspaces are sized by the *_workspace_bytes() queries.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *  - Return 0 on success, negative on error: -1 bad argument, -2 misaligned pointer/stride
 *    (16-byte alignment is required for activations), -3-16*hipError launch failure, -4 unsupported
 *    shape (channel counts must be multiples of 64 on the matrix-core kernels).
 *  - Re-entrant and thread-safe: no global state (kernel-variant choices are per-call `select` arguments; the
 *    diagnostic entry points at the end exist only in libsisr_hip_diag.so, built with -DSISR_DIAG).
 *  - fp32 everywhere.  Activations are NHWC in 64-channel chunks described by a 6-element int64
 *    "view": {sB, sH, sW, chi, clo, cdiv}; element (b,h,w,chunk q,c) is at
 *        base + b*sB + h*sH + w*sW + (q / cdiv)*chi + (q % cdiv)*clo + c        (strides in floats)
 *    Plain NHWC with C channels: {H*W*C, W*C, C, 0, 64, 1<<30}.  The output of
 *    conv(64 -> 64 r^2) + PixelShuffle(r) is the view {rH*rW*64, r*rW*64, r*64, rW*64, 64, r} of the
 *    [B][rH][rW][64] tensor, with chunk q holding original channels {c*r*r + q}: the shuffle is an
 *    address map, never a copy (ref: advanced/common.py:28-31).
 *  - Conv weights stay in the reference's OIHW layout in the caller's parameters; kernels take
 *    either a packed copy (sisr_pack_conv3x3) or generic strides: element (o, i, tap) is read at
 *    w[o*so + i*si + (flip ? 8-tap : tap)], with channel maps o = n*perm_n + q*perm_q for in-chunk
 *    index n of chunk q.  Forward: so = Cin*9, si = 9, flip = 0.  Input gradient: roles swapped
 *    (so = 9, si = Cin*9) and flip = 1.
 */
#ifndef SISR_HIP_H
#define SISR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- 3x3 convolution, Cin and Cout multiples of 64 (fp32 MFMA 32x32x2) ------------------------
 * ref: advanced/common.py:5-8 default_conv -> nn.Conv2d(k=3, pad=1) forward, and (with role-swapped,
 * flipped packed weights) its input gradient.  Fused:  y = mask( relu?( conv(x*in_scale+in_shift) + bias )
 * * alpha * out_scale ) + res, optional per-wave GAP partial sums of y.
 *   bias index    = n*bias_n + q*bias_q   (nullable)
 *   in_scale/shift: [B][cin]  (nullable; applied to in-image pixels only, zero padding stays zero)
 *   out_scale     : [B][cout] (nullable)          res, mask: same view as y (nullable)
 *   gap_partial   : [B][sisr_conv3x3_c64_gap_parts(H,W)][cout] (nullable)
 *   relu          : 0 none, 1 ReLU, 2 LeakyReLU(0.2); + 4: `mask` is LeakyReLU(0.2)'s derivative (slope 0.2 where the
 *                   masking map is <= 0) instead of ReLU's.  Codes 2 and 4 exist on the fp32 kernels only (SFTMD). */
int sisr_pack_conv3x3(const float* w, float* packed, int cout, int cin, int64_t so, int64_t si, int flip_taps,
                      int out_perm_n, int out_perm_q, int in_perm_n, int in_perm_q, void* stream);
/* forward and input-gradient packings of one weight in one launch (shuffle_r > 1: conv feeds PixelShuffle(r)) */
int sisr_pack_conv3x3_both(const float* w, float* packed_fwd, float* packed_dgrad, int cout, int cin, int shuffle_r,
                           void* stream);
/* every conv weight of a network in ONE launch (a training step repacks them all after the optimiser update).
 * jobs_device: device array of n_jobs records {const float* w; void* packed_fwd; void* packed_dgrad; int32 cout, cin,
 * shuffle_r, first_block} (sisr_pack_job_bytes() each; the caller builds it once per network); job j owns blocks
 * [first_block_j, first_block_j+1) of 256 packed elements each; total_blocks = their sum.  bf16 = 1: the bf16
 * packings of sisr_pack_conv3x3_bf16_both; 2: the three-plane packings of sisr_pack_conv3x3_x3_both. */
size_t sisr_pack_job_bytes(void);
int sisr_pack_conv3x3_many(const void* jobs_device, int n_jobs, int total_blocks, int bf16, void* stream);
int sisr_conv3x3_c64_gap_parts(int H, int W);
int sisr_conv3x3_c64(const float* x, const int64_t* xview, const float* wpacked, const float* bias, int bias_n,
                     int bias_q, float* y, const int64_t* yview, const float* res, const float* mask,
                     const float* in_scale, const float* in_shift, const float* out_scale, float alpha, int relu,
                     float* gap_partial, const float* gate_add, float* gate_out, const float* dot, int B, int H,
                     int W, int cin, int cout, const void* ca_tail, int select, void* stream);
/* ca_tail (nullable HOST pointer to a sisr_ca_tail, copied into the launch; 64 -> 64 only): the workgroup that finishes a
 * sample last turns the partial sums this launch writes into the channel-attention gate (backward = 0: the forward gate
 * from gap_partial -- sisr_ca_gate_fwd's outputs) or into the gate's backward (backward = 1: from the `dot` partial sums
 * -- sisr_ca_gate_bwd's outputs; the last sample's finisher sums the parameter gradients over the batch).  Same
 * arithmetic and summation order as those two entry points.  counter: B + 1 zero-initialised device words, returned to
 * zero; workspace (backward): B rows of 80 floats (per sample: dz2 [64], then dz1 [hidden <= 16]), so the workspace of a
 * launch over samples b0 .. b1 is rows b0 .. b1 of the whole batch's.
 * head != 0 turns the same record into a gate HEAD: the launch that CONSUMES a gate computes it first -- every workgroup for
 * its own sample, from the partial sums a previous launch wrote (head_part [B][head_parts][64]) -- instead of a gate launch of
 * its own between the two convs: backward = 0 on a gate_add / gate_out launch (in_scale must be g_out: it is filled here, with
 * s_out / hid_out / ca_out), backward = 1 on an in_scale + in_shift + mask launch (in_shift must be `shift`: filled here, with
 * dmul and the workspace; parameter gradients: sisr_ca_gate_bwd_params_batch).  counter and dw / db fields unused. */
typedef struct {
  int backward, hidden;
  float inv_hw;
  const float *w1, *b1, *w2, *b2, *mul; /* gate parameters (b1, b2 unused backward), optional second factor [B][64] */
  const float *s, *hid, *ca;            /* backward: what the forward kept */
  float *s_out, *hid_out, *ca_out, *g_out;       /* forward outputs */
  float *shift, *dmul, *dw1, *db1, *dw2, *db2;   /* backward outputs */
  float* workspace;
  unsigned* counter;
  const float* head_part;
  int head_parts, head;
} sisr_ca_tail;
size_t sisr_ca_tail_bytes(void);
/* select: 0 (= 4) issue-lean kernel, tile height chosen by grid size, general kernel as fallback; 5 / 6 the same with
 * the 4-row / 2-row tile forced (bit-identical results; A/B measurements and tests); 2 general kernel only.
 * 8 / 9 / 10: the caller asserts structural zeros in the packed weight (SFTMD's merged convs) and the kernel skips them:
 *   8  64 -> 128 block-diagonal (output chunk q contracts input channels 32q .. 32q+31 only), plain epilogue;
 *   9  128 -> 64 whose input channels >= 80 are zero (second chunk: first 16 channels only), LeakyReLU epilogue;
 *  10  128 -> 64, the transpose of 8 (input chunk c feeds output channels 32c .. 32c+31 only), LeakyReLU' mask, no bias. */
/* gate_add / gate_out / dot (all nullable; 64 -> 64, x and y in one layout) fuse the gated-residual chain of
 * RCAB / QRCAB stacks (ref: advanced/architectures.py:68-71, :107-110) into the neighbouring convs:
 *   gate_add + gate_out : the conv reads  x * in_scale[b,c] + gate_add  (the previous block's `res * y + x`) and
 *                         writes that map to gate_out once -- no separate gate pass in forward;
 *   dot                 : gap_partial receives sum(v * dot) instead of sum(v): the gate gradient sum(dY * t) of
 *                         the block below, taken while its dY is being produced -- no separate reduction pass. */

/* ---- 3x3 convolution weight + bias gradient (fp32 MFMA), deterministic two-stage reduction ------
 * ref: autograd's convolution_backward for default_conv (loss.backward(), SISR/models/__init__.py:483).
 *   dW = alpha * sum X (x) dY',  db = alpha * sum dY',  dY' = dY*dy_scale[b][co] + dy_shift[b][co] */
size_t sisr_wgrad3x3_c64_workspace_bytes(int B, int H, int W, int cin, int cout);
int sisr_wgrad3x3_c64(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                      const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so, int64_t si,
                      int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n, int in_perm_q, float* dbias,
                      int bias_n, int bias_q, float* workspace, size_t workspace_bytes, int B, int H, int W, int cin,
                      int cout, unsigned long long active_units, void* stream);
/* active_units: 0 = the whole gradient.  Otherwise bit ((cin_chunk * cout_chunks + cout_chunk) * 4 + ci_half * 2 + co_half)
 * selects the 32 x 32-channel blocks (x 9 taps) to compute; the others are neither computed nor written (a caller whose
 * weight is structurally sparse -- SFTMD's merged convs -- never reads them).  At most 64 blocks; every bias half needs
 * one active block. */
/* Several 64 -> 64 weight gradients of ONE geometry in one launch (plain OIHW dw [64][64][3][3], db [64], alpha 1): at a few
 * tiles per GPU a single gradient's launch is mostly ramp, slab epilogue and drain; batched, every workgroup walks a long
 * run of tiles of its own job.  jobs: HOST array of njobs <= sisr_wgrad3x3_c64_batch_max() records
 * { x, dy, dy_scale (nullable), dy_shift (nullable), dw, dbias (nullable) } of device pointers, all with the views given. */
typedef struct {
  const float* x;
  const float* dy;
  const float* dy_scale;
  const float* dy_shift;
  float* dw;
  float* dbias;
} sisr_wgrad_job;
size_t sisr_wgrad_job_bytes(void);
int sisr_wgrad3x3_c64_batch_max(void);
size_t sisr_wgrad3x3_c64_batch_workspace_bytes(int njobs, int B, int H, int W);
int sisr_wgrad3x3_c64_batch(const void* jobs, int njobs, const int64_t* xview, const int64_t* dyview, float* workspace,
                            size_t workspace_bytes, int B, int H, int W, void* stream);

/* ---- RGB-side 3x3 convolutions (fp32 VALU, HBM-bound) ------------------------------------------
 * ref: head = default_conv(3, n_feats), tail[-1] = default_conv(n_feats, 3)
 * (advanced/architectures.py:141,150-152).  x/y of the 3-channel side are planar NCHW. */
int sisr_conv3x3_cin3(const float* x, const float* w, int64_t so, int64_t si, int flip_taps, const float* bias,
                      float* y, const int64_t* yview, int B, int H, int W, int cout, void* stream);
int sisr_conv3x3_cout3(const float* x, const int64_t* xview, const float* w, int64_t so, int64_t si, int flip_taps,
                       const float* bias, float* y, int B, int H, int W, int cin, void* stream);
size_t sisr_corr3x3_c3_workspace_bytes(int B, int H, int W, int channels);
int sisr_corr3x3_c3(const float* P, const float* Q, const int64_t* qview, float alpha, float* dw, int64_t so,
                    int64_t si, int flip_taps, int a_is_out, float* dbias, float* workspace, size_t workspace_bytes,
                    int B, int H, int W, int channels, void* stream);

/* ---- channel attention gate (64 channels) --------------------------------------------------------
 * ref: advanced/architectures.py:13-32 CALayer; attention_manipulators/architectures.py:125 QCALayer 'standard'.
 * fwd: s = inv_hw*sum(partials); hid = relu(W1 s + b1); ca = sigmoid(W2 hid + b2); g = ca * (mul or 1)
 * bwd: from partial sums of dg = sum_hw dOut*t: shift = dL/ds * inv_hw, dmul = dg*ca, dW1 db1 dW2 db2 */
int sisr_ca_gate_fwd(const float* gap_partial, int parts, int B, float inv_hw, const float* w1, const float* b1,
                     const float* w2, const float* b2, int channels, int hidden, const float* mul, float* s,
                     float* hid, float* ca, float* g, void* stream);
size_t sisr_ca_gate_bwd_workspace_bytes(int B);
int sisr_ca_gate_bwd(const float* dg_partial, int parts, int B, float inv_hw, const float* w1, const float* w2,
                     int channels, int hidden, const float* s, const float* hid, const float* ca, const float* mul,
                     float* shift, float* dmul, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                     unsigned* counter, void* stream);
/* workspace: B rows of 80 floats (sisr_ca_gate_bwd_workspace_bytes), per sample dz2 [64] then dz1 [hidden].
 * dw1 = db1 = dw2 = db2 = NULL (counter may be NULL too): only the per-sample part (shift, dmul, dz2 / dz1 into the
 * workspace); the parameter gradients of up to sisr_ca_gate_bwd_params_batch_max() such calls are then taken in ONE launch.
 * jobs: HOST array of { that call's workspace, hid, s, dw1, db1, dw2, db2 } (sisr_ca_param_job_bytes() each). */
int sisr_ca_gate_bwd_params_batch_max(void);
size_t sisr_ca_param_job_bytes(void);
int sisr_ca_gate_bwd_params_batch(const void* jobs, int njobs, int B, int hidden, void* stream);
/* counter: one zero-initialised device word owned by the caller and reused across calls on a stream (the kernel
 * returns it to zero): the sample blocks count themselves on it and the last one sums the parameter gradients,
 * so the whole gate backward is ONE launch on the block's serial backward chain. */

/* ---- meta-attention gate --------------------------------------------------------------------------
 * ref: attention_manipulators/q_layer.py:4-43 ParaCALayer: m = sigmoid(V2 act(V1 md + c1) + c2) */
int sisr_meta_gate_fwd(const float* md, int B, int M, int hidden, int channels, const float* v1, const float* c1,
                       const float* v2, const float* c2, int relu, float* hid, float* m, void* stream);
size_t sisr_meta_gate_bwd_workspace_bytes(int B, int hidden, int channels);
int sisr_meta_gate_bwd(const float* dm, const float* m, const float* hid, const float* md, int B, int M, int hidden,
                       int channels, const float* v1, const float* v2, int relu, float* dv1, float* dc1, float* dv2,
                       float* dc2, float* dmd, float* workspace, void* stream);
/* All L meta-attention layers of a network at once (the gates depend on the metadata and each layer's own weights
 * only -- ref: attention_manipulators/architectures.py:172-180 applies q_node to the same `metadata` in every
 * QRCAB): *_table are device arrays of L device pointers to the layers' parameters; hid [L][B][hidden], m and dm
 * [L][B][channels]; gradients come back as [L][hidden*M], [L][hidden], [L][channels*hidden], [L][channels].
 * No metadata gradient. */
int sisr_meta_gate_many_fwd(const float* md, int B, int M, int hidden, int channels, int layers,
                            const float* const* v1_table, const float* const* c1_table, const float* const* v2_table,
                            const float* const* c2_table, int relu, float* hid, float* m, void* stream);
size_t sisr_meta_gate_many_bwd_workspace_bytes(int B, int hidden, int channels, int layers);
int sisr_meta_gate_many_bwd(const float* dm, const float* m, const float* hid, const float* md, int B, int M, int hidden,
                            int channels, int layers, const float* const* v1_table, const float* const* v2_table,
                            int relu, float* dv1, float* dc1, float* dv2, float* dc2, float* workspace, void* stream);
/* the same, with each layer's four gradients written to the addresses in four device tables of `layers` pointers (e.g.
 * slices of an optimiser's flat gradient arena) instead of into [layers][...] arrays */
int sisr_meta_gate_many_bwd_scatter(const float* dm, const float* m, const float* hid, const float* md, int B, int M,
                                    int hidden, int channels, int layers, const float* const* v1_table,
                                    const float* const* v2_table, int relu, float* const* dv1_table, float* const* dc1_table,
                                    float* const* dv2_table, float* const* dc2_table, float* workspace, void* stream);

/* ---- generic gate MLP: the metadata-mixing QCALayer styles ---------------------------------------------
 * ref: attention_manipulators/architectures.py:105-127 (QCALayer.forward after avg_pool: 'modulate', 'max_concat',
 * 'softmax', 'mini_concat', 'extended_attention').  Layer k: z = W_k in_k + b_k with in_k = cat(v_k, metadata if cat[k]),
 * ReLU'd first if relu_in[k]; v_{k+1} = act[k](z), act 0 none / 1 ReLU / 2 sigmoid; final_mode 0 none / 1 softmax over
 * channels / 2 multiply by the metadata (M == C); then an optional per-(b,c) factor `mul` (the meta-attention gate).
 * desc: HOST pointer, copied into the launch; w/b device pointers (Conv2d 1x1 weights, [nout][nin + cat*M]).
 * acts [B][nin[0] + sum nout], yfin [B][C] are kept for the backward; workspace [B][sum nout]; dw/db HOST arrays of L
 * device pointers; dmd, dmul nullable. */
#define SISR_GATE_MLP_MAX_LAYERS 4
typedef struct {
  const float* w[SISR_GATE_MLP_MAX_LAYERS];
  const float* b[SISR_GATE_MLP_MAX_LAYERS];
  int nin[SISR_GATE_MLP_MAX_LAYERS], nout[SISR_GATE_MLP_MAX_LAYERS], cat[SISR_GATE_MLP_MAX_LAYERS],
      relu_in[SISR_GATE_MLP_MAX_LAYERS], act[SISR_GATE_MLP_MAX_LAYERS];
  int L, M, C, final_mode;
} sisr_gate_mlp;
size_t sisr_gate_mlp_desc_bytes(void);
int sisr_gate_mlp_fwd(const float* pool, const float* md, const float* mul, int B, const void* desc, float* acts,
                      float* yfin, float* y, void* stream);
int sisr_gate_mlp_bwd(const float* dy, const float* md, const float* mul, int B, const void* desc, const float* acts,
                      const float* yfin, float* workspace, float* dpool, float* dmd, float* dmul, float* const* dw,
                      float* const* db, void* stream);

/* ---- gated residual: y = t*g[b,c] + shift[b,c] + x  (g, shift, x nullable)
 * ref: the `x * y` / `res += x` tails of CALayer, RCAB, QRCAB, ParamResBlock; with shift it is also the
 * CALayer input gradient dy*g + dL/ds/HW.  gate_dg_partial: per-slice sums of dy*t (t NULL: of dy = GAP). */
int sisr_gate_residual_fwd(const float* t, const float* g, const float* shift, const float* x, float* y, int B,
                           long hw, int channels, void* stream);
int sisr_gate_dg_parts(long hw);
int sisr_gate_dg_partial(const float* dy, const float* t, float* part, int B, long hw, int channels, void* stream);
int sisr_sum_partials(const float* part, int parts, int B, int channels, float scale, float* out, void* stream);

/* ---- pixel attention, 64 channels, hidden 8 (ref: attention_manipulators/architectures.py:13-26 PALayer):
 * y = x * sigmoid(w2 . relu(W1 x + b1) + b2) per pixel; x, y contiguous [npix][64]. */
int sisr_pa_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, long npix,
                int channels, int hidden, void* stream);
size_t sisr_pa_bwd_workspace_bytes(long npix);
int sisr_pa_bwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* dy,
                float* dx, float* dw1, float* db1, float* dw2, float* db2, float* workspace, long npix, int channels,
                int hidden, void* stream);

/* ---- HAN attention modules ---------------------------------------------------------------------
 * ref: advanced/HAN_blocks.py:7-37 LAM_Module: x [B][N][chw] (N layer maps, any common layout);
 *      attn = softmax_j(max_j E_ij - E_ij), E = X X^T;  y = gamma*attn X + x.   N in {2..6, 8, 11}.
 * ref: advanced/HAN_blocks.py:40-76 CSAM_Module: x NHWC 64 channels; w27 = Conv3d weight [dc][dh][dw];
 *      y = x*(1 + gamma*sigmoid(conv3d(x) + bias)).  gamma / bias are DEVICE pointers (1-element parameters). */
size_t sisr_lam_workspace_bytes(int B, int N, long chw);
int sisr_lam_fwd(const float* x, const float* gamma, float* y, float* attn, float* workspace, int B, int N, long chw,
                 void* stream);
int sisr_lam_bwd(const float* x, const float* attn, const float* gamma, const float* dy, float* dx, float* dgamma,
                 float* workspace, int B, int N, long chw, void* stream);
int sisr_csam_fwd(const float* x, const float* w27, const float* bias, const float* gamma, float* y, int B, int H,
                  int W, int C, void* stream);
size_t sisr_csam_bwd_workspace_bytes(int B, int H, int W, int C);
int sisr_csam_bwd(const float* x, const float* w27, const float* bias, const float* gamma, const float* dy, float* dx,
                  float* dw27, float* dbias, float* dgamma, float* workspace, int B, int H, int W, int C, void* stream);

/* ---- bf16 matrix-core variants of the 64-channel-chunk conv and its weight gradient ----------------
 * Same arguments, views, prologue / epilogue options and results layout as sisr_conv3x3_c64 / sisr_wgrad3x3_c64
 * (ref: advanced/common.py:5-8 default_conv and its autograd backward).  Feature maps, biases and gradients stay
 * fp32 in HBM; the two MFMA operands are rounded to bf16 (round-to-nearest-even) on their way into LDS --
 * activations after the optional affine prologue, weights once at pack time -- products are exact and
 * accumulation is fp32 (the semantics of a bf16 autocast convolution).  The bias gradient is summed from the
 * unrounded fp32 values.  pack_conv3x3_bf16_both writes cout*cin*9 bf16 elements per packing. */
int sisr_pack_conv3x3_bf16_both(const float* w, void* packed_fwd, void* packed_dgrad, int cout, int cin, int shuffle_r,
                                void* stream);
int sisr_conv3x3_c64_bf16(const float* x, const int64_t* xview, const void* wpacked_bf16, const float* bias, int bias_n,
                          int bias_q, float* y, const int64_t* yview, const float* res, const float* mask,
                          const float* in_scale, const float* in_shift, const float* out_scale, float alpha, int relu,
                          float* gap_partial, const float* gate_add, float* gate_out, const float* dot,
                          int B, int H, int W, int cin, int cout, int select, void* stream);
/* select: 0 persistent double-buffered tile loop where it applies (64 -> 64, >= 1024 tiles), 1 per-tile kernel */
size_t sisr_wgrad3x3_c64_bf16_workspace_bytes(int B, int H, int W, int cin, int cout);
int sisr_wgrad3x3_c64_bf16(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                           const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so, int64_t si,
                           int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n, int in_perm_q, float* dbias,
                           int bias_n, int bias_q, float* workspace, size_t workspace_bytes, int B, int H, int W,
                           int cin, int cout, void* stream);

/* ---- bf16 STORAGE of the maps a residual group keeps (opt-in on top of the bf16 operand mode; BASELINE config 5) -------
 * ref: the maps are the reference's fp32 activations of advanced/architectures.py:48-71, :94-110 (RCAB / ResidualGroup);
 * the reference has no reduced-precision mode -- parity is pinned to the oracle's restatement of exactly this rounding.
 * A bf16 map has the View of its fp32 twin (strides in ELEMENTS) and 2-byte elements; it is passed as float* (16-byte
 * aligned).  storage bits of the conv: 1 = x / gate_add / gate_out, 2 = y, 4 = mask / dot, 8 = res are bf16 maps.  64 -> 64
 * only, always the persistent tile loop; only the combinations the fused group node launches exist (others: unsupported).
 * Weight gradient: storage 1 = x is a bf16 map, 3 = x and dY are.  sisr_f32_to_bf16 rounds a map to nearest even. */
int sisr_conv3x3_c64_bf16s(const float* x, const int64_t* xview, const void* wpacked_bf16, const float* bias, int bias_n,
                           int bias_q, float* y, const int64_t* yview, const float* res, const float* mask,
                           const float* in_scale, const float* in_shift, float alpha, int relu, float* gap_partial,
                           const float* gate_add, float* gate_out, const float* dot, int B, int H, int W, int storage,
                           void* stream);
int sisr_wgrad3x3_c64_bf16s(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                            const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so, int64_t si,
                            int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n, int in_perm_q, float* dbias,
                            int bias_n, int bias_q, float* workspace, size_t workspace_bytes, int B, int H, int W,
                            int cin, int cout, int storage, void* stream);
int sisr_f32_to_bf16(const float* src, void* dst, long n, void* stream); /* n a multiple of 8 */

/* ---- SAN attention modules ---------------------------------------------------------------------
 * Second-order channel attention, ref: advanced/SAN_blocks.py:244-302 SOCA + advanced/mpncov.py:12-112.
 *   covpool_fwd : cov[b] = (1/M) sum_p (x_p - mean[b]) x_p^T      x [B][M][64] channels-last, mean [B][64]
 *                 (= Covpool.forward's X I^ X^T without the M x M matrix)
 *   sqrtm_fwd   : Newton-Schulz square root (trace pre-normalised, `iters` iterations, post-compensated) and the
 *                 column means SOCA feeds to its gate: pooled [B][64]; `saved` keeps the iterates for backward
 *   sqrtm_bwd   : Sqrtm.backward from dL/dpooled; returns G + G^T (the form Covpool.backward consumes)
 *   soca_bwd_apply : dx = dy * gate[b,c] + (1/M) (G + G^T)(x - mean)   -- `y_cov * x` plus Covpool.backward
 * Non-local attention, ref: advanced/SAN_blocks.py:126-141 (_embedded_gaussian: f = theta^T phi, softmax over
 * keys, y = f g): theta [nb][nq][8], phi / g [nb][nk][8]; streaming softmax, lse [nb][nq] kept for backward,
 * dsum [nb][nq] is scratch. */
size_t sisr_covpool_workspace_bytes(int B, long hw);
int sisr_covpool_fwd(const float* x, const float* mean, float* cov, float* workspace, int B, long hw, int channels,
                     void* stream);
size_t sisr_sqrtm_saved_bytes(int B, int dim, int iters);
int sisr_sqrtm_fwd(const float* cov, float* saved, float* pooled, int B, int dim, int iters, void* stream);
int sisr_sqrtm_bwd(const float* cov, const float* saved, const float* dpooled, float* dcov_sym, int B, int dim,
                   int iters, void* stream);
int sisr_soca_bwd_apply(const float* dy, const float* gate, const float* x, const float* mean, const float* dcov_sym,
                        float* dx, int B, long hw, int channels, void* stream);
int sisr_nl_attn_fwd(const float* theta, const float* phi, const float* g, float* y, float* lse, int nb, int nq, int nk,
                     int dim, void* stream);
int sisr_nl_attn_bwd(const float* theta, const float* phi, const float* g, const float* y, const float* lse,
                     const float* dy, float* dtheta, float* dphi, float* dg, float* dsum, int nb, int nq, int nk,
                     int dim, void* stream);

/* ---- loss and optimiser ---------------------------------------------------------------------------
 * ref: SISR/models/__init__.py:268 nn.L1Loss, :299-308 optim.Adam, :481-489 standard_update */
size_t sisr_l1_loss_workspace_bytes(void);
int sisr_l1_loss(const float* a, const float* b, long n, float* loss, float* grad, float* workspace, void* stream);
/* torch.optim.Adam's update over one flat range (parameters, gradients, exp_avg, exp_avg_sq laid out alike):
 * one_minus_beta*, step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) computed by the host in double;
 * gradients are read as g * grad_scale */
int sisr_adam_flat(float* p, const float* g, float* m, float* v, long n, float beta2, float one_minus_beta1,
                   float one_minus_beta2, float eps, float step_size, float bc2_sqrt, float grad_scale, void* stream);

/* ---- progress words for hipGraph replays (data-parallel reducer; ref: SISR/models/__init__.py:344-347 DataParallel) ----
 * sisr_host_flags_alloc: n zeroed 32-bit words of pinned, host-coherent, device-visible memory (nullptr on failure; the
 * one allocation the library makes on the caller's behalf, returned to sisr_host_flags_free).  sisr_signal_host enqueues
 * a one-thread kernel that adds 1 to *flag once every earlier kernel of the stream has completed; under stream capture it
 * becomes a graph node, so a replay reports how far it has got without host round trips inside the graph. */
void* sisr_host_flags_alloc(int n);
void sisr_host_flags_free(void* flags);
int sisr_signal_host(void* flag, void* stream);
/* device-side wait: work enqueued on `stream` after this call starts once *flag >= value (hipStreamWaitValue32) */
int sisr_stream_wait_flag(void* flag, unsigned value, void* stream);
/* the same as a one-lane polling kernel on `stream` (bounded: after about a minute it stores 1 to *timed_out, nullable) */
int sisr_stream_spin_flag(void* flag, unsigned value, void* timed_out, void* stream);

/* ---- training tiles cut on the device (the step in front of the path) ---------------
 * ref: sr_tools/image_manipulation.py:233-257 random_flip_rotate + random_matched_crop, in the order
 * data_handler.py:500-513 applies them.  src: B device pointers to planar [C][H_b][W_b] fp32 images; params: B records
 * of 8 ints {H, W, top, left, hflip, vflip, transpose, 0} (top / left in the augmented image); dst [B][C][crop][crop]. */
int sisr_crop_augment(const float* const* src, const int* params, float* dst, int B, int C, int crop, void* stream);

/* ---- diagnostics (not on the product path): sustained fp32-MFMA rate and in-kernel clock ---------- */
#ifdef SISR_DIAG /* libsisr_hip_diag.so only (csrc/build.sh diag): not part of the product library */
int sisr_diag_mfma_peak(int blocks, int iters, float* out, unsigned long long* clk, void* stream);
/* resident workgroups per CU the runtime computes for the plain bf16x3 (which = 0) / fp32 (1) conv kernel */
int sisr_diag_conv_occupancy(int which);
/* every later conv3x3_c64_v4_kernel launch writes, per wave, 8 uint32 {start (shader clock), start (100 MHz clock), staging,
 * K loop, epilogue (shader cycles), HW_ID, XCC_ID, lifetime (100 MHz ticks)} into buf (device, workgroups*4*8 words); nullptr switches it off
 * (tools/conv_timeline.py) */
void sisr_diag_conv_stamp(void* buf);
/* the cost of `count` (0 with kind 1, else 8 / 16 / 32) filler instructions of one kind per eight v_mfma_f32_32x32x2_f32
 * (kinds: csrc/diag.hip); clk = {shader cycles, 100 MHz ticks} of block 0's loop (tools/mfma_fill.py) */
int sisr_diag_mfma_fill(int blocks, int iters, int kind, int count, float* out, const float* src, unsigned long long* clk,
                        void* stream);
#endif

/* ---- fp32 through the bf16 matrix cores ("bf16x3", opt-in; the reference has no such mode) -----------------------
 * Same contract as sisr_conv3x3_c64_bf16, but every fp32 operand is split exactly into three bf16 numbers (hi + mid +
 * lo) and the six products of weight >= 2^-16 run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (hi*hi and the
 * five corrections in separate accumulators): fp32-class error (the dropped products are <= 2^-24 relative each) at
 * 6/16 of the fp32 MFMA's cycles.  Packed weights: three bf16 planes, cout*cin*9 elements apart. */
int sisr_pack_conv3x3_x3_both(const float* w, void* packed_fwd, void* packed_dgrad, int cout, int cin, int shuffle_r,
                              void* stream);
int sisr_conv3x3_c64_x3(const float* x, const int64_t* xview, const void* wpacked_x3, const float* bias, int bias_n,
                        int bias_q, float* y, const int64_t* yview, const float* res, const float* mask,
                        const float* in_scale, const float* in_shift, const float* out_scale, float alpha, int relu,
                        float* gap_partial, const float* gate_add, float* gate_out, const float* dot, int B, int H,
                        int W, int cin, int cout, int select /* must be 0 */, void* stream);

/* weight / bias gradient in the same arithmetic (contract of sisr_wgrad3x3_c64) */
size_t sisr_wgrad3x3_c64_x3_workspace_bytes(int B, int H, int W, int cin, int cout);
int sisr_wgrad3x3_c64_x3(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview,
                         const float* dy_scale, const float* dy_shift, float alpha, float* dw, int64_t so, int64_t si,
                         int flip_taps, int out_perm_n, int out_perm_q, int in_perm_n, int in_perm_q, float* dbias,
                         int bias_n, int bias_q, float* workspace, size_t workspace_bytes, int B, int H, int W, int cin,
                         int cout, void* stream);

/* ---- channel padding and RGB pixel-shuffle (SRMD: conv(3+M -> nc) ... conv(nc -> 3 r^2) + PixelShuffle(r))
 * ref: advanced/architectures.py:380-425, advanced/SRMD_blocks.py:33-126.  The MFMA convs work on 64-channel chunks:
 * the (3+M)-channel NCHW input becomes a zero-padded NHWC map, head / tail weights zero-padded OIHW copies (crop != 0:
 * the inverse, for their gradients), and the tail's NHWC result is shuffled into the NCHW image (adjoint != 0: the
 * gradient map, padded channels zero). */
int sisr_nchw_to_nhwc_pad(const float* x, float* y, int B, int C, int H, int W, int C_padded, void* stream);
int sisr_pad_oihw(const float* src, float* dst, int cout, int cin, int cout_padded, int cin_padded, int taps, int crop,
                  void* stream);
int sisr_shuffle_rgb(const float* src, float* dst, int B, int C, int r, int H, int W, int C_padded, int adjoint,
                     void* stream);
/* HAN's map stack (ref: advanced/architectures.py:357-362 torch.cat of the 11 intermediate maps): map [B][hw][64] -> slot k
 * of stack [B][N][hw][64] (unstack != 0: the reverse, for the stack's gradient). */
int sisr_stack_maps(const float* src, float* dst, int B, long hw, int N, int k, int unstack, void* stream);
/* nn.PixelShuffle(r) on a channels-last map (ref: advanced/common.py:20-45; C = 64 upsamplers fuse it into the conv's store):
 * src [B][H][W][C r^2] -> dst [B][rH][rW][C]; adjoint != 0: dst [B][H][W][C r^2] <- src [B][rH][rW][C]. */
int sisr_pixel_shuffle_cl(const float* src, float* dst, int B, int H, int W, int C, int r, int adjoint, void* stream);

/* ---- on-the-fly LR synthesis (online_degradations) -----------------------------------------------------------
 * ref: sr_tools/gaussian_utils.py:346-368 BatchBlur (reflection pad + per-channel l x l correlation), :52-53
 * ToPILImage quantisation `mul(255).byte()`, sr_tools/image_manipulation.py:32-53 downsample (PIL BICUBIC resize).
 * sisr_blur_quant: planar [C][H][W] fp32, one [l][l] kernel (l <= 21) -> uint8 (and / or the unquantised fp32 blur).
 * sisr_pil_resample: ONE pass of libImaging/Resample.c's 8-bit resampler over a planar uint8 image (vertical = 0:
 * taps along x; 1: along y, optionally written as fp32 / 255 = ToTensor): out = clip8((2^21 + sum in*coef) >> 22);
 * bounds [out][2] (first tap, taps) and coef [out][ksize] int32 are device copies of the host tables. */
int sisr_blur_quant(const float* x, const float* kernel, unsigned char* y_u8, float* y_f32, int C, int H, int W, int l,
                    void* stream);
/* y = byte(255 * clamp(noise * sigma + blur, 0, 1)) over n values: b_GaussianNoising on the blurred image followed by
 * ToPILImage's quantisation (ref: sr_tools/gaussian_utils.py:306-312, 409-411); `noise` is the host-drawn N(0,1) field */
int sisr_noise_quant(const float* blur, const float* noise, float sigma, unsigned char* y_u8, long n, void* stream);
int sisr_pil_resample(const unsigned char* in, void* out, const int* bounds, const int* coef, int ksize, int C, int Hin,
                      int Win, int Hout, int Wout, int vertical, int to_float, void* stream);

/* ---- SPARNet / QSPARNet pieces (csrc/sparnet.hip) --------------------------------------------------------------
 * ref: SPARNet/blocks.py:69-103 ConvLayer ([nearest x2] -> ReflectionPad2d(1) -> Conv2d(3x3, stride 1 | 2) -> [BatchNorm2d]
 * -> [LeakyReLU(0.2)]), :106-174 ResidualBlock, :177-243 HourGlassBlock.  Maps are NHWC with C a multiple of 64 (channels
 * >= C_real zero).  The convs themselves are sisr_conv3x3_c64 / sisr_wgrad3x3_c64 on the padded geometry.
 * sisr_pad_reflect_up: adjoint == 0: x (B,H,W,C) -> y (B, up H + 2, up W + 2, C) = ReflectionPad2d(1)(nearest_up(x)), up 1 | 2;
 *   adjoint != 0: x is the gradient of that padded map, y (B,H,W,C) its fold back (sums in index order).
 * sisr_crop_stride: embed == 0: src (B,Hf,Wf,C) -> dst (B,Ho,Wo,C), dst[h][w] = src[1 + s h][1 + s w], Ho = (Hf-3)/s + 1 --
 *   the interior of the "same" conv over the padded map = the reference's unpadded (strided) conv; embed != 0: the adjoint.
 * sisr_bn_act_fwd: y = act(BatchNorm2d(x)), act = LeakyReLU(slope) (slope 1: none).  training != 0: batch statistics
 *   (biased variance, two passes), mean_out / invstd_out [C] saved for the backward, running_mean / running_var (nullable)
 *   updated with `momentum` and the unbiased variance as torch does; training == 0: the running statistics.
 * sisr_bn_act_bwd: dx, dgamma [C_real], dbeta [C_real] from x, dy (gradient AFTER the activation), mean, invstd.
 *   workspace: sisr_bn_workspace_bytes(npix, C) for both.
 * sisr_spar_combine_fwd: y = identity (nullable) + x * a, a = sigmoid(logits[p][0]) (logits NHWC with C_logits channels),
 *   att[p] = a.  _bwd: dx = dy * a; dlogits[p][0] = (sum_c dy x) a (1 - a), the other channels of dlogits zero. */
/* The stride-1 ConvLayer conv WITHOUT the gathers (round 4): reflection and nearest upsampling as address arithmetic in the
 * MFMA kernels' staging (csrc/conv3x3_mfma.hip template GEO, csrc/wgrad3x3_mfma.hip GEO bodies).
 * sisr_conv3x3_c64_geo  mode 1: y (B,H,W,cout) = Conv2d(3x3, no padding)(ReflectionPad2d(1)(nearest_up^up(x))) + bias,
 *     x (B, H >> up, W >> up, cin), up 0 | 1 (a SHIFT: 1 = nearest x2);
 *   mode 2: y (B,H,W,cout) = zero-padded 3x3 conv of x (B,H-2,W-2,cin) placed at (1,1) of an H x W zero map -- with the
 *     input-gradient packing of the weight, the gradient of mode 1's padded map; sisr_pad_reflect_up(adjoint) folds it back.
 *   wpacked as for sisr_conv3x3_c64; gap_partial (nullable): per-strip channel sums of y in sisr_conv3x3_c64's layout;
 *   kreal > 0 (cin == 64 only): the caller's promise that input channels >= kreal are zero (<= 8 / <= 32 run 1 / 4 of the 8
 *   channel octets per tap).  Results equal the gather -> sisr_conv3x3_c64 -> gather composition bit for bit.
 * sisr_wgrad3x3_c64_geo: dw (co_real, ci_real, 3, 3) and db (co_real, nullable) of mode 1 from x (B, H >> up, W >> up, cin) and
 *   dy (B,H,W,cout); workspace: sisr_wgrad3x3_c64_workspace_bytes(B, H, W, cin, cout); active_units as in sisr_wgrad3x3_c64
 *   (blocks that hold only channel padding may be masked; bias halves >= co_real need no unit). */
int sisr_conv3x3_c64_geo(const float* x, const int64_t* xview, const float* wpacked, const float* bias, float* y,
                         const int64_t* yview, float* gap_partial, int B, int H, int W, int cin, int cout, int mode, int up,
                         int kreal, void* stream);
int sisr_wgrad3x3_c64_geo(const float* x, const int64_t* xview, const float* dy, const int64_t* dyview, float* dw, int co_real,
                          int ci_real, float* dbias, float* workspace, size_t workspace_bytes, int B, int H, int W, int cin,
                          int cout, int up, unsigned long long active_units, void* stream);
/* Any number of such weight gradients of DIFFERENT geometries, eight per launch (plain NHWC maps: x (B, H >> up, W >> up, cin),
 * dy (B, H, W, cout)); jobs: HOST array of njobs records, read before the call returns.  Per job the result is that of
 * sisr_wgrad3x3_c64_geo up to the summation order of its K-split. */
typedef struct sisr_wgrad_geo_job {
  const float* x;
  const float* dy;
  float* dw;
  float* dbias; /* nullable */
  int B, H, W, cin, cout, up, co_real, ci_real;
  unsigned long long active_units;
} sisr_wgrad_geo_job;
size_t sisr_wgrad_geo_job_bytes(void);
size_t sisr_wgrad3x3_c64_geo_batch_workspace_bytes(const void* jobs, int njobs);
int sisr_wgrad3x3_c64_geo_batch(const void* jobs, int njobs, float* workspace, size_t workspace_bytes, void* stream);
/* Non-default ConvLayer options (ref: SPARNet/blocks.py:17-33 norm_type 'in' | 'gn' | 'pixel', :50-64 relu_type 'prelu' | 'selu',
 * :147-151 att_name 'spar3d').  Maps NHWC, C a multiple of 4, channels >= C_real zero padding (written as zero).
 * sisr_group_norm_fwd / _bwd: statistics per (sample, group of cg consecutive real channels) over hw pixels (InstanceNorm2d: cg = 1;
 *   GroupNorm(32, C): cg = C / 32), biased variance, two passes; mean / invstd: [B][C_real / cg]; backward writes dx and the
 *   per-sample sums dgamma_b, dbeta_b [B][C_real] (add them over the batch with sisr_sum_partials).
 * sisr_pixel_norm: backward == 0: out = x / max(||x||_2 over the pixel's channels, 1e-12) (F.normalize(p = 2, dim = 1));
 *   else out = dx from x and dy.  C / 4 a power of two <= 64.
 * sisr_act: mode 0 PReLU (slope [C_real]), 1 SELU; backward == 0: out = act(x); else out = dx and, for PReLU, dyx = dy min(x, 0)
 *   (its per-channel sum is the slope's gradient).
 * sisr_spar3d: backward == 0: out0 = identity (nullable) + x sigmoid(logits); else (third argument = dy) out0 = dx = dy a,
 *   out1 = dlogits = dy x a (1 - a); n elements (a multiple of 4). */
int sisr_group_norm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* mean_out, float* invstd_out, int B,
                        long hw, int C, int C_real, int cg, float eps, void* stream);
int sisr_group_norm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* invstd, float* dx,
                        float* dgamma_b, float* dbeta_b, int B, long hw, int C, int C_real, int cg, void* stream);
int sisr_pixel_norm(const float* x, const float* dy, float* out, long npix, int C, int backward, void* stream);
int sisr_act(const float* x, const float* dy, const float* slope, float* out, float* dyx, long npix, int C, int C_real, int mode,
             int backward, void* stream);
int sisr_spar3d(const float* x, const float* logits, const float* identity_or_dy, float* out0, float* out1, long n, int backward,
                void* stream);
/* sisr_nearest_up: nn.Upsample(scale_factor = up, 'nearest') on an NHWC map, up 1 .. 4 (ref: advanced/SRMD_blocks.py:58-63, the
 * 'upconv' tail of SRMD); adjoint != 0: the gradient summed back over the up x up replicas. */
int sisr_nearest_up(const float* src, float* dst, int B, int H, int W, int C, int up, int adjoint, void* stream);
size_t sisr_bn_workspace_bytes(long npix, int C);
int sisr_pad_reflect_up(const float* x, float* y, int B, int H, int W, int C, int up, int adjoint, void* stream);
int sisr_crop_stride(const float* src, float* dst, int B, int Hf, int Wf, int C, int stride, int embed, void* stream);
int sisr_bn_act_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* mean_out, float* invstd_out, long npix, int C, int C_real, int training,
                    float momentum, float eps, float slope, float* workspace, size_t workspace_bytes, void* stream);
int sisr_bn_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                    const float* invstd, float* dx, float* dgamma, float* dbeta, long npix, int C, int C_real, float slope,
                    float* workspace, size_t workspace_bytes, void* stream);
int sisr_spar_combine_fwd(const float* x, const float* logits, const float* identity, float* y, float* att, long npix, int C,
                          int C_logits, void* stream);
int sisr_spar_combine_bwd(const float* dy, const float* x, const float* att, float* dx, float* dlogits, long npix, int C,
                          int C_logits, void* stream);

/* ---- SFTMD pieces (csrc/sft.hip) ------------------------------------------------------------------------------
 * ref: SFTMD_variants/architectures.py:25-56 StandardSft (x * sigmoid(mul) + add), :110-176 SFTMD (LeakyReLU(0.2),
 * 9x9 64 -> 3 output conv, clamp).  The network's 3x3 convs run on sisr_conv3x3_c64 with `relu` = 2 (LeakyReLU(0.2)
 * epilogue) or `relu` | 4 (the `mask` operand carries LeakyReLU's derivative: slope 0.2 where the map is <= 0).
 * sisr_sft_compose: split == 0: WA [64][128][9] = rows (mul_conv1 | add_conv1) over input channels 0 .. 63 + M (rest
 *   zero), bA [64], WB [128][64][9] = block-diagonal (mul_conv2 on inputs 0..31, add_conv2 on inputs 32..63), bB [128]
 *   from the layer's eight parameters; split != 0: the reverse copy (gradients).  M = metadata maps (<= 64).
 * sisr_sft_combine_fwd: out = [relu](x * sigmoid(y2[:, :64]) + y2[:, 64:]); x / out with pixel strides (floats), y2
 *   [npix][128]; md (nullable) [npix][64] is copied into out's second 64-channel chunk.  _bwd: dx [npix][64], dy2.
 * sisr_map64: 64-channel maps with pixel strides: op 0 copy, 1 a + b, 2 LeakyReLU(a), 3 b * LeakyReLU'(a), 4 a * b,
 *   5 ReLU(a), 6 b * ReLU'(a), 7 a * (b's channel 0, broadcast).
 * sisr_conv9_*: 9x9 conv 64 -> 3 (OIHW weight), x NHWC [B][H][W][64], y NCHW; dgrad optionally masked by LeakyReLU'
 *   of `leaky_mask` (the activated map that fed the conv); wgrad: ordered two-stage sums (dw OIHW, db).
 * sisr_clamp01: backward == 0: out = clamp(a, 0, 1); else out = grad * [0 <= a <= 1]. */
int sisr_sft_compose(float* mul_w1, float* mul_b1, float* add_w1, float* add_b1, float* mul_w2, float* mul_b2, float* add_w2,
                     float* add_b2, float* WA, float* bA, float* WB, float* bB, int M, int split, void* stream);
/* all SFT layers of a network in one launch: table = n records of 12 device pointers (the arguments of sisr_sft_compose in
 * order) + int M + int pad, sisr_sft_compose_record_bytes() each, in device memory */
size_t sisr_sft_compose_record_bytes(void);
int sisr_sft_compose_many(const void* table, int n, void* stream);
int sisr_sft_combine_fwd(const float* x, long x_stride, const float* y2, const float* md, float* out, long out_stride,
                         long npix, int relu, void* stream);
int sisr_sft_combine_bwd(const float* dout, long dout_stride, const float* x, long x_stride, const float* y2, float* dx,
                         float* dy2, long npix, int relu, void* stream);
int sisr_map64(const float* a, long a_stride, const float* b, long b_stride, float* out, long out_stride, long npix, int op,
               void* stream);
int sisr_conv9_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, void* stream);
int sisr_conv9_dgrad(const float* dy, const float* w, const float* leaky_mask, float* dx, int B, int H, int W, void* stream);
size_t sisr_conv9_wgrad_workspace_bytes(int B, int H, int W);
int sisr_conv9_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, size_t workspace_bytes, int B,
                     int H, int W, void* stream);
int sisr_clamp01(const float* a, const float* grad, float* out, long n, int backward, void* stream);

/* ---- around the non-local attention: 1x1 projections, pooling, output projection (csrc/nonlocal.hip) ---------------
 * ref: advanced/SAN_blocks.py:104-148 (theta / phi / g = Conv2d(64, 8, 1), phi and g through MaxPool2d(2), W = Conv2d(8, 64, 1),
 * z = W(y) + x), :305-336 (the block applied per quadrant).  x, z, dz, dx: channels-last [npix][64]; proj, dproj:
 * [npix][24] (theta | phi | g).  `domains`: 9 ints in host memory B, H, W, y0, x0, hq, wq, nqy, nqx = B * nqy * nqx
 * rectangles of hq x wq positions starting at (y0, x0); domain index (b * nqy + iy) * nqx + ix, rows
 * [domain][position][8], pooled rows [domain][(hq / 2) * (wq / 2)][8] (floor mode).
 * sisr_nl_project_bwd: dx = dproj . Wp + dz, and part[sisr_nl_project_bwd_parts(npix)][33][64] ordered partial sums
 *   (rows 0..23 = d(theta | phi | g weight)[n][c], row 32 = the 24 bias gradients): sum with sisr_sum_partials.
 * sisr_nl_output_bwd: dy rows, and part[sisr_nl_output_bwd_parts(domains)][64 * 8 + 64] (dW [64][8], then db [64]). */
int sisr_nl_project_fwd(const float* x, const float* w_theta, const float* b_theta, const float* w_phi, const float* b_phi,
                        const float* w_g, const float* b_g, float* proj, long npix, void* stream);
int sisr_nl_project_bwd_parts(long npix);
int sisr_nl_project_bwd(const float* x, const float* dproj, const float* dz, const float* w_theta, const float* w_phi,
                        const float* w_g, float* dx, float* part, long npix, void* stream);
int sisr_nl_split_pool_fwd(const float* proj, float* theta, float* phi, float* g, const int* domains, void* stream);
int sisr_nl_split_pool_bwd(const float* proj, const float* dtheta, const float* dphi, const float* dg, float* dproj,
                           const int* domains, void* stream);
int sisr_nl_output_fwd(const float* y, const float* x, const float* w, const float* bias, float* z, const int* domains,
                       void* stream);
int sisr_nl_output_bwd_parts(const int* domains);
int sisr_nl_output_bwd(const float* dz, const float* y, const float* w, float* dy, float* part, const int* domains,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif
