/* smml.h - C-ABI of the MI355X (gfx950) kernels for the multimodal-MIL attention / fusion hot path.
 *
 * Drop-in boundary for helenypzhang/Subspace-Multimodal-Learning.  The reference has no FFI layer: the
 * path sits behind plain PyTorch nn.Module classes.  This library is what the
 * host-side mirror of those modules (package `subspace-multimodal-learning_amd`) binds through ctypes;
 * each entry point names the reference op sequence (file:line) it replaces.
 *
 * Conventions
 *   - all tensors are fp32, contiguous, token-major (channel-last), and live in HBM of ONE device;
 *     every pointer is a raw device pointer owned by the caller (PyTorch caching allocator on the
 *     Python side); the library never allocates, frees or retains device memory;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and re-entrant;
 *   - return value 0 = ok, negative = error; the message is read with smml_last_error()
 *     (thread-local: the one piece of per-thread state the library keeps); no exception crosses the boundary;
 *   - "accumulated into" outputs must be zeroed (or hold a running sum) by the caller;
 *   - fixed widths of the two fused attention families (no other value is built, the entry points return an error):
 *     head dim 64; position-bias MLP posdim -> 32 -> 32 -> heads / groups (the reference's dim = 128), heads / groups <= 2.
 */
#ifndef SMML_H_
#define SMML_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* smml_last_error(void);
int smml_abi_version(void);
/* 0 iff `device` is a usable gfx950 device of this process */
int smml_device_check(int device);
/* HIP events as plain handles: time one kernel on the stream it is launched on (bench.py roofline leg) */
void* smml_event_create(void);
int smml_event_destroy(void* ev);
int smml_event_record(void* ev, void* stream);
int smml_event_elapsed_ms(void* start, void* stop, float* ms);

/* ------------------------------------------------------------------------------------------------
 * Strided-batched fp32 GEMM on the f32 matrix cores (exact fp32):
 *   C[b0,b1](m,n) (+)= act(alpha * sum_k A[b0,b1](m,k) B[b0,b1](k,n) + bias) + beta * residual
 * A(m,k) at A + b0*sa0 + b1*sa1 + m*sam + k*sak;  B(k,n) at B + b0*sb0 + b1*sb1 + k*sbk + n*sbn;
 * C(m,n) at C + b0*sc0 + b1*sc1 + m*ldc + n;  residual shares C's batch offsets with row stride ldr.
 * bias_mode 0 none | 1 bias[n] | 2 bias[(m / rows_per_bias) * bias_ld + n];  act 0 none | 1 relu | 2 tanh.
 * splitk > 1 or accumulate != 0: products are atomically added into C (caller zeroes C or keeps a running
 * sum there; no act / residual); batch strides of 0 on C fold a batch dimension into one output.
 * Replaces nn.Linear / 1x1 nn.Conv sites of the path: models/DeformCrossTransMIL.py:35-37,83,93-95,
 * 196; models/DeformableAttention2D.py:218-221; models/DeformableAttention1D.py:150-153;
 * models/NystromAttention.py:60-65,86,122-140 - and their backward products. */
int smml_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                  int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                  long long ldc, long long ldr, int nb0, int nb1, long long sa0, long long sa1,
                  long long sb0, long long sb1, long long sc0, long long sc1, long long sbias0,
                  long long sbias1, int bias_mode, int rows_per_bias, long long bias_ld, int act,
                  int splitk, int accumulate, float alpha, float beta, void* stream);
/* test hook: 1 routes every product through the generic (any stride / any K) kernel instead of the tiled
 * fast path (K % 16 == 0, 16-byte aligned operands with a unit stride on k or on the row index). */
void smml_gemm_force_generic(int on);
/* precision / tuning switch of the tiled kernels: 0 automatic (default: exact fp32 on the f32 MFMA, split-bf16 for large
 * square-ish products), 1 fp32-MFMA kernel only, 2 split-bf16 (three terms, fp32-grade) wherever it applies, 3 single-term bf16
 * (operands rounded to bf16 when staged, fp32 accumulation: the 16-bit compute mode of the Nystrom block), 4 single-term fp16 (operands
 * rounded to fp16 when staged - 11 mantissa bits, fp16's exponent range: for forward-range operands; the fp16 compute modes issue their
 * gradient products in mode 3). */
void smml_gemm_set_mode(int mode);
int smml_gemm_get_mode(void);
/* tile-height switch of the tiled kernels: 0 automatic (64-row tiles for launches that the 128-row tiling leaves below two
 * workgroups per CU, e.g. the batched 256^3 products of the Nystrom pseudo-inverse), 1 never, 2 wherever it applies. */
void smml_gemm_set_small_tile(int v);

/* bf16-STORAGE GEMM (16-bit matrix pipe, fp32 accumulation): A and B are bf16 in memory, C is bf16 (out_bf16 = 1) or fp32.
 *   trans = 0   C[M, N] = A[M, K] B[N, K]^T   (row strides lda / ldb >= K, in elements; K % 8 == 0)
 *   trans = 1   C[M, N] = A[K, M]^T B[K, N]   (row strides lda >= M, ldb >= N; M % 8 == 0, N % 8 == 0)
 * bias (fp32 [N] or NULL) is added once.  splitk > 1 (fp32 output only): the K range is cut into slices whose partial products are
 * ADDED to C atomically - the caller zeroes C; splitk = 0: chosen by the library (one round of workgroups; C must then be zeroed too).  Operands 16-byte aligned, lda / ldb multiples of 8.
 * The projections of the Nystrom block's bf16 compute mode (models/NystromAttention.py:88 to_qkv, :147 to_out) and their backward
 * products with bf16 bags (BASELINE configs 2 / 4): activations stay bf16 end to end. */
int smml_gemm_b16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb,
                  long long ldc, int trans, int out_bf16, int splitk, void* stream);
/* tile selection of the bf16-storage GEMM: 0 automatic (the 256-row, eight-wave tile where its grid still covers the chip), 1 the
 * 128 x 128 tile only, 2 the 256-row tile wherever M >= 256 and N >= 128 (test / measurement switch; SMML_B16_TILE presets it). */
void smml_gemm_b16_set_tile(int mode);
/* measurement switch: 1 (default) places the tiles of one slice of a split reduction on one XCD (they share the slice's operands through one
 * L2), 0 spreads them over the XCDs like any other launch. */
void smml_gemm_b16_set_slice_major(int on);
/* nb problems of that shape at element strides sa / sb / sc (multiples of 8) between them; sc = 0 (fp32 output only): the problems' products
 * are ADDED into the one zeroed output, like split-K slices.  A bag's n real rows are addressed in place as one batch item (the zero rows
 * the reference pads in front of a bag, NystromAttention.py:82, are never multiplied). */
int smml_gemm_b16_batched(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda, long long ldb,
                          long long ldc, int trans, int out_bf16, int splitk, int nb, long long sa, long long sb, long long sc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim C (<= 1024) of x [R, C]; saves mean / rstd per row.
 * Replaces nn.LayerNorm at models/DeformCrossTransMIL.py:44,71,75,90,144 and mil.py:175,187. */
int smml_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                           float* rstd, long long R, int C, float eps, void* stream);
/* dy row for x row i is dy[(i / rows_per_dy) * C + c] * dy_scale (rows_per_dy > 1 broadcasts one
 * gradient row over a token block, e.g. the gradient of Pooler's mean, DeformCrossTransMIL.py:193);
 * dx is overwritten (accumulate_dx = 0) or added to; dgamma / dbeta are accumulated into. */
int smml_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* mean,
                           const float* rstd, float* dx, float* dgamma, float* dbeta, long long R, int C,
                           long long rows_per_dy, float dy_scale, int accumulate_dx, void* stream);

/* out[b, c] += scale * sum_r x[b, r, c] for x [nb, R, C]; out accumulated into.
 * Pooler mean (DeformCrossTransMIL.py:193) and bias gradients. */
int smml_colsum_f32(const float* x, float* out, int nb, long long R, int C, float scale, void* stream);

/* OrthogonalLoss (models/cmta_utils.py:1212-1228) on rows P, P_hat, G, G_hat [B, D]: loss [B] (nullable) and, when
 * dloss [B] is given, dP / dPh / dG / dGh [B, D]; one wave per sample, wavefront reductions. */
int smml_orth_loss_f32(const float* P, const float* Ph, const float* G, const float* Gh, const float* dloss, float* loss,
                       float* dP, float* dPh, float* dG, float* dGh, int B, int D, float gamma, void* stream);

/* Tail of BatchLoss after its two Gram products (utils/loss.py:26-40): S = go / ||go||_row, V = mean_g gv[g] / ||gv[g]||_row,
 * loss = (S - V)^2 / n_total on [R, R] (dloss NULL), or the gradients dgo [R, R], dgv [nv, R, R] for a given dloss [R, R]; R = batch x world
 * <= 64, nv <= 16 (the reference's 8 offset groups).  One launch per direction instead of ~30 launch-bound elementwise kernels. */
int smml_batchloss_tail_f32(const float* go, const float* gv, const float* dloss, float* loss, float* dgo, float* dgv, int R, int nv,
                            float n_total, void* stream);

/* dx = dy * (y > 0)  (ReLU of _fc1, DeformCrossTransMIL.py:83) */
int smml_relu_bwd_f32(const float* dy, const float* y, float* dx, long long n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Offset network of the deformable attention (DeformableAttention2D.py:207-213,255-266;
 * DeformableAttention1D.py:139-146,176-183): depthwise strided conv (kernel ks, stride r,
 * pad (ks-r)/2) -> GELU -> 1x1 (dg -> posdim) -> tanh -> * offset_scale; vgrid = meshgrid + offsets
 * (returned to the caller's loss un-normalised); vs = 2 vgrid / max(t-1, 1) - 1.
 *   q [B, Hh, Ww, G*dg] (1-D module: Hh = 1)   w0 [dg, 1, ks(, ks)]  b0 [dg]  w2 [posdim, dg]
 *   vgrid [(B G), posdim, th, tw]               vs [(B G), th*tw, posdim] */
int smml_offsets_out_len(int s, int ks, int r);
int smml_offsets_fwd_f32(const float* q, const float* w0, const float* b0, const float* w2, float* vgrid,
                         float* vs, int B, int Hh, int Ww, int G, int dg, int ks, int r, int posdim,
                         float offset_scale, void* stream);
/* upstream gradient = dvgrid (direct, nullable) + 2/max(t-1,1) * dvs (nullable); dw0, db0, dw2 are overwritten (gather formulation, no
 * atomics); dq is overwritten, or - accumulate_dq = 1 - ADDED to: dq then already holds the gradient q received from its other consumer,
 * the fused attention core (one pass over the [B, N, 512] tensor instead of a store plus an elementwise add).
 * workspace: smml_offsets_bwd_workspace_bytes(...) bytes, 16-byte aligned. */
size_t smml_offsets_bwd_workspace_bytes(int B, int Hh, int Ww, int G, int dg, int ks, int r, int posdim);
int smml_offsets_bwd_f32(const float* q, const float* w0, const float* b0, const float* w2,
                         const float* dvgrid, const float* dvs, float* dq, float* dw0, float* db0,
                         float* dw2, void* workspace, size_t workspace_bytes, int B, int Hh, int Ww, int G, int dg,
                         int ks, int r, int posdim, float offset_scale, int accumulate_dq, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bilinear sampling = F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False) of the
 * grouped key/value stream (DeformableAttention2D.py:268-274; DeformableAttention1D.py:36-43,185-190).
 *   x [B, Hh, Ww, G*cg]   vs [(B G), J, posdim]   kv [B, J, G*cg]
 * The 1-D module's degenerate sampling is posdim = 1 with the map laid out [Hh = n, Ww = 1]. */
int smml_bilinear_sample_fwd_f32(const float* x, const float* vs, float* kv, int B, int Hh, int Ww, int G,
                                 int cg, int J, int posdim, void* stream);
/* dx accumulated into (float atomics); dvs accumulated into (+=). */
int smml_bilinear_sample_bwd_f32(const float* x, const float* vs, const float* dkv, float* dx, float* dvs,
                                 int B, int Hh, int Ww, int G, int cg, int J, int posdim, void* stream);
/* integer path only (bit-exact contract): corner indices cx, cy [n, 4] and in-bounds masks cm [n, 4]
 * in the order (x0,y0) (x1,y0) (x0,y1) (x1,y1) for n sample points vs [n, posdim]. */
int smml_bilinear_corners_f32(const float* vs, int* cx, int* cy, unsigned char* cm, int n, int Hh, int Ww,
                              int posdim, void* stream);

#ifndef SMML_DEFORM_OPTS_DEFINED
#define SMML_DEFORM_OPTS_DEFINED
/* Optional behaviour of ONE fused deformable-attention launch (trailing `opts` argument of the entry points below; NULL = all defaults).
 * Passed by the caller with every call: the library keeps no per-thread or global launch state. */
typedef struct SmmlDeformOpts {
  const unsigned long long* seed_offset; /* device-resident offset hashed into dropout_seed when the kernel RUNS (a launch captured in a hipGraph
                                            bakes dropout_seed in; the offset lets every replay draw a new mask), or NULL */
  int raw_distance;                      /* 1: posdim-1 launches feed the bias MLP the raw offset gq - vs (DeformableAttention1D.py:92,
                                            cpb_log_distance = False); 0: its signed log (the reference's default; posdim 2 has no such switch) */
  const unsigned short* mask_table;      /* smml_deform_attn16_bwd with relu_masks == NULL: layer-2 ReLU decisions from this table (filled by
                                            smml_cpb_mask_table with mask_table_pmax), or NULL = recompute layer 2 per pair */
  float mask_table_pmax;
  unsigned short* export_masks;          /* tests: smml_deform_attn16_bwd with relu_masks == NULL also writes the decisions it used here
                                            ([B, H, nst / 32, J, 2, 32] u16, the forward's layout), or NULL */
  int region_lds_cap;                    /* tests: the region entry points keep only regions with an id below this in LDS and take the others from
                                            global memory (the path of parameter sets with more than 2048 linear regions); 0 = the default (2048) */
} SmmlDeformOpts;
#endif

/* ------------------------------------------------------------------------------------------------
 * Fused deformable cross-attention core: softmax(scale q k^T + CPB(gq - vs)) v with the continuous
 * position bias MLP posdim -> 32 -> 32 -> heads/groups evaluated on the matrix cores per
 * (query, key) pair and never materialised.
 * Replaces DeformableAttention2D.py:120-157,284-312 and DeformableAttention1D.py:60-102,205-232.
 *   q [B, N, H*64] (unscaled)  k, v [B, J, H*64]  vs [(B G), J, posdim]  gq [N, posdim]
 *   w1 [32, posdim] b1 [32] w2 [32, 32] b2 [32] w3 [H/G, 32] b3 [H/G]
 *   out [B, N, H*64]  lse [B, H, N]  logits_t [B, H, nst / 32, J, 32] and relu_masks [B, H, nst / 32, J, 2, 32] uint16 with
 *   nst = smml_deform_attn_nst(N) (both nullable together: only needed for backward).  Score-shaped tensors are stored per 32-query
 *   tile - the unit one wave owns - so that every pass streams contiguous memory; relu_masks holds the ReLU decisions of the
 *   position bias's second layer, 32 bits per (query, key, head), in the bit order the backward kernel consumes.  Both are
 *   opaque to the caller (scratch it allocates and hands back to the backward).  H/G <= 2.
 * ev_start / ev_stop (nullable, handles of smml_event_create) are recorded on `stream` around the fused
 * forward kernel, resp. around the position-bias backward kernel (the dominant kernel of the step). */
int smml_deform_attn_nst(int N);
int smml_deform_attn_fwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                             const float* w1, const float* b1, const float* w2, const float* b2,
                             const float* w3, const float* b3, float* out, float* lse, float* logits_t,
                             unsigned short* relu_masks, int B, int N, int J, int H, int G, int posdim, float scale,
                             float dropout_p, unsigned long long dropout_seed, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts);
/* scratch the backward needs: position-bias gradient slabs, the per-wave d vs rows and the query-sliced dK / dV partial sums;
 * 16-byte aligned */
size_t smml_deform_attn_bwd_workspace_bytes(int B, int N, int J, int H);
/* dlogits_t: scratch of logits_t's size and layout (receives d scores); dq / dk / dv / dw* / db* overwritten;
 * dvs overwritten.  Every output is reduced in a fixed order (no float atomics): run-to-run identical results. */
int smml_deform_attn_bwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                             const float* w1, const float* b1, const float* w2, const float* b2,
                             const float* w3, const float* b3, const float* out, const float* dout,
                             const float* lse, const float* logits_t, const unsigned short* relu_masks,
                             float* dlogits_t, float* dq, float* dk, float* dv, float* dvs, float* dw1, float* db1, float* dw2, float* db2, float* dw3,
                             float* db3, void* workspace, size_t workspace_bytes, int B, int N, int J, int H,
                             int G, int posdim, float scale, float dropout_p, unsigned long long dropout_seed,
                             void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts);
/* The same fused core with the position bias evaluated EXACTLY PER LINEAR REGION of its MLP (csrc/cpb_regions.h; round 5).  The MLP of
 * models/DeformableAttention2D.py:129-152 is piecewise affine in the signed-log offsets (:148): smml_cpb_regions_build tabulates, for the
 * current parameters, which linear piece every cell of [-pmax, pmax]^2 lies in (cells a ReLU kink crosses carry the kink's line, cells
 * several kinks cross are refined once, what is left evaluates the MLP itself), and the forward / backward below replace the per-pair MLP
 * by a lookup + 2 FMAs, resp. by three fixed-point moment sums per region from which all six parameter gradients follow linearly.
 * Results are those of smml_deform_attn_fwd_f32 / _bwd_f32 to fp32 rounding (no tolerance added: tests/test_gpu_parity.py runs both);
 * posdim = 2, signed-log offsets, one head per offset group (G = H), J <= 16384.
 *   tables: caller-allocated scratch of smml_cpb_regions_bytes() bytes, 256-byte aligned; built per forward call (the parameters change
 *           every step), handed unchanged to the backward of the same call.  pmax >= max |slog(gq - vs)| for speed only: pairs
 *           outside the square evaluate the MLP.
 *   region_ids [B, H, nst / 32, J, 32] uint16: the linear piece of every pair (0xFFFF: none), saved in place of relu_masks.
 *   workspace of the backward: smml_deform_attn_region_bwd_workspace_bytes, 256-byte aligned.
 * Every output is run-to-run identical (slab sums in a fixed order; the region moments are integer sums). */
size_t smml_cpb_regions_bytes(void);
int smml_cpb_regions_build(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                           float pmax, void* tables, size_t tables_bytes, void* stream);
int smml_deform_attn_region_fwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                                    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                    const float* b3, const void* tables, float* out, float* lse, float* logits_t,
                                    unsigned short* region_ids, int B, int N, int J, int H, float scale, float dropout_p,
                                    unsigned long long dropout_seed, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts);
size_t smml_deform_attn_region_bwd_workspace_bytes(int B, int N, int J, int H);
int smml_deform_attn_region_bwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                                    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                    const float* b3, const void* tables, const float* out, const float* dout, const float* lse,
                                    const float* logits_t, const unsigned short* region_ids, float* dlogits_t, float* dq, float* dk,
                                    float* dv, float* dvs, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                                    void* workspace, size_t workspace_bytes, int B, int N, int J, int H, float scale, float dropout_p,
                                    unsigned long long dropout_seed, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts);

/* The same two entry points in the 16-bit compute modes of smml_deform_attn16_fwd / _bwd (dtype 0 = bf16, 1 = fp16: single-term operands on
 * the matrix pipe, scores saved as fp16 [B, H, nst / 32, J, 32], d scores as bf16): the position bias stays the fp32 lookup per linear
 * region, so it is MORE exact than the 16-bit per-pair MLP it replaces, and the launch pair costs about half of it.  Same conditions as
 * above; tables from smml_cpb_regions_build; workspace of the backward: smml_deform_attn_region_bwd_workspace_bytes, 256-byte aligned. */
int smml_deform_attn16_region_fwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, const void* tables,
                                  float* out, float* lse, unsigned short* logits16, unsigned short* region_ids, int B, int N, int J, int H,
                                  float scale, float dropout_p, unsigned long long dropout_seed, int dtype, void* ev_start, void* ev_stop,
                                  void* stream, const SmmlDeformOpts* opts);
int smml_deform_attn16_region_bwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, const void* tables,
                                  const float* out, const float* dout, const float* lse, const unsigned short* logits16,
                                  const unsigned short* region_ids, unsigned short* dlogits16, float* dq, float* dk, float* dv, float* dvs,
                                  float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, void* workspace,
                                  size_t workspace_bytes, int B, int N, int J, int H, float scale, float dropout_p,
                                  unsigned long long dropout_seed, int dtype, void* ev_start, void* ev_stop, void* stream,
                                  const SmmlDeformOpts* opts);

/* 16-bit compute mode of the same fused core (csrc/deform_attn16.hip; BASELINE config 4 names bf16, config 5 fp16): the op sequence of
 * smml_deform_attn_fwd_f32 / _bwd_f32 (models/DeformableAttention2D.py:120-157,284-312; DeformableAttention1D.py:60-102,205-232) with
 * single-term 16-bit operands on the matrix pipe - dtype 0 = bf16, 1 = fp16 for forward-range operands (q, k, v, probabilities, the
 * hidden layer and W2 of the position-bias MLP); gradient-range operands are always bf16 - and 16-bit score storage:
 *   logits16  [B, H, nst / 32, J, 32] fp16 in BOTH modes (the pre-softmax scores incl. bias are of forward range; in training the forward's
 *             own softmax runs on the rounded values, and with dropout the keep decision rides in the lowest mantissa bit)
 *   dlogits16 [B, H, nst / 32, J, 32] bf16 (scratch of the backward: d scores = d bias)
 * q / k / v / out and every gradient stay fp32 in memory; layer 1 of the position-bias MLP, the softmax statistics and all
 * accumulators are fp32; relu_masks, the workspace (smml_deform_attn_bwd_workspace_bytes) and the decision export
 * (smml_deform_attn_relu1_masks) are shared with the fp32 path.  ev_start / ev_stop as in the fp32 entry points. */
int smml_deform_attn16_fwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                           const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, float* out,
                           float* lse, unsigned short* logits16, unsigned short* relu_masks, int B, int N, int J, int H, int G,
                           int posdim, float scale, float dropout_p, unsigned long long dropout_seed, int dtype, void* ev_start,
                           void* ev_stop, void* stream, const SmmlDeformOpts* opts);
int smml_deform_attn16_bwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                           const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, const float* out,
                           const float* dout, const float* lse, const unsigned short* logits16, const unsigned short* relu_masks,
                           unsigned short* dlogits16, float* dq, float* dk, float* dv, float* dvs, float* dw1, float* db1,
                           float* dw2, float* db2, float* dw3, float* db3, void* workspace, size_t workspace_bytes, int B, int N,
                           int J, int H, int G, int posdim, float scale, float dropout_p, unsigned long long dropout_seed,
                           int dtype, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts);
/* relu_masks == NULL in smml_deform_attn16_bwd: the forward ran without the MLP (smml_deform_attn_table_fwd below) and saved no ReLU bits -
 * the backward recomputes layer 2 per pair (one bf16 term) and differentiates the per-pair MLP exactly as with saved bits
 * (opts->export_masks: tests read the decisions it used). */
/* Mask table: instead of recomputing layer 2 per pair, a relu_masks == NULL backward can take the layer-2 decisions from a table of the sign
 * pattern of W2 relu(W1 p + b1) + b2 at the centres of cells^posdim cells over [-pmax, pmax]^posdim (cells = smml_cpb_mask_table_cells(posdim):
 * 1024 per axis in 2-D, 16384 in 1-D; `table` = cells^posdim 32-bit words = two u16 lane-half words in relu_masks' bit layout).  A pair's
 * decision then differs from its own pre-activation's sign only where a layer-2 kink crosses its cell (|x2| <= |grad x2| x 1.8e-3), the size of the
 * decision noise of the single-term bf16 product.  Handed to the backward as opts->mask_table / opts->mask_table_pmax. */
int smml_cpb_mask_table_cells(int posdim);
int smml_cpb_mask_table(const float* w1, const float* b1, const float* w2, const float* b2, unsigned short* table, int posdim, float pmax,
                        void* stream);
/* Table mode of the 16-bit core (csrc/deform_attn16.hip, "table mode"): the continuous position bias CPB(slog(gq - vs)) of
 * models/DeformableAttention2D.py:120-157 / DeformableAttention1D.py:60-102 is ONE function of the posdim signed-log offsets for every
 * pair of a launch, so the caller evaluates the MLP once on a grid - `table` [H / G, points^posdim] fp32, point (i0, i1) at index
 * i1 * points + i0 <-> p = -table_pmax + i * 2 table_pmax / (points - 1), points = smml_deform_attn_table_points(posdim) - and the
 * kernels interpolate it (bi)linearly per pair (positions beyond +-table_pmax take the edge value).  The backward returns
 * d table (the interpolation's adjoint: a histogram of d bias over the cells, fp32 atomics in LDS - the one order-dependent sum) in
 * place of the six parameter gradients, which follow from it through the table's own small backward; dq / dk / dv / dvs as in
 * smml_deform_attn16_bwd (fixed-order reductions).  grid_h, grid_w > 0 (posdim 2, grid_h * grid_w = N, both <= 128) assert that the
 * queries sit on a regular grid - gq[y * grid_w + x] = (X[x], Y[y]), as the 2-D module's do: the weight of a pair then factorises over
 * the axes and d table becomes two small dense products per key on the matrix pipe (no atomics: fixed-order sums); 0, 0 = any queries.
 * An additive mode: values differ from the per-pair MLP by the interpolation error
 * (tests/test_gpu_deform_table.py states the bounds); everything else - logits16 / dlogits16 layouts, dtype, dropout, events - as above. */
int smml_deform_attn_table_points(int posdim);
size_t smml_deform_attn_table_bwd_workspace_bytes(int B, int N, int J, int H, int posdim);
int smml_deform_attn_table_fwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* table,
                               float* out, float* lse, unsigned short* logits16, int B, int N, int J, int H, int G, int posdim,
                               int table_g, float table_pmax, float scale, float dropout_p, unsigned long long dropout_seed, int dtype,
                               void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts);
int smml_deform_attn_table_bwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* table,
                               const float* out, const float* dout, const float* lse, const unsigned short* logits16,
                               unsigned short* dlogits16, float* dq, float* dk, float* dv, float* dvs, float* dtable, void* workspace,
                               size_t workspace_bytes, int B, int N, int J, int H, int G, int posdim, int table_g, float table_pmax,
                               int grid_h, int grid_w, float scale, float dropout_p, unsigned long long dropout_seed, int dtype,
                               void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts);


/* attention dropout (nn.Dropout on the probabilities, DeformableAttention2D.py:309 / 1D :229): dropout_p in
 * [0, 1) and a 64-bit seed select a counter-based keep decision per (b, h, query, key); pass the same pair to
 * forward and backward.  The mask itself (0 / 1 floats [B, H, N, J]) is only materialised for tests: */
int smml_deform_attn_dropout_mask_f32(float* mask, int B, int N, int J, int H, float dropout_p,
                                      unsigned long long dropout_seed, void* stream, const SmmlDeformOpts* opts);

/* decision export (parity tests only): the ReLU decisions [W1 p + b1 > 0] of the position bias's FIRST layer
 * (DeformableAttention2D.py:129-131, 1D :69-71) bit for bit as the fused forward and the position-bias backward evaluate them,
 * masks [(B G), J, 2, smml_deform_attn_nst(N)] uint16 in the bit order of relu_masks above.  Together with relu_masks (layer 2)
 * and smml_bilinear_corners_f32 (the sampler's cells) this is every piecewise-linear decision of the module; tests impose
 * them on the fp64 oracle so that gradient comparisons do not depend on rounding-level ties. */
int smml_deform_attn_relu1_masks(const float* vs, const float* gq, const float* w1, const float* b1, unsigned short* masks,
                                 int B, int N, int J, int G, int posdim, void* stream, const SmmlDeformOpts* opts);

/* ------------------------------------------------------------------------------------------------
 * Nystrom landmark self-attention, HBM-bound pieces (the contractions of models/NystromAttention.py:86,
 * 122-140 and the pinv iteration :20-35 run through smml_gemm_f32 with alpha / beta epilogues).
 * Row softmax over the last dim L of x [rows, L] (:137) and its backward dx = y (dy - sum(dy y)). */
int smml_softmax_fwd_f32(const float* x, float* y, long long rows, int L, void* stream);
int smml_softmax_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int L, void* stream);
/* dst[b, r, c] = scale * src[b, c]: backward of the landmark segment mean (:102-118; the forward is
 * smml_colsum_f32 over [B*h*m, l, d] with scale 1/l). */
int smml_tile_rows_f32(const float* src, float* dst, long long nb, int R, int C, float scale, void* stream);
/* depthwise residual convolution along the tokens (:72,144-145): v [B, H, n, D], w [H, KW] ->
 * out_merged [B, n, H*D] (heads merged, the layout the output projection consumes). */
/* bf16-STORAGE forms of the n'-sized kernels of the block's bf16 compute mode; qkv / dqkv are the token-major bf16 buffers
 * [B, n, 3, H, D] the projection GEMM writes / its backward reads (n = m l).
 *   segment_mean: qmean, kmean fp32 [B, H, m, D] = landmark means of q and of k (models/NystromAttention.py:102-118)
 *   segment_mean_bwd_add: dqkv(q, k parts) += (dqmean, dkmean) broadcast / l
 *   resconv: 33-tap depthwise convolution over tokens (:62-66,144-145); element (b, t, h, d) of in at b i_bs + t i_rs + h D + d (elements),
 *            of out at b o_bs + t o_rs + h D + d; flip = 1 applies the reversed taps (gradient w.r.t. the input)
 *   resconv_wgrad: dw fp32 [H, 33] += sum dout[b, t, h, d] v[b, t + k - 16, h, d] */
int smml_segment_mean_b16(const void* qkv, float* qmean, float* kmean, int B, int n, int l, int H, int D, void* stream);
int smml_segment_mean_bwd_add_b16(void* dqkv, const float* dqmean, const float* dkmean, int B, int n, int l, int H, int D, void* stream);
int smml_resconv_b16(const void* in, const float* w, void* out, int B, int H, int n, int D, int KW, long long i_bs, long long i_rs,
                     long long o_bs, long long o_rs, int flip, void* stream);
int smml_resconv_wgrad_b16(const void* dout, const void* v, float* dw, int B, int H, int n, int D, int KW, long long g_bs, long long g_rs,
                           long long v_bs, long long v_rs, void* stream);

/* Newton-Schulz pseudo-inverse iteration (models/NystromAttention.py:28-33; dup cmta_utils.py:152-157), `iters` times, on NB problems of
 * m x m (row-major, contiguous): z <- 1/4 z (13 I - x z (15 I - x z (7 I - x z))).  One host call issues the whole chain of batched
 * products (4 per iteration forward, 8 + one update backward) on `stream`.  reduced != 0: the caller accepts products with 16-bit
 * mantissas (the block's 16-bit compute mode; m = 256 then runs as two bf16 planes per matrix on the 16-bit matrix pipe); reduced = 0:
 * exact fp32 products.  The forward and the backward of one call take the same `reduced`.
 * fwd: saved (smml_newton_schulz_saved_floats floats) receives what the backward needs of every iteration; z_out [NB, m, m] = the result.
 * bwd: dx, dz0 [NB, m, m] overwritten; scratch = smml_newton_schulz_scratch_floats floats.  All buffers 16-byte aligned. */
size_t smml_newton_schulz_saved_floats(int NB, int m, int iters, int reduced);     /* sizes, in floats, of `saved` and `scratch` below */
size_t smml_newton_schulz_scratch_floats(int NB, int m, int iters, int reduced);
/* measurement / test switch: 0 every product through smml_gemm_f32; 1 the exact-fp32 chain kernel for m = 256 whatever `reduced`; 2 (default) */
void smml_newton_schulz_set_fast(int on);
int smml_newton_schulz_fwd(const float* x, const float* z0, float* saved, float* z_out, int NB, int m, int iters, int reduced, void* stream);
int smml_newton_schulz_bwd(const float* x, const float* z0, const float* saved, const float* dz_in, float* dx, float* dz0,
                           float* scratch, int NB, int m, int iters, int reduced, void* stream);

int smml_resconv_fwd_f32(const float* v, const float* w, float* out_merged, int B, int H, int n, int D, int KW,
                         void* stream);
/* dv [B, H, n, D] overwritten, dw [H, KW] accumulated into */
int smml_resconv_bwd_f32(const float* dout_merged, const float* v, const float* w, float* dv, float* dw, int B, int H,
                         int n, int D, int KW, void* stream);
/* PPEG (models/mil.py:192-206): depthwise 7x7 + 5x5 + 3x3 + identity as ONE merged 7x7 depthwise pass on
 * channel-last maps x [B, H, W, C]; wm [C, 49]; flip = 1 applies the rotated kernel (data gradient). */
int smml_dwconv7_fwd_f32(const float* x, const float* wm, const float* bias, float* y, int B, int H, int W, int C, int flip,
                         void* stream);
/* dwm [C, 49], db [C] accumulated into */
int smml_dwconv7_bwd_weight_f32(const float* x, const float* dy, float* dwm, float* db, int B, int H, int W, int C,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * Train-step glue: the gradient-modulation block of the reference's training loop,
 * train_test.py:87-184 (task types diag2021 / grade / subtype; 'survival' below), as one launch with no host synchronisation:
 * per-branch logits feat_x W_x^T + b / 2, score_x = sum_b softmax(.)[label_b], ratio_t = score_t / score_i, and for
 * every class row of classifier.weight.grad [C, 2 hs] whose tumor / immune halves have negative cosine similarity the
 * projection + renormalisation of the weaker branch's half (:158-183), IN PLACE in weight_grad.
 *   feat_t, feat_i [B, hs]   weight [C, 2 hs]   bias [C]   label [B] int64   weight_grad [C, 2 hs]
 *   info (nullable) [4 + 2 C]: score_t, score_i, ratio_t, ratio_i, then (cosine similarity, branch taken 0 / 1 / 2) per row. */
int smml_grad_modulate_f32(const float* feat_t, const float* feat_i, const float* weight, const float* bias,
                           const long long* label, float* weight_grad, float* info, int B, int C, int hs, void* stream);
/* task_type 'survival' of the same block (train_test.py:99-102,121-149): the two branch scores are the concordance indices of
 * risk_b = -sum_t cumprod_t (1 - sigmoid(out_b)) against (censor, survtime) - what the reference obtains on the host from scikit-survival's
 * concordance_index_censored(event = 1 - censor, survtime, risk, tied_tol = 1e-8) (utils/utils.py:315-317): a sample with an event is
 * comparable with every sample of later time and with every sample censored at the same time; concordant if its risk is larger, tied within
 * 1e-8 counts one half.  censor, survtime [B] fp32 (censor 1 = censored).  All samples censored (:127-133) or no comparable pair: no edit.
 * info[0..1] = cindex_t, cindex_i, the rest as above.  scikit-survival is absent from the build image: parity of this entry is unpinned. */
int smml_grad_modulate_survival_f32(const float* feat_t, const float* feat_i, const float* weight, const float* bias,
                                    const float* censor, const float* survtime, float* weight_grad, float* info, int B, int C, int hs,
                                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-format step in front of the path: bags stored as packed bf16 rows, resampled to the fixed
 * instance count on the device by the reference's index rule (data/dataset.py:151-175: i mod n when the bag is short,
 * int(np.around(i * (n / fixdim))) when it is long - IEEE double, round-half-even; bit-exact integer path).
 *   src [n_rows, dim] bf16 (as uint16)   dst [fixdim, dim] fp32 (out_is_f32 != 0) or bf16; dim % 8 == 0, 16-byte aligned.
 * smml_fixdim_indices writes the fixdim source-row indices alone (int64), for the bit-exactness tests. */
int smml_fixdim_indices(long long* out, long long n_rows, long long fixdim, void* stream);
int smml_fixdim_gather_bf16(const unsigned short* src, long long n_rows, void* dst, int out_is_f32, long long fixdim, int dim,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused softmax attention on the 16-bit matrix pipe: O = softmax(scale Q K^T) V with fp32 storage, bf16 (use_fp16 = 0) or
 * fp16 (use_fp16 = 1) operands converted on the fly and fp32 accumulation; the [Lq, Lk] probability matrix is never written
 * (forward keeps one base-2 log-sum-exp per query in lse2, backward recomputes).  Serves the two attention-shaped products of
 * the Nystrom block in its 16-bit compute mode (models/NystromAttention.py:122-140): softmax(q kl^T) (z (attn3 v)) with
 * (Lq, Lk) = (n', m) and softmax(ql k^T) v with (Lq, Lk) = (m, n').
 *   q [BH, Lq, 64]   k, v [BH, Lk, 64]   dq [BH, Lq, 64]   dk, dv [BH, Lk, 64]   lse2 [BH, Lq]   (D must be 64)
 *   out / dout: head-major [BH, Lq, 64] (heads_merged = 0) or heads merged [BH / H, Lq, H * 64] (heads_merged = H: the layout
 *   the block's output projection consumes).  accumulate != 0: out += result (out already holds the residual convolution of
 *   v, :144-145); the backward then takes that residual (same layout, nullable) to recover the attention output.
 * Backward: dq / dk / dv overwritten; workspace of smml_attn16_bwd_workspace_bytes(...) bytes, 16-byte aligned. */
size_t smml_attn16_fwd_workspace_bytes(int BH, int Lq, int Lk);   /* 0 unless the shape runs key-split (few queries, many keys) */
int smml_attn16_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse2, void* workspace,
                        size_t workspace_bytes, int BH, int Lq, int Lk, int D, float scale, int use_fp16, int heads_merged,
                        int accumulate, void* stream);
/* forward kernel choice for Lk <= 256 (measurement / test hook): 0 (default) = always the online-softmax kernel, 1 = the two-pass
 * kernel with all keys resident in LDS wherever its 256-query workgroups fill the chip, 2 = wherever Lk <= 256.  The two-pass kernel
 * halves the vector instructions per MFMA but the launch is HBM-bound with fp32 storage (csrc/attn16.hip). */
void smml_attn16_set_fewkeys(int mode);
/* queries-long bf16-storage forward (smml_attn16_fwd_b16, long_side 1): 0 = one 32-query block per wave, 1 (default) = two blocks per wave
 * (the K / V fragments of a tile feed two accumulator sets: half the LDS fragment reads, barriers and K / V staging per MFMA) wherever the
 * 256-query workgroups still cover the chip twice over, 2 = always (test / measurement switch). */
void smml_attn16_set_query_blocks(int mode);
/* bf16-STORAGE forms of the two attention-shaped products (bf16 compute mode with bf16 bags): the LONG side (long_side 0: keys / values -
 * softmax(ql k^T) v; 1: queries - softmax(q kl^T) w) is bf16 in memory at element strides (s_bs, s_hs, s_rs) per (bag, head, row) - q / k / v
 * are read in the token-major [b, n', 3, h, 64] buffer the projection wrote - the short (landmark) side is fp32 [B H, L, 64].
 *   long_side 0: q fp32, k / v bf16, out fp32 [B H, Lq, 64]; residual must be NULL
 *   long_side 1: q bf16, k / v fp32, out = attention + residual, both bf16 at strides (o_bs, o_hs, o_rs) (residual may be NULL)
 * Backward: gradients of the long side are bf16 at strides (g_bs, g_hs, g_rs) (e.g. inside the gradient of the qkv buffer), those of the
 * short side fp32.  long_side 0: dout / out fp32, dq fp32, dk written, dv written or added to (dv_accumulate);
 * long_side 1: out / residual / dout bf16 at the o strides, dq bf16 written, dk / dv fp32.  Workspaces as for the fp32-storage entries. */
int smml_attn16_fwd_b16(const void* q, const void* k, const void* v, void* out, const void* residual, float* lse2, void* workspace,
                        size_t workspace_bytes, int B, int H, int Lq, int Lk, float scale, int long_side, long long s_bs, long long s_hs,
                        long long s_rs, long long o_bs, long long o_hs, long long o_rs, void* stream);
int smml_attn16_bwd_b16(const void* q, const void* k, const void* v, const void* out, const void* residual, const void* dout,
                        const float* lse2, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes, int B, int H, int Lq,
                        int Lk, float scale, int long_side, long long s_bs, long long s_hs, long long s_rs, long long o_bs, long long o_hs,
                        long long o_rs, long long g_bs, long long g_hs, long long g_rs, int dv_accumulate, void* stream);
size_t smml_attn16_bwd_workspace_bytes(int BH, int Lq, int Lk);
int smml_attn16_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* residual, const float* dout,
                        const float* lse2, float* dq, float* dk, float* dv, void* workspace, size_t workspace_bytes, int BH, int Lq,
                        int Lk, int D, float scale, int use_fp16, int heads_merged, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMML_H_ */
