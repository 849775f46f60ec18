/*
 * pcgc.h — C ABI of the MI355X-native PCGCv1 hot path.
 *
 * Two shared libraries implement it:
 *   libpcgc_hip.so   (hipcc, gfx950)  every pcgc_* function that takes a stream
 *   libpcgc_host.so  (g++)            the sequential host tail: range coder, CDF
 *                                      quantiser, partition / merge, ply text
 *
 * The reference is pure Python on TensorFlow 1.13; its "FFI" for this path is
 * the set of TF ops its Python calls.  Each entry point below names the
 * reference call site it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - extern "C"; every function returns int: 0 = ok, < 0 = error
 *     (pcgc_last_error() / pcgc_host_last_error() give a thread-local message).
 *   - Caller owns every buffer.  Device pointers are plain pointers obtained
 *     from any HIP allocator (the Python host passes torch allocations).
 *     Device memory the library allocates itself: the weight copy inside a pcgc_net
 *     (freed by pcgc_net_destroy), a stream-ordered temporary for the repacked weights of a
 *     single pcgc_conv3d_fwd call on a matrix-core shape (hipMallocAsync / hipFreeAsync on the
 *     caller's stream), and one 512 KiB log2 table per device (first pcgc_laplace_cdf call).
 *     Everything else comes from the caller, sized by the *_workspace_bytes functions.
 *   - Every device function enqueues on the caller's stream (hipStream_t passed
 *     as void*), asynchronously, with no hidden synchronisation.
 *   - Activations are NDHWC float32, contiguous.  Weights are passed in
 *     TensorFlow layouts: Conv3D [kd,kh,kw,Cin,Cout], Conv3DTranspose
 *     [kd,kh,kw,Cout,Cin] (models/model_voxception.py:21-54, 164-182).
 *   - Empty inputs (B = 0 cubes / n = 0 elements / rows = 0) are valid no-ops of the forward, likelihood, CDF and
 *     top-k entry points: they return 0 without touching any pointer (which may then be NULL).
 *   - No floating-point atomics and no grid-size-dependent reduction order
 *     anywhere: results are bit-identical for any batch size / batch slot /
 *     GPU count (the reference's known enc/dec mismatch, README.md:111-114).
 */
#ifndef PCGC_H_
#define PCGC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pcgc_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------ */
/* libpcgc_hip.so                                                      */
/* ------------------------------------------------------------------ */
int pcgc_version(void);
const char* pcgc_last_error(void);

/* One Keras Conv3D / Conv3DTranspose (padding='same') + bias + optional ReLU.
 * Replaces tf.keras.layers.Conv3D/Conv3DTranspose.__call__
 * (models/model_voxception.py:21-54, 83-122, 153-192, 224-244, 263-297).
 *   x [B,D,D,D,Cin] -> y [B,Do,Do,Do,Cout];  Do = D (stride 1), D/2 (stride 2),
 *   2D (transposed).  ksize odd, 1..9 (3 and 1 in model_voxception.py, 5 and 9 in model_simple.py:20-41, 56-86:
 *   those run on the generic direct kernel); stride 1 or 2; transposed implies stride 2.  TF 'SAME' padding: a
 *   stride-2 conv pads ksize-2 voxels in total, the smaller half in front; the transposed conv is its adjoint.
 *   bias may be NULL (down_1/down_2, model_voxception.py:99,111).
 *   algo: 0 = auto (MFMA kernel when the shape has one, else direct),
 *         1 = force the direct (VALU) kernel, 2 = force MFMA (error if none). */
int pcgc_conv3d_fwd(const float* x, const float* kernel, const float* bias, float* y,
                    int B, int D, int Cin, int Cout, int ksize, int stride,
                    int transposed, int relu, int algo, pcgc_stream_t stream);

/* One _VoxceptionResNet block (model_voxception.py:11-68, call 56-68):
 *   out = relu(x + concat[ relu(conv1_2(relu(conv1_1(x)))),
 *                          relu(conv2_3(relu(conv2_2(relu(conv2_1(x)))))) ])
 * x, out: [B, D, D, D, C] NDHWC fp32 (out may alias x).  params = the block's ten tensors in the
 * order of its layers {conv1_1, conv1_2, conv2_1, conv2_2, conv2_3}, kernel then bias each, kernels in
 * the Keras layout [kd,kh,kw,Cin,Cout].  C = 16 at D = 64, C = 32 at D = 32 and C = 64 at D = 16 (the three stages
 * of both transforms) run the v_mfma_f32_4x4x1 row kernels that pcgc_net_forward uses for those stages;
 * any other C % 4 == 0 runs the generic layer kernels.  Workspace from pcgc_vrn_workspace_bytes. */
size_t pcgc_vrn_workspace_bytes(int B, int D, int C);
int pcgc_vrn_fwd(const float* x, const float* const* params, float* out, int B, int D, int C,
                 void* workspace, size_t workspace_bytes, pcgc_stream_t stream);

/* Whole transforms.  kind selects the layer table (pcgcv1_amd/models/spec.py,
 * restating model_voxception.py:71-308). */
enum {
  PCGC_NET_ANALYSIS = 0,      /* AnalysisTransform.call   model_voxception.py:125-144 */
  PCGC_NET_SYNTHESIS = 1,     /* SynthesisTransform.call  model_voxception.py:195-214 */
  PCGC_NET_HYPER_ENCODER = 2, /* HyperEncoder.call        model_voxception.py:246-252 */
  PCGC_NET_HYPER_DECODER = 3  /* HyperDecoder.call        model_voxception.py:299-308 */
};
typedef struct pcgc_net pcgc_net;

/* Number of parameter tensors `params` must hold for `kind`: for every layer of
 * the table, in order, its kernel then (if the layer has one) its bias. */
int pcgc_net_param_count(int kind);
/* params[i] are DEVICE pointers in TF layouts; they are repacked into the
 * library's MFMA operand layout on `stream`; the caller may free them after
 * synchronising the stream.  Replaces tf.train.Checkpoint.restore binding
 * (transform.py:107-112, 214-218). */
int pcgc_net_create(int kind, const float* const* params, int n_params,
                    pcgc_stream_t stream, pcgc_net** out);
void pcgc_net_destroy(pcgc_net* net);
/* algo 0 (default): MFMA kernels wherever a shape has one; 1: direct (VALU) kernels only
 * (the on-device cross-check the parity tests use). */
int pcgc_net_set_algo(pcgc_net* net, int algo);
/* Exact skipping of empty space (AnalysisTransform at cube size 64; model_voxception.py:125-131 = conv_in + the three
 * C = 16 blocks).  A 64^3 cube of a voxelised surface is ~98 % zeros; wherever the receptive field of a wave tile of a
 * layer's output holds no occupied voxel the tile equals, bit for bit, the same tile of that layer's response to an
 * all-zero cube (kept per net, made by the same kernels at pcgc_net_create), and the wave copies it instead of computing
 * it.  Results are identical with and without (environment PCGC_SKIP_EMPTY=0 computes every tile).
 * PCGC_SKIP_EMPTY=3 (the default): conv_in and the three C = 16 blocks work on SLOTS of 8 planes x 2 rows x 16 voxels instead
 * of whole-row tiles — four slots to a wave, only the heavy ones launched, a slot nobody wrote read from the empty-cube
 * response by its reader (csrc/vrn_seg.hip); 1: whole-row tiles that are not written, 2: whole-row tiles that are copied.
 * The workspace (pcgc_net_workspace_bytes) then also holds the slot lists and room for a copy of the empty-cube responses,
 * which is made only when the net's own copy and the chunk's tensors do not fit one 2 GiB buffer window.
 * pcgc_rowocc: rowocc[b * 64 + d] bit h = row (d, h) of cube b holds a voxel that is not +0.0 — what the row-tile forms test.
 * pcgc_net_set_skip_counter: test aid — a device word that receives +1 per skipped wave tile / slot (NULL: off). */
int pcgc_rowocc(const float* x, unsigned long long* rowocc, int B, pcgc_stream_t stream);
int pcgc_net_set_skip_counter(pcgc_net* net, unsigned* device_counter);
/* Per-launch timing for bench.py's roofline line: when on, every layer launch of pcgc_net_forward is
 * bracketed by hipEvents on the caller's stream.  pcgc_net_profile_report drains them as text, one line per
 * launch: "layer_index layer_name mfma|direct Cin Cout k mode B Din milliseconds".  Call with buf = NULL to
 * get the size in *needed (the records are consumed by the call that copies them). */
int pcgc_net_set_profiling(pcgc_net* net, int on);
int pcgc_net_profile_report(pcgc_net* net, char* buf, size_t cap, size_t* needed);
/* Scratch the forward pass needs for a batch of B cubes whose INPUT spatial
 * size is D (64 for analysis, 16 for synthesis / hyper encoder, 8 for hyper
 * decoder at cube_size 64). */
size_t pcgc_net_workspace_bytes(const pcgc_net* net, int B, int D);
/* Forward pass.  out1 is only used by the hyper decoder: out0 = loc,
 * out1 = max(|scale|, scale_lower_bound)  (model_voxception.py:308 +
 * transform.py:145-146, 232-233; train_hyper.py:189).
 * Replaces tf.map_fn(loop_analysis | loop_synthesis | loop_hyper_*),
 * transform.py:116-147, 224-257 — batched instead of one cube per call. */
int pcgc_net_forward(const pcgc_net* net, const float* x, float* out0, float* out1,
                     int B, int D, float scale_lower_bound, void* workspace,
                     size_t workspace_bytes, pcgc_stream_t stream);

/* ---- entropy-model kernels -------------------------------------------- */

/* tf.math.round (half-to-even) of n floats + per-segment min / max of the
 * rounded values (segments = consecutive runs of seg_len elements).
 * Replaces _quantize(..., "symbols") + reduce_min/max,
 * conditional_entropy_model.py:151-154 (per cube) and entropy_model.py:246-250
 * (one segment = whole batch).  q may be NULL.  seg_min/seg_max: int32[n/seg_len]. */
int pcgc_repro_eval(int fn, const float* x, float* y, int64_t n, pcgc_stream_t stream);

int pcgc_round_minmax(const float* x, float* q, int32_t* seg_min, int32_t* seg_max,
                      int64_t n, int64_t seg_len, pcgc_stream_t stream);
/* The same with the rounded values as int16 — the coder's input type (entropy_model.py:253-258: cast, - min_v, cast to int16,
 * range_encode): no float tensor, no conversion pass.  Values outside 16 bits saturate; the caller reads seg_min / seg_max
 * anyway and refuses such a range. */
int pcgc_round_minmax_i16(const float* x, int16_t* q, int32_t* seg_min, int32_t* seg_max,
                          int64_t n, int64_t seg_len, pcgc_stream_t stream);
/* Decoder side of the same cast (entropy_model.py:298-304: range_decode -> + min_v -> float32):
 * out[i] = (float)sym[i] + offset. */
int pcgc_symbols_to_values(const int16_t* sym, int offset, float* out, int64_t n, pcgc_stream_t stream);
/* Per-cube form (conditional_entropy_model.py:195-199: each cube's decoded symbols + that cube's min_v):
 * out[i] = (float)sym[i] + seg_offset[i / seg_len], seg_offset float32 [n / seg_len] on the device. */
int pcgc_symbols_to_values_seg(const int16_t* sym, const float* seg_offset, float* out, int64_t n, int64_t seg_len,
                               pcgc_stream_t stream);

/* SymmetricConditional.__call__ (conditional_entropy_model.py:71-93), eval or
 * training (noise = U(-.5,.5) supplied by the caller, may be NULL for eval):
 * values = round(y) or y+noise; likelihood = max(|c(up')-c(lo')|, bound). */
int pcgc_laplace_likelihood(const float* y, const float* loc, const float* scale,
                            const float* noise, float* values, float* likelihood,
                            int64_t n, float likelihood_bound, pcgc_stream_t stream);

/* SymmetricConditional._get_cdf (conditional_entropy_model.py:95-124) fused with
 * pmf_to_quantized_cdf (TF 1.13 contrib/coder, call site :122) for `rows`
 * (voxel,channel) rows grouped in segments of seg_rows rows (one segment = one
 * cube); segment s uses the integer support [seg_min[s], seg_max[s]].
 *   cdf_lower: uint16 [rows, ncols]; entry k is the quantised CDF value at
 *              symbol k (cdf[0] = 0); the implicit entry past the last symbol
 *              is 65536.  ncols >= max over segments of (max-min+1).
 *   If symbols != NULL (float, already rounded) also writes
 *   lohi[row] = lower | (upper-1) << 16 for that row's symbol (what range_encode
 *   consumes, conditional_entropy_model.py:161).  Either output may be NULL.
 * The first call on a device allocates and uploads a 512 KiB table of log2(v), v in [0, 65536] (computed on the
 * host in double, like the oracle) and synchronises once; later calls are asynchronous on `stream`. */
int pcgc_laplace_cdf(const float* loc, const float* scale, const int32_t* seg_min,
                     const int32_t* seg_max, int64_t rows, int64_t seg_rows, int ncols,
                     float likelihood_bound, const float* symbols, uint16_t* cdf_lower,
                     uint32_t* lohi, pcgc_stream_t stream);

/* EntropyBottleneck.__call__ (entropy_model.py:153-181).  params: the 12 tensors
 * matrix_0,bais_0,factor_0,...,matrix_3,bais_3,factor_3 packed back to back
 * (C*(3+3+3 + 9+3+3 + 9+3+3 + 3+1+1) floats, entropy_model.py:50-66), device. */
int pcgc_factorized_likelihood(const float* z, const float* params, const float* noise,
                               float* values, float* likelihood, int64_t n, int C,
                               float likelihood_bound, pcgc_stream_t stream);
/* EntropyBottleneck._get_cdf pmf part (entropy_model.py:199-214): pmf [C, N] over
 * the integers min_v..max_v (device output). */
int pcgc_factorized_pmf(const float* params, int C, int min_v, int max_v,
                        float likelihood_bound, float* pmf, pcgc_stream_t stream);

/* ---- decoder tail ------------------------------------------------------ */
/* select_voxels + get_adaptive_thres (dataprocess/inout_points.py:147-179):
 * per cube b: k = k_per_cube[b]; threshold = k-th largest of the values > -2.0
 * (all values if fewer than k; k = 0 -> the smallest candidate); or
 * fixed_thres when use_fixed != 0.  Writes thresholds[b] and, if mask != NULL,
 * mask[b, v] = (x >= thres) as uint8.  vox = voxels per cube. */
int pcgc_topk_threshold(const float* x, const int32_t* k_per_cube, int B, int64_t vox,
                        int use_fixed, float fixed_thres, float* thresholds,
                        uint8_t* mask, void* workspace, size_t workspace_bytes,
                        pcgc_stream_t stream);
size_t pcgc_topk_workspace_bytes(int B, int64_t vox);

/* loss.get_bce_loss (loss.py:8-33): sums[0..3] = sum_{label=0} -log(1-o),
 * count(label=0), sum_{label>0} -log(o), count(label>0); o = clip(sigmoid(pred)).
 * Deterministic two-stage reduction (double). */
int pcgc_bce_sums(const float* pred, const float* label, int64_t n, double* sums4,
                  void* workspace, size_t workspace_bytes, pcgc_stream_t stream);
size_t pcgc_bce_workspace_bytes(int64_t n);

/* loss.get_confusion_matrix / get_classify_metrics (loss.py:35-78).  pred, label: n floats; a voxel is positive
 * when its value is > th.  pcgc_classify_sums: sums3 = {TP, FP, FN} as exact counts (wavefront ballot + popcount,
 * fixed-order two-stage sum); precision = TP/(TP+FP), recall = TP/(TP+FN), IoU = TP/(TP+FP+FN) are formed by the
 * caller.  pcgc_confusion_matrix writes the three 0/1 maps TP = p*l, FP = p*(1-l), FN = (1-p)*l. */
size_t pcgc_classify_workspace_bytes(void);
int pcgc_classify_sums(const float* pred, const float* label, int64_t n, float th, double* sums3,
                       void* workspace, size_t workspace_bytes, pcgc_stream_t stream);
int pcgc_confusion_matrix(const float* pred, const float* label, int64_t n, float th, float* tp,
                          float* fp, float* fn, pcgc_stream_t stream);

/* loss.get_focal_loss (loss.py:83-93): y_pred are probabilities, y_true is 0 / 1.
 *   loss = - sum alpha (1 - pt_1)^gamma log(pt_1) - sum (1 - alpha) pt_0^gamma log(1 - pt_0),
 *   pt_1 = clip(y_true == 1 ? y_pred : 1, 1e-3, .999), pt_0 = clip(y_true == 0 ? y_pred : 0, 1e-3, .999)
 * (float32 per element, wavefront butterfly + fixed-order double sum).  _bwd: dy_pred = grad_scale * dloss/dy_pred,
 * zero where y_pred is outside the clip range. */
size_t pcgc_focal_workspace_bytes(void);
int pcgc_focal_loss(const float* y_pred, const float* y_true, int64_t n, float gamma, float alpha, double* loss,
                    void* workspace, size_t workspace_bytes, pcgc_stream_t stream);
int pcgc_focal_loss_bwd(const float* y_pred, const float* y_true, int64_t n, float gamma, float alpha,
                        float grad_scale, float* dy_pred, pcgc_stream_t stream);

/* D1 (point-to-point) distortion of MPEG pc_error as the reference's eval uses it
 * (myutils/pc_error_wrapper.py:26-75; eval.py:194-207): out2[0] = mean over the points of A of the squared
 * distance to the nearest point of B, out2[1] = the largest such squared distance (squared Hausdorff).
 * Points are int32 xyz with 0 <= coordinate < res.  Call twice (A->B, B->A); PSNR = 10 log10(3 peak^2 / max mse).
 * Exact (integer distances); deterministic. */
size_t pcgc_d1_workspace_bytes(int res);
int pcgc_d1_mse(const int32_t* pa, int64_t na, const int32_t* pb, int64_t nb, int res, double* out2,
                void* workspace, size_t workspace_bytes, pcgc_stream_t stream);

/* D2 (point-to-plane) distortion of MPEG pc_error 0.13.4 (`-n normal1`, neighborsProc 1, averageNormals 1;
 * myutils/pc_error_wrapper.py:46-51).  The target cloud Q is passed as its linear keys (x*res + y)*res + z, int64,
 * SORTED ascending and unique; per-point arrays of Q (normals) are in that order.
 *   pcgc_d2_transfer_normals: normals_q[j] = mean of normals_p[i] over every i whose nearest-neighbour set in Q
 *     (all points at the minimal distance) contains j; zero where no point of P maps to j.
 *   pcgc_d2_mse: out2[0] = mean over p of mean_{q in T(p)} ((p-q).normal_q)^2, out2[1] = the largest such value.
 * Deterministic (integer atomics / fixed-order double sums). */
size_t pcgc_d2_workspace_bytes(int res, int64_t nq);
int pcgc_d2_transfer_normals(const int32_t* p, int64_t np, const float* normals_p, const int64_t* qkeys, int64_t nq,
                             int res, float* normals_q, void* workspace, size_t workspace_bytes, pcgc_stream_t stream);
int pcgc_d2_mse(const int32_t* p, int64_t np, const int64_t* qkeys, int64_t nq, const float* normals_q, int res,
                double* out2, void* workspace, size_t workspace_bytes, pcgc_stream_t stream);

/* points2voxels (dataprocess/inout_points.py:116-132) on device: scatter
 * n points (cube index, x, y, z as int32 x4) into zero-initialised float cubes. */
int pcgc_voxelize(const int32_t* cube_xyz, int64_t n, int cube_size, float* cubes,
                  int B, pcgc_stream_t stream);
/* The same straight from pcgc_partition's outputs (process.py:33-36 runs partition and points2voxels back to back):
 * points int32 [n,3] global coordinates, cube_of_point int32 [n] index into the key-sorted cubes (-1 = point of a
 * dropped cube); fills the zero-initialised cubes [cube_hi - cube_lo, cs, cs, cs] of the cubes cube_lo <= c < cube_hi
 * (a rank's block; 0, B for all).  Coordinates are reduced mod cube_size here. */
int pcgc_voxelize_points(const int32_t* points, const int32_t* cube_of_point, int64_t n, int cube_size,
                         int cube_lo, int cube_hi, float* cubes, pcgc_stream_t stream);

/* ---- training step (train_hyper.py:174-214) ------------------------------ */
/* Gradient of one Conv3D / Conv3DTranspose layer.  D = spatial size of the layer's INPUT x; dz = gradient w.r.t.
 * the layer's pre-activation output (apply pcgc_relu_bwd first), contiguous [B,Do^3,Cout].  Replaces what
 * tf.GradientTape derives for tf.keras.layers.Conv3D/Conv3DTranspose (train_hyper.py:202-207).
 * bwd_data: dx [B,D^3,Cin];  bwd_weight: dkernel in the layer's TF layout, dbias [Cout] or NULL.
 * Deterministic (two-stage fixed-order reductions).  workspace: pcgc_conv3d_bwd_workspace_bytes. */
size_t pcgc_conv3d_bwd_workspace_bytes(int Cin, int Cout, int ksize);
int pcgc_conv3d_bwd_data(const float* dz, const float* kernel, float* dx, int B, int D, int Cin, int Cout,
                         int ksize, int stride, int transposed, void* workspace, size_t workspace_bytes,
                         pcgc_stream_t stream);
/* bwd_data with the elementwise neighbours of the reverse pass fused into its epilogue:
 *   dx = (relu_mask > 0) ? (add_to + conv_bwd_data(dz)) : 0
 * relu_mask (or NULL) = the layer's forward input x when x is a ReLU output — the gradient of the producing ReLU, so the
 * caller needs no pcgc_relu_bwd for it; add_to (or NULL, may alias dx) = gradient already accumulated for x from its
 * other consumers (the residual branch of a VRN block).  Both [B,D^3,Cin] like dx. */
int pcgc_conv3d_bwd_data_fused(const float* dz, const float* kernel, float* dx, const float* relu_mask,
                               const float* add_to, int B, int D, int Cin, int Cout, int ksize, int stride,
                               int transposed, void* workspace, size_t workspace_bytes, pcgc_stream_t stream);
int pcgc_conv3d_bwd_weight(const float* x, const float* dz, float* dkernel, float* dbias, int B, int D,
                           int Cin, int Cout, int ksize, int stride, int transposed, void* workspace,
                           size_t workspace_bytes, pcgc_stream_t stream);
/* dz[v,c] = dy[v*dy_cs + dy_co + c] * (y[v,c] > 0)  (ReLU backward on a channel slice of dy; y NULL = copy). */
int pcgc_relu_bwd(const float* dy, int dy_cs, int dy_co, const float* y, float* dz, int64_t nvox, int C,
                  pcgc_stream_t stream);
/* out = relu(x + concat(t12, t23))  — the block tail of _VoxceptionResNet.call (model_voxception.py:65-67); C = channels
 * of x, a multiple of 8. */
int pcgc_vrn_merge(const float* x, const float* t12, const float* t23, float* out, int64_t nvox, int C,
                   pcgc_stream_t stream);
int pcgc_add_inplace(float* a, const float* b, int64_t n, pcgc_stream_t stream);
/* Reverse of the block tail out = relu(x + concat(t12, t23)) in one pass (model_voxception.py:65-67 differentiated):
 *   dpre = dout * (out > 0)            (premasked != 0: dout already carries that mask, dpre is not written and may be NULL)
 *   dz12 = dpre[:, :C/2] * (t12 > 0),  dz23 = dpre[:, C/2:] * (t23 > 0)     — the gradients w.r.t. the pre-activation
 * outputs of conv1_2 / conv2_3, what pcgc_conv3d_bwd_* take.  C = channels of out. */
int pcgc_vrn_bwd_split(const float* dout, const float* out, const float* t12, const float* t23, float* dpre, float* dz12,
                       float* dz23, int64_t nvox, int C, int premasked, pcgc_stream_t stream);
/* t23 == NULL: t12 is the concatenated [nvox, C] tensor (`pre` of pcgc_vrn_fwd_train). */
/* The same pair for blocks whose forward kept only the SIGNS of the pre-residual output: pre_signs int32 [B,D,D,D],
 * bit c = (pre[c] > 0) — all the reverse pass reads of `pre` (2 B of information per voxel instead of 64 B written and
 * read back).  For C = 16 the word also carries the other ReLU masks of the block's reverse: bits 16-19 = (t22 > 0),
 * 20-23 = (t11 > 0), 24-27 = (t21 > 0); pcgc_vrn_bwd_tail_split takes them from there and does not read t11 / t21 / t22
 * (their pointers must still be valid tensors; PCGC_DEBUG_SIGNS=1 makes the entry points compare the bits with the tensors first
 * and refuse words that do not carry the masks).  pcgc_vrn_fwd_train_signs: as pcgc_vrn_fwd_train, where pcgc_vrn_fwd_train_signs_supported(D, C) != 0
 * (D = 64 with C = 16, D = 32 with C = 32); pcgc_vrn_bwd_split_signs: as pcgc_vrn_bwd_split with the masks (t12 > 0), (t23 > 0) taken from
 * the bits (C <= 32). */
int pcgc_vrn_fwd_train_signs_supported(int D, int C);
int pcgc_vrn_fwd_train_signs(const float* x, const float* const* params, float* t11, float* t21, float* t22,
                             int32_t* pre_signs, float* out, int B, int D, int C, pcgc_stream_t stream);
int pcgc_vrn_bwd_split_signs(const float* dout, const float* out, const int32_t* pre_signs, float* dpre, float* dz12,
                             float* dz23, int64_t nvox, int C, int premasked, pcgc_stream_t stream);
/* The Q4 layout of the training step's 64^3 stage.  The C = 16 blocks' row kernels read a voxel row as one VGPR per
 * channel; on NDHWC tensors that is 16 B per lane at a 64 B stride, on Q4 [b][d][h][C/4][w][4] one contiguous KiB per
 * wave instruction (the inference path's layout).  With Trainer(q4=True) the 16-channel tensors of the stage (conv_in's
 * output ... down_1's input, up_2's output ... deconv_out's input, and their gradients) and the blocks' 8-channel
 * gradients are Q4; 4-channel tensors are the same in both layouts.  pcgc_vrn_fwd_train_q4 / pcgc_vrn_bwd_tail_split_q4 /
 * pcgc_vrn_bwd_input_q4 are the _signs / _split / _input entry points on such tensors; the boundary layers and the
 * weight gradients learn the layout per layer through pcgc_train_plan_set_layout.  pcgc_layout_q4 converts
 * (tests, tools): to_q4 = 1 NDHWC -> Q4, 0 back; C a multiple of 4. */
int pcgc_vrn_fwd_train_q4(const float* x, const float* const* params, float* t11, float* t21, float* t22,
                          int32_t* pre_signs, float* out, int B, int D, int C, pcgc_stream_t stream);
int pcgc_layout_q4(const float* src, float* dst, int B, int D, int C, int to_q4, pcgc_stream_t stream);

/* Reverse of the block's inner convolutions in one pass (model_voxception.py:59-60, 62-64 differentiated), from the
 * outputs of pcgc_vrn_bwd_split(_signs):
 *   dt11 = [t11 > 0] * conv1_2^T(dz12),   dt22 = [t22 > 0] * conv2_3^T(dz23),   dt21 = [t21 > 0] * conv2_2^T(dt22)
 * dz12 / dz23 [B,D,D,D,C/2]; t11 / t21 / t22 (the forward's saved ReLU outputs) and dt11 / dt21 / dt22 [B,D,D,D,C/4];
 * kernel12 [3,3,3,C/4,C/2], kernel22 [3,3,3,C/4,C/4], kernel23 [1,1,1,C/4,C/2] in the TensorFlow layouts.  dt11 / dt21
 * feed pcgc_vrn_bwd_input; the three layers' dW stay with pcgc_train_conv_bwd_weight (x = t11 / t21 / t22,
 * dz = dz12 / dt22 / dz23).  Only where pcgc_vrn_bwd_tail_supported(D, C) != 0 (D = 64 with C = 16). */
int pcgc_vrn_bwd_tail_supported(int D, int C);
int pcgc_vrn_bwd_tail(const float* dz12, const float* dz23, const float* t11, const float* t21, const float* t22,
                      const float* kernel12, const float* kernel22, const float* kernel23, float* dt11, float* dt21,
                      float* dt22, int B, int D, int C, pcgc_stream_t stream);

/* pcgc_vrn_bwd_split_signs (premasked form) and pcgc_vrn_bwd_tail in ONE pass: dout [B,D,D,D,C] already carries the mask
 * (out > 0); dz12 / dz23 are made from it and the sign bits for every row the kernel touches and written once (the
 * weight gradients of conv1_2 / conv2_3 read them), dt11 / dt21 / dt22 as above.  pcgc_vrn_bwd_tail_split_supported:
 * D = 64 with C = 16 (other blocks: pcgc_vrn_bwd_split_signs, then pcgc_vrn_bwd_tail). */
int pcgc_vrn_bwd_tail_split_supported(int D, int C);
int pcgc_vrn_bwd_tail_split(const float* dout, const int32_t* pre_signs, const float* t11, const float* t21,
                            const float* t22, const float* kernel12, const float* kernel22, const float* kernel23,
                            float* dz12, float* dz23, float* dt11, float* dt21, float* dt22, int B, int D, int C,
                            pcgc_stream_t stream);
int pcgc_vrn_bwd_tail_split_q4(const float* dout, const int32_t* pre_signs, const float* t11, const float* t21,
                               const float* t22, const float* kernel12, const float* kernel22, const float* kernel23,
                               float* dz12, float* dz23, float* dt11, float* dt21, float* dt22, int B, int D, int C,
                               pcgc_stream_t stream);                       /* dout, dz12, dz23 Q4 */

/* Reverse of the block head in one pass: the three contributions to the gradient of the block input
 * (x feeds conv1_1, conv2_1 and the skip connection, model_voxception.py:57-58, 61, 65-67):
 *   dx = [x > 0] * ( dpre + conv1_1^T(dt11) + conv2_1^T(dt21) )
 * dt11 / dt21 [B,D,D,D,C/4]: gradients w.r.t. the pre-activation outputs of conv1_1 / conv2_1; dpre / dx [B,D,D,D,C]
 * (dx may alias dpre); x_mask = the block input where it is a ReLU output, NULL for no mask; kernel11 / kernel21 in the
 * TensorFlow layouts [3,3,3,C,C/4] / [1,1,1,C,C/4].  The dW of the two layers stay with pcgc_train_conv_bwd_weight.
 * Only where pcgc_vrn_bwd_input_supported(D, C) != 0 (D = 64 with C = 16). */
int pcgc_vrn_bwd_input_supported(int D, int C);
int pcgc_vrn_bwd_input(const float* dt11, const float* dt21, const float* dpre, const float* x_mask, const float* kernel11,
                       const float* kernel21, float* dx, int B, int D, int C, pcgc_stream_t stream);
int pcgc_vrn_bwd_input_q4(const float* dt11, const float* dt21, const float* dpre, const float* x_mask, const float* kernel11,
                          const float* kernel21, float* dx, int B, int D, int C, pcgc_stream_t stream);   /* dpre / x_mask / dx Q4 */

/* Forward of one _VoxceptionResNet block for the training step (train_hyper.py:184-196 runs the same
 * model_voxception.py:56-68 call under the tape): pcgc_vrn_fwd's row kernels on NDHWC tensors, keeping what the reverse
 * pass reads — t11 = relu(conv1_1(x)), t21 = relu(conv2_1(x)), t22 = relu(conv2_2(t21)) [B,D,D,D,C/4] each, and
 * pre = concat[relu(conv1_2(t11)), relu(conv2_3(t22))] [B,D,D,D,C] (out = relu(x + pre)).  Only where
 * pcgc_vrn_fwd_train_supported(D, C) != 0 (D = 64 with C = 16, D = 32 with C = 32); other blocks run layer by layer. */
int pcgc_vrn_fwd_train_supported(int D, int C);
int pcgc_vrn_fwd_train(const float* x, const float* const* params, float* t11, float* t21, float* t22, float* pre,
                       float* out, int B, int D, int C, pcgc_stream_t stream);

/* ---- training plan: the step's per-layer housekeeping batched (csrc/train_plan.hip) ----
 * One entry per Conv3D / Conv3DTranspose of the trained sub-models (train_hyper.py:202-214: the variables the tape
 * differentiates and the optimiser updates).  kernel / dkernel / dbias point into the caller's parameter and gradient
 * buffers and must stay valid and in place for the plan's lifetime; dbias NULL = layer without bias. */
typedef struct pcgc_train_layer {
  const float* kernel;
  float* dkernel;
  float* dbias;
  int Cin, Cout, ksize, stride, transposed;
} pcgc_train_layer;
typedef struct pcgc_train_plan pcgc_train_plan;
int pcgc_train_plan_create(const pcgc_train_layer* layers, int n_layers, pcgc_train_plan** out);
void pcgc_train_plan_destroy(pcgc_train_plan* plan);
int pcgc_train_plan_layers(const pcgc_train_plan* plan);
/* Once per step, after the optimiser update and before the forward pass: packs / flips every filter from the current
 * parameter values (two launches for all layers) and resets the pool of weight-gradient partial sums. */
int pcgc_train_plan_prepare(pcgc_train_plan* plan, pcgc_stream_t stream);
/* pcgc_conv3d_fwd / pcgc_conv3d_bwd_data_fused / pcgc_conv3d_bwd_weight of layer `layer` on the prepared filters;
 * D = spatial size of the layer's input.  Results are bit-identical to those entry points.  bwd_weight only produces
 * partial sums: dkernel / dbias of every layer are written by pcgc_train_plan_finish_weights (one launch per 56
 * pending reductions), to be called once after the last bwd_weight of the step. */
/* x_q4 / y_q4 != 0: this layer's input / output tensor (and the gradients laid out like them) are Q4 in every later
 * call on the layer (see pcgc_vrn_fwd_train_q4); shapes without a kernel for that fail with an error, never silently. */
int pcgc_train_plan_set_layout(pcgc_train_plan* plan, int layer, int x_q4, int y_q4);
int pcgc_train_conv_fwd(const pcgc_train_plan* plan, int layer, const float* x, const float* bias, float* y, int B, int D,
                        int relu, pcgc_stream_t stream);
int pcgc_train_conv_bwd_data(const pcgc_train_plan* plan, int layer, const float* dz, float* dx, const float* relu_mask,
                             const float* add_to, int B, int D, pcgc_stream_t stream);
int pcgc_train_conv_bwd_weight(pcgc_train_plan* plan, int layer, const float* x, const float* dz, int B, int D,
                               pcgc_stream_t stream);
/* Two independent stride-1 layers of a VRN block (model_voxception.py:56-66) in ONE launch where a pair kernel exists — the
 * 16^3 blocks of a training batch, whose layers alone are 512 workgroups of 11-37 us: conv1_1 | conv2_1 (xa == xb),
 * conv1_2 | conv2_2, and in reverse conv1_2^T | conv2_3^T (dx_i = (relu_mask_i > 0) * conv_i^T(dz_i), nothing accumulated).
 * Every other pair of shapes runs as exactly the two single calls above; results are bit-identical either way. */
int pcgc_train_conv_fwd_pair(const pcgc_train_plan* plan, int layer_a, int layer_b, const float* xa, const float* xb,
                             const float* bias_a, const float* bias_b, float* ya, float* yb, int B, int D, int relu_a, int relu_b,
                             pcgc_stream_t stream);
int pcgc_train_conv_bwd_data_pair(const pcgc_train_plan* plan, int layer_a, int layer_b, const float* dz_a, const float* dz_b,
                                  float* dx_a, float* dx_b, const float* relu_mask_a, const float* relu_mask_b, int B, int D,
                                  pcgc_stream_t stream);
/* Two layers in one launch where the second needs from the first only what the same workgroup wrote (16^3 blocks again):
 *  _bwd_data_chain: the reverse of the block's two input layers (layer_b 1x1x1), dx = m * (m * (dx + conv_a^T(dz_a)) + conv_b^T(dz_b))
 *                   in place on dx (the gradient the skip connection brought), m = (relu_mask > 0), or 1 when relu_mask is NULL;
 *  _fwd_merge:      conv2_3 (1x1x1, y = tensor2_3) and the block's merge out = relu(blk_x + concat(t12, y)) (pcgc_vrn_merge).
 * Other shapes run as the single calls; bit-identical either way (the same sums in the same order). */
int pcgc_train_conv_bwd_data_chain(const pcgc_train_plan* plan, int layer_a, int layer_b, const float* dz_a, const float* dz_b,
                                   float* dx, const float* relu_mask, int B, int D, pcgc_stream_t stream);
int pcgc_train_conv_fwd_merge(const pcgc_train_plan* plan, int layer, const float* x, const float* bias, float* y, int relu,
                              const float* blk_x, const float* t12, float* out, int C, int B, int D, pcgc_stream_t stream);
/* conv1_1 (3x3x3) and conv2_1 (1x1x1) of a VRN block (model_voxception.py:56-62) read the same tensor: both layers'
 * partial sums in one pass over x where the fused kernel exists (16 | Cin, Cout 4 or 8), else exactly the two calls above.
 * The 3x3x3 layer's sums are those of pcgc_train_conv_bwd_weight bit for bit; the 1x1x1 layer's are added in another
 * (fixed) order. */
int pcgc_train_conv_bwd_weight_pair(pcgc_train_plan* plan, int layer3, int layer1, const float* x, const float* dz3,
                                    const float* dz1, int B, int D, pcgc_stream_t stream);
int pcgc_train_plan_finish_weights(pcgc_train_plan* plan, pcgc_stream_t stream);
/* on != 0 (off by default): the weight gradients of the stride-1 layers at D <= 16 — the 16^3 stage and the hyperprior nets
 * of the train_hyper step (train_hyper.py:200-207), ~35 launches of 128-512 workgroups per step — are recorded by
 * pcgc_train_conv_bwd_weight and launched by pcgc_train_plan_finish_weights, equal shapes as the jobs of ONE launch.  The
 * caller must then keep x and dz of such calls alive and unchanged until pcgc_train_plan_finish_weights.  Same kernels and
 * sums per layer: the gradients are bit-identical.  Switch it between steps only. */
int pcgc_train_plan_defer_small(pcgc_train_plan* plan, int on);
/* dscale == NULL: out = max(|s_raw|, lower_bound) (model_voxception.py:308 + train_hyper.py:189);
 * else out = dscale * sign(s_raw) * (|s_raw| >= lower_bound)  (its gradient, TF conventions). */
int pcgc_abs_max(const float* s_raw, float lower_bound, const float* dscale, float* out, int64_t n,
                 pcgc_stream_t stream);
/* Gradients of coef * sum(log(max(likelihood, bound))) w.r.t. the (noisy) values, loc and scale
 * (conditional_entropy_model.py:34-56 differentiated; sign() has zero gradient). */
int pcgc_laplace_likelihood_bwd(const float* values, const float* loc, const float* scale, float coef,
                                float likelihood_bound, float* dvalues, float* dloc, float* dscale, int64_t n,
                                pcgc_stream_t stream);
/* Same for the factorized prior (entropy_model.py:72-98, 114-151): dvalues [n] and dparams in the packed
 * tensor order of pcgc_factorized_likelihood.  C must divide 256. */
size_t pcgc_factorized_bwd_workspace_bytes(int C);
int pcgc_factorized_likelihood_bwd(const float* values, const float* params, float coef, float likelihood_bound,
                                   float* dvalues, float* dparams, int64_t n, int C, void* workspace,
                                   size_t workspace_bytes, pcgc_stream_t stream);
/* d/dpred of w0 * mean_{label=0}(-log(1-o)) + w1 * mean_{label>0}(-log o) (loss.py:8-33); pass w0/n0, w1/n1. */
int pcgc_bce_bwd(const float* pred, const float* label, float w0_over_n0, float w1_over_n1, float* dpred,
                 int64_t n, pcgc_stream_t stream);
/* The three reverse kernels above with their coefficients formed ON THE DEVICE from counts that are still there: the
 * reference divides the loss terms by the number of occupied / empty voxels (train_hyper.py:193-199), which the step only
 * knows after its forward pass — a host that reads them back stalls the stream in the middle of the step.
 *   pcgc_bce_bwd_dev:            w0/n0 = (float)(a0 / sums4[1]), w1/n1 = (float)(a1 / sums4[3]), sums4 = pcgc_bce_sums' output;
 *   pcgc_*_likelihood_bwd_dev:   coef = (float)(num / (mul * *count))   (e.g. num = delta, mul = -ln 2, count = &sums4[3]).
 * The same double-precision expressions the host forms: identical gradients. */
int pcgc_bce_bwd_dev(const float* pred, const float* label, const double* sums4, double a0, double a1, float* dpred,
                     int64_t n, pcgc_stream_t stream);
int pcgc_laplace_likelihood_bwd_dev(const float* values, const float* loc, const float* scale, double num, double mul,
                                    const double* count, float likelihood_bound, float* dvalues, float* dloc,
                                    float* dscale, int64_t n, pcgc_stream_t stream);
int pcgc_factorized_likelihood_bwd_dev(const float* values, const float* params, double num, double mul,
                                       const double* count, float likelihood_bound, float* dvalues, float* dparams,
                                       int64_t n, int C, void* workspace, size_t workspace_bytes, pcgc_stream_t stream);
/* out = sum(log(p)) in double, fixed order (train_hyper.py:194-196). */
size_t pcgc_sum_log_workspace_bytes(void);
int pcgc_sum_log(const float* p, int64_t n, double* out, void* workspace, size_t workspace_bytes,
                 pcgc_stream_t stream);
/* The training step's three loss reductions in two launches (train_hyper.py:193-199: the BCE sums of the logits against
 * the occupancy, sum log(likelihood) of y and of z): sums4 as pcgc_bce_sums, logs2 = {pcgc_sum_log(lik_y), pcgc_sum_log(lik_z)},
 * bit-identical to the three separate calls (the same blocks run the same sums). */
size_t pcgc_train_loss_sums_workspace_bytes(int64_t n);
int pcgc_train_loss_sums(const float* pred, const float* label, int64_t n, const float* lik_y, int64_t n_y,
                         const float* lik_z, int64_t n_z, double* sums4, double* logs2, void* workspace,
                         size_t workspace_bytes, pcgc_stream_t stream);
/* tf.train.AdamOptimizer update (TF1 form, train_hyper.py:104, 209-214); lr_t = lr*sqrt(1-b2^t)/(1-b1^t). */
int pcgc_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1,
                   float beta2, float epsilon, pcgc_stream_t stream);
/* The same update, skipped ON THE DEVICE when the step's BCE sums (sums4 of pcgc_train_loss_sums, still on the device) say the
 * batch had no empty or no occupied voxel (sums4[1] == 0 or sums4[3] == 0: loss.py:8-33 divides by both counts, the gradients
 * are inf / NaN).  Lets the host queue the update before it reads the loss terms back instead of idling the GPU for the
 * read-back; the host still raises on such a batch, with the parameters untouched. */
int pcgc_adam_step_guarded(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1,
                           float beta2, float epsilon, const double* bce_sums4, pcgc_stream_t stream);

/* ------------------------------------------------------------------ */
/* libpcgc_host.so — sequential host tail (no HIP dependency)          */
/* ------------------------------------------------------------------ */
const char* pcgc_host_last_error(void);

/* coder_ops.pmf_to_quantized_cdf (entropy_model.py:218). pmf [rows,n] -> cdf int32 [rows,n+1]. */
int pcgc_pmf_to_quantized_cdf(const float* pmf, int64_t rows, int n, int precision, int32_t* cdf);
/* coder_ops.range_encode (entropy_model.py:258; conditional_entropy_model.py:161).
 * data int16 [rows, cols]; cdf int32 [(rows|1)*cols, n+1] (broadcast_rows=1 for the
 * [1,C,N+1] table).  Writes up to cap bytes, returns the length in *out_len (if it
 * exceeds cap the function returns -2 and the caller retries with a bigger buffer). */
int pcgc_range_encode(const int16_t* data, int64_t rows, int cols, const int32_t* cdf, int n,
                      int broadcast_rows, int precision, uint8_t* out, int64_t cap, int64_t* out_len);
/* The same stream from the rounded VALUES: symbol = data[i] - offset (data int8 or int16, elem_bytes 1 or 2) — the z string
 * of entropy_model.py:249-259 without a pass over the batch to subtract min_v first (millions of symbols for a large cloud,
 * on the one thread everything else then waits for). */
int pcgc_range_encode_values(const void* data, int elem_bytes, int64_t rows, int cols, int offset, const int32_t* cdf, int n,
                             int broadcast_rows, int precision, uint8_t* out, int64_t cap, int64_t* out_len);
/* coder_ops.range_decode (entropy_model.py:298; conditional_entropy_model.py:195). */
int pcgc_range_decode(const uint8_t* str, int64_t len, int64_t rows, int cols, const int32_t* cdf,
                      int n, int broadcast_rows, int precision, int16_t* out);
/* Same, publishing the number of completed rows in *progress (release stores, every 1024 rows and at the end; -1 on a
 * corrupt stream): the single z stream of a file is sequential (entropy_model.py:249-259), so its consumers start on
 * the first cubes' symbols while a helper thread is still decoding the rest. */
int pcgc_range_decode_progress(const uint8_t* str, int64_t len, int64_t rows, int cols, const int32_t* cdf,
                               int n, int broadcast_rows, int precision, int16_t* out, int64_t* progress);

/* Batched forms used by compress_hyper / decompress_hyper: n_streams independent
 * cubes coded on n_threads host threads (each stream stays sequential).
 * Encode consumes the device-produced lohi words (pcgc_laplace_cdf), sym_per_stream
 * per cube; out is n_streams slots of cap_per_stream bytes; out_lens[n_streams]. */
int pcgc_range_encode_lohi_batch(const uint32_t* lohi, int n_streams, int64_t sym_per_stream,
                                 int precision, uint8_t* out, int64_t cap_per_stream,
                                 int64_t* out_lens, int n_threads);
/* Decode consumes the device-produced uint16 lower-CDF rows [n_streams*sym_per_stream, ncols];
 * n_sym[s] = number of valid symbols (max-min+1) of stream s. Output int16 symbols. */
int pcgc_range_decode_u16_batch(const uint8_t* strings, const int64_t* offsets, const int64_t* lens,
                                int n_streams, int64_t sym_per_stream, const uint16_t* cdf_lower,
                                int ncols, const int32_t* n_sym, int precision, int16_t* out,
                                int n_threads);

/* load_points partition (dataprocess/inout_points.py:50-90) on an in-memory cloud.
 * points int32 [n,3].  Two-call protocol: first call with outputs NULL returns the
 * number of surviving cubes in *n_cubes; second call fills
 *   cube_positions int64 [n_cubes,3]  (first-appearance order, as the reference returns it)
 *   sorted_positions int64 [n_cubes,3] (key order = order of the cubes)
 *   cube_of_point int32 [n] (index into the SORTED cube list, -1 if its cube was dropped)
 *   points_numbers (unique voxels per cube, uint16 wrap like process.py:45). */
int pcgc_partition(const int32_t* points, int64_t n, int cube_size, int min_num, int64_t* n_cubes,
                   int64_t* cube_positions, int64_t* sorted_positions, int32_t* cube_of_point);

/* load_ply_data (dataprocess/inout_points.py:8-28) on a whole file image: every line whose first three
 * single-space-separated tokens read as Python floats is a point (header lines do not and are skipped, as are lines
 * with fewer than three tokens, where the reference raises), truncated to int32 like ndarray.astype.  out holds
 * cap x 3 values (cap >= number of lines is always enough); *n_points receives the count. */
int pcgc_parse_ply_points(const char* text, int64_t len, int32_t* out, int64_t cap, int64_t* n_points, int n_threads);

/* Body of write_ply_data (dataprocess/inout_points.py:43-44) for integer coordinates: "x y z\n" per point, digits
 * as Python's str(int).  *out_len always receives the exact text length (never more than 63 bytes per point); with a
 * smaller cap (or out == NULL) nothing is written and the call returns -2. */
int pcgc_format_points_int(const int64_t* pts, int64_t n, char* out, int64_t cap, int64_t* out_len);

/* CRC-32C (Castagnoli, reflected 0x82F63B78), the checksum of TensorFlow's tensor-bundle checkpoints
 * (tf.train.Checkpoint files restored at transform.py:107-112; written at train_hyper.py:255-268).
 * Returns the crc of `crc_in`-continued data (pass 0 to start); not masked. */
uint32_t pcgc_crc32c(uint32_t crc_in, const void* data, int64_t n);

/* Reproducible elementary functions behind every pmf that becomes a range-coder CDF (Laplace:
 * conditional_entropy_model.py:21-56, 95-124; factorized: entropy_model.py:72-98, 114-151, 183-221).
 * They are fixed sequences of IEEE-754 binary32 +, -, *, /, floor (no FMA), written out in
 * pcgcv1_amd/csrc/repro_math.h and restated in numpy by oracle/entropy.py, so the HIP kernels, this host
 * library and the oracle produce identical bits; a stream written on one ROCm version decodes on another.
 *   fn: 0 exp, 1 log (normal x > 0), 2 tanh, 3 sigmoid, 4 softplus.   y[i] = fn(x[i]).
 * pcgc_host_repro_eval: libpcgc_host.so (CPU);  pcgc_repro_eval: libpcgc_hip.so (device pointers). */
int pcgc_host_repro_eval(int fn, const float* x, float* y, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* PCGC_H_ */
