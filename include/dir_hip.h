/* dir_hip.h — C-ABI of libdir_hip.so: MI355X (gfx950) kernels for the LDS/FDS hot path of
 * YyzHarry/imbalanced-regression (imdb-wiki-dir / agedb-dir).
 *
 * The reference has no FFI (it is 100 % Python); its boundary is the Python module
 * surface fds.py / loss.py / utils.py / datasets.py. Each entry point below names the reference
 * lines whose device work it replaces. Conventions:
 *   - extern "C", plain pointers and sizes; no torch types.
 *   - every pointer is a DEVICE pointer owned by the caller unless the comment says "host";
 *     nothing is allocated inside; scratch comes from a caller-provided workspace.
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no host synchronisation inside; re-entrant (no global state).
 *   - return 0 on success, DIR_E* (< 0) for argument errors, hipError_t (> 0) for runtime errors.
 *   - row-major contiguous arrays; "f32" = IEEE float32, "f64" = IEEE float64.
 *   - kernels are compiled with -ffp-contract=off: every float32 expression below is evaluated
 *     exactly as written (one rounding per operation), like the reference's eager torch ops.
 */
#ifndef DIR_HIP_H
#define DIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIR_ABI_VERSION 4

#define DIR_OK            0
#define DIR_EINVAL       (-1)   /* bad argument (null pointer, non-positive size, ks even, ...) */
#define DIR_EUNSUPPORTED (-2)   /* valid request this build does not implement (dtype, mode)    */
#define DIR_EWORKSPACE   (-3)   /* workspace too small                                          */

/* feature dtypes */
#define DIR_F32  0
#define DIR_BF16 1

/* label-scan flag bits written by dir_fds_label_flags */
#define DIR_FLAG_HAS_LO      1u  /* some label == bucket_start        (fds.py:94, :123)          */
#define DIR_FLAG_HAS_HI      2u  /* some label == bucket_num - 1      (fds.py:96, :130)          */
#define DIR_FLAG_NONINTEGER  4u  /* an in-range label with a fractional part        */
#define DIR_FLAG_NAN         8u  /* a NaN label                                                   */

typedef void* dir_stream_t;      /* hipStream_t */

int         dir_abi_version(void);
const char* dir_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * K1  label -> table row ("bin").  Replaces the host loop `for label in torch.unique(labels)` and
 * its per-label branch (imdb-wiki-dir/fds.py:91-99 and :120-137):
 *   l > num-1  -> rows used only if some label == num-1 in this call (then bin = num-1-start)
 *   l < start  -> rows used only if some label == start in this call (then bin = 0)
 *   otherwise  -> bin = (int)(l - start)      (float32 subtraction, truncation; fds.py:104)
 *   bin = -1   -> row untouched / not counted.
 * The presence flags are a separate step so that a data-parallel caller can OR-reduce them across
 * ranks between the two calls.  flags: one uint32 on the device, OR-ed into (the
 * caller zeroes it, or calls dir_fds_bin_index which does).
 */
int dir_fds_label_flags(const float* labels, int n, int bucket_start, int bucket_num,
                        uint32_t* flags, dir_stream_t stream);
int dir_fds_assign_bins(const float* labels, int n, int bucket_start, int bucket_num,
                        const uint32_t* flags, int32_t* bins, dir_stream_t stream);
/* both steps, single rank; zeroes *flags first */
int dir_fds_bin_index(const float* labels, int n, int bucket_start, int bucket_num,
                      int32_t* bins, uint32_t* flags, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  per-bin sufficient statistics of the feature matrix.  Replaces, for every label, the boolean
 * mask + gather copy + torch.mean + torch.var of imdb-wiki-dir/fds.py:95-102.
 *   feats [n, C] (dtype), bins [n] int32 from K1 (-1 rows are skipped).
 *   out: count [nb] f64 (exact integers), mean [nb, C] f64, m2 [nb, C] f64 where
 *        m2 = sum_rows (x - mean)^2   (so var_unbiased = m2 / (count - 1)); bins with no rows get
 *        count = 0, mean = 0, m2 = 0.
 * One streaming pass over feats: rows are grouped by bin with a stable counting sort, every
 * (bin, <=rows_per_piece rows) piece is reduced in registers in float64 around a per-bin shift
 * (the bin's first row, so a constant column has m2 == 0 exactly, like torch.var — SURVEY A.7/A.9),
 * pieces are combined in a fixed order: results are bit-reproducible run to run.
 * (count, mean, m2) triples from several ranks merge with Chan's formula.
 * workspace: >= dir_fds_scatter_stats_workspace(n, C, nb) bytes, 256-byte aligned.
 */
size_t dir_fds_scatter_stats_workspace(int n, int C, int nb);
int dir_fds_scatter_stats(const void* feats, int dtype, const int32_t* bins, int n, int C, int nb,
                          double* count, double* mean, double* m2,
                          void* workspace, size_t workspace_bytes, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  momentum update of the running tables.  Replaces imdb-wiki-dir/fds.py:101-111 for all bins
 * at once.  For every bin with count > 0:
 *   curr_mean = (f32) mean;  curr_var = count == 1 ? 0 : (f32)(m2 / (count - 1))      (fds.py:101-102)
 *   tracked  += (f32) count                                                           (fds.py:104)
 *   factor    = mode 0: 0 (epoch == start_update, fds.py:107)
 *               mode 1: momentum                                                      (fds.py:105)
 *               mode 2: 1 - count / tracked   (momentum is None, fds.py:105-106; float64)
 *   running   = (f32)(1 - factor) * curr + (f32)factor * running                      (fds.py:108-111)
 * Bins with count == 0 keep their values.  In place on running_mean / running_var [nb, C] f32 and
 * num_samples_tracked [nb] f32.
 */
#define DIR_FACTOR_ZERO      0
#define DIR_FACTOR_MOMENTUM  1
#define DIR_FACTOR_COUNT     2
int dir_fds_finalize_update(const double* count, const double* mean, const double* m2, int nb, int C,
                            int factor_mode, double momentum,
                            float* running_mean, float* running_var, float* num_samples_tracked,
                            dir_stream_t stream);
/* K3 for NON-INTEGER labels (ABI 3).  imdb-wiki-dir/fds.py:91-111 loops over torch.unique(labels): every distinct label VALUE
 * inside [bucket_start, bucket_num - 1] is its own row group and triggers its own blend of bin int(value - bucket_start), in ascending value
 * order (the boundary values lump the out-of-range rows, fds.py:94-97).  count / mean / m2 hold the statistics per value GROUP
 * (dir_fds_scatter_stats with the group index as the "bin": [ngroups], [ngroups, C]; groups sorted by value), bin_ptr [nb + 1] (device) the
 * first group of every bin.  For every bin the blends of dir_fds_finalize_update are applied one after the other, num_samples_tracked
 * advancing group by group (it enters the count-based factor of mode 2). */
int dir_fds_finalize_update_groups(const double* count, const double* mean, const double* m2, int ngroups, int C,
                                   const int32_t* bin_ptr, int nb, int factor_mode, double momentum,
                                   float* running_mean, float* running_var, float* num_samples_tracked, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4  smoothing across the bin axis.  Replaces the two F.pad(reflect) + F.conv1d calls of
 * imdb-wiki-dir/fds.py:58-67:  out[b, c] = sum_k window[k] * in[reflect(b + k - ks/2), c],
 * float32, taps accumulated k = 0..ks-1, for the mean and the var table in one launch.
 * window [ks] f32 on the device; ks odd, ks/2 < nb.  Outputs are contiguous [nb, C].
 */
int dir_fds_smooth_bins(const float* mean, const float* var, const float* window, int ks, int nb, int C,
                        float* smoothed_mean, float* smoothed_var, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5a calibration multiplier table.  Replaces the per-call recomputation of
 * torch.sqrt(torch.clamp(v2 / v1, clip_min, clip_max)) and the two guards of
 * imdb-wiki-dir/utils.py:98-104, once per table change instead of once per label per step:
 *   scale[b, c] = -1                        if sum_c v1[b, :] < 1e-10      (utils.py:98-99)
 *               = -1                        if v1[b, c] == 0               (utils.py:100-104)
 *               = sqrtf(clamp(v2/v1))       otherwise (NaN propagates)     (utils.py:106-107)
 * -1 means "leave the element untouched" (a real multiplier is never negative).
 */
int dir_fds_prepare_scale(const float* v1, const float* v2, int nb, int C, float clip_min, float clip_max,
                          float* scale, dir_stream_t stream);

/* K5  calibration, forward, IN PLACE.  Replaces the per-label
 * `features[mask] = calibrate_mean_var(features[mask], m1[b], v1[b], m2[b], v2[b])` of
 * imdb-wiki-dir/fds.py:120-143 + utils.py:97-107 (11 076 aten dispatches / 919 host syncs at B=256):
 *   x[r, c] = scale[b, c] < 0 ? x[r, c] : (x[r, c] - m1[b, c]) * scale[b, c] + m2[b, c],  b = bins[r] >= 0.
 */
int dir_fds_calibrate_fwd(void* x_inout, int dtype, const int32_t* bins, int B, int C,
                          const float* m1, const float* scale, const float* m2, dir_stream_t stream);
/* K6  its autograd:  dx[r, c] = (bins[r] < 0 || scale < 0) ? dy : dy * scale.  dx may alias dy. */
int dir_fds_calibrate_bwd(const void* dy, void* dx, int dtype, const int32_t* bins, int B, int C,
                          const float* scale, dir_stream_t stream);
/* K5 with the bin-statistic tables STAGED IN LDS (large batches, B >= 4096; imdb-wiki-dir/fds.py:120-143): a workgroup keeps a
 * 128-column tile of all nb rows of (m1, scale, m2) resident (150 KB at nb = 100) and streams feature rows in their natural order,
 * four rows in flight per lane — no table re-reads through the L2, no sort.  Same arithmetic, bit-identical to dir_fds_calibrate_fwd.
 * nb = rows of the tables; bins in [0, nb) or < 0 (row untouched).  Falls back to dir_fds_calibrate_fwd when the layout does not
 * allow 16-byte accesses or nb x 16 columns x 12 B exceed a CU's LDS. */
int dir_fds_calibrate_fwd_lds(void* x_inout, int dtype, const int32_t* bins, long long B, int C, int nb,
                              const float* m1, const float* scale, const float* m2, dir_stream_t stream);
/* K5 / K6 of the NYUD2-DIR dense variant on the network's own NCHW map (nyud2-dir/models/fds.py:128-149 permutes the map to
 * [B*H*W, C] and back around the calibration: two copies of 284 MB).  x, y: [N, C, HW] float32, HW % 4 == 0, 16-byte aligned;
 * bins [N * HW] int32 per pixel (< 0: copied unchanged); y may alias x.  nb <= 128: short-lived workgroups in address order, each one
 * contiguous 16 KB chunk of ONE channel plane with that channel's column of the tables (3 x nb floats) in LDS; larger nb: persistent
 * workgroups per (4096-pixel block, channel group) with the group's table slices transposed in LDS ([C / groups][nb]).
 * DIR_EUNSUPPORTED when neither fits or the layout does not allow 16-byte accesses. */
int dir_fds_calibrate_fwd_nchw(const void* x, void* y, int dtype, const int32_t* bins, long long N, int C, int HW, int nb,
                               const float* m1, const float* scale, const float* m2, dir_stream_t stream);
int dir_fds_calibrate_bwd_nchw(const void* dy, void* dx, int dtype, const int32_t* bins, long long N, int C, int HW, int nb,
                               const float* scale, dir_stream_t stream);
/* K1+K5 in one launch for a training batch (FDS.smooth, fds.py:115-144): labels [B] f32;
 * bins_out [B + 1] int32: the first B entries receive the bins for the backward, the last one is
 * scratch.  Any B (falls back to K1 then K5 above a batch-size threshold). */
int dir_fds_smooth_fwd(void* x_inout, int dtype, const float* labels, int B, int C,
                       int bucket_start, int bucket_num,
                       const float* m1, const float* scale, const float* m2,
                       int32_t* bins_out, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K7  weighted regression losses, forward + gradient in one launch.  Replaces
 * imdb-wiki-dir/loss.py:5-48 (weighted_{mse,l1,focal_mse,focal_l1,huber}_loss) and their autograd.
 *   x, y, w [n] f32 (w may be NULL = unweighted);  loss[0] = mean_i(per_i * w_i)  f32;
 *   dx_unit [n] = d loss / d x_i for an upstream gradient of 1 (may be NULL).
 *   beta:  focal: loss.py default .2 ; huber: loss.py default 1.   gamma: focal exponent.
 *   activate: 0 = sigmoid (default), 1 = tanh.
 * workspace: >= dir_weighted_loss_workspace(n) bytes (may be NULL when that is 0).
 */
#define DIR_LOSS_MSE        0
#define DIR_LOSS_L1         1
#define DIR_LOSS_FOCAL_MSE  2
#define DIR_LOSS_FOCAL_L1   3
#define DIR_LOSS_HUBER      4
size_t dir_weighted_loss_workspace(int n);
int dir_weighted_loss(int kind, const float* x, const float* y, const float* w, int n,
                      float beta, float gamma, int activate,
                      float* loss, float* dx_unit, void* workspace, size_t workspace_bytes,
                      dir_stream_t stream);
/* out[i] = in[i] * scalar[0]  (scalar on the device: the upstream gradient of the loss) */
int dir_scale_by_device_scalar(const float* in, const float* scalar, float* out, int n, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K8  LDS per-sample weights — HOST function, all pointers are host pointers.  Replaces
 * imdb-wiki-dir/datasets.py:55-83 (_prepare_weights), bit for bit:
 *   bin = min(max_target-1, (int)label); counts -> sqrt (f64) | clip(5,1000) (int64);
 *   if lds: scipy.ndimage.convolve1d(mode='constant') with `window` [ks] f64
 *           (symmetric-kernel accumulation order, integer counts stay integer — SURVEY A.5/E.1);
 *   w_i = (f32)(1 / value[bin_i]); scaling = (f32)n / pairwise_f32_sum(w) (numpy order, E.2);
 *   weights[i] = scaling * w_i.
 * reweight: 1 = sqrt_inv, 2 = inverse.  window may be NULL when lds == 0.
 */
#define DIR_REWEIGHT_SQRT_INV 1
#define DIR_REWEIGHT_INVERSE  2
int dir_lds_weights(const double* labels, int64_t n, int max_target, int reweight, int lds,
                    const double* window, int ks, float* weights);

/* ---------------------------------------------------------------------------------------------
 * K9a  BatchNorm2d (+ residual add) (+ ReLU), NHWC activations, training and eval forward, backward.
 * Replaces, for each of the 53 BN layers of imdb-wiki-dir/resnet.py (Bottleneck.forward :57-70, stem :128-130,
 * downsample :114-117), the eager chain  bn -> [+= residual] -> relu  and its autograd
 * (MIOpen BatchNorm fwd/bwd + 3 element-wise torch kernels: 58 % of the round-1 step time).
 *   x, residual, y, dout, out, dx, dres: [M, C] with C contiguous (a channels_last NCHW tensor), dtype
 *   DIR_BF16 or DIR_F32, 16-byte aligned; M = N*H*W.  C % 8 == 0 (bf16) / % 4 (f32).
 *   gamma, beta, running_*, save_*, dgamma, dbeta: [C] f32.
 * train fwd:  mean/var over M (biased var for the normalisation; running_var gets the unbiased one, torch
 *             semantics; running_* may both be NULL), save_mean / save_rstd for the backward,
 *             y = [relu]((x - mean) * rstd * gamma + beta [+ residual]).
 * eval fwd :  same with the running statistics.
 * bwd      :  g = relu ? dout * (out > 0) : dout;  dbeta = sum g;  dgamma = sum g * xhat;
 *             dx = gamma * rstd * (g - dbeta/M - xhat * dgamma/M);  dres (optional) = g.
 * workspace: >= dir_bn_workspace(dtype, M, C) bytes (0 = unsupported shape), 256-byte aligned.
 */
size_t dir_bn_workspace(int dtype, int64_t M, int C);
int dir_bn_fwd_train(const void* x, const void* residual, void* y, int dtype, int64_t M, int C,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     double momentum, double eps, int relu, float* save_mean, float* save_rstd,
                     void* workspace, size_t workspace_bytes, dir_stream_t stream);
/* same as dir_bn_fwd_train, but the per-channel (sum, sum of squares) partials [partial_rows][2][C] f32 were already
 * produced by dir_conv_fwd's epilogue: no statistics pass over x. */
int dir_bn_fwd_train_partials(const void* x, const void* residual, void* y, int dtype, int64_t M, int C,
                              const float* partial, int partial_rows,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              double momentum, double eps, int relu, float* save_mean, float* save_rstd,
                              void* workspace, size_t workspace_bytes, dir_stream_t stream);
int dir_bn_fwd_eval(const void* x, const void* residual, void* y, int dtype, int64_t M, int C,
                    const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                    double eps, int relu, void* workspace, size_t workspace_bytes, dir_stream_t stream);
/* `out` may be NULL for a ReLU layer WITHOUT residual: the mask is then recomputed from x with the forward's own
 * coefficients (needs beta; bit-identical decision, one tensor read less per pass). */
int dir_bn_bwd(const void* dout, const void* x, const void* out, void* dx, void* dres, int dtype,
               int64_t M, int C, const float* gamma, const float* beta, const float* save_mean,
               const float* save_rstd, float* dgamma, float* dbeta, int relu, void* workspace,
               size_t workspace_bytes, dir_stream_t stream);
/* dir_bn_fwd_train[_partials] (relu = 1; partial may be NULL) and dir_bn_apply (relu = 1) for bf16 tensors that ALSO emit the
 * ReLU's backward mask as one bit per element: relu_bits_out [M][C / 8] bytes, bit j of byte (m, g) = y[m][8 g + j] > 0.  The
 * block outputs relu(bn3(.) + shortcut) of imdb-wiki-dir/resnet.py:66-68 are masks of the next block's data gradient
 * (dir_conv_dgrad_ex): reading the bits instead of the tensor saves 15/16 of that read. */
int dir_bn_fwd_train_bits(const void* x, const void* residual, void* y, int64_t M, int C, const float* partial, int partial_rows,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, double momentum,
                          double eps, float* save_mean, float* save_rstd, void* relu_bits_out, void* workspace,
                          size_t workspace_bytes, dir_stream_t stream);
int dir_bn_apply_bits(const void* x, const void* residual, const float* residual_coef, void* y, int64_t M, int C,
                      const float* coef, void* relu_bits_out, dir_stream_t stream);
/* Backward of the projection-shortcut join relu(bn3(x) + bn_d(r)) (imdb-wiki-dir/resnet.py:63-68) on its gradient g with the
 * ReLU backward ALREADY applied (the consumer's data-gradient kernel masked it, dir_conv_dgrad_ex): both BatchNorm backwards in
 * one reduction pass + one apply pass — g is read twice instead of four times; bit-identical to two dir_bn_bwd(relu = 0) calls.
 * workspace: 2 x dir_bn_workspace(dtype, M, C) bytes. */
int dir_bn_bwd_join(const void* g, const void* x, const void* r, void* dx, void* dr, int dtype, int64_t M, int C,
                    const float* gamma, const float* save_mean, const float* save_rstd, const float* gamma_r,
                    const float* save_mean_r, const float* save_rstd_r, float* dgamma, float* dbeta, float* dgamma_r,
                    float* dbeta_r, void* workspace, size_t workspace_bytes, dir_stream_t stream);
/* dir_bn_bwd minus its first pass: the per-channel sums of g and g*x arrive as `partial` [partial_rows][2][C] f32 from the
 * data-gradient kernel that produced dout (dir_conv_dgrad_bnstats / dir_conv_dgrad_s2_bnstats), so dout and x are read once
 * (apply pass) instead of twice.  relu != 0: ReLU layer without residual, mask recomputed from x (as dir_bn_bwd with
 * out == NULL).  Replaces the BatchNorm2d backward of imdb-wiki-dir/resnet.py:45-50 (bn1/bn2/bn3 of a Bottleneck). */
int dir_bn_bwd_partials(const void* dout, const void* x, void* dx, int dtype, int64_t M, int C, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                        int relu, const float* partial, int partial_rows, void* workspace, size_t workspace_bytes,
                        dir_stream_t stream);
/* The two halves of dir_bn_fwd_train[_partials] on their own, for the projection-shortcut join
 * relu(bn3(x) + bn_d(r)) of imdb-wiki-dir/resnet.py:63-68 (bn_d = downsample[1]): both BatchNorms are prepared
 * (statistics -> save_mean/save_rstd, running statistics, coef[2][C] = scale/shift), then ONE apply pass normalises
 * x and r and adds them, so bn_d(r) is never written or re-read.
 *   dir_bn_prepare_train: partial == NULL -> statistics pass over x; else partial [partial_rows][2][C] from dir_conv_fwd.
 *   dir_bn_apply: y = [relu](x * coef[0] + coef[1] [+ residual | + residual * residual_coef[0] + residual_coef[1]]). */
int dir_bn_prepare_train(const void* x, int dtype, int64_t M, int C, const float* partial, int partial_rows,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, double momentum,
                         double eps, float* save_mean, float* save_rstd, float* coef, void* workspace,
                         size_t workspace_bytes, dir_stream_t stream);
int dir_bn_apply(const void* x, const void* residual, const float* residual_coef, void* y, int dtype, int64_t M, int C,
                 const float* coef, int relu, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K9  convolution as MFMA implicit GEMM, NHWC bf16, fp32 accumulation.  Replaces nn.Conv2d (bias=False) of
 * imdb-wiki-dir/resnet.py:44-49,79,114-115 for every layer with Cin % 64 == 0 and Cout % 64 == 0 (all but the stem).
 *   x [N, H, W, Cin] bf16;  w [Cout, R, S, Cin] bf16 (= a channels_last [Cout, Cin, R, S] tensor);
 *   y [N, Ho, Wo, Cout] bf16,  Ho = (H + 2 pad - R) / stride + 1.
 *   stats (optional, may be NULL): [dir_conv_stats_rows(N, Ho, Wo)][2][Cout] f32 — per 128-row tile, the sum and the
 *   sum of squares of the (bf16-rounded) outputs of each channel: the partial-sum buffer dir_bn_* consumes, so the
 *   BatchNorm that follows needs no statistics pass of its own.
 * The same entry point computes the data gradient of a stride-1 convolution when given dY and the 180-degree
 * rotated, in/out-transposed weights.
 */
size_t dir_conv_stats_rows(int N, int Ho, int Wo);
/* Kernel of a launch, chosen per launch (there is no process-wide switch): DIR_CONV_AUTO = the product heuristic — the 256 x 256
 * CU-tile kernel (one 1024-thread workgroup, 16 wavefronts with 64 x 64 wave tiles sharing one 64 KB K-step stage) for launches
 * with >= 16 K-steps and >= 150 tiles and at most one fused addend; the patch-staged kernel for 3x3 / stride-1 / pad-1 layers on
 * 56^2, 28^2, 14^2 maps (the input patch of a chunk of whole image rows is staged in LDS once per 64-channel block and all nine
 * taps read it there); 128-row tiles otherwise.  The other values force one kernel (tests, A/B measurements): the launch returns
 * DIR_EUNSUPPORTED when the geometry is not that kernel's. */
enum { DIR_CONV_AUTO = 0, DIR_CONV_TILE_REG = 1 /* 128-row tiles, register-staged K loop */, DIR_CONV_TILE_DMA = 2 /* 128-row tiles,
       LDS-DMA K loop */, DIR_CONV_PATCH3 = 3 /* patch-staged 3x3 */, DIR_CONV_BIG = 5 /* 256 x 256 CU tile: Cout % 256 == 0, M % 256 == 0 */ };
/* Rows of the `stats` list of ONE launch of this geometry through `variant`: the kernels tile M differently (128 output pixels per
 * row for the tile and CU-tile kernels, one row per chunk of whole image rows for the patch-staged kernel), and which kernel AUTO
 * picks also depends on whether the launch carries BOTH fused addends (two_addends).  The host sizes `stats` with this function and
 * passes the same number as `stats_rows` to the launch, which re-derives it from the kernel it is about to run and returns
 * DIR_EINVAL on a mismatch (a list sized for another tiling would be overrun or half-filled).  0 = invalid / not applicable. */
size_t dir_conv_plan_rows(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int two_addends, int variant);
/* float32 master weight [Cout][R][S][Cin] -> bf16 copy (same layout) and, if w16_rot != NULL, the data-gradient
 * weight [Cin][R][S][Cout] with the taps rotated by 180 degrees.  One launch per layer per optimizer step. */
int dir_conv_prep_weights(const float* w, int Cout, int R, int S, int Cin, void* w16, void* w16_rot,
                          dir_stream_t stream);
/* The same for all layers of a network in ONE launch (52 launches per optimizer step otherwise).  table: device
 * array [nlayers][7] of int64: { w (const float*), w16, w16_rot (0 = none), Cout, R*S, Cin, rot_mode }. */
int dir_conv_prep_weights_batched(const void* table, int nlayers, dir_stream_t stream);
/* torch.optim.Adam.step() (imdb-wiki-dir/train.py:161-162 builds the optimizer, :259-260 steps it) for ALL parameter tensors in one
 * launch, fused with the bf16 operand preparation of the convolution weights (the job of dir_conv_prep_weights_batched).
 * table: device [ntensors][12] int64 = (param, grad, exp_avg, exp_avg_sq, numel, w16, w16_rot, Cout, R*S, Cin, rot_mode, 0); rows with
 * w16 == 0 are plain tensors (BatchNorm affine, the linear layer), the others float32 channels_last conv weights whose bf16
 * operands are rewritten from the updated value.  step >= 1: the step count after this step (bias corrections on the host in
 * float64, like torch's non-capturable path).  Arithmetic = torch's single-tensor Adam (no amsgrad / maximize). */
int dir_adam_step(const void* table, int ntensors, double lr, double beta1, double beta2, double eps, double weight_decay,
                  long long step, dir_stream_t stream);
/* torch.optim.SGD.step() (imdb-wiki-dir/train.py:163-164: --optimizer sgd, momentum + weight decay) the same way: one launch for all
 * parameter tensors + the bf16 operand preparation.  table rows: (param, grad, momentum_buffer or 0, 0, numel, w16, w16_rot, Cout, R*S, Cin,
 * rot_mode, 0).  first != 0: the buffers hold no value yet (torch's first step sets them to the gradient).  Arithmetic = torch's
 * single-tensor SGD (weight decay, momentum, dampening, nesterov; no maximize). */
int dir_sgd_step(const void* table, int ntensors, double lr, double momentum, double dampening, double weight_decay, int nesterov, int first,
                 dir_stream_t stream);
/* rot_mode 0: as dir_conv_prep_weights.  rot_mode 1 (3x3): w16_rot receives the four parity-class weights of the
 * STRIDE-2 data gradient, packed back to back (class (a, b), a = row parity, b = column parity of the output pixel:
 * (1 + a)(1 + b) taps, [Cin][taps][Cout]; bases at 0, 1, 3, 5 taps x Cin x Cout) — the operand of dir_conv_dgrad_s2. */
int dir_conv_prep_weights_ex(const float* w, int Cout, int R, int S, int Cin, void* w16, void* w16_rot, int rot_mode,
                             dir_stream_t stream);
/* Data gradient of a 3x3 / stride-2 / pad-1 convolution (conv2 of the first block of stages 2-4, resnet.py:46-47) as
 * four stride-1 launches of the implicit-GEMM kernel, one per output-pixel parity class: dx[n, 2i+a, 2j+b, :] =
 * sum over the class's taps of dY[n, i+dr, j+ds, :] * W.  Every dx element is written exactly once: no zero fill, no
 * atomics.  dy [N, Ho, Wo, Cy] bf16, dx [N, 2 Ho, 2 Wo, Cx] bf16, wcls from dir_conv_prep_weights_ex(rot_mode = 1). */
int dir_conv_dgrad_s2(const void* dy, const void* wcls, void* dx, int N, int Ho, int Wo, int Cy, int Cx, dir_stream_t stream);
int dir_conv_fwd(const void* x, const void* w, void* y, float* stats, int stats_rows, int N, int H, int W, int Cin,
                 int Cout, int R, int S, int stride, int pad, dir_stream_t stream);
/* y = relu'(relu_mask) * bf16(bf16(conv(x, w)) + addend): the gradient accumulation autograd would run as a separate add kernel at
 * a fan-out (block input feeding conv1 and the identity shortcut) and the ReLU backward, fused into the data-gradient convolution's
 * store loop.  With x = dY
 * of a block's first convolution, addend = the shortcut gradient and relu_mask = the block input (the output of the
 * previous block's relu(bn3(.) + shortcut), resnet.py:66-68), y is the gradient that BatchNorm node needs with its ReLU
 * backward already applied, so dir_bn_bwd runs with relu = 0 and re-reads neither its saved output nor writes a
 * separate shortcut gradient.  addend, relu_mask: [N, Ho, Wo, Cout] bf16, either may be NULL; stats must be NULL when
 * one of them is given. */
int dir_conv_fwd_fused(const void* x, const void* w, const void* addend, const void* relu_mask, void* y, float* stats, int stats_rows,
                       int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, dir_stream_t stream);
/* Data gradient at a projection block's input (imdb-wiki-dir/resnet.py:57-68: x feeds conv1 AND the stride-2 1x1
 * downsample conv): y = relu'(relu_mask) * bf16(bf16(conv(x, w)) + addend + up2(addend_s2)), stride 1.  addend_s2 is the
 * COMPACT data gradient of the downsample conv, [N, Ho/2, Wo/2, Cout] bf16 (= its dY times its transposed weight, a
 * plain 1x1 stride-1 launch of this kernel): it is added at the even (ho, wo) only, so the 4x larger zero-filled
 * scatter a strided transposed convolution would write (and this launch re-read) never exists.  Ho, Wo even. */
int dir_conv_dgrad_join(const void* x, const void* w, const void* addend, const void* addend_s2, const void* relu_mask,
                        void* y, int N, int H, int W, int Cin, int Cout, int R, int S, int pad, dir_stream_t stream);
/* dir_conv_dgrad_join / dir_conv_dgrad_s2 with the FIRST pass of the BatchNorm backward that consumes the result fused into
 * the store loop: the result y is the gradient of the OUTPUT of a BatchNorm (bn1 / bn2 / bn3 of a Bottleneck, resnet.py:45-50)
 * whose input is bn_x ([N, Ho, Wo, Cout] bf16, the geometry of y); stats [rows][2][Cout] f32 receives, per 128-row tile, the
 * sums of g and g * bn_x over the rows (g = y as stored, in bf16) — the `partial` operand of dir_bn_bwd_partials.
 * stats_rows = dir_conv_plan_rows(...) for the stride-1 forms, 4 x dir_conv_stats_rows(N, Ho, Wo) for the stride-2 form (one block of
 * rows per parity class); checked by the launch.
 * bn_gamma / bn_beta / bn_mean / bn_rstd non-NULL: the BatchNorm is followed by a ReLU (no residual); g is taken under that
 * ReLU's mask, recomputed as bn_x * a + b > 0 with the forward's coefficients (the stored y stays unmasked). */
int dir_conv_dgrad_bnstats(const void* x, const void* w, const void* addend, const void* addend_s2, const void* relu_mask,
                           void* y, int N, int H, int W, int Cin, int Cout, int R, int S, int pad, const void* bn_x,
                           const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_rstd,
                           float* stats, int stats_rows, dir_stream_t stream);
/* The general stride-1 data gradient: dir_conv_dgrad_join, plus dir_conv_dgrad_bnstats when bn_x != NULL (then stats != NULL),
 * with the ReLU backward mask given either as the tensor itself (relu_mask, as above) or as the bit mask that
 * dir_bn_fwd_train_bits / dir_bn_apply_bits emitted for it (relu_mask_bits, [N*H*W][Cout / 8] bytes, bit j of a byte = channel
 * 8 b + j was positive): 1/16 of the bytes for the same decision.  At most one of the two.  variant: DIR_CONV_* (0 = product). */
int dir_conv_dgrad_ex(const void* x, const void* w, const void* addend, const void* addend_s2, const void* relu_mask,
                      const void* relu_mask_bits, void* y, int N, int H, int W, int Cin, int Cout, int R, int S, int pad,
                      const void* bn_x, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                      const float* bn_rstd, float* stats, int stats_rows, int variant, dir_stream_t stream);
int dir_conv_dgrad_s2_bnstats(const void* dy, const void* wcls, void* dx, int N, int Ho, int Wo, int Cy, int Cx,
                              const void* bn_x, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                              const float* bn_rstd, float* stats, int stats_rows, dir_stream_t stream);
/* dir_conv_dgrad_s2 / _bnstats (bn_x and stats both NULL or both given) with the kernel of the four launches forced (variant). */
int dir_conv_dgrad_s2_ex(const void* dy, const void* wcls, void* dx, int N, int Ho, int Wo, int Cy, int Cx,
                         const void* bn_x, const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                         const float* bn_rstd, float* stats, int stats_rows, int variant, dir_stream_t stream);
/* Stem convolution 7x7 / stride 2 / pad 3, 3 -> 64 channels (imdb-wiki-dir/resnet.py:79,129), MFMA without an im2col
 * buffer.  x [N, H, W, 3] bf16 (channels_last image), wpack = dir_stem_conv_prep_weights(w) with w the float32 master
 * weight [64][7][7][3] (= channels_last [64, 3, 7, 7]); y [N, Ho, Wo, 64] bf16; stats (may be NULL)
 * [dir_stem_conv_stats_rows(N, H)][2][64] f32 partial sums / sums of squares of the rounded outputs for dir_bn_*.
 * W % 8 == 0, Wo <= 128 (W <= 256). */
size_t dir_stem_conv_stats_rows(int N, int H);
int dir_stem_conv_prep_weights(const float* w, void* wpack /* [64][176] bf16 */, dir_stream_t stream);
int dir_stem_conv_fwd(const void* x, const void* wpack, void* y, float* stats, int N, int H, int W, dir_stream_t stream);
/* Weight gradient of the stem convolution: dy [N, Ho, Wo, 64] bf16, x [N, H, W, 3] bf16 -> dw [64][7][7][3] f32 (the
 * channels_last layout of the [64, 3, 7, 7] parameter).  Persistent MFMA kernel (K = the pixels of an output row) + a
 * fixed-order reduction over workgroups: deterministic.  workspace >= dir_stem_conv_wgrad_workspace(N, H) bytes. */
size_t dir_stem_conv_wgrad_workspace(int N, int H);
int dir_stem_conv_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, void* workspace,
                        size_t workspace_bytes, dir_stream_t stream);

/* dir_conv_fwd with the kernel forced (variant = DIR_CONV_*), for tests and A/B measurements; stats_rows =
 * dir_conv_plan_rows(..., 0, variant). */
int dir_conv_fwd_variant(const void* x, const void* w, void* y, float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout,
                         int R, int S, int stride, int pad, int variant, dir_stream_t stream);

/* K9w  weight gradient of the same convolution:  dw[co, r, s, ci] = sum_m dy[m, co] * x[gather(m, r, s), ci]
 * (float32 output, layout [Cout][R][S][Cin] = a channels_last [Cout, Cin, R, S] tensor).  MFMA GEMM with the
 * batch*pixel axis as K, operands transposed in registers on their way into LDS, deterministic split-K through
 * `workspace` (>= dir_conv_wgrad_workspace(...) bytes, 256-byte aligned; 0 = unsupported shape).
 * Replaces the weight half of nn.Conv2d's autograd (imdb-wiki-dir/resnet.py conv layers with Cin, Cout % 64 == 0).
 */
/* form (per launch, like `variant` of the forward entry points): DIR_WGRAD_AUTO = the product's choice — the 1x1 / stride-1 layers with
 * 128-multiples of channels take the register-lean form (both operands staged by LDS-DMA in their natural [pixel][channel] layout,
 * MFMA fragments by the transposing LDS read ds_read_b64_tr_b16: <= 128 registers, four workgroups per CU), everything else the
 * transposing kernel; the other values force one form (tests, A/B): DIR_EUNSUPPORTED / workspace 0 when it does not take the geometry. */
enum { DIR_WGRAD_AUTO = 0, DIR_WGRAD_TRANSPOSE = 1, DIR_WGRAD_DMA1 = 2 /* one LDS stage, four workgroups per CU */, DIR_WGRAD_DMA2 = 3 /* two stages */ };
size_t dir_conv_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int form);
int dir_conv_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout,
                   int R, int S, int stride, int pad, int form, void* workspace, size_t workspace_bytes,
                   dir_stream_t stream);
/* The 3x3 / stride-1 / pad-1 weight gradients (conv2 of every Bottleneck but the three strided ones, resnet.py:46-47) in ONE
 * pass over dY and X for all nine taps: H == W in {56, 28, 14, 7}, Cin % 64 == 0, Cout % 64 == 0 (workspace 0 = not this
 * shape; dir_conv_wgrad then runs its per-tap form).  A workgroup keeps a 64 x 64 (co x ci) block of all nine taps in
 * registers and walks over chunks of whole image rows staged in LDS in their natural layout (LDS-DMA); the taps are row
 * offsets of the transposing LDS reads (ds_read_b64_tr_b16).  Same deterministic split-K reduction as dir_conv_wgrad. */
size_t dir_conv_wgrad3x3_workspace(int N, int H, int W, int Cin, int Cout);
int dir_conv_wgrad3x3(const void* dy, const void* x, float* dw, int N, int H, int W, int Cin, int Cout, void* workspace,
                      size_t workspace_bytes, dir_stream_t stream);
/* dw[i] = sum over s < splits, in order, of part[s * n + i] (the reduction pass of both weight-gradient forms); n % 4 == 0. */
int dir_conv_wgrad_reduce_splits(const float* part, int splits, size_t n, float* dw, dir_stream_t stream);
/* Round 5: ONE reduction launch per backward pass instead of one per layer (52 launches of 5-8 us at ResNet-50).  The two *_partials entry
 * points run the split-K GEMM of dir_conv_wgrad / dir_conv_wgrad3x3 only: the partials stay at the start of `workspace` as
 * [*splits][Cout * R * S * Cin] float32 (the caller keeps the workspace alive).  dir_conv_wgrad_reduce_batched then sums the layers listed
 * in `table` (device memory, [nlayers][4] int64 = partials, splits, n, dw) exactly as dir_conv_wgrad_reduce_splits does per layer — same
 * order per element, bit-identical gradients (imdb-wiki-dir/train.py:260, loss.backward(): the weight half of every conv layer). */
int dir_conv_wgrad_partials(const void* dy, const void* x, int* splits, int N, int H, int W, int Cin, int Cout,
                            int R, int S, int stride, int pad, int form, void* workspace, size_t workspace_bytes, dir_stream_t stream);
int dir_conv_wgrad3x3_partials(const void* dy, const void* x, int* splits, int N, int H, int W, int Cin, int Cout, void* workspace,
                               size_t workspace_bytes, dir_stream_t stream);
int dir_conv_wgrad_reduce_batched(const void* table, int nlayers, dir_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * §8f-4  training-time image augmentation on the GPU.  Replaces, for a decoded and resized uint8 batch, the transform chain
 * of imdb-wiki-dir/datasets.py:38-53 behind Resize: RandomCrop(S, padding = pad, fill 0) -> RandomHorizontalFlip ->
 * ToTensor -> Normalize([.5]*3, [.5]*3) (torchvision.transforms; not vendored in the reference, semantics restated in
 * oracle/augment_oracle.py).  img [B, S, S, 3] uint8 (HWC); params [B][3] int32 = (top, left, flip) with top, left in
 * [0, 2 pad] — the draws RandomCrop.get_params / RandomHorizontalFlip make — or NULL for the evaluation transform (no crop,
 * no flip); out [B, S, S, 3] float32 or bf16 (dtype) = a channels_last [B, 3, S, S] tensor, float32 arithmetic exactly as
 * ToTensor / Normalize execute it ((u8 / 255 - 0.5) / 0.5; padding pixels -> -1). */
int dir_augment_u8(const void* img, const int* params, void* out, int dtype, int B, int S, int pad, dir_stream_t stream);
/* The Resize((S, S)) in front of that chain (imdb-wiki-dir/datasets.py:41,49: torchvision Resize on a PIL image = Pillow's bilinear
 * resize, antialiased when shrinking) for a RAGGED batch of decoded uint8 RGB images, bit for bit Pillow's result (fixed-point
 * two-pass algorithm of Pillow's Resample.c; oracle/resize_oracle.py): loader workers only decode.
 *   src: device bytes holding the B images back to back (HWC, 3 channels); table [B][4] int64 (device) = (byte offset of image b in
 *   src, H_b, W_b, byte offset of its [H_b][S][3] intermediate inside the workspace's tmp area); out [B, S, S, 3] uint8 — the input of
 *   dir_augment_u8.  hmax = max H_b; kmax = max over images and axes of dir_resize_ksize(in, S) (taps of the widest window);
 *   workspace >= dir_resize_u8_workspace(B, S, kmax, tmp_bytes), tmp_bytes = end of the last intermediate (sum of H_b * S * 3, offsets
 *   chosen by the host).  Three launches, no host sync. */
int    dir_resize_ksize(int in_size, int out_size);
size_t dir_resize_u8_workspace(int B, int S, int kmax, size_t tmp_bytes);
int    dir_resize_u8(const void* src, size_t src_bytes, const long long* table, void* out, int B, int S, int hmax, int kmax, void* workspace,
                     size_t workspace_bytes, dir_stream_t stream);    /* (ABI 3: src_bytes — table entries are checked against the source buffer too) */



/* 3x3 / stride 2 / pad 1 max pooling, NHWC bf16 (resnet.py:82,131 nn.MaxPool2d) with one argmax byte (0..8 = r*3+s,
 * first maximum in scan order like torch) per output element; backward is a gather over the <= 2x2 windows that
 * contain an input pixel (no atomics).  x [N,H,W,C], y/argmax [N,Ho,Wo,C], Ho = (H-1)/2+1;  C % 8 == 0. */
int dir_maxpool3x3s2_fwd(const void* x, void* y, void* argmax, int N, int H, int W, int C, dir_stream_t stream);
int dir_maxpool3x3s2_bwd(const void* dy, const void* argmax, void* dx, int N, int H, int W, int C, dir_stream_t stream);
/* Global average pool of the last feature map (imdb-wiki-dir/resnet.py:85,136-137: AvgPool2d(7) on the 7x7 map +
 * view): x [N, HW, C] bf16 -> y [N, C] float32 (mean kept in float32 for the FDS / linear / loss tail), and its
 * backward dx [N, HW, C] bf16 = dy / HW.  C % 8 == 0. */
int dir_avgpool_fwd(const void* x, float* y, int N, int HW, int C, dir_stream_t stream);
int dir_avgpool_bwd(const float* dy, void* dx, int N, int HW, int C, dir_stream_t stream);
/* Stem tail relu(bn1(x)) -> MaxPool2d(3, 2, 1) (imdb-wiki-dir/resnet.py:80-82,129-131) in one pass over the BatchNorm
 * INPUT x [N, H, W, C] bf16: the normalised map is never materialised.
 *   fwd: coef [2][C] from dir_bn_prepare_train; y [N, Ho, Wo, C] bf16; argmax [N, Ho, Wo, C] u8 (0..8 window position,
 *        9 = clipped by the ReLU).
 *   bwd: dy = gradient of y; dx = gradient of x (BatchNorm backward included: dgamma, dbeta out).
 *        workspace >= dir_bn_relu_maxpool_bwd_workspace(C) bytes.  C % 8 == 0, C <= 128. */
int dir_bn_relu_maxpool_fwd(const void* x, const float* coef, void* y, void* argmax, int N, int H, int W, int C,
                            dir_stream_t stream);
size_t dir_bn_relu_maxpool_bwd_workspace(int C);
int dir_bn_relu_maxpool_bwd(const void* dy, const void* argmax, const void* x, void* dx, int N, int H, int W, int C,
                            const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                            void* workspace, size_t workspace_bytes, dir_stream_t stream);
/* The same pair with the BatchNorm input AT THE ARGMAX kept by the forward (xmax [N, Ho, Wo, C] bf16, the bits of x as read): the
 * backward's reduction (sum g, sum g x) then streams three pooled-size tensors instead of gathering 2-byte elements out of the 4x
 * larger x, and the apply pass runs with one thread per 2 x 2 input block (four windows loaded once) instead of one per pixel.
 * xmax == NULL: exactly the functions above (gather reduction, per-pixel apply).  Identical results either way (same values, same
 * summation order): the older pair is the oracle of the newer one in tests/test_hip_bn.py. */
int dir_bn_relu_maxpool_fwd_xmax(const void* x, const float* coef, void* y, void* argmax, void* xmax, int N, int H, int W, int C,
                                 dir_stream_t stream);
int dir_bn_relu_maxpool_bwd_xmax(const void* dy, const void* argmax, const void* x, const void* xmax, void* dx, int N, int H, int W,
                                 int C, const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                                 void* workspace, size_t workspace_bytes, dir_stream_t stream);
/* Second third of dir_bn_bwd on its own: partial [rows][2][C] f32 of (sum g, sum g*x) -> dgamma, dbeta and
 * coef [3][C] = (a, p, q) with dx = a g + p x + q. */
int dir_bn_bwd_finalize(const float* partial, int rows, int64_t M, int C, const float* gamma, const float* save_mean,
                        const float* save_rstd, float* dgamma, float* dbeta, float* coef, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * STS-B-DIR FDS variant (sts-b-dir/fds.py, sts-b-dir/util.py:63-73) — SURVEY.md §8f-2.  Same scatter / finalize /
 * smooth / calibrate kernels as the age variant, plus:
 *   dir_fds_bin_edges:          bin = bucket_num-1 if label == edges[nedges-1], else
 *                               max((first i with edges[i] > label) - 1, bucket_start)   (fds.py:51-57); output is the
 *                               table row (bin - bucket_start), -1 where the reference raises (label > last edge, NaN).
 *                               edges: device f32 [nedges] = np.histogram(bins=bucket_num, range=(0,5)) edges.
 *   dir_fds_fill_empty_buckets: buckets with count == 0 in this update take (left+right)/2 (ends: the neighbour),
 *                               in increasing bucket order (fds.py:112-125).
 *   dir_fds_prepare_scale_ex:   guard_mode 0 = the age variant (column untouched when v1 == 0);
 *                               guard_mode 1 = util.py:66-70 as it EXECUTES on torch >= 1.2: `(v1 > 0) + (v2 >= 0)` is a
 *                               bool (logical or) and `== 2` is never true, so a row with ANY column v1 <= 0 or v2 < 0 is
 *                               returned unchanged as a whole;
 *                               guard_mode 2 = the intent (torch 0.4.1 semantics): only those columns stay untouched.
 */
int dir_fds_bin_edges(const float* labels, int n, const float* edges, int nedges, int bucket_start,
                      int bucket_num, int32_t* bins, dir_stream_t stream);
int dir_fds_fill_empty_buckets(const double* count, int nb, int C, float* running_mean, float* running_var,
                               dir_stream_t stream);
int dir_fds_prepare_scale_ex(const float* v1, const float* v2, int nb, int C, float clip_min, float clip_max,
                             int guard_mode, float* scale, dir_stream_t stream);

/* NYUD2-DIR dense FDS variant (nyud2-dir/models/fds.py, nyud2-dir/util.py:151-162) — SURVEY.md §8f-1: per-pixel buckets
 * bucket = clamp(int(depth * mult), bucket_start, bucket_num - 1) (float32 product, truncation; fds.py:51-53,138-139);
 * output = table row (bucket - bucket_start), -1 for NaN.  The [B,128,H,W] feature maps are handled as [B*H*W, 128] rows
 * by the shared kernels (narrow-row paths inside dir_fds_scatter_stats / dir_fds_calibrate_*), calibration uses
 * dir_fds_prepare_scale_ex with clip [0.2, 5]. */
int dir_fds_bin_scaled(const float* labels, long long n, float mult, int bucket_start, int bucket_num,
                       int32_t* bins, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Exact-float32 convolutions + pools: the PARITY MODE of the ResNet-50 stack (resnet50 run without autocast) and the
 * general fallback for shapes the bf16 MFMA kernels do not take (channel counts not multiples of 64, odd-sized strided
 * data gradients, tensors beyond their 32-bit offset range).  Replace nn.Conv2d forward / autograd for ANY layer of
 * imdb-wiki-dir/resnet.py:44-49,79,112-116 (cuDNN in the reference) with implicit GEMMs on v_mfma_f32_32x32x2_f32, which
 * is bit-for-bit a k-ordered fmaf chain (float32 products, float32 accumulation).  No library convolution is called
 * anywhere in the package.
 *   x  [N, H, W, Cin]  f32 NHWC        w  [Cout, R, S, Cin] f32 (= a channels_last [Cout, Cin, R, S] tensor)
 *   y / dy [N, Ho, Wo, Cout] f32 NHWC, Ho = (H + 2 pad - R) / stride + 1
 *   dir_conv_f32_dgrad: dx [N, H, W, Cin]   (every element written; no zero fill needed)
 *   dir_conv_f32_wgrad: dw [Cout, R, S, Cin]; deterministic split-K through `workspace`
 *                       (>= dir_conv_f32_wgrad_workspace(...) bytes, 256-byte aligned). */
int dir_conv_f32_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                     int stride, int pad, dir_stream_t stream);
int dir_conv_f32_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                       int stride, int pad, dir_stream_t stream);
/* data gradient with the fused store epilogue of the bf16 path (dir_conv_fwd_fused / dir_conv_dgrad_join): dx = dgrad
 * (+ addend [N,H,W,Cin]) (+ addend_s2, COMPACT [N,H/2,W/2,Cin], at the even pixels) then zeroed where !(relu_mask > 0);
 * any of the three may be NULL.  This is what lets the float32 parity mode run the SAME fused autograd graph as the bf16
 * product path (block-input gradient accumulation, deferred ReLU backward, projection pair). */
int dir_conv_f32_dgrad_fused(const float* dy, const float* w, const float* addend, const float* addend_s2,
                             const float* relu_mask, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                             int stride, int pad, dir_stream_t stream);
size_t dir_conv_f32_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad);
int dir_conv_f32_wgrad(const float* dy, const float* x, float* dw, int N, int H, int W, int Cin, int Cout, int R, int S,
                       int stride, int pad, void* workspace, size_t workspace_bytes, dir_stream_t stream);
/* The same three with the kernel forced (ABI 3; tests and A/B measurements).  variant 0 = the product's choice: the TILE kernels
 * (128 x 128 workgroup tile, 64 x 64 per wavefront, LDS-DMA operand staging, two LDS stages; 4x the gather kernels' rate) wherever
 * the K axis comes in whole 16-channel steps (Cin resp. Cout % 16 == 0; weight gradient: channels % 4 == 0), the element-GATHER
 * kernels otherwise (7x7 stem, odd channel counts).  Forward and data gradient of the two are bit-identical (same k order on the
 * same MFMA); the weight gradients differ in their split-K boundaries.  DIR_EUNSUPPORTED: variant TILE on a geometry it does not take. */
#define DIR_CONV_F32_GATHER 1
#define DIR_CONV_F32_TILE 2
/* ABI 4: the tile kernels with SPLIT-bf16 arithmetic on the bf16 matrix pipe — every float32 operand element split in registers into three
 * (X3) or two (X2) bf16 terms, six (three) v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block instead of eight v_mfma_f32_32x32x2_f32:
 * float32-GRADE results (X3: ~2^-23 per product, like float32's own rounding; X2: 16 significand bits), NOT bit-equal to the exact kernels.
 * `train.py --amp fp32x3` / the first epochs of `--amp_switch_epoch` run on X3; the parity mode (`--amp fp32`) stays on the exact MFMA. */
#define DIR_CONV_F32_TILE_X3 3
#define DIR_CONV_F32_TILE_X2 4
int dir_conv_f32_fwd_variant(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int R, int S,
                             int stride, int pad, int variant, dir_stream_t stream);
int dir_conv_f32_dgrad_variant(const float* dy, const float* w, const float* addend, const float* addend_s2,
                               const float* relu_mask, float* dx, int N, int H, int W, int Cin, int Cout, int R, int S,
                               int stride, int pad, int variant, dir_stream_t stream);
/* forward + BatchNorm statistics (ABI 3): the per-channel (sum, sum of squares) partials of y, one row per 64-row slab of the output
 * ([stats_rows][2][Cout] f32, stats_rows = dir_conv_f32_stats_rows(N, Ho, Wo)), formed in the tile kernel's store loop — what
 * dir_bn_fwd_train_partials consumes instead of a statistics pass over y (as dir_conv_fwd does for the bf16 path).  Tile-kernel
 * geometries with Cout % 4 == 0 only; DIR_EUNSUPPORTED otherwise (the caller then lets the BatchNorm count). */
size_t dir_conv_f32_stats_rows(int N, int Ho, int Wo);
int dir_conv_f32_fwd_stats(const float* x, const float* w, float* y, float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout,
                           int R, int S, int stride, int pad, dir_stream_t stream);
int dir_conv_f32_wgrad_variant(const float* dy, const float* x, float* dw, int N, int H, int W, int Cin, int Cout, int R, int S,
                               int stride, int pad, void* workspace, size_t workspace_bytes, int variant, dir_stream_t stream);
/* dir_conv_f32_fwd_stats with the kernel / arithmetic chosen (ABI 4): variant 0, DIR_CONV_F32_TILE, _TILE_X3 or _TILE_X2. */
int dir_conv_f32_fwd_stats_variant(const float* x, const float* w, float* y, float* stats, int stats_rows, int N, int H, int W, int Cin, int Cout,
                                   int R, int S, int stride, int pad, int variant, dir_stream_t stream);
/* float32 NHWC pools of the parity mode: MaxPool2d(3, 2, 1) with an argmax byte (resnet.py:82,131) and the global
 * average pool (resnet.py:85,136-137; sequential window sum / HW like torch's AvgPool2d). */
int dir_maxpool3x3s2_f32_fwd(const float* x, float* y, void* argmax, int N, int H, int W, int C, dir_stream_t stream);
int dir_maxpool3x3s2_f32_bwd(const float* dy, const void* argmax, float* dx, int N, int H, int W, int C, dir_stream_t stream);
int dir_avgpool_f32_fwd(const float* x, float* y, int N, int HW, int C, dir_stream_t stream);
int dir_avgpool_f32_bwd(const float* dy, float* dx, int N, int HW, int C, dir_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused network tail: AvgPool2d(7) + view -> FDS.smooth -> Linear(2048, 1)
 * (imdb-wiki-dir/resnet.py:136-148, fds.py:115-144) in ONE launch, and its backward.
 *   x [B, HW, C] (dtype DIR_BF16 or DIR_F32, NHWC map of the last stage), weight [C] f32, bias [1] f32.
 *   calibration: m1 / scale / m2 [nb, C] f32 (scale from dir_fds_prepare_scale) — all three NULL = no calibration
 *   (epoch < start_smooth, eval mode, FDS off).  The sample's bucket comes from `labels` [B] f32 (boundary presence
 *   flags over THIS batch, SURVEY A.3, like dir_fds_smooth_fwd; use for B <= 2048) or from `bins_in` [B] (dir_fds_bin_index).
 *   out: encoding [B, C] f32 = the CALIBRATED pooled features (the tensor resnet.py returns as `encoding`, A.2),
 *        pred [B] f32 = encoding . weight + bias,  bins_out [B] int32 (required when calibrating; saved for the backward).
 * dir_tail_bwd: dx [B, HW, C] (dtype) = ((dpred[b] * weight[c] (+ dencoding[b, c])) * s[bin_b, c]) / HW   (dx NULL = skip),
 *        dweight [C] = sum_b dpred[b] * encoding[b, c], dbias [1] = sum_b dpred[b]  (dweight NULL = skip both); fixed
 *        summation order.  workspace >= dir_tail_bwd_workspace(B, C) bytes.  bins NULL = no calibration in the forward.
 * Arithmetic is float32 and written like the unfused chain (dir_avgpool_* -> dir_fds_smooth_fwd / dir_fds_calibrate_bwd),
 * so encoding and dx are bit-identical to it; pred / dweight differ from a library gemv by summation order only. */
int dir_tail_fwd(const void* x, int dtype, const float* labels, const int32_t* bins_in, int B, int HW, int C,
                 int bucket_start, int bucket_num, const float* m1, const float* scale, const float* m2,
                 const float* weight, const float* bias, float* encoding, float* pred, int32_t* bins_out,
                 dir_stream_t stream);
size_t dir_tail_bwd_workspace(int B, int C);
int dir_tail_bwd(const float* dpred, const float* dencoding, const int32_t* bins, const float* scale, const float* weight,
                 const float* encoding, int B, int HW, int C, int dtype, void* dx, float* dweight, float* dbias,
                 void* workspace, size_t workspace_bytes, dir_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIR_HIP_H */
