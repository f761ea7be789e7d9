/* sggan.h -- C ABI of the MI355X-native SG-GAN train-step kernels (libsggan.so).
 *
 * Drop-in boundary for the hot path of fhfonsecaa/SG-GAN-TF2.
 * The reference has no FFI layer of its own: its hot path is a chain of stock
 * TensorFlow/Keras ops called from Python (module.py:208-318, model.py:149-200).
 * Each entry point below replaces the TF op(s) at the cited reference call site;
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *     the caller owns all buffers including workspaces; nothing is allocated,
 *     freed or synchronised inside; calls are stream-ordered on `stream`
 *     (a hipStream_t passed as void*).
 *   - activations are NHWC with the channel count padded to a multiple of 8
 *     (SGG_CPAD); padded channels hold zeros.  dtype selects the storage type
 *     of activations and packed weights: SGG_F32 (parity path, f32 MFMA) or
 *     SGG_BF16 (performance path, bf16 MFMA, f32 accumulate).  Parameters,
 *     gradients of parameters, statistics, logits and losses are always f32.
 *   - return value: SGG_OK (0) or a negative sgg_status; sgg_strerror() names it.
 */
#ifndef SGGAN_H
#define SGGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGG_VERSION 100          /* major*10000 + minor*100 + patch */
#define SGG_CPAD 8               /* activation channel granule */

typedef enum { SGG_OK = 0, SGG_EINVAL = -1, SGG_EUNSUPPORTED = -2, SGG_ELAUNCH = -3, SGG_EWORKSPACE = -4 } sgg_status;
typedef enum { SGG_F32 = 0, SGG_BF16 = 1 } sgg_dtype;
typedef enum { SGG_ACT_NONE = 0, SGG_ACT_RELU = 1, SGG_ACT_LRELU = 2, SGG_ACT_TANH = 3 } sgg_act;
typedef enum { SGG_PAD_ZERO = 0, SGG_PAD_REFLECT = 1 } sgg_pad_mode;

/* Geometry of one convolution  y[n,ho,wo,k] = sum_{r,s,c} xpad[n, ho*stride - pad_t + r, wo*stride - pad_l + s, c] * w[r,s,c,k].
 * pad_mode ZERO : out-of-range taps read 0 (TF 'SAME' is expressed by the caller as
 *                 pad_t/pad_l = the LEADING pads; the trailing pad is implied by Ho/Wo).
 * pad_mode REFLECT: tf.pad(...,"REFLECT") of pad_t (= pad_l) followed by a VALID conv.
 * C and K are the PADDED channel counts (multiples of SGG_CPAD). */
typedef struct sgg_conv_desc {
    int32_t N, H, W, C;      /* conv input  (NHWC)  */
    int32_t K, R, S;         /* conv output channels, kernel height/width */
    int32_t stride;          /* 1 or 2 */
    int32_t pad_t, pad_l;
    int32_t Ho, Wo;          /* conv output spatial size */
    int32_t pad_mode;        /* sgg_pad_mode */
    int32_t dtype;           /* sgg_dtype */
} sgg_conv_desc;

int sgg_version(void);
const char* sgg_strerror(int status);

/* ---- measurement hooks (bench.py's roofline leg; not part of the drop-in surface) -----------------------------------
 * sgg_time_next_launch(start, stop) arms one pair of HIP events for the calling host thread: the next launch of a timed
 * kernel family's MAIN kernel made by that thread -- the 3x3 halo GEMMs (sgg_conv2d_fwd* / _bwd_data* on the
 * residual-block shape), the all-taps weight gradient (sgg_conv2d_bwd_weight*), the instance-norm apply pass
 * (sgg_instnorm_fwd*) -- carries them on its own dispatch packet (hipExtLaunchKernel), so sgg_event_elapsed_ms gives that
 * kernel's begin-to-end time on the launch stream, as rocprofv3's kernel trace does.  Helper launches of the same call
 * (side-tensor gather, slab reduce, norm finalize) are not included.  (NULL, NULL) disarms and returns 1 if the armed
 * pair was consumed by a launch, 0 if not; arming returns the same for the previous pair.  Not for use while a stream is
 * being captured. */
int sgg_event_create(void** ev);
int sgg_event_destroy(void* ev);
int sgg_event_elapsed_ms(void* start, void* stop, float* ms);
int sgg_time_next_launch(void* start, void* stop);
/* Number of nodes captured so far into the graph `stream` is being captured to (-1: the stream is not capturing).  The
 * host-side step recorder (sggan_amd/graph.py) uses it to see that a graph segment is still empty, so that consecutive
 * host actions (gradient all-reduce launches / waits) share one cut instead of recording empty graphs. */
int sgg_stream_capture_nodes(void* stream, int* n);

/* ---- weights -------------------------------------------------------------------
 * Keras kernel (HWIO f32, module.py:211 etc.; for Conv2DTranspose the (kh,kw,out,in)
 * kernel of module.py:254,258 IS the HWIO kernel of the equivalent conv) ->
 *   w_fwd  [Kpad][R*S*Cpad]  (K-major rows, reduction contiguous)  used by conv fwd / deconv bwd-data
 *   w_dgrad[Cpad][R*S*Kpad]                                         used by conv bwd-data / deconv fwd
 * in `dtype`, zero-filled padding.  Either output may be NULL. */
int sgg_pack_conv_weights(const float* w_hwio, int R, int S, int C, int K, int Cpad, int Kpad,
                          int dtype, void* w_fwd, void* w_dgrad, void* stream);

/* The same for every conv layer of a network in one launch (all packed weights go stale together at each optimizer step).
 * items_dev: DEVICE array of n_items entries; pointers are device pointers, w_fwd / w_dgrad may be NULL;
 * max_elems = max over items of taps*Cpad*Kpad. */
typedef struct sgg_pack_item {
    const float* w;      /* HWIO f32 kernel of the layer */
    void* w_fwd;
    void* w_dgrad;
    int32_t taps, C, K, Cpad, Kpad, reserved;
} sgg_pack_item;
int sgg_pack_conv_weights_batch(const sgg_pack_item* items_dev, int n_items, int64_t max_elems, int dtype, void* stream);

/* ---- conv2d: tf.keras.layers.Conv2D (+ tf.pad REFLECT) ---- module.py:210-216,230-232,236,240,262-264,284-311
 * fwd: y = act(conv(x) + bias);  bias may be NULL; bias has Kpad f32 entries.
 * ws: sgg_conv2d_fwd_workspace() bytes (non-zero only for small outputs, which are computed split-K). */
size_t sgg_conv2d_fwd_workspace(const sgg_conv_desc* d);
int sgg_conv2d_fwd(const sgg_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                   void* y, int act, float leak, void* ws, size_t ws_bytes, void* stream);

/* conv2d forward that also emits, for the instance norm that follows it (module.py:211-212 etc.), the per-image
 * per-channel (sum, sum of squares) of the stored output, split over pixel chunks:
 *   partial[N][chunks][Kpad][2] f32,  chunks = sgg_conv2d_fwd_stats_chunks(d)  (0: this shape has no such epilogue;
 *   currently the bf16 3x3 stride-1 kernel).  No activation (the norm applies it).  Feed partial to
 *   sgg_instnorm_fwd_partial(), which then skips its own statistics pass over the tensor. */
size_t sgg_conv2d_fwd_stats_chunks(const sgg_conv_desc* d);
int sgg_conv2d_fwd_stats(const sgg_conv_desc* d, const void* x, const void* w_fwd, const float* bias, void* y,
                         float* partial, void* ws, size_t ws_bytes, void* stream);
/* bwd_data: dx = conv^T(dy) including the MirrorPadGrad fold for REFLECT (gen_tape.gradient, model.py:196).
 * ws: sgg_conv2d_bwd_data_workspace() bytes (REFLECT: pre-folded gather rows of the border pixels; small outputs: split-K slabs). */
size_t sgg_conv2d_bwd_data_workspace(const sgg_conv_desc* d);
/* addend (nullable, same shape/dtype as dx): dx = conv^T(dy) + addend -- the skip-connection gradient of a
 * residual block (module.py:217) folded into the epilogue. */
int sgg_conv2d_bwd_data(const sgg_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, void* dx,
                        void* ws, size_t ws_bytes, void* stream);

/* conv2d data gradient that also emits the first pass of the instance-norm BACKWARD that consumes dx (the norm in front of
 * this conv in the forward direction, module.py:212-215): partial[N][chunks][Cpad][2] = per pixel chunk (sum g, sum g*xhat),
 * g = dx * act'(gamma*xhat + beta), xhat = (norm_x - mean) * rstd from norm_stats (as written by sgg_instnorm_fwd).
 * chunks = sgg_conv2d_bwd_data_stats_chunks(d) (0: unsupported shape; currently the bf16 3x3 stride-1 kernel).
 * Feed partial to sgg_instnorm_bwd_partial().  ws as for sgg_conv2d_bwd_data. */
size_t sgg_conv2d_bwd_data_stats_chunks(const sgg_conv_desc* d);
int sgg_conv2d_bwd_data_stats(const sgg_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, void* dx,
                              const void* norm_x, const float* norm_stats, const float* norm_gamma, const float* norm_beta,
                              int norm_act, float norm_leak, float* partial, void* ws, size_t ws_bytes, void* stream);
/* Mixed-precision data gradient (an extension; the reference trains in f32 throughout, gen_tape.gradient model.py:196):
 * bf16 operands, dx written in F32 -- and the addend read as f32 when addend_is_f32 -- so that the gradient chain
 * between the instance norms of the residual blocks is not re-rounded to bf16 at every layer (the norm backward
 * subtracts most of dy; a rounding relative to dy is amplified there).  Supported where
 * sgg_conv2d_bwd_data_mixed_supported(d) returns 1 (the bf16 3x3 stride-1 kernel); ws as for sgg_conv2d_bwd_data. */
int sgg_conv2d_bwd_data_mixed_supported(const sgg_conv_desc* d);
int sgg_conv2d_bwd_data_mixed(const sgg_conv_desc* d, const void* dy, const void* w_dgrad, const void* addend, int addend_is_f32,
                              float* dx, void* ws, size_t ws_bytes, void* stream);
/* The lockstep pair (see sgg_instnorm_*_pair): d describes the STACKED batch (N = both networks' images); images
 * 0..nsplit-1 are convolved with (w, bias), the rest with (w2, bias2), in ONE launch -- where sgg_conv2d_pair_supported(d)
 * returns 1 (the bf16 3x3 stride-1 kernels); elsewhere call the one-network entry points on the two halves. */
int sgg_conv2d_pair_supported(const sgg_conv_desc* d);
int sgg_conv2d_fwd_stats_pair(const sgg_conv_desc* d, const void* x, const void* w_fwd, const float* bias, const void* w_fwd2,
                              const float* bias2, int nsplit, void* y, float* partial, void* ws, size_t ws_bytes, void* stream);
int sgg_conv2d_bwd_data_pair(const sgg_conv_desc* d, const void* dy, const void* w_dgrad, const void* w_dgrad2, int nsplit,
                             const void* addend, void* dx, void* ws, size_t ws_bytes, void* stream);
/* "Normalise on load" -- conv -> InstanceNormalization -> ReLU -> conv (the residual block, module.py:211-215) without the
 * norm's apply pass over the tensor: x_raw is the FIRST conv's raw output, x_stats its (mean, rstd)[N][C]
 * (sgg_instnorm_finalize over the first conv's statistics rows), x_gamma / x_beta the norm's parameters.  The second conv
 * applies relu(x * gamma * rstd + beta - mean * gamma * rstd) to its operand tiles after they land in LDS and writes the
 * normalised tensor to x_norm on the way (the backward pass needs it as this conv's weight-gradient operand).  y, partial:
 * as sgg_conv2d_fwd_stats.  w2 == NULL: one network; else the lockstep pair (images >= nsplit use w2 / bias2 / x_gamma2 /
 * x_beta2).  x_norm, y and partial are bit-identical to sgg_instnorm_fwd_partial(act = RELU) followed by sgg_conv2d_fwd_stats.
 * Supported where sgg_conv2d_fwd_normload_supported(d) returns 1 (bf16, REFLECT 3x3 stride 1, C <= 512). */
int sgg_conv2d_fwd_normload_supported(const sgg_conv_desc* d);
int sgg_conv2d_fwd_stats_normload(const sgg_conv_desc* d, const void* x_raw, const float* x_stats, const float* x_gamma, const float* x_beta,
                                  const float* x_gamma2, const float* x_beta2, void* x_norm, const void* w_fwd, const float* bias,
                                  const void* w_fwd2, const float* bias2, int nsplit, void* y, float* partial,
                                  void* ws, size_t ws_bytes, void* stream);
/* bwd_weight: dw_hwio[R][S][C_real][K_real] f32, overwritten (accumulate=0) or added to (accumulate=1: a network
 * applied twice in one step, model.py:186-187).  ws: sgg_conv2d_bwd_weight_workspace() bytes. */
size_t sgg_conv2d_bwd_weight_workspace(const sgg_conv_desc* d);
int sgg_conv2d_bwd_weight(const sgg_conv_desc* d, const void* x, const void* dy, float* dw_hwio,
                          int C_real, int K_real, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* Weight gradient of TWO applications of one layer in a step (same desc): dw (+)= wgrad(x0, dy0) + wgrad(x1, dy1) with one
 * launch, one set of split slabs and one reduce (the cycle step applies each generator twice, model.py:120-121 by name).
 * sgg_conv2d_bwd_weight_pair_supported() tells whether the shape has this path (the all-taps 3x3 stride-1 kernel);
 * otherwise call sgg_conv2d_bwd_weight twice.  Workspace: sgg_conv2d_bwd_weight_workspace(d). */
int sgg_conv2d_bwd_weight_pair_supported(const sgg_conv_desc* d);
int sgg_conv2d_bwd_weight_pair(const sgg_conv_desc* d, const void* x0, const void* dy0, const void* x1, const void* dy1,
                               float* dw, int C_real, int K_real, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* ... of TWO networks of one architecture, each applied twice (the upstream cycle step's generators): four (x, dy) sets, two dW, ONE
 * launch in which each network gets half the blocks -- half as many f32 slabs to write and reduce.  Same result as two
 * sgg_conv2d_bwd_weight_pair calls up to f32 summation order.  Workspace as for one network. */
int sgg_conv2d_bwd_weight_pair2(const sgg_conv_desc* d, const void* xa0, const void* dya0, const void* xa1, const void* dya1, float* dwa,
                                const void* xb0, const void* dyb0, const void* xb1, const void* dyb1, float* dwb,
                                int Cr, int Kr, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---- grouped launches: the same call site of TWO networks of one architecture in one call (the upstream cycle step runs
 * G_A->B beside G_B->A and D_A beside D_B; module.py:219-318 built twice).  `d` describes ONE network's call (N images per
 * network); x / y (dy / dx, addend) are the STACKED tensors of 2N images, the first network's images first; (w, bias) belong to
 * the first network, (w2, bias2) to the second.  The result is bit for bit that of two single calls -- each group runs the
 * single call's tiles and split-K -- but in ONE launch for the generic GEMM and the stacked-batch halo kernels (two half-size
 * launches of the discriminators' small maps are latency bound); the special 7x7 / narrow kernels run as two launches.
 * ws >= 2 x the single call's workspace.  (3x3 stride-1 shapes with the norm-statistics epilogue or REFLECT fold:
 * sgg_conv2d_fwd_stats_pair / sgg_conv2d_bwd_data_pair.) */
int sgg_conv2d_fwd_group2(const sgg_conv_desc* d, const void* x, const void* w, const float* bias, const void* w2,
                          const float* bias2, void* y, int act, float leak, void* ws, size_t ws_bytes, void* stream);
int sgg_conv2d_bwd_data_group2(const sgg_conv_desc* d, const void* dy, const void* w, const void* w2, const void* addend,
                               void* dx, void* ws, size_t ws_bytes, void* stream);
/* weight gradients of the two networks: their main kernels run back to back into two sets of split slabs, ONE reduce launch sums
 * both (each in the single call's order).  dw / dw2: the two networks' f32 gradients, layouts as sgg_conv2d_bwd_weight.
 * Exception -- the 3x3 all-taps halo shapes (stride 1 pad 1, or stride 2; 64 | C, 128 | K; also as Conv2DTranspose) and the
 * LDS-DMA kernel's shapes (K >= 256 and R*S*C >= 256): ONE main launch in which each network
 * gets half the blocks and half the split slabs (the slabs are that kernel's main memory traffic); the result then equals two
 * single calls up to f32 summation order (~1e-6 relative), not bit for bit. */
int sgg_conv2d_bwd_weight_group2(const sgg_conv_desc* d, const void* x, const void* dy, float* dw, float* dw2, int C_real, int K_real,
                                 int accumulate, void* ws, size_t ws_bytes, void* stream);
int sgg_deconv2d_bwd_weight_group2(const sgg_conv_desc* d, const void* x, const void* dy, float* dw, float* dw2, int C_real, int K_real,
                                   int accumulate, void* ws, size_t ws_bytes, void* stream);
int sgg_deconv2d_fwd_group2(const sgg_conv_desc* d, const void* x, const void* w_dgrad, const float* bias, const void* w_dgrad2,
                            const float* bias2, void* y, int act, float leak, void* ws, size_t ws_bytes, void* stream);
int sgg_deconv2d_bwd_data_group2(const sgg_conv_desc* d, const void* dy, const void* w_fwd, const void* w_fwd2, void* dx,
                                 void* ws, size_t ws_bytes, void* stream);

/* ---- deconv2d: tf.keras.layers.Conv2DTranspose(3x3, s2, 'same') ---- module.py:254,258
 * `d` describes the EQUIVALENT FORWARD CONV whose input is the deconv OUTPUT:
 *   (d->N,H,W,C) = deconv output, (d->Ho,Wo,K) = deconv input, pad_t/pad_l = TF SAME leading pads of that conv.
 * fwd: y[N,H,W,C] = act(conv^T(x[N,Ho,Wo,K]) + bias[C]). */
size_t sgg_deconv2d_fwd_workspace(const sgg_conv_desc* d);
int sgg_deconv2d_fwd(const sgg_conv_desc* d, const void* x, const void* w_dgrad, const float* bias,
                     void* y, int act, float leak, void* ws, size_t ws_bytes, void* stream);
/* the same (no activation) + the per-chunk (sum, sumsq) rows partial[N][chunks][C][2] of the STORED output for the InstanceNormalization
 * behind the layer (module.py:255,259), so that sgg_instnorm_fwd_partial() skips its statistics pass; chunks = sgg_deconv2d_fwd_stats_chunks(d),
 * 0 where the shape has no such epilogue.  w2 != NULL: a stacked batch of two networks, images >= nsplit use w2 / bias2. */
size_t sgg_deconv2d_fwd_stats_chunks(const sgg_conv_desc* d);
int sgg_deconv2d_fwd_stats(const sgg_conv_desc* d, const void* x, const void* w_dgrad, const float* bias, const void* w_dgrad2, const float* bias2,
                           int nsplit, void* y, float* partial, void* ws, size_t ws_bytes, void* stream);
size_t sgg_deconv2d_bwd_data_workspace(const sgg_conv_desc* d);
int sgg_deconv2d_bwd_data(const sgg_conv_desc* d, const void* dy, const void* w_fwd, void* dx,
                          void* ws, size_t ws_bytes, void* stream);
/* dw has the Keras transpose-kernel layout [R][S][C_real(out)][K_real(in)]. */
int sgg_deconv2d_bwd_weight(const sgg_conv_desc* d, const void* x, const void* dy, float* dw,
                            int C_real, int K_real, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* bias gradient: db[c] (+)= sum_p dy[p][c], c < C_real (f32).  ws >= sgg_bias_grad_workspace(P, C) bytes. */
size_t sgg_bias_grad_workspace(int64_t P, int C);
int sgg_bias_grad(const void* dy, float* db, int64_t P, int C, int C_real, int accumulate, int dtype,
                  void* ws, size_t ws_bytes, void* stream);
/* grouped call (see sgg_conv2d_fwd_group2): dy holds two networks' tensors back to back, P pixels each; ws >= 2 x the single call's */
int sgg_bias_grad_group2(const void* dy, float* db, float* db2, int64_t P, int C, int C_real, int accumulate, int dtype,
                         void* ws, size_t ws_bytes, void* stream);

/* ---- instance_norm: tfa.layers.InstanceNormalization ---- module.py:212,216,233,...,308 (spec by name: ops.py:13-22)
 * y = act(gamma*(x-mean)*rstd + beta) (+ residual, added AFTER act; module.py:217 uses act NONE).
 * stats[N][C][2] = (mean, rstd) f32 is written by fwd and read by bwd.
 * ws >= sgg_instnorm_workspace(N, HW, C) bytes. */
size_t sgg_instnorm_workspace(int N, int64_t HW, int C);
int sgg_instnorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                     float* stats, int N, int64_t HW, int C, float eps, int act, float leak, int dtype,
                     void* ws, size_t ws_bytes, void* stream);
/* same, with the statistics pass replaced by precomputed partial sums partial[N][chunks][C][2] (sgg_conv2d_fwd_stats) */
int sgg_instnorm_fwd_partial(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                             float* stats, const float* partial, int chunks, int N, int64_t HW, int C, float eps,
                             int act, float leak, int dtype, void* stream);
/* only the finalize step of the above: stats[N][C][2] = (mean, rstd) from the partial sums (no pass over the tensor) */
int sgg_instnorm_finalize(const float* partial, int chunks, float* stats, int N, int64_t HW, int C, float eps, void* stream);
/* dx = d/dx of the above given dy (w.r.t. the post-activation output); dgamma/dbeta[C_real] f32 overwritten or
 * (accumulate=1) added to.  gamma/beta/stats are indexed over the padded C. */
int sgg_instnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats,
                     void* dx, float* dgamma, float* dbeta, int N, int64_t HW, int C, int C_real, int accumulate,
                     int act, float leak, int dtype, void* ws, size_t ws_bytes, void* stream);
/* the same for an F32 dy against bf16 x / dx (mixed mode, see sgg_conv2d_bwd_data_mixed) */
int sgg_instnorm_bwd_mixed(const float* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                           float* dgamma, float* dbeta, int N, int64_t HW, int C, int C_real, int accumulate, int act, float leak,
                           void* ws, size_t ws_bytes, void* stream);
/* same, with the statistics pass replaced by precomputed partial sums partial[N][chunks][C][2] (sgg_conv2d_bwd_data_stats);
 * ws >= N*C*4 floats */
int sgg_instnorm_bwd_partial(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                             float* dgamma, float* dbeta, const float* partial, int chunks, int N, int64_t HW, int C, int C_real,
                             int accumulate, int act, float leak, int dtype, void* ws, size_t ws_bytes, void* stream);

/* Two networks of the same shape in lockstep on ONE stacked batch (the cycle step's G_A->B / G_B->A and D_A / D_B pairs,
 * model.py:114-133 applied to both translation directions): images 0..nsplit-1 belong to the first network (gamma, beta,
 * dgamma, dbeta), images nsplit..N-1 to the second (gamma2, ...).  Instance norm is per image, so one launch over the 2x larger
 * tensor serves both -- same arithmetic per image as two separate calls (bit-identical), at a higher fraction of HBM bandwidth
 * and half the launches. */
int sgg_instnorm_fwd_pair(const void* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2, int nsplit,
                          const void* residual, void* y, float* stats, int N, int64_t HW, int C, float eps, int act, float leak,
                          int dtype, void* ws, size_t ws_bytes, void* stream);
int sgg_instnorm_fwd_partial_pair(const void* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2, int nsplit,
                                  const void* residual, void* y, float* stats, const float* partial, int chunks, int N, int64_t HW, int C,
                                  float eps, int act, float leak, int dtype, void* stream);
int sgg_instnorm_bwd_pair(const void* dy, const void* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                          int nsplit, const float* stats, void* dx, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2,
                          int N, int64_t HW, int C, int C_real, int accumulate, int act, float leak, int dtype,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- lrelu / relu / tanh: tf.keras.layers.LeakyReLU / Activation ---- module.py:213,265,285 (ops.py:36-37) */
int sgg_act_fwd(const void* x, void* y, int64_t n, int act, float leak, int dtype, void* stream);
/* dx = dy * act'(.) evaluated from the OUTPUT y (relu/lrelu: sign of y; tanh: 1-y^2). */
int sgg_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int act, float leak, int dtype, void* stream);
/* out = a + b (residual join of gradients). */
int sgg_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);

/* ---- semantic mask over the adversarial map ---- module.py:312-314
 * out[n,i,j] = sum_{c<C_real} h4[n,i',j',c] * mask[n,i,j,c];  h4 is (N,hh,hw,Cpad) in dtype, mask (N,mh,mw,C_real) f32;
 * (hh,hw) == (mh,mw), or (1,1) broadcast over the mask grid (the 128x128 reference case).  out f32 (N,mh,mw). */
int sgg_mask_reduce_fwd(const void* h4, const float* mask, float* out, int N, int hh, int hw, int mh, int mw,
                        int C_real, int Cpad, int dtype, void* stream);
int sgg_mask_reduce_bwd(const float* dout, const float* mask, void* dh4, int N, int hh, int hw, int mh, int mw,
                        int C_real, int Cpad, int dtype, void* stream);

/* ---- losses ---- model.py:149-166
 * BCE-with-logits against a constant label, global mean:  *loss (+)= weight*mean(...);
 * dlogits[i] (+)= weight*gscale*(sigmoid(x_i)-label)/n.  accumulate bit0: add into *loss; bit1: add into dlogits. */
int sgg_bce_logits(const float* logits, int64_t n, float label, float weight, float gscale, float* loss, float* dlogits,
                   int accumulate, void* stream);
/* L1 (abs_criterion, module.py:336-337): *loss (+)= weight*mean_{p,c<C_real}|a-b|;  db (+)= -weight*gscale*sign(a-b)/(P*C_real)
 * (0 in padded channels).  accumulate bit0: add into *loss; bit1: add into db. */
int sgg_l1_loss(const void* a, const void* b, int64_t P, int C_real, int Cpad, float weight, float gscale, float* loss,
                void* db, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);
size_t sgg_l1_loss_workspace(int64_t P, int Cpad);

/* ---- defined-not-wired SG-GAN criteria, used by the cycle-mode step ----
 * LSGAN criterion against a constant target -- mae_criterion (module.py:340-341; squared error despite the name):
 *   *loss (+)= weight*mean((x-t)^2);  dx (+)= weight*gscale*2(x-t)/n.  accumulate bits as sgg_bce_logits. */
int sgg_mse_const(const float* x, int64_t n, float target, float weight, float gscale, float* loss, float* dx,
                  int accumulate, void* stream);
/* segmentation-edge indicator (model.py:108-119): 1 where the REFLECT-padded colour segmentation has a non-zero central
 * difference in x or y on any channel, else 0.  seg (N,H,W,Cpad) in dtype -> out f32 (N,H,W). */
int sgg_seg_edge_weight(const void* seg, float* out, int N, int H, int W, int C_real, int Cpad, int dtype, void* stream);
/* gradient-sensitive loss -- tf_deriv + gradloss_criterion (module.py:325-351): Sobel gx/gy (depthwise, SAME zero pad) of
 * `in` and `target`;  *loss (+)= lambda * mean_pixels( weight * mean_{2C}| |d(in)| - |d(target)| | );
 * din (+)= gscale * d(lambda*loss)/d(in) (NULL: loss only).  accumulate bit0: loss, bit1: din. */
size_t sgg_gradloss_workspace(int N, int H, int W, int C_real);
int sgg_gradloss(const void* in, const void* target, const float* weight, int N, int H, int W, int C_real, int Cpad,
                 float lambda, float gscale, float* loss, void* din, int accumulate, int dtype,
                 void* ws, size_t ws_bytes, void* stream);

/* ---- optimizer: tf.keras.optimizers.Adam.apply_gradients ---- model.py:199-200,205-207
 * Keras form, over one flat f32 buffer:  m=b1*m+(1-b1)*g; v=b2*v+(1-b2)*g^2;
 * theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps);   g is first multiplied by grad_scale (1/world for DP). */
int sgg_adam(float* theta, const float* g, float* m, float* v, int64_t n, int t, float lr, float beta1, float beta2,
             float eps, float grad_scale, void* stream);
/* The same update with the step number kept on the DEVICE: `state` is int64[2] -- state[0] is the counterpart of Keras'
 * `optimizer.iterations` variable (t = state[0] + 1 is used and state[0] incremented), state[1] is scratch.  Two launches
 * (a 1-thread one that evaluates lr_t, then the update); no host-side step state, so the call can be captured into a HIP
 * graph and replayed (the eager per-op dispatch of model.py:168 is what this removes). */
int sgg_adam_iter(float* theta, const float* g, float* m, float* v, int64_t n, int64_t* state, float lr, float beta1,
                  float beta2, float eps, float grad_scale, void* stream);

/* ---- data side of the step ----
 * segment_class.py:60-70,95-97: colour -> class index, bit exact.  rgb: uint8 [n_pixels][channels>=3]. */
int sgg_seg_class_map(const uint8_t* rgb, int channels, int64_t n_pixels, uint8_t* out, void* stream);
/* host-side copy of the kernel's colour table (keys = R<<16|G<<8|B); returns the entry count (21). No GPU needed. */
int sgg_seg_class_table(uint32_t* keys_host, uint8_t* vals_host, int capacity);
/* utils.py:158-165 + :197-199 (deviation D1): one-hot of the align-corners nearest resample of the
 * class-index map:  mask[n,i,j,c] = (idx[n, round(i*(H-1)/(oh-1)), round(j*(W-1)/(ow-1))] == c). */
int sgg_onehot_resample(const uint8_t* idx, float* mask, int N, int H, int W, int oh, int ow, int n_classes, void* stream);
/* ---- evaluation (next-row SURVEY 8(f)4): metric._fast_hist (metric.py:18-24) and scores_seg_fake (metric.py:71-77), bit exact
 * hist[n_class*t + p] += 1 for pixels with 0 <= t,p < n_class (uint64 counts, caller zeroes);
 * labels[i] = argmax_c uint8(255*x[i][c]) over the first C_real channels (first maximum wins). */
int sgg_confusion_hist(const int32_t* label_true, const int32_t* label_pred, int64_t n, int n_class, uint64_t* hist, void* stream);
int sgg_argmax_u8_labels(const void* x, int32_t* labels, int64_t P, int C_real, int Cpad, int dtype, void* stream);
/* f32 [P][Cs] -> dtype [P][Cd] with zero fill (Cd >= Cs), and back (drops padded channels). */
int sgg_pad_channels(const float* src, void* dst, int64_t P, int Cs, int Cd, int dtype, void* stream);
int sgg_unpad_channels(const void* src, float* dst, int64_t P, int Cs, int Cd, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGGAN_H */
