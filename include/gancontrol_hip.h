/*
 * gancontrol_hip.h -- C ABI of the MI355X (gfx950) StyleGAN2 hot-path kernels for gan-control.
 *
 * This is the drop-in boundary.  Every entry point takes raw device pointers, explicit
 * extents and a HIP stream (as void*); the library never allocates or frees memory the
 * caller can see, keeps no mutable global state, launches asynchronously on the stream it is
 * given and never synchronises.  All tensors are contiguous NCHW float32 unless stated.
 *
 * Return value: 0 on success, negative on failure (GC_ERR_*).  gc_last_error() returns a
 * thread-local, human-readable description of the last failure on the calling thread.
 *
 * The reference (amazon-science/gan-control) ships no native code: its operator socket is the
 * module-level "if FUSED:" block of src/gan_control/models/gan_model.py:19-50, which binds the
 * names FusedLeakyReLU / fused_leaky_relu / upfirdn2d, and falls back to ATen calls
 * (F.conv2d, F.conv_transpose2d, F.pad, F.leaky_relu).  Each function below cites the reference
 * lines whose arithmetic it replaces.  INTEGRATION.md shows the binding a maintainer adds.
 */
#ifndef GANCONTROL_HIP_H
#define GANCONTROL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct below changes layout or an entry point changes its signature (2: gc_conv_desc.in_pitch / out_pitch and
 * gc_conv_epilogue.residual, added during ABI 1 without a bump, + gc_struct_sizes).  A binding must check BOTH gc_abi_version() and
 * gc_struct_sizes() against its own mirrors before the first call: a stale library otherwise mis-reads every descriptor. */
#define GC_ABI_VERSION 2

#define GC_OK 0
#define GC_ERR_BAD_ARG (-1)      /* null pointer, non-positive extent, inconsistent geometry */
#define GC_ERR_UNSUPPORTED (-2)  /* legal request outside what the kernels implement */
#define GC_ERR_HIP (-3)          /* the HIP runtime reported a launch error */
#define GC_ERR_WORKSPACE (-4)    /* workspace too small */

typedef void* gc_stream_t; /* hipStream_t */

int gc_abi_version(void);
const char* gc_last_error(void);
/* sizeof() of the library's own view of every struct of this header, in declaration order: gc_conv_desc, gc_conv_epilogue, gc_wlayout_group,
 * gc_wpack_group, gc_glin_group, gc_wsq_group.  Writes min(n, count) entries and returns count. */
#define GC_STRUCT_COUNT 6
int gc_struct_sizes(size_t* sizes, int n);
/* sha256 (first 16 hex digits) over the kernel sources and build flags this library was compiled from, stamped at build time (csrc/Makefile,
 * tools/build_alt.sh): measurements are tagged with it so that counters collected on one build are never quoted for another. */
const char* gc_source_hash(void);

/* ------------------------------------------------------------------------------------------
 * K1  upfirdn2d: zero-stuff by (up) -> pad / crop -> 2-D FIR -> decimate by (down).
 *
 * Replaces upfirdn2d() gan_model.py:45-50 -> upfirdn2d_native() pytorch_upfirdn2d.py:9-51
 * (callers: Blur gan_model.py:126-129, Upsample :86-89, Downsample :107-110,
 * non_leaking.py:338,359).
 *
 *   y[p,oy,ox] = sum_{a<kh, b<kw} T[a][b] * U[oy*down_y + a - pad_y0][ox*down_x + b - pad_x0]
 *   U[v][u]    = x[p][v/up_y][u/up_x] if up_y | v, up_x | u and the source is inside, else 0
 *   T[a][b]    = taps[kh-1-a][kw-1-b] if flip_taps (the forward op: a true convolution)
 *              = taps[a][b]           otherwise   (the adjoint uses the un-flipped kernel)
 *
 * planes = N*C.  taps is a DEVICE pointer to kh*kw floats (the module's `kernel` buffer).
 * out_h/out_w are the caller's: (in*up + pad0 + pad1 - k)/down + 1; negative pads crop.
 * The gradient w.r.t. x is the same call with up<->down, flip_taps toggled and
 * pad0' = k - 1 - pad0; the double-backward is the forward call again.
 */
int gc_upfirdn2d_f32(const float* x, const float* taps, float* y,
                     int planes, int in_h, int in_w, int out_h, int out_w,
                     int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_y0, int flip_taps, gc_stream_t stream);

/* K1 with the NoiseInjection + FusedLeakyReLU that follow the Blur of an up-sampling StyledConv
 * (gan_model.py:295-307 then :402-408) applied before the result is stored:
 *
 *   y[b,c,o] = gain * lrelu( FIR(x)[b,c,o] + noise_w[0] * noise[b,o] + bias[c], slope )
 *
 * in the arithmetic order of gc_bias_act_f32 (bit-identical to the two-pass result, one write + one read of the tensor
 * saved).  4x4 taps, up = down = 1, planes at least 64 x 16: GC_ERR_UNSUPPORTED otherwise (callers then run K1 + K2).
 * bias ([channels]) may be NULL; noise ([batch, out_h*out_w]) and noise_w go together. */
/* gc_upfirdn2d_f32 / gc_upfirdn2d_act_f32 over an input whose rows are in_pitch >= in_w floats apart (planes in_h * in_pitch apart): the
 * pitched output of a transposed convolution (gc_conv_desc.out_pitch) read by the Blur that follows it (gan_model.py:304-307) -- and / or
 * writing rows out_pitch >= out_w floats apart (0 = dense): the (H + 1)-wide output of the Blur in front of a stride-2 convolution
 * (ConvLayer gan_model.py:866-872), whose 16-byte stores otherwise straddle cache lines (4.4 vs 5.6 TB/s).
 * up = down = 1 with 4 x 4 taps on planes the tile kernel takes (out_w >= 64, out_h >= 16); bias / noise / noise_w NULL and
 * slope = gain = 1, activate = 0 give the plain FIR. */
int gc_upfirdn2d_pitched_f32(const float* x, const float* taps, float* y, int batch, int channels, int in_h, int in_w, int in_pitch,
                             int out_h, int out_w, int out_pitch, int kh, int kw, int pad_x0, int pad_y0, int flip_taps, int activate,
                             const float* bias, const float* noise, const float* noise_w, float slope, float gain, gc_stream_t stream);

int gc_upfirdn2d_act_f32(const float* x, const float* taps, float* y,
                         int batch, int channels, int in_h, int in_w, int out_h, int out_w,
                         int kh, int kw, int pad_x0, int pad_y0, int flip_taps,
                         const float* bias, const float* noise, const float* noise_w, float slope, float gain,
                         gc_stream_t stream);

/* y = FIR(x) * (mask_ref > 0 ? gain : gain * slope): the Blur ADJOINT followed by the backward of the bias + leaky-ReLU whose output the
 * Blur read (ResBlock: conv1 -> FusedLeakyReLU -> Blur -> stride-2 conv2, gan_model.py:893-922, 844-890) in one pass -- the product is
 * applied to the fp32 FIR result, so the values equal gc_upfirdn2d_f32 followed by gc_bias_act_bwd_f32 bit for bit, while the
 * gradient makes one round trip to HBM less.  mask_ref: [batch, channels, out_h, out_w] dense (the activation output); x rows in_pitch
 * floats apart (0 = dense); 4 x 4 taps, up = down = 1, planes the tile kernel takes (out_w >= 64, out_h >= 16). */
int gc_upfirdn2d_mask_f32(const float* x, const float* taps, float* y, int batch, int channels, int in_h, int in_w, int in_pitch,
                          int out_h, int out_w, int kh, int kw, int pad_x0, int pad_y0, int flip_taps,
                          const float* mask_ref, float slope, float gain, gc_stream_t stream);

/* The backward of StyledConv's up-sampling tail (Blur -> NoiseInjection -> FusedLeakyReLU, gan_model.py:402-408 after :304-307) in one pass
 * over the gradient:  g_pre = gy * (y_ref > 0 ? gain : gain * slope) is formed while gy is staged, gx = FIR(g_pre) (the Blur adjoint) is
 * stored, and per (plane, tile)  psum = sum g_pre,  pdot = sum g_pre * noise[b]  over the input elements the tile owns (every element once):
 * the bias gradient is the sum of psum over batch and tiles, the noise-strength gradient the sum of all pdot.  g_pre never travels to HBM
 * (gc_bias_act_bwd_reduce_f32 + gc_upfirdn2d_f32 write and re-read it).  gy, y_ref: [batch, channels, in_h, in_w] dense; noise
 * [batch, in_h * in_w] with pdot, or both NULL; gx rows out_pitch floats apart (0 = dense); psum / pdot: [batch * channels,
 * gc_upfirdn2d_actbwd_tiles(out_h, out_w)].  4 x 4 taps on planes the tile kernel takes. */
int gc_upfirdn2d_actbwd_tiles(int out_h, int out_w);
int gc_upfirdn2d_actbwd_f32(const float* gy, const float* y_ref, const float* noise, const float* taps, float* gx, float* psum, float* pdot,
                            int batch, int channels, int in_h, int in_w, int out_h, int out_w, int out_pitch, int kh, int kw,
                            int pad_x0, int pad_y0, int flip_taps, float slope, float gain, gc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2  fused (noise +) bias + leaky-ReLU * gain.
 *
 * Replaces FusedLeakyReLU.forward gan_model.py:32-35, fused_leaky_relu gan_model.py:39-41 and,
 * when `noise` is given, the NoiseInjection add that always precedes it
 * (gan_model.py:340-345, 402-408).
 *
 *   y[b,c,i] = gain * lrelu(x[b,c,i] + bias[c] + noise_w[0] * noise[b,i], slope)
 *
 * inner = product of the dims after the channel dim (1 for [B,F] inputs).  bias may be null;
 * noise ([batch, inner]) and noise_w (DEVICE scalar) are both null or both set.
 */
int gc_bias_act_f32(const float* x, const float* bias, const float* noise, const float* noise_w,
                    float* y, int batch, int channels, int64_t inner,
                    float slope, float gain, gc_stream_t stream);

/* Gradient (and, being linear in dy, also the double-backward) of K2 through the activation,
 * masked by the sign of the forward OUTPUT:
 *   dx[i] = dy[i] * (y_ref[i] > 0 ? gain : gain * slope)
 */
int gc_bias_act_bwd_f32(const float* dy, const float* y_ref, float* dx, int64_t count,
                        float slope, float gain, gc_stream_t stream);

/* The same gradient fused with the reductions its caller needs (one pass instead of three):
 *   dx = dy * mask(y_ref);   psum[(b*C + c)*chunks + j] = sum over chunk j of dx[b,c,:]
 *   pdot[(b*C + c)*chunks + j] = sum over chunk j of dx[b,c,:] * noise[b,:]      (only when noise != null)
 * with chunks = gc_bias_act_bwd_chunks(inner).  Bias gradient = psum summed over b and j; noise-strength
 * gradient = pdot summed over everything.  Deterministic (fixed-order block reductions, no atomics). */
int gc_bias_act_bwd_chunks(int64_t inner);
int gc_bias_act_bwd_reduce_f32(const float* dy, const float* y_ref, const float* noise, float* dx,
                               float* psum, float* pdot, int batch, int channels, int64_t inner,
                               float slope, float gain, gc_stream_t stream);

/* The same pass with one more reduction, for a convolution whose activation ran in its epilogue (gc_conv_epilogue) so
 * that the pre-activation tensor was never written:
 *   pself[(b*C + c)*chunks + j] = sum over chunk j of dx[b,c,:] * x_pre[b,c,:],
 *   x_pre = lrelu^-1(y_ref / gain) - bias[c] - noise_w[0] * noise[b,:]      (the activation's input, rebuilt from its output)
 * Summed over the chunks and divided by out_scale[b,c] this is the out_scale (demodulation) gradient of K3, which would
 * otherwise need gc_plane_dot_f32 over (dx, pre-activation).  bias / noise / noise_w / pdot / pself may be NULL
 * (noise_w is required with noise when pself is requested); slope and gain must be non-zero for pself. */
int gc_bias_act_bwd_reduce_self_f32(const float* dy, const float* y_ref, const float* noise, const float* bias, const float* noise_w,
                                    float* dx, float* psum, float* pdot, float* pself, int batch, int channels, int64_t inner,
                                    float slope, float gain, gc_stream_t stream);

/* ADJOINT of gc_bias_act_bwd_reduce_self_f32 with respect to dy (and to y_ref through pself): the second-order pass of the R1 and
 * path-length regularisers (d_r1_loss / g_path_regularize, generator_trainer.py / gan_model.py) through an activation backward.
 * Cotangents: ggx [B,C,inner] of dx (may be NULL), cs / cd / cw [B,C,chunks] of psum / pdot / pself (each may be NULL).  With
 *   total = ggx + cs + cd * noise + cw * x_pre      (x_pre as above)
 *   g_dy   = total * (y_ref > 0 ? gain : gain * slope)
 *   g_yref = cw * dx * (y_ref > 0 ? 1 / gain : 1 / (gain * slope))            (only with cw; dx = the first pass's output)
 *   pgb[(b*C + c)*chunks + j] = sum over chunk j of cw * dx          (bias gradient   = -sum over b, j)
 *   pgn[(b*C + c)*chunks + j] = sum over chunk j of cw * dx * noise  (noise_w gradient = -sum over everything)
 * one read of each input and one write of each output instead of ~14 elementwise passes.  g_yref / pgb / pgn may be NULL;
 * dx is required when cw is given. */
int gc_bias_act_bwd_reduce_adjoint_f32(const float* ggx, const float* cs, const float* cd, const float* cw, const float* y_ref, const float* dx,
                                       const float* noise, const float* bias, const float* noise_w, float* g_dy, float* g_yref, float* pgb, float* pgn,
                                       int batch, int channels, int64_t inner, float slope, float gain, gc_stream_t stream);

/* partial[p*chunks + j] = sum over chunk j of a[p,:] * b[p,:], chunks = gc_bias_act_bwd_chunks(inner); planes = B*C.
 * Gradients of the per-sample modulation / demodulation factors of K3 (sum_hw x * dx and sum_hw dy * y). */
int gc_plane_dot_f32(const float* a, const float* b, float* partial, int planes, int64_t inner, gc_stream_t stream);

/* The same reduction over planes of `rows` x `width` elements whose rows are a_pitch / b_pitch floats apart (planes rows * pitch apart):
 * partial[p * chunks + j] with chunks = gc_plane_dot_pitched_chunks(rows). */
int gc_plane_dot_pitched_chunks(int rows);
int gc_plane_dot_pitched_f32(const float* a, const float* b, float* partial, int planes, int rows, int width, int a_pitch, int b_pitch, gc_stream_t stream);

/* out[r] = (sum_j partial[r*chunks + j]) / den[r]; den may be NULL (plain sum), a zero denominator counts as one.
 * The second stage of gc_plane_dot_f32 / the pself sums of gc_bias_act_bwd_reduce_self_f32 fused with the division by the modulation
 * (in_scale) / demodulation (out_scale) factor: `sum_hw x * dx / s` of ModulatedConv2d's weight algebra (gan_model.py:284-293) in one
 * launch.  chunks = 1 is an element-wise safe division of two [rows] vectors. */
int gc_rows_sum_div_f32(const float* partial, const float* den, float* out, int rows, int chunks, gc_stream_t stream);

/* Per-channel sum over batch and inner dims: out[c] = sum_{b,i} x[b,c,i]  (bias gradient).
 * Deterministic two-stage reduction; workspace must hold gc_channel_sum_workspace() bytes. */
size_t gc_channel_sum_workspace(int batch, int channels, int64_t inner);
int gc_channel_sum_f32(const float* x, float* out, int batch, int channels, int64_t inner,
                       void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K3/K4  generalised 2-D convolution on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 *
 * One contraction covers F.conv2d (EqualConv2d.forward gan_model.py:154-160 and the plain
 * ModulatedConv2d branch gan_model.py:325-329: up = 1, down = stride), F.conv_transpose2d
 * (ModulatedConv2d up-sampling branch gan_model.py:295-306 and every input-gradient:
 * up = stride, down = 1) and, through in_scale / out_scale, StyleGAN2 weight (de)modulation
 * WITHOUT materialising per-sample weights (gan_model.py:284-293):
 *
 *   y[b,n,oy,ox] = out_scale[b,n] * sum_{ty,tx,k} w[ty,tx,k,n] * in_scale[b,k]
 *                                   * U(x)[b,k, oy*down + ty - pad_y, ox*down + tx - pad_x]
 *
 * U zero-stuffs x by `up` as in K1.  w is [kh, kw, in_ch, out_ch] (out_ch contiguous) in
 * CORRELATION order; in_scale [batch, in_ch] and out_scale [batch, out_ch] may be null (= 1).
 * Supported: kh = kw in {1, 3}; (up, down) in {(1,1), (1,2), (2,1)}.
 */
typedef struct gc_conv_desc {
    int32_t batch;
    int32_t in_ch, out_ch;
    int32_t in_h, in_w;
    int32_t out_h, out_w;
    int32_t kh, kw;
    int32_t up, down;
    int32_t pad_y, pad_x;
    int32_t in_pitch;    /* floats between the starts of two INPUT rows (planes in_h * in_pitch apart); 0 (or in_w) = dense.  Honoured by the
                          * stride-2 launches gc_conv2d_in_pitch_ok() names (forward and weight gradient), whose input is the (H + 1)-wide
                          * output of a Blur written with aligned rows (gc_upfirdn2d_pitched_f32, out_pitch). */
    int32_t out_pitch;   /* floats between the starts of two output rows; 0 (or out_w) = dense.  Only the launches gc_conv2d_out_pitch()
                          * names honour another value (a plane is then out_h * out_pitch floats): rows of a (2H + 1)-wide transposed-
                          * convolution output are never 16-byte aligned, and their partial-line stores bound that kernel. */
} gc_conv_desc;

/* Products with a tiny INNER extent (the weight gradients of the style path: EqualLinear gan_model.py:171-202 on [B, 512] latents, the
 * modulation / demodulation algebra of ModulatedConv2d gan_model.py:284-293 -- g[B, C]^T @ x[B, 512], a rank-B update):
 *
 *   out[M, N] (dense, row-major) = alpha * (a[M, K] @ b[K, N]) + beta * bias[N]        K <= 8
 *
 * a and b are addressed through element strides (transposed views need no copy); bias may be NULL.
 * torch.addmm(bias, a, b, beta=beta, alpha=alpha) for the shape where a library GEMM is all latency.
 * gc_small_gemm_ok() says whether a shape is taken (1: K <= 8 and M * N <= 2^19) or should go to a GEMM library (0). */
int gc_small_gemm_ok(int M, int K, int N, int64_t sb0, int64_t sb1);
int gc_small_gemm_f32(const float* a, int64_t sa0, int64_t sa1, const float* b, int64_t sb0, int64_t sb1,
                      const float* bias, float beta, float alpha, float* out, int M, int K, int N, gc_stream_t stream);

/* The row pitch (floats, a multiple of 32) the convolution of `d` in arithmetic `mode` (0 f32, 1 bf16x3, 2 bf16) can write its output with
 * when that pays -- the fused transposed 3x3 convolution with an odd output width -- or 0: write dense rows. */
int gc_conv2d_out_pitch(const gc_conv_desc* d, int mode);

/* 1 when the forward convolution (wgrad = 0) / weight gradient (wgrad = 1) of `d` in arithmetic `mode` reads a row-pitched input
 * (gc_conv_desc.in_pitch) in place: the split-bf16 stride-2 kernels. */
int gc_conv2d_in_pitch_ok(const gc_conv_desc* d, int mode, int wgrad);

int gc_conv2d_f32(const gc_conv_desc* d, const float* x, const float* w,
                  const float* in_scale, const float* out_scale, float* y, gc_stream_t stream);

/* Fused epilogue: the (noise +) bias + leaky-ReLU pass K2 that follows every convolution of the path (EqualConv2d ->
 * FusedLeakyReLU in ConvLayer gan_model.py:844-890; ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU in StyledConv
 * gan_model.py:402-408; `+ self.bias` in ToRGB gan_model.py:430) applied to the accumulators before they are stored:
 *
 *   y[b,n,o] = A( out_scale[b,n] * sum(...) + noise_w[0] * noise[b,o] + bias[n] ) + residual[b,n,o],
 *   A(v) = activate ? gain * lrelu(v, slope) : v
 *
 * `residual` (same shape as y, may be NULL) is the second operand of an add that follows the convolution: the
 * `out + skip` of ResBlock.forward (gan_model.py:919-921) and -- on the backward side -- the gradient arriving from the
 * other consumer of a tensor that feeds two layers (what autograd would sum with a separate elementwise pass).
 *
 * with the arithmetic of gc_bias_act_f32 in the same order, so conv + epilogue equals gc_conv2d_f32 followed by
 * gc_bias_act_f32 bit for bit while saving one write and one read of the activation tensor.  ep == NULL: plain convolution.
 */
typedef struct gc_conv_epilogue {
    const float* bias;     /* [out_ch] or NULL */
    const float* noise;    /* [batch, out_h * out_w] or NULL */
    const float* noise_w;  /* DEVICE scalar; set exactly when noise is */
    float slope, gain;     /* used when activate != 0 */
    int32_t activate;
    const float* residual; /* [batch, out_ch, out_h, out_w] or NULL; must not alias y */
} gc_conv_epilogue;

int gc_conv2d_fused_f32(const gc_conv_desc* d, const float* x, const float* w,
                        const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep,
                        float* y, gc_stream_t stream);

/* gc_conv2d_fused_f32 with scratch memory: with gc_conv2d_f32_workspace(d) bytes (0 for most shapes) the layers whose output planes are
 * <= 16 pixels wide are split over the input channels -- one 16 / 32-channel weight slab per workgroup over every sample (in groups of
 * samples when their planes exceed the LDS) for planes <= 8 x 8 and for transposed 3x3 onto planes <= 10 x 10 (conv_f32_small_kernel),
 * K slices of conv_mfma_kernel otherwise -- and a fixed-order pass adds the slices and applies
 * out_scale and the epilogue: same arithmetic (exact fp32), 256 workgroups instead of 8..64.  Without workspace (or with too little) it
 * is gc_conv2d_fused_f32. */
size_t gc_conv2d_f32_workspace(const gc_conv_desc* d);
int gc_conv2d_fused_f32_ws(const gc_conv_desc* d, const float* x, const float* w,
                           const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep,
                           float* y, void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* The same contraction on the bf16 matrix cores with split-bf16 ("bf16x3") arithmetic: every fp32
 * operand is split into bf16 hi + lo parts and a*b is formed as hi*hi + hi*lo + lo*hi with fp32
 * accumulation (~5e-6 relative error per layer, 5.3x the fp32 MFMA rate).  Inputs and outputs stay
 * fp32; `workspace` (gc_conv2d_bf16x3_workspace() bytes, 16-byte aligned) receives the split weights.
 * Shapes the fast kernel does not cover (in_ch < 16, planes <= 8 px wide) run on gc_conv2d_f32.  Transposed 3x3 launches of the
 * (2H + 1) x (2W + 1) geometry with >= 256 input channels run as an H x W main region plus an edge launch (last row and column).
 */
size_t gc_conv2d_bf16x3_workspace(const gc_conv_desc* d);
int gc_conv2d_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* w,
                         const float* in_scale, const float* out_scale, float* y,
                         void* workspace, size_t workspace_bytes, gc_stream_t stream);

int gc_conv2d_fused_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* w,
                               const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep, float* y,
                               void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* The split weights as a separate, reusable object.  A parameter changes once per optimiser step but is used by several
 * launches in between (forward and input gradient of D over the fake and real batches, R1, the generator passes of the D and G
 * steps), so the caller may split a weight tensor ONCE -- gc_conv2d_pack_weights_bf16x3 into a buffer of
 * gc_conv2d_bf16x3_packed_bytes(d) bytes, 16-byte aligned; the buffer depends on (kh, kw, in_ch, out_ch) and w only -- and hand
 * it to every launch that uses the same w (the reference recomputes `weight * scale` per call, gan_model.py:154, 284).
 * gc_conv2d_bf16x3_packed_bytes returns 0 for shapes that run on the fp32 kernel: pass packed = NULL and a workspace of
 * gc_conv2d_bf16x3_workspace(d) bytes then.  Same results as gc_conv2d_fused_bf16x3_f32, bit for bit.
 */
size_t gc_conv2d_bf16x3_packed_bytes(const gc_conv_desc* d);
/* Small planes (9 .. 32 pixels wide, >= 128 input channels) are split over the input channels to fill the chip; the slices are summed in
 * a fixed order by a finish pass that applies out_scale and the epilogue.  gc_conv2d_bf16x3_splitk_bytes(d) is the workspace that takes
 * (0 when the shape is not split); without it the launch runs unsplit -- same result up to fp32 summation order. */
size_t gc_conv2d_bf16x3_splitk_bytes(const gc_conv_desc* d);
int gc_conv2d_pack_weights_bf16x3(const gc_conv_desc* d, const float* w, void* packed, size_t packed_bytes, gc_stream_t stream);
int gc_conv2d_fused_bf16x3_packed_f32(const gc_conv_desc* d, const float* x, const float* w, const void* packed, size_t packed_bytes,
                                      const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep, float* y,
                                      void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* Plain bf16 arithmetic: the kernels of the split-bf16 path with ONE MFMA per product on the bf16-rounded operands (fp32 accumulate,
 * fp32 in HBM on both sides; ~3e-3 relative error per layer) -- the precision BASELINE.json's config[1] names.  Same arguments,
 * packed weights (gc_conv2d_pack_weights_bf16x3; the lo half is not read) and workspaces as the split-bf16 entry points. */
int gc_conv2d_fused_bf16_packed_f32(const gc_conv_desc* d, const float* x, const float* w, const void* packed, size_t packed_bytes,
                                    const float* in_scale, const float* out_scale, const gc_conv_epilogue* ep, float* y,
                                    void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* Name of the kernel variant the dispatcher picks for a shape ("conv_bf16x3_kernel<1,4,2,2>|up1,down1,k3", ...), written by the
 * dispatch code itself (the launchers run in a no-launch probe mode): what a host-side profiler keys its per-kernel statistics on.
 * mode: 0 = exact fp32, 1 = split-bf16, 2 = plain bf16.  name must hold >= 128 bytes. */
int gc_conv2d_variant_name(const gc_conv_desc* d, int mode, char* name, int name_bytes);

/* Weight gradient of the same contraction (up must be 1):
 *
 *   dw[ty,tx,k,n] = sum_{b,oy,ox} in_scale[b,k] * x[b,k, oy*down + ty - pad_y, ox*down + tx - pad_x]
 *                                 * out_scale[b,n] * dy[b,n,oy,ox]
 *
 * Replaces the weight half of aten::convolution_backward reached from gan_model.py:154,304,327.
 * Deterministic: partial sums go to `workspace` (gc_conv2d_wgrad_workspace() bytes) and are
 * reduced in a fixed order.
 */
size_t gc_conv2d_wgrad_workspace(const gc_conv_desc* d);
int gc_conv2d_wgrad_f32(const gc_conv_desc* d, const float* x, const float* dy,
                        const float* in_scale, const float* out_scale, float* dw,
                        void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* Split-bf16 variant of the weight gradient (same contract; shapes it does not cover -- fewer than 32 channels, planes < 4 px
 * wide, and 3x3 gradients onto planes <= 8 x 8, which one exact-fp32 launch computes without a workspace -- run on
 * gc_conv2d_wgrad_f32, so size the workspace with this function). */
size_t gc_conv2d_wgrad_bf16x3_workspace(const gc_conv_desc* d);
int gc_conv2d_wgrad_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* dy,
                               const float* in_scale, const float* out_scale, float* dw,
                               void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* Plain-bf16 variant (see gc_conv2d_fused_bf16_packed_f32); workspace as for the split-bf16 weight gradient. */
int gc_conv2d_wgrad_bf16_f32(const gc_conv_desc* d, const float* x, const float* dy,
                             const float* in_scale, const float* out_scale, float* dw,
                             void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* Weight gradient together with each sample's share of it:
 *
 *   dw_samples[b,ty,tx,k,n] = in_scale[b,k] * out_scale[b,n] * sum_{oy,ox} x[b,k, oy*down + ty - pad_y, ...] * dy[b,n,oy,ox]
 *   dw = sum_b dw_samples[b]                                       (both reduced in a fixed order)
 *
 * What the shares are for: ModulatedConv2d (gan_model.py:281-331) multiplies its weight by the per-sample style,
 * `weight = self.scale * self.weight * style` (and by the demodulation factor), so the gradients of those factors are contractions of
 * the SAME per-sample products with the weight (gc_wgrad_samples_contract_f32 below) -- autograd's route through
 * aten::convolution_backward of the grouped convolution materialises them per sample as well.  Taking them from here replaces two
 * full-plane products per layer, sum_p x * dx and sum_p dy * y (gc_plane_dot_f32), by reads of [B, taps, K, N].
 * The pixel splits of the plain weight gradient are regrouped so that every split stays inside one sample; same kernels, same
 * arithmetic.  gc_conv2d_wgrad_samples_workspace(): bytes of scratch for `mode` (0 = fp32, 1 = split-bf16, 2 = plain bf16), 0 when
 * the shape has no per-sample form: fp32 has one for the thin 1x1 shapes only (<= 4 channels on one side: ToRGB), the bf16 modes
 * additionally for everything their own weight-gradient kernels take (>= 32 channels on both sides, planes >= 4 px wide). */
size_t gc_conv2d_wgrad_samples_workspace(const gc_conv_desc* d, int mode);
int gc_conv2d_wgrad_samples_f32(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale,
                                float* dw, float* dw_samples, void* workspace, size_t workspace_bytes, gc_stream_t stream);
int gc_conv2d_wgrad_samples_bf16x3_f32(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale,
                                       float* dw, float* dw_samples, void* workspace, size_t workspace_bytes, gc_stream_t stream);
int gc_conv2d_wgrad_samples_bf16_f32(const gc_conv_desc* d, const float* x, const float* dy, const float* in_scale, const float* out_scale,
                                     float* dw, float* dw_samples, void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* The scale gradients from the per-sample shares (w and dw_samples[b] are [taps, a, c], the kernel layout [kh*kw, K, N]):
 *
 *   g_a[b,a] = sum_{t,c} w[t,a,c] * dw_samples[b,t,a,c] / scale_a[b,a]          (d loss / d in_scale:  the modulation, gan_model.py:283-284)
 *   g_c[b,c] = sum_{t,a} w[t,a,c] * dw_samples[b,t,a,c] / scale_c[b,c]          (d loss / d out_scale: the demodulation, gan_model.py:286-288)
 *
 * (dw_samples carries both scales as factors; dividing one out leaves the derivative with respect to it.  A scale of exactly 0 divides
 * by 1, as gc_rows_sum_div_f32 does.)  g_a or g_c may be null, a null scale divides by 1.  Fixed summation order.
 * workspace: gc_wgrad_samples_contract_workspace(batch, a, c) bytes, needed for g_c only. */
size_t gc_wgrad_samples_contract_workspace(int batch, int a, int c);
int gc_wgrad_samples_contract_f32(const float* dw_samples, const float* w, const float* scale_a, const float* scale_c, float* g_a, float* g_c,
                                  int batch, int taps, int a, int c, void* workspace, size_t workspace_bytes, gc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weight re-layout: scale + permute + optional tap mirror in one pass.
 *
 *   dst[t' * dst_stride[0] + k * dst_stride[1] + n * dst_stride[2]]
 *       = scale * src[t * src_stride[0] + k * src_stride[1] + n * src_stride[2]],   t' = flip_taps ? taps - 1 - t : t
 *
 * for t < taps (= kh*kw), k < in_ch, n < out_ch; strides in ELEMENTS, non-negative.  Replaces the ATen passes the
 * reference spends per call on `self.weight * self.scale` (gan_model.py:154, 284), the transpose / view of the
 * up-sampling branch (gan_model.py:295-303) and the permutes inside aten::convolution_backward:
 *   parameter [N,K,taps] -> kernel layout [taps,K,N]      src_stride = {1, taps, K*taps}, dst_stride = {K*N, N, 1}
 *   kernel layout -> input-gradient weights [taps',N,K]   src_stride = {K*N, N, 1},       dst_stride = {N*K, 1, K}, flip
 *   weight gradient [taps,K,N] -> parameter layout        src_stride = {K*N, N, 1},       dst_stride = {1, taps, K*taps}
 * src and dst must not overlap.
 */
int gc_weight_layout_f32(const float* src, float* dst, int taps, int k, int n,
                         const int64_t src_stride[3], const int64_t dst_stride[3],
                         int flip_taps, float scale, gc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Image-space pieces of the ADA augmentation (random_apply_affine, non_leaking.py:316-371) besides its two FIR passes.
 *
 * gc_affine_warp_bilinear_f32: bilinear resampling of x [batch, channels, in_h, in_w] into y [batch, channels, out_h, out_w]
 * under a per-sample affine map mat [batch, 6]: output pixel (ox, oy) reads the input at
 *     sx = m0*ox + m1*oy + m2,  sy = m3*ox + m4*oy + m5        (input PIXEL coordinates, pixel centres at integers)
 * with zeros outside -- what F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False) computes for the grid the
 * reference builds with make_grid / affine_grid (non_leaking.py:244-263, 338-357: an affine function of the output pixel index,
 * which the host folds into mat).  adjoint != 0: the transposed map -- x is then the gradient of the OUTPUT
 * [batch, channels, out_h, out_w] and y receives the gradient of the input [batch, channels, in_h, in_w] (scatter-add).
 *
 * gc_reflect_pad_f32: y = F.pad(x, (left, right, top, bottom), mode='reflect') on `planes` planes of in_h x in_w
 * (non_leaking.py:288-313); adjoint != 0: x is the gradient of the padded tensor, y receives the folded-back gradient.
 */
int gc_affine_warp_bilinear_f32(const float* x, const float* mat, float* y, int batch, int channels,
                                int in_h, int in_w, int out_h, int out_w, int adjoint, gc_stream_t stream);
int gc_reflect_pad_f32(const float* x, float* y, int planes, int in_h, int in_w, int left, int right, int top, int bottom,
                       int adjoint, gc_stream_t stream);

/* The derived weight forms of a whole network in a few launches.  After an optimiser step every convolution weight of a network is needed
 * again as kernel layout, as input-gradient layout and (split-bf16 mode) as hi / lo packs of both: ~4 launches of 4 - 6 us per layer and
 * weight version when done one tensor at a time (gc_weight_layout_f32, gc_conv2d_pack_weights_bf16x3).  The grouped entry points take a
 * table of the same arguments and produce bit-identical results (the per-element arithmetic is shared). */
typedef struct gc_wlayout_group {
    const float* src;
    float* dst;
    int32_t taps, k, n, flip_taps;
    int64_t src_stride[3], dst_stride[3];
    float scale;
} gc_wlayout_group;
int gc_weight_layout_grouped_f32(const gc_wlayout_group* groups, int n_groups, gc_stream_t stream);

typedef struct gc_wpack_group {
    gc_conv_desc desc;        /* as for gc_conv2d_pack_weights_bf16x3 */
    const float* w;           /* [kh, kw, in_ch, out_ch] */
    void* packed;             /* gc_conv2d_bf16x3_packed_bytes(&desc) bytes, 16-byte aligned */
    size_t packed_bytes;
} gc_wpack_group;
int gc_conv2d_pack_weights_bf16x3_grouped(const gc_wpack_group* groups, int n_groups, gc_stream_t stream);

/* The backward of a fused (k <= 3 -> n) 1x1 convolution + bias + leaky-ReLU -- the discriminator's FromRGB ConvLayer (gan_model.py:955,
 * 844-890), whose output [B, 32, 1024, 1024] is the largest activation of D -- without a separate activation-backward pass: the gradient dy
 * that arrives at the activation output is multiplied by the mask (y_ref > 0 ? gain : gain * slope) while it is loaded.
 *   gc_pw_act_wgrad_f32  dw_db[j, :] = sum_{b,p} x[b, j, p] * dy_masked[b, :, p]  for j < k, and dw_db[k, :] = sum_{b,p} dy_masked[b, :, p]
 *                        (the weight gradient in [k, n] order followed by the bias gradient; fixed summation order)
 *   gc_pw_act_dgrad_f32  gx[b, j, p] = sum_i w[i, j] * dy_masked[b, i, p]          w: [n, k] (the input-gradient weights), k <= 4
 * x: [batch, k, plane], dy / y_ref: [batch, n, plane], gx: [batch, k, plane]; planes dense. */
size_t gc_pw_act_wgrad_workspace(int batch, int k, int n, int64_t plane);
int gc_pw_act_wgrad_f32(const float* x, const float* dy, const float* y_ref, float* dw_db, int batch, int k, int n, int64_t plane,
                        float slope, float gain, void* workspace, size_t workspace_bytes, gc_stream_t stream);
int gc_pw_act_dgrad_f32(const float* dy, const float* y_ref, const float* w, float* gx, int batch, int n, int k, int64_t plane,
                        float slope, float gain, gc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Grouped dense layers of the style path: every EqualLinear of one kind in ONE launch.
 *
 * Replaces, per generator pass, the 26 `self.modulation(style)` calls of ModulatedConv2d.forward (gan_model.py:281-283; EqualLinear
 * :171-202: F.linear(input, weight * scale, bias * lr_mul)) and the 18 demodulation sums `rsqrt((weight ** 2).sum([2, 3, 4]) + 1e-8)`
 * (:284-293; here `scale^2 * (s^2) @ (sum_taps W^2)^T + eps`), and their ATen backward / double-backward GEMMs.
 *
 * Group g:   y_g[b, j] = alpha_g * sum_k x_g[b, k] * w_g[j, k] + beta_g * bias_g[j]        b < batch, j < n_g, k < k_g
 *   x_g: rows x_stride floats apart (>= k), w_g: [n, k] row-major (the EqualLinear parameter layout), bias_g: [n] or NULL, y_g: [batch, n]
 *   contiguous.  k and x_stride multiples of 4, x / w 16-byte aligned.  Any number of groups (the table is cut into launches of 32).
 * The three entry points are each other's derivatives and share the table type; the comments give the role of each pointer:
 *   gc_grouped_linear_f32        reads x, w, bias          writes y
 *   gc_grouped_linear_bwd_x_f32  reads y (= dL/dy), w      writes x (= dL/dx = alpha * gy @ w)
 *   gc_grouped_linear_bwd_w_f32  reads y (= dL/dy), x      writes w (= dL/dw = alpha * gy^T @ x) and, when non-NULL, bias (= beta * sum_b gy)
 * Fixed summation order: bit-identical from run to run. */
typedef struct gc_glin_group {
    float* x;
    float* w;
    float* bias;
    float* y;
    int32_t n, k;
    int64_t x_stride;
    float alpha, beta;
} gc_glin_group;
int gc_grouped_linear_f32(const gc_glin_group* groups, int n_groups, int batch, gc_stream_t stream);
int gc_grouped_linear_bwd_x_f32(const gc_glin_group* groups, int n_groups, int batch, gc_stream_t stream);
int gc_grouped_linear_bwd_w_f32(const gc_glin_group* groups, int n_groups, int batch, gc_stream_t stream);

/* sum over the taps of W^2 -- the weight half of ModulatedConv2d's demodulation `rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8)`
 * (gan_model.py:289-290), which only changes with the weight -- for many weight tensors in one launch, and its gradient:
 *   gc_weight_sq_grouped_f32      out[r] = sum_t w[r, t]^2                 r < rows (= out_ch * in_ch), t < taps;   g unused
 *   gc_weight_sq_bwd_grouped_f32  out[r, t] = 2 * w[r, t] * g[r] */
typedef struct gc_wsq_group {
    const float* w;
    const float* g;
    float* out;
    int32_t rows, taps;
} gc_wsq_group;
int gc_weight_sq_grouped_f32(const gc_wsq_group* groups, int n_groups, gc_stream_t stream);
int gc_weight_sq_bwd_grouped_f32(const gc_wsq_group* groups, int n_groups, gc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * f-2  Forward pass of the FID feature network (inference, fp32): src/gan_control/fid_utils/inception.py:17-165 over
 * overwrite_inception.py.  Replaces BasicConv2d.forward (conv -> BatchNorm(eval) -> ReLU, overwrite_inception.py:424-434), the
 * F.avg_pool2d / F.max_pool2d calls of the Inception blocks (inception.py:204-305, overwrite_inception.py:252-341), the
 * nn.MaxPool2d / nn.AdaptiveAvgPool2d of the wrapper (inception.py:92-125) and its F.interpolate + `2 * x - 1` (inception.py:147-154).
 * ------------------------------------------------------------------------------------------ */

/* y[b, chan_off + n, :, :] = relu?(scale[n] * conv(x, w)[b, n] + shift[n]);  w is [out_ch, in_ch, kh, kw] (the reference layout),
 * taps up to 7 x 7 (also 1 x 7, 7 x 1, 1 x 3, 3 x 1, 5 x 5), stride 1 or 2, zero padding; scale / shift (the folded BatchNorm) may be
 * NULL (1 / 0).  The output tensor has out_channels planes per sample: a block's branches write their slices of the concatenation. */
int gc_conv2d_bn_relu_f32(const float* x, const float* w, const float* scale, const float* shift, float* y,
                          int batch, int in_ch, int out_ch, int in_h, int in_w, int kh, int kw, int stride, int pad_y, int pad_x,
                          int relu, int out_channels, int chan_off, gc_stream_t stream);

/* k x k pooling with stride / zero padding; mode 0 = max, 1 = average over the taps inside the image (count_include_pad = False,
 * the FID patch of inception.py:204-207).  Output at a channel offset as above. */
int gc_pool2d_f32(const float* x, float* y, int batch, int channels, int in_h, int in_w, int k, int stride, int pad, int mode,
                  int out_channels, int chan_off, gc_stream_t stream);

/* y[p] = mean(x[p, :]): nn.AdaptiveAvgPool2d((1, 1)) over planes = B * C. */
int gc_global_avgpool_f32(const float* x, float* y, int planes, int inner, gc_stream_t stream);

/* y = mul * bilinear(x) + add, F.interpolate(mode='bilinear', align_corners=False) semantics. */
int gc_resize_bilinear_f32(const float* x, float* y, int planes, int in_h, int in_w, int out_h, int out_w, float mul, float add, gc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GANCONTROL_HIP_H */
