/* libscrabble_hip.so -- C-ABI of the MI355X (gfx950) kernels behind the ScrabbleGAN train step.
 *
 * The reference (UtkuKaradeniz/scrabble-gan) has no FFI of its own: every op below is a TensorFlow
 * primitive it calls from Python.  Each entry point names the reference call site it replaces
 * (paths under /root/reference/src).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (fp32 unless typed otherwise); nothing
 *     is retained after the call; no allocation happens inside (scratch is passed in)
 *   - `stream` is a hipStream_t; all work is asynchronous on it
 *   - tensors are NHWC contiguous; kernels use the TF layouts: Conv2D [kh,kw,Cin,Cout],
 *     Conv2DTranspose [kh,kw,Cout,Cin], Dense [in,out]
 *   - return 0 on success, <0 on error (SG_ERR_*); never throws
 *   - weight/bias gradient outputs ACCUMULATE (+=): the caller zeroes the flat gradient buffer once
 *     per step, which is what lets D/S passes on several batches share one buffer
 */
#ifndef SCRABBLE_HIP_H
#define SCRABBLE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define SG_OK 0
#define SG_ERR_ARG (-1)
#define SG_ERR_LAUNCH (-2)
#define SG_ERR_UNSUPPORTED (-3)

/* flags */
#define SG_RELU_IN 1   /* max(x,0) applied to the activation operand while it is loaded        */
#define SG_ACCUM 2     /* out += result                                                          */
#define SG_RELU_OUT 4  /* max(.,0) applied to the result                                          */
#define SG_TANH_OUT 8  /* tanh applied to the result (Cout == 1 path)                             */
#define SG_UPS2_IN 64   /* sg_conv2d_bwd_data_wino / sg_conv2d_bwd_weight_wino with tile = 4 only: the gradient operand dy is given at HALF
                         * resolution [B,H/2,W/2,Cout] and stands for 0.25 * upsample2x2(dy) -- the backward of tf.nn.pool(AVG) (resnet_ops.py:105-106)
                         * folded into the gradient transforms; H, W stay the convolution's (full) size */
#define SG_POOL2_OUT 16 /* sg_conv2d_fwd_wino / sg_wino_output with tile = 4 only: y [B,H/2,W/2,N] (+)= avg_pool2x2(conv + bias) -- the
                         * conv2 -> tf.nn.pool(AVG) pair of a ResNetBlockDown (resnet_ops.py:102-106) without the full-resolution tensor */
#define SG_MMA_BF16 256 /* sg_conv2d_bwd_weight: bf16 matrix-core operands, fp32 accumulation (config c3)  */

/* ---- convolutions: layers.Conv2D stride 1 (bigacgan/resnet_ops.py:65,98,103,109;
 *      net_architecture.py:28-49,283; arch_ops.py:38-65 1x1) ------------------------------- */
/* y = conv(relu?(x), w) + bias + bias2 ; x [B,H,W,Cin], y [B,Ho,Wo,Cout]; pad_same: 1 SAME, 0 VALID */
int sg_conv2d_fwd(const float* x, const float* w, const float* bias, const float* bias2, float* y,
                  int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);
/* dx = conv_data_grad(dy, w) ; result zeroed where mask <= 0 (mask has dx's shape; ReLU backward);
 * SG_ACCUM adds the previous dx AFTER masking (sum of the main and shortcut branches) */
int sg_conv2d_bwd_data(const float* dy, const float* w, const float* mask, float* dx,
                       int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);
/* dw += conv_weight_grad(relu?(x), dy) ; dbias (nullable, Cout > 1) += sum over pixels of dy, fused into the same sweep;
 * sample_scale (nullable, [B]) weights sample b's contribution to dw / dbias by sample_scale[b] (shared backward sweeps) */
int sg_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, const float* sample_scale,
                         int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);

/* data-grad with a pre-transposed filter copy wt [kh,kw,Cout,Cin] = sg_transpose_filter(w, kh*kw, Cin, Cout): same
 * contract and result as sg_conv2d_bwd_data, but the launch uses the forward pass's straight filter loader */
int sg_transpose_filter(const float* w, float* out, int taps, int K, int N, void* stream);
int sg_conv2d_bwd_data_wt(const float* dy, const float* wt, const float* mask, float* dx,
                          int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);

/* ---- bf16 matrix-core variants (BASELINE config c3: bf16 MFMA convs, fp32 accumulation; activations, biases and
 *      results stay fp32).  The filter is passed as a packed bf16 copy [tap][N][K] written by sg_pack_filter_bf16:
 *        forward : pack(w, taps = kh*kw, K = Cin,  N = Cout, transpose = 1)
 *        data-grad: pack(w, taps = kh*kw, K = Cout, N = Cin,  transpose = 0)
 *      Restrictions: K % 8 == 0 and N > 32 (otherwise SG_ERR_UNSUPPORTED: use the fp32 entry point). ------------- */
int sg_pack_filter_bf16(const float* w, void* out, int taps, int K, int N, int transpose, void* stream);
int sg_conv2d_fwd_bf16(const float* x, const void* wp_fwd, const float* bias, const float* bias2, float* y,
                       int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);
int sg_conv2d_bwd_data_bf16(const float* dy, const void* wp_bwd, const float* mask, float* dx,
                            int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);

/* ---- second-generation fp32 convolutions (round 2): the DMA-fed 256-row-tile kernel of conv_bf16v2.hip with fp32 operands
 *      (v_mfma_f32_32x32x2_f32, exact fp32 products, fp32 accumulation: same numerics as sg_conv2d_fwd / _bwd_data up to the
 *      summation order).  wt_fwd = the filter transposed to [tap][Cout][Cin] by sg_transpose_filter(w, taps, K = Cin, N = Cout);
 *      the data-grad takes w [kh,kw,Cin,Cout] itself.  Same contracts and flags as sg_conv2d_fwd / sg_conv2d_bwd_data
 *      (resnet_ops.py:65,98,103,109); SG_ERR_UNSUPPORTED unless reduction channels % 32 == 0 and output channels % 64 == 0. */
int sg_conv2d_fwd_v2(const float* x, const float* wt_fwd, const float* bias, const float* bias2, float* y,
                     int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);
int sg_conv2d_bwd_data_v2(const float* dy, const float* w, const float* mask, float* dx,
                          int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream);

/* ---- Winograd-domain fp32 3x3 convolutions (round 3; conv_winograd.hip): stride-1 SAME 3x3 convolutions with reduction
 *      channels % 32 == 0 and output channels % 128 == 0 as (tile + 2)^2 independent [tiles x K] x [K x N] products between an
 *      input and an output transform.  tile = 2: F(2x2, 3x3), 16 products per 2x2 outputs (16 / 36 of the direct multiplies),
 *      H and W even;  tile = 4: F(4x4, 3x3), 36 products per 4x4 outputs (9 / 36), H % 4 == W % 4 == 0.  Same contracts, flags
 *      and results as sg_conv2d_fwd / sg_conv2d_bwd_data (resnet_ops.py:65,98,103) up to fp32 rounding (tolerances in
 *      tests/test_winograd_gpu.py); SG_ERR_UNSUPPORTED for other shapes (caller: the direct entry points).
 *        sg_wino_filter(w_nk [3][3][N][K], u [(tile+2)^2][N][K], N, K, flip, tile): transformed filter, once per optimizer step --
 *            forward : w_nk = sg_transpose_filter(w, 9, K = Cin, N = Cout) ([tap][Cout][Cin]), flip = 0;
 *            data-grad: w_nk = w [3,3,Cin,Cout] itself (N = Cin, K = Cout),              flip = 1 (taps mirrored);
 *        workspace: sg_wino_workspace_bytes(B, H, W, Cin, Cout, tile) bytes of device memory, caller-provided, contents scratch:
 *            V [P][Tp][K] followed by Mt [P][Tp][N], P = (tile+2)^2, Tp = sg_wino_plane_rows(B, H, W, tile) = B H/tile W/tile
 *            rounded up to 128.
 *      The three steps are also exported one by one (the host side times the HBM-bound transforms apart from the
 *      matrix-bound product): sg_wino_input (x [B,H,W,C] -> V, relu != 0 applies max(.,0) to x), sg_wino_gemm
 *      (Mt[f] = V[f] U[f]^T for all planes in one grouped launch), sg_wino_output (Mt -> y with bias + bias2, ReLU
 *      mask, SG_ACCUM, SG_RELU_OUT). */
long sg_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout, int tile);
long sg_wino_plane_rows(int B, int H, int W, int tile);
int sg_wino_filter(const float* w_nk, float* u, int N, int K, int flip, int tile, void* stream);
int sg_wino_input(const float* x, float* V, int B, int H, int W, int C, int relu, int tile, void* stream);
int sg_wino_input_ups(const float* x_half, float* V, int B, int H, int W, int C, int tile, void* stream);   /* V of 0.25 * upsample2x2(x_half [B,H/2,W/2,C]); tile = 4 */
int sg_wino_gemm(const float* V, const float* u, float* Mt, int B, int H, int W, int K, int N, int tile, void* stream);
int sg_wino_output(const float* Mt, float* y, const float* bias, const float* bias2, const float* mask,
                   int B, int H, int W, int N, int flags, int tile, void* stream);
int sg_conv2d_fwd_wino(const float* x, const float* u_fwd, const float* bias, const float* bias2, float* y,
                       int B, int H, int W, int Cin, int Cout, int flags, int tile, void* workspace, long workspace_bytes, void* stream);
int sg_conv2d_bwd_data_wino(const float* dy, const float* u_bwd, const float* mask, float* dx,
                            int B, int H, int W, int Cin, int Cout, int flags, int tile, void* workspace, long workspace_bytes, void* stream);
/*      Weight gradient in the same domain (contract of sg_conv2d_bwd_weight, resnet_ops.py:65,98,103 under the tape: dw += ,
 *      db += when non-null, sample_scale [B] nullable, SG_RELU_IN on x): dU[f] = V[f]^T Qt[f] over the tiles with
 *      Qt = A dy A^T (sg_wino_grad_input: rows scaled by sample_scale[b], db += their column sums), then dw += G^T dU G
 *      (sg_wino_filter_grad).  sg_wino_wgrad_gemm overwrites dU [P][Cin][Cout]; partial sums of the tile chunks meet through
 *      float atomics (one chunk in deterministic mode).  db_scratch (nullable): 64 x N floats of scratch through which the bias
 *      gradient's column sums are folded (short atomic chains); null: every workgroup adds into db itself.
 *      sg_wino_wgrad_gemm's v_plane_rows: rows between two planes of V (0 = this batch's Tp) -- the forward launch's V, kept by the
 *      host, serves the weight gradient of a batch slice without a second transform sweep.
 *      Workspace: sg_wino_wgrad_workspace_bytes = V | Qt | dU | those 64 x Cout floats. */
long sg_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int tile);
int sg_wino_grad_input(const float* dy, float* Qt, const float* sample_scale, float* db, float* db_scratch,
                       int B, int H, int W, int N, int tile, void* stream);
int sg_wino_grad_input_ups(const float* dy_half, float* Qt, const float* sample_scale, float* db, float* db_scratch, int B, int H, int W,
                           int N, int tile, void* stream);      /* the same for dy = 0.25 * upsample2x2(dy_half [B,H/2,W/2,N]); tile = 4 */
int sg_wino_wgrad_gemm(const float* V, const float* Qt, float* dU, int B, int H, int W, int K, int N, int tile, long v_plane_rows, void* stream);
int sg_wino_filter_grad(const float* dU, float* dw, int K, int N, int tile, void* stream);
int sg_conv2d_bwd_weight_wino(const float* x, const float* dy, float* dw, float* db, const float* sample_scale,
                              int B, int H, int W, int Cin, int Cout, int flags, int tile, void* workspace, long workspace_bytes, void* stream);

/* ---- second-generation bf16 path: bf16 ACTIVATIONS in HBM, operand tiles moved global -> LDS by DMA (round 2).
 *      sg_cvt_bf16: fp32 [n] -> bf16 [n] (round to nearest even), n % 8 == 0; relu != 0 applies max(.,0) first; rowscale
 *      (nullable, [n / rowlen], rowlen % 8 == 0) multiplies row r by rowscale[r] first (the per-sample factors of the shared
 *      backward sweep, data_utils.py:449-468 of the reference run as one sweep).
 *      sg_conv2d_fwd_bf16v2 / sg_conv2d_bwd_data_bf16v2: contracts of sg_conv2d_fwd / sg_conv2d_bwd_data
 *      (resnet_ops.py:65,98,103,109) with the activation operand given as a bf16 NHWC tensor (x16 / dy16), the filter as the
 *      packed copy of sg_pack_filter_bf16, an fp32 result and, when y16 / dx16 is non-null, a bf16 copy of the result
 *      for the next launch (the forward entry points accept y == NULL with y16 != NULL: an operand-only result, no fp32
 *      tensor is written; not with SG_ACCUM); mask16 (nullable) may replace the fp32 ReLU mask by its bf16 copy.  SG_ERR_UNSUPPORTED unless
 *      reduction channels % 64 == 0 and output channels % 64 == 0:
 *      the caller then uses sg_conv2d_fwd_bf16 / sg_conv2d_bwd_data_bf16 on the fp32 tensor. ----------------------- */
int sg_cvt_bf16(const float* x, void* out, long n, int relu, const float* rowscale, long rowlen, void* stream);
/* sg_cvt_bf16 of a [M, C] gradient fused with its bias gradient (resnet_ops.py:65 bias of the conv whose weight-grad
 * follows): out bf16 = rowscale[m / rows_per_sample] * x (rowscale nullable), out_plain (nullable) bf16 = x unscaled (the
 * data-grad operand of the same gradient), dbias[c] += the fp32 column sums of the scaled values.  C % 8 == 0. */
int sg_cvt_bf16_bias(const float* x, void* out, void* out_plain, long M, int C, const float* rowscale, long rows_per_sample,
                     float* dbias, void* stream);
/* amax_y / amax_dx (nullable, 2 floats, zeroed by the caller; config c5): the epilogue also takes max |result| (element 0) and
 * max |amax_rowscale[b] * result| (element 1; amax_rowscale nullable [B] = no factor) with atomicMax while it writes the result:
 * the per-tensor scales of the fp8 launches that read this result next, without a separate amax sweep over it. */
int sg_conv2d_fwd_bf16v2(const void* x16, const void* wp_fwd, const float* bias, const float* bias2, float* y, void* y16,
                         int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, float* amax_y, void* stream);
int sg_conv2d_bwd_data_bf16v2(const void* dy16, const void* wp_bwd, const float* mask, const void* mask16, float* dx, void* dx16,
                              int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, float* amax_dx,
                              const float* amax_rowscale, void* stream);
/* dw [kh,kw,Cin,Cout] (fp32) += weight gradient of the SAME stride-1 convolution from bf16 operands x16 [B,H,W,Cin] and
 * dy16 [B,H,W,Cout] (tape of d_loss / s_loss / g_final, data_utils.py:449-468; per-sample factors already folded into dy16 by
 * sg_cvt_bf16).  flags: SG_RELU_IN on x16.  No bias gradient (sg_bias_grad).  SG_ERR_UNSUPPORTED unless (Cin % 64 == 0 and
 * Cout % 256 == 0) or Cin == Cout == 64: the caller then uses sg_conv2d_bwd_weight. */
int sg_conv2d_bwd_weight_bf16v2(const void* x16, const void* dy16, float* dw, int B, int H, int W, int Cin, int Cout,
                                int kh, int kw, int pad_same, int flags, void* stream);
/* bf16 weight gradient of layers.Conv2DTranspose (resnet_ops.py:57,69; contract of sg_conv2d_transpose_bwd_weight with bf16
 * operand copies): x16 [B,H,W,Cin], dy16 [B,sh*H,sw*W,Cout], dw fp32 [kh,kw,Cout,Cin] +=.  (Cout, Cin) % (256,256), (64,256) or
 * (64,64) == 0 and H*W >= 64, else SG_ERR_UNSUPPORTED. */
int sg_conv2d_transpose_bwd_weight_bf16v2(const void* x16, const void* dy16, float* dw, int B, int H, int W, int Cin, int Cout,
                                          int kh, int kw, int sh, int sw, void* stream);

/* ---- fp8 (OCP e4m3) operands for the forward / data-grad convolutions (BASELINE config c5): per-tensor
 *      scaling operand8 = e4m3(value * 448 / amax), amax = max|tensor| kept as a DEVICE scalar (no host sync), fp32
 *      accumulation on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales, result = sums * amax_a * amax_w / 448^2.
 *      sg_amax_f32: amax[0] = max(amax[0], max|x|) (zero it first; n % 4 == 0).  sg_cvt_fp8: fp32 -> fp8 with that scale,
 *      relu != 0 applies max(.,0) first (n % 8 == 0).  sg_pack_filter_fp8: as sg_pack_filter_bf16, bytes instead of bf16.
 *      sg_conv2d_fwd_fp8 / sg_conv2d_bwd_data_fp8: contracts of the bf16v2 entry points (resnet_ops.py:98,103 call sites
 *      of the D-shaped trunks); SG_RELU_IN is not accepted (fold it into sg_cvt_fp8); SG_ERR_UNSUPPORTED unless reduction
 *      channels % 128 == 0 and output channels % 256 == 0.  amax_y / amax_dx / amax_rowscale: as for the bf16v2 entry points
 *      (the amax of the result for the NEXT fp8 launch, taken in the epilogue). ------------------------------------ */
int sg_amax_f32(const float* x, long n, float* amax, void* stream);
int sg_cvt_fp8(const float* x, void* out, long n, int relu, const float* amax, void* stream);
/* the same conversion from a bf16 tensor: the fp8 operand of a conv whose input is an OPERAND-ONLY result of the previous conv
 * (sg_conv2d_fwd_bf16v2 / _fp8 with y == NULL and y16 != NULL: conv1 -> conv2 of a ResNetBlockDown, resnet_ops.py:97-104 -- the
 * activation between two chained convolutions exists as bf16 only in configs c3 / c5).  n % 8 == 0. */
int sg_cvt_fp8_bf16(const void* x16, void* out, long n, int relu, const float* amax, void* stream);
int sg_pack_filter_fp8(const float* w, void* out, const float* amax, int taps, int K, int N, int transpose, void* stream);
int sg_conv2d_fwd_fp8(const void* x8, const float* amax_x, const void* wp8, const float* amax_w, const float* bias, const float* bias2,
                      float* y, void* y16, int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, float* amax_y,
                      void* stream);
int sg_conv2d_bwd_data_fp8(const void* dy8, const float* amax_dy, const void* wp8, const float* amax_w, const float* mask,
                           const void* mask16, float* dx, void* dx16, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                           int pad_same, int flags, float* amax_dx, const float* amax_rowscale, void* stream);
/* fp8 weight gradient (config c5; resnet_ops.py:98,103,109 call sites of the D-shaped trunks, reverse pass of the tapes at
 * data_utils.py:449-468): dw fp32 [kh,kw,Cin,Cout] += sum over pixels of x8 (e4m3, the forward launch's operand copy, ReLU
 * folded in) x dy8 (OCP e5m2 = e5m2(clamp(rowscale * dy * 57344 / amax_dy)), per-sample factors folded in) times
 * amax_x amax_dy / (448 * 57344), on v_mfma_scale_f32_32x32x64_f8f6f4 (A e4m3, B e5m2), fp32 accumulation.  SAME stride-1
 * or 1x1 convolutions with Cin % 256 == 0 and Cout % 256 == 0, else SG_ERR_UNSUPPORTED (caller: the bf16 entry point).
 * sg_amax2_f32: amax2[0] = max(amax2[0], max |x|), amax2[1] = max(amax2[1], max |rowscale[i / rowlen] x_i|) in ONE read
 * sweep (zero amax2 first; n % 4 == 0).  sg_cvt_fp8_grad: one sweep over a gradient [M, C] -> the e5m2 weight-grad
 * operand (scaled rows), the e4m3 data-grad operand (nullable; unscaled rows, scale 448 / amax2[0]) and dbias [C] +=
 * fp32 column sums of the scaled values (nullable) -- the fp8 counterpart of sg_cvt_bf16_bias. */
int sg_conv2d_bwd_weight_fp8(const void* x8, const float* amax_x, const void* dy8, const float* amax_dy, float* dw, int B, int H,
                             int W, int Cin, int Cout, int kh, int kw, int pad_same, void* stream);
int sg_amax2_f32(const float* x, long n, const float* rowscale, long rowlen, float* amax2, void* stream);
int sg_cvt_fp8_grad(const float* x, void* out_e5m2, void* out_e4m3, long M, int C, const float* rowscale, long rows_per_sample,
                    const float* amax2, float* dbias, void* stream);

/* bf16 variants of the transposed convolution (w [kh,kw,Cout,Cin]): forward wp = pack(w, kh*kw, K = Cin, N = Cout,
 * transpose = 0); data-grad wp = pack(w, kh*kw, K = Cout, N = Cin, transpose = 1).  SG_ERR_UNSUPPORTED for a stride /
 * kernel combination with a tap-less parity class (1x1, stride 2) and for K % 8 != 0 or N <= 32: use the fp32 entry. */
int sg_conv2d_transpose_fwd_bf16(const float* x, const void* wp, const float* bias, const float* bias2, float* y,
                                 int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int flags, void* stream);
int sg_conv2d_transpose_bwd_data_bf16(const float* dy, const void* wp, const float* mask, float* dx,
                                      int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int flags, void* stream);

/* ---- layers.Conv2DTranspose(padding='same', strides=(sh,sw)) (resnet_ops.py:57,69) ------- */
/* x [B,H,W,Cin] -> y [B,sh*H,sw*W,Cout]; k=3,s=2: y[2i+k] += x[i] w[k] cropped to 2n; bias everywhere */
int sg_conv2d_transpose_fwd(const float* x, const float* w, const float* bias, const float* bias2, float* y,
                            int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int flags, void* stream);
int sg_conv2d_transpose_bwd_data(const float* dy, const float* w, const float* mask, float* dx,
                                 int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int flags, void* stream);
int sg_conv2d_transpose_bwd_weight(const float* x, const float* dy, float* dw,
                                   int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int flags, void* stream);

/* db[n] += sum_m dy[m,n]  (bias gradient of any of the above; M = B*Ho*Wo rows) */
int sg_bias_grad(const float* dy, float* db, long M, int N, void* stream);

/* ---- pooling / elementwise ------------------------------------------------------------------- */
/* out = avgpool2x2(a) + avgpool2x2(b) (b may be null): tf.nn.pool AVG SAME s2 + `net += shortcut`
 * (resnet_ops.py:105-114) */
int sg_avgpool2_add_fwd(const float* a, const float* b, float* out, int B, int H, int W, int C, void* stream);
int sg_avgpool2_bwd(const float* dout, float* dx, int B, int H, int W, int C, void* stream);   /* H,W of dx */
/* The same backward (resnet_ops.py:105-106 under the tapes of data_utils.py:449-468) writing the bf16 / fp8 OPERAND COPIES of dx
 * directly instead of dx itself (configs c3 / c5: dx feeds only conv2's weight-grad and data-grad launches, which read operand
 * copies): dx16 = bf16(0.25 dout[b,y/2,x/2,c]) and dx16_scaled = bf16(rowscale[b] * that) (each nullable, rowscale nullable);
 * fp8: dx_e4m3 and dx_e5m2 (rowscale folded in; each nullable) with per-tensor scales 448 / amax, 57344 / amax, where
 * amax_dout = {max |dout|, max |rowscale dout|} from sg_amax2_f32 and amax_dx[0..1] receives 0.25 * those.  Bit-identical
 * to sg_avgpool2_bwd followed by sg_cvt_bf16 / sg_cvt_fp8_grad.  C % 8 == 0. */
int sg_avgpool2_bwd_bf16(const float* dout, void* dx16, void* dx16_scaled, const float* rowscale, int B, int H, int W, int C, void* stream);
int sg_avgpool2_bwd_fp8(const float* dout, void* dx_e4m3, void* dx_e5m2, const float* rowscale, const float* amax_dout, float* amax_dx,
                        int B, int H, int W, int C, void* stream);
int sg_add(const float* a, const float* b, float* out, long n, void* stream);
int sg_relu_mask(const float* dy, const float* ref, float* dx, long n, void* stream);          /* dx = ref>0 ? dy : 0 */
int sg_tanh_bwd(const float* y, const float* dy, float* dx, long n, void* stream);             /* net_architecture.py:289 */
/* layers.MaxPool2D(pool_size=(ph,pw)) (arch_ops.py:47,58; net_architecture.py:29,32,38,47) */
int sg_maxpool_fwd(const float* x, float* y, unsigned char* idx, int B, int H, int W, int C, int ph, int pw, void* stream);
int sg_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int ph, int pw, int accum, void* stream);
/* tf.nn.relu + GlobalAveragePooling2D (net_architecture.py:249-250,340-341,399-400) */
int sg_gap_fwd(const float* x, float* out, int B, int HW, int C, int relu, void* stream);
int sg_gap_bwd(const float* dout, const float* x, float* dx, int B, int HW, int C, int relu, void* stream);
/* NonLocalBlock residual `sigma * attn_g + input` (arch_ops.py:67) and its pieces */
int sg_scale_add(const float* o, const float* x, const float* sigma, float* out, long n, void* stream);
int sg_scale(const float* a, const float* s, float* out, long n, void* stream);
int sg_dot_accum(const float* a, const float* b, float* out, long n, void* stream);            /* out[0] += a.b */
int sg_rowscale(const float* x, const float* s, float* out, long rows, int rowlen, void* stream);

/* ---- Dense layers (resnet_ops.py:18,24; net_architecture.py:55,251,342,401) ---------------- */
/* C = alpha*op(A)*op(B) + beta*C (+bias[n]); op(A) is MxK, op(B) is KxN; row-major with leading dims */
int sg_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb, int ldc,
            int transA, int transB, float alpha, float beta, void* stream);

/* ---- BatchNormalization / ConditionalBatchNorm (resnet_ops.py:13-28; net_architecture.py:42,46,281) */
long sg_bn_stats_workspace_floats(long M, int C);
/* sums[0:C] = sum x, sums[C:2C] = sum x^2 over the M rows (fp64; all-reduce these for SyncBN) */
int sg_bn_stats_sums(const float* x, long M, int C, float* workspace, double* sums, void* stream);
int sg_bn_stats_finalize(const double* sums, double count, float* mean, float* var, int C, void* stream);
/* y = (x-mean)*rsqrt(var+eps)*gamma[b*gstride+c] + beta[b*gstride+c] (+ReLU); gstride=C: per-sample (CBN), 0: per-channel */
int sg_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, int gstride,
                float* y, int B, int HW, int C, float eps, int relu, void* stream);
/* dgamma/dbeta [B,C] += per-sample sums (caller zeroes); chan (fp64 [4C]): sum dxhat, sum dxhat*xhat, sum_b dgamma, sum_b dbeta;
 * dgamma_c/dbeta_c [C] (nullable) += sum_b dgamma/dbeta: the gradients of a per-channel gamma/beta */
int sg_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* var, const float* gamma,
                     int gstride, float* dgamma, float* dbeta, double* chan, float* dgamma_c, float* dbeta_c, int B, int HW, int C,
                     float eps, int relu, void* stream);
int sg_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* var, const float* gamma,
                    int gstride, const double* chan, double count, float* dx, int B, int HW, int C, float eps, int relu,
                    int use_stats, void* stream);
int sg_bn_update_moving(float* mm, float* mv, const float* mean, const float* var, double count, float momentum, int C, void* stream);

/* ---- SpatialEmbedding filter bank + z0 contraction + seed layout (arch_ops.py:77-95;
 *      net_architecture.py:229-231,259-271).  z [B,128] (z0 = z[:, :32]); y int32 [B,L];
 *      table [vocab,32,8192]; seed [B,4,4L,512] ------------------------------------------------ */
int sg_filterbank_fwd(const float* z, const int* y, const float* table, float* seed, int B, int L, int vocab, void* stream);
int sg_filterbank_bwd(const float* z, const int* y, const float* table, const float* dseed, float* dtable, float* dz,
                      int B, int L, int vocab, void* stream);

/* ---- NonLocalBlock attention core (arch_ops.py:51-52,61): out = softmax(theta phi^T) g ------ */
int sg_attention_fwd(const float* theta, const float* phi, const float* g, float* out, float* lse,
                     int B, int Nq, int Nk, int dk, int dv, void* stream);
int sg_attention_bwd(const float* theta, const float* phi, const float* g, const float* out, const float* lse, const float* dout,
                     float* dtheta, float* dphi, float* dg, float* delta_scratch, int B, int Nq, int Nk, int dk, int dv, void* stream);

/* ---- recognizer tail: softmax + K.ctc_batch_cost (net_architecture.py:55-64) ----------------
 * logits [B,T,C] (pre-softmax Dense output), labels int32 [B,label_stride], blank = C-1.
 * loss [B]; dlogits [B,T,C] = d loss_b / d logits_b, or null for forward only */
int sg_softmax_ctc(const float* logits, const int* labels, int label_stride, float* loss, float* dlogits,
                   int B, int T, int C, int input_length, int label_length, void* stream);

/* ---- losses + gradient balancing + statistics (net_loss.py:4-54; data_utils.py:418-442,476-490) */
/* mode 0 hinge, 1 not_saturating.  sums: fp64 [12] (all-reduce across ranks between the two calls) */
int sg_loss_sums(const float* d_r, const float* d_f, const float* s_my, const float* s_f, const float* s_r, const float* r_f,
                 const float* r_r, int B, int mode, double* sums, void* stream);
int sg_loss_grads(const float* d_r, const float* d_f, const float* s_my, const float* s_f, const float* s_r, const float* r_f,
                  int B, int mode, int balance, float alpha, const double* sums, float* scalars16, float* gD_r, float* gD_f,
                  float* gS_my, float* gS_f, float* gG_d, float* gG_s, float* gG_r, float* shD3, float* shS3, void* stream);
/* shD3 / shS3 (nullable, [3,B]): upstream u, weight-gradient scale gD_f/u and image-gradient scale gG_d/u of the ONE shared
 * backward sweep through D(x_f) / S(x_f) that serves both sum(d_loss) and sum(g_final) (backprop is linear per sample) */

/* out7 [7,B] = d_loss, d_loss_real, d_loss_fake, g_loss, s_loss, s_a, s_b per sample (the 7 tensors loss_fn returns) */
int sg_loss_terms(const float* d_r, const float* d_f, const float* s_my, const float* s_f, const float* s_r, int B, int mode,
                  float* out7, void* stream);

/* ---- make_my_recognizer extras (net_architecture.py:82-179): LeakyReLU(0.01) :104, Dropout masks :112-153,
 *      Bidirectional(LSTM(256, dropout=0.5)) :146-150 (Keras gate order i,f,c~,o; sigmoid/tanh) ------------- */
int sg_leaky_relu_fwd(const float* x, float* y, long n, float alpha, void* stream);
int sg_leaky_relu_bwd(const float* dy, const float* x, float* dx, long n, float alpha, void* stream);
/* out[r,c] = x[r,c] * mask[r / rows_per_mask, c]  (rows_per_mask = T: one input-dropout mask per sample, all timesteps) */
int sg_mul_mask(const float* x, const float* mask, float* out, long rows, int cols, int rows_per_mask, void* stream);
/* z [B,4H] (row stride ldz): pre-activations in, gate activations out; c = f c_prev + i c~; h = o tanh(c) */
int sg_lstm_cell_fwd(float* z, int ldz, const float* c_prev, float* c_out, float* h_out, int ldh, float* h_copy, int ldc,
                     int B, int H, void* stream);
/* gates [B,4H]: activations in, d(pre-activations) out; dh = dh_a (stride lda) + dh_b; dc_prev out */
int sg_lstm_cell_bwd(float* gates, int ldz, const float* c_prev, const float* c_t, const float* dh_a, int lda, const float* dh_b,
                     const float* dc_next, float* dc_prev, int B, int H, void* stream);

/* ---- optimizers (main.py:27-33; Keras Adam / RMSprop) and spectral_norm (arch_ops.py:98-126) */
int sg_adam_update(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta_1, float beta_2, float eps, void* stream);
/* sg_adam_update with the bias-corrected step size lr_t read from device memory (lr_t_dev[0]): a step captured into a HIP graph
 * (scrabble_gan_amd/graph_step.py) replays its launches with the arguments of the capture, and lr_t changes every step */
int sg_adam_update_dlr(float* p, const float* g, float* m, float* v, long n, const float* lr_t_dev, float beta_1, float beta_2, float eps,
                       void* stream);
int sg_rmsprop_update(float* p, const float* g, float* ms, long n, float lr, float rho, float eps, void* stream);
long sg_spectral_norm_workspace_floats(int K, int N);
int sg_spectral_norm(const float* w, const float* u, float* out, float* workspace, int K, int N, int power_iteration, void* stream);
/* backward of spectral_norm(w, u, power_iteration = 1) (arch_ops.py:107-126, no stop-gradient): dw [K,N] += d/dw given
 * g = gradient w.r.t. the normalised weight; used by kernel_reg = 'applied' (SURVEY Appendix C-3). */
long sg_spectral_norm_bwd_workspace_floats(int K, int N);
int sg_spectral_norm_bwd(const float* w, const float* u, const float* g, float* dw, float* workspace, int K, int N, void* stream);

/* ---- host-pipeline helpers ------------------------------------------------------------------------------
 * sg_normalize_u8: out[i] = (float(u8[i]) - 127.5) / 127.5, the pixel normalisation of load_prepare_data (data_utils.py:82)
 * run on the GPU on bytes that arrived through a pinned staging buffer; bit-identical to the numpy expression; n % 16 == 0.
 * sg_bias_add: y[m, c] += bias[c] (C % 4 == 0), the bias of the strided Conv2D layers of make_my_discriminator
 * (net_architecture.py:425-443), whose contractions run on the transposed-convolution kernels. */
int sg_normalize_u8(const unsigned char* u8, float* out, long n, void* stream);
int sg_bias_add(float* y, const float* bias, long M, int C, void* stream);

/* ---- collective of the data-parallel step (the reference has none: BASELINE.json north_star) ----------
 * SUM all-reduce, in place, of `n` elements of a flat device buffer over RCCL on `stream`; gradients are sums over the
 * batch (reference data_utils.py:450,454,458,467 differentiate [B,1] targets), so ranks ADD.  comm = the ncclComm_t made by
 * sg_rccl_comm_init_rank (or any RCCL communicator of the caller).  dtype: the enum below.  Asynchronous on `stream`.
 * SG_ERR_UNSUPPORTED when librccl cannot be resolved or the dtype is not reducible. */
#define SG_DTYPE_F32 0
#define SG_DTYPE_BF16 1
#define SG_DTYPE_FP8_E4M3 2
#define SG_DTYPE_F64 3
int sg_allreduce_sum(void* buf, long n, int dtype, void* comm, void* stream);
int sg_rccl_unique_id(void* id128);                                             /* 128 bytes, created by one rank */
int sg_rccl_comm_init_rank(void** comm, int nranks, const void* id128, int rank);
int sg_rccl_comm_destroy(void* comm);

/* ---- reproducibility (the reference's GradientTapes are deterministic on the TF CPU path, data_utils.py:449-468) ----------
 * sg_set_deterministic(1): process-wide switch.  Every convolution launch (first- and second-generation kernels, fp32 /
 * bf16 / fp8) then runs ONE workgroup per output tile -- no reduction splits, no float atomics in forward / data-grad --
 * so a sample's activations do not depend on the batch size it is launched in.  sg_set_deterministic(0) restores the
 * CU-quantum tail split.  Returns the previous setting. */
int sg_set_deterministic(int on);

/* ---- self-test of the status convention -------------------------------------------------------------------------------
 * Every entry point returns SG_ERR_LAUNCH if ANY of the launches it queued failed, not only the last one.
 * sg_selftest_launch_status queues: a valid launch, a launch with an impossible block size (2048 threads), a valid
 * launch -- and returns what an entry point would: SG_ERR_LAUNCH.  scratch: >= 256 floats (overwritten). */
int sg_selftest_launch_status(float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif
