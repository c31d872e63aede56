// Optimizer updates, spectral-norm power iteration and the loss head of the train step.
//   Adam / RMSprop : tf.keras.optimizers.{Adam,RMSprop} (/root/reference/src/main.py:27-33) on the flat
//                    per-network parameter / gradient / slot buffers (28 B of HBM traffic per weight)
//   spectral_norm  : /root/reference/src/bigacgan/arch_ops.py:98-126 (one power iteration, fresh u)
//   loss head      : net_loss.py:4-54 + apply_gradient_balancing (data_utils.py:476-490) + the 13
//                    reduce_mean / 2 reduce_std of data_utils.py:430-442, and the per-sample upstream
//                    gradients of the four tape.gradient calls (data_utils.py:450-467)
#include "sg_common.h"

#define F4(p) (*reinterpret_cast<float4*>(p))
#define CF4(p) (*reinterpret_cast<const float4*>(p))

// Keras Adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m / (sqrt(v) + eps)
// lr_dev (nullable): the step size lr_t read from device memory instead of the kernel argument -- a captured HIP graph replays
// the launch with the arguments it was captured with, and lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) changes every step
__global__ __launch_bounds__(256) void k_adam(float* p, const float* g, float* m, float* v, long n, float lr_arg, const float* lr_dev, float b1,
                                              float b2, float eps) {
  const float lr_t = lr_dev ? lr_dev[0] : lr_arg;
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride) {
    const float4 gv = CF4(g + 4 * e);
    float4 mv = CF4(m + 4 * e), vv = CF4(v + 4 * e), pv = CF4(p + 4 * e);
    mv.x = b1 * mv.x + (1.f - b1) * gv.x; mv.y = b1 * mv.y + (1.f - b1) * gv.y;
    mv.z = b1 * mv.z + (1.f - b1) * gv.z; mv.w = b1 * mv.w + (1.f - b1) * gv.w;
    vv.x = b2 * vv.x + (1.f - b2) * gv.x * gv.x; vv.y = b2 * vv.y + (1.f - b2) * gv.y * gv.y;
    vv.z = b2 * vv.z + (1.f - b2) * gv.z * gv.z; vv.w = b2 * vv.w + (1.f - b2) * gv.w * gv.w;
    pv.x -= lr_t * mv.x / (sqrtf(vv.x) + eps); pv.y -= lr_t * mv.y / (sqrtf(vv.y) + eps);
    pv.z -= lr_t * mv.z / (sqrtf(vv.z) + eps); pv.w -= lr_t * mv.w / (sqrtf(vv.w) + eps);
    F4(m + 4 * e) = mv; F4(v + 4 * e) = vv; F4(p + 4 * e) = pv;
  }
  for (long e = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const float gv = g[e];
    const float mv = b1 * m[e] + (1.f - b1) * gv, vv = b2 * v[e] + (1.f - b2) * gv * gv;
    m[e] = mv; v[e] = vv;
    p[e] -= lr_t * mv / (sqrtf(vv) + eps);
  }
}

// Keras RMSprop, TF 2.1 non-fused dense path (momentum 0, not centered; rmsprop.py _resource_apply_dense):
//   ms = rho ms + (1-rho) g^2 ; p -= lr g / (sqrt(ms) + eps)      -- eps OUTSIDE the root
__global__ __launch_bounds__(256) void k_rmsprop(float* p, const float* g, float* ms, long n, float lr, float rho, float eps) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float gv = g[e];
    const float s = rho * ms[e] + (1.f - rho) * gv * gv;
    ms[e] = s;
    p[e] -= lr * gv / (sqrtf(s) + eps);
  }
}

extern "C" int sg_adam_update(float* p, const float* g, float* m, float* v, long n, float lr_t, float beta_1, float beta_2,
                              float eps, void* stream) {
  if (!p || !g || !m || !v) return SG_ERR_ARG;
  if (n <= 0) return SG_OK;
  SG_KERNEL(k_adam, dim3(sg_grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, (const float*)nullptr, beta_1,
            beta_2, eps);
  return sg_launch_status();
}

// the same update with lr_t taken from device memory (lr_t_dev[0]): for steps replayed from a captured HIP graph
extern "C" int sg_adam_update_dlr(float* p, const float* g, float* m, float* v, long n, const float* lr_t_dev, float beta_1, float beta_2,
                                  float eps, void* stream) {
  if (!p || !g || !m || !v || !lr_t_dev) return SG_ERR_ARG;
  if (n <= 0) return SG_OK;
  SG_KERNEL(k_adam, dim3(sg_grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f, lr_t_dev, beta_1, beta_2, eps);
  return sg_launch_status();
}

extern "C" int sg_rmsprop_update(float* p, const float* g, float* ms, long n, float lr, float rho, float eps, void* stream) {
  if (!p || !g || !ms) return SG_ERR_ARG;
  if (n <= 0) return SG_OK;
  SG_KERNEL(k_rmsprop, dim3(sg_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, ms, n, lr, rho, eps);
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
// spectral norm: W2 = reshape(w, [K, N]);  ws = [v(K) | u_(N) | nv2 | nu2] floats
// ------------------------------------------------------------------------------------------
// v[k] = sum_n u[n] * rs_u * W[k,n] ; nv2 += v[k]^2      (one wave per row, rs_u = 1 or rsqrt(max(|u_|^2,1e-12)))
__global__ __launch_bounds__(256) void k_sn_rows(const float* w, const float* u, const float* u_norm2, float* v, float* nv2, int K, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= K) return;
  const float rs = u_norm2 ? rsqrtf(fmaxf(u_norm2[0], 1e-12f)) : 1.f;
  float s = 0.f;
  for (int n = lane; n < N; n += 64) s += u[n] * w[(size_t)row * N + n];
  s = sg_wave_sum(s) * rs;
  if (lane == 0) {
    v[row] = s;
    atomicAdd(nv2, s * s);
  }
}
// u_[n] += sum_{k in slab} v[k] * rsqrt(max(nv2,1e-12)) * W[k,n]
__global__ __launch_bounds__(256) void k_sn_cols(const float* w, const float* v, const float* nv2, float* u_, int K, int N, int rows_per_block) {
  const float rs = rsqrtf(fmaxf(nv2[0], 1e-12f));
  const int k0 = blockIdx.x * rows_per_block, k1 = min(K, k0 + rows_per_block);
  for (int n = threadIdx.x; n < N; n += 256) {
    float s = 0.f;
    for (int k = k0; k < k1; ++k) s += v[k] * w[(size_t)k * N + n];
    atomicAdd(u_ + n, s * rs);
  }
}
__global__ __launch_bounds__(256) void k_sn_norm2(const float* x, float* out, int n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  s = sg_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}
// sigma = (v_hat W) u_hat^T = |u_|^2 * rsqrt(max(|u_|^2,1e-12)) ; out = w / sigma
__global__ __launch_bounds__(256) void k_sn_scale(const float* w, const float* nu2, float* out, long n) {
  const float a = nu2[0];
  const float inv = 1.f / (a * rsqrtf(fmaxf(a, 1e-12f)));
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) out[e] = w[e] * inv;
}

extern "C" long sg_spectral_norm_workspace_floats(int K, int N) { return (long)K + N + 4; }

// w, out: [K, N] row-major (K = prod of the leading kernel dims); u: the N(0,1) draw of arch_ops.py:110
extern "C" int sg_spectral_norm(const float* w, const float* u, float* out, float* workspace, int K, int N, int power_iteration,
                                void* stream) {
  if (!w || !u || !out || !workspace || K < 1 || N < 1 || power_iteration < 1) return SG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* v = workspace;
  float* u_ = workspace + K;
  float* nv2 = workspace + K + N;
  float* nu2 = nv2 + 1;
  const float* u_cur = u;
  const float* u_n2 = nullptr;
  for (int it = 0; it < power_iteration; ++it) {
    (void)hipMemsetAsync(nv2, 0, sizeof(float), s);
    SG_KERNEL(k_sn_rows, dim3(sg_cdiv(K, 4)), dim3(256), 0, s, w, u_cur, u_n2, v, nv2, K, N);
    // u_cur may alias u_ on later iterations: k_sn_rows has consumed it before the memset below (stream order)
    (void)hipMemsetAsync(u_, 0, sizeof(float) * N, s);
    const int rpb = 64;
    SG_KERNEL(k_sn_cols, dim3(sg_cdiv(K, rpb)), dim3(256), 0, s, w, v, nv2, u_, K, N, rpb);
    SG_KERNEL(k_sn_norm2, dim3(1), dim3(256), 0, s, u_, nu2, N);
    u_cur = u_;
    u_n2 = nu2;
  }
  SG_KERNEL(k_sn_scale, dim3(sg_grid_for((long)K * N, 256)), dim3(256), 0, s, w, nu2, out, (long)K * N);
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
// spectral norm, backward (kernel_reg 'applied' mode: the forward convolves with w~ = w / sigma(w, u)).  One power iteration:
//   a = W u,  v^ = a / |a|,  b = v^ W,  sigma = |b|,  u^ = b / |b|            (arch_ops.py:110-121; no stop-gradient)
//   d sigma / dW = v^ u^T + g_a u^T... precisely  v^[k] u^[n] + g_a[k] u[n],   g_a = (t - (t . v^) v^) / |a|,  t = W u^
//   dW += G / sigma - (<G, W> / sigma^2) * d sigma / dW                      (G = gradient w.r.t. w~)
// ws = [a(K) | b(N) | t(K) | na2 | nb2 | nt_unused | s | c]   (floats)
// ------------------------------------------------------------------------------------------
// one wave per row: t[k] = sum_n W[k,n] b[n] rs_b ; c += t[k] a[k] rs_a ; s += sum_n G[k,n] W[k,n]
__global__ __launch_bounds__(256) void k_snb_rows(const float* w, const float* g, const float* a, const float* b, const float* na2,
                                                  const float* nb2, float* t, float* c, float* sdot, int K, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= K) return;
  const float rs_b = rsqrtf(fmaxf(nb2[0], 1e-12f)), rs_a = rsqrtf(fmaxf(na2[0], 1e-12f));
  float tt = 0.f, ss = 0.f;
  for (int n = lane; n < N; n += 64) {
    const float wv = w[(size_t)row * N + n];
    tt += wv * b[n];
    ss += wv * g[(size_t)row * N + n];
  }
  tt = sg_wave_sum(tt) * rs_b;
  ss = sg_wave_sum(ss);
  if (lane == 0) {
    t[row] = tt;
    atomicAdd(c, tt * a[row] * rs_a);
    atomicAdd(sdot, ss);
  }
}
__global__ __launch_bounds__(256) void k_snb_apply(const float* g, const float* u, const float* a, const float* b, const float* t,
                                                   const float* na2, const float* nb2, const float* c, const float* sdot, float* dw, int K, int N) {
  const float nb = nb2[0], na = na2[0];
  const float rs_b = rsqrtf(fmaxf(nb, 1e-12f)), rs_a = rsqrtf(fmaxf(na, 1e-12f));
  const float sigma = nb * rs_b;                 // (v^ W) u^T = |b|^2 rsqrt(max(|b|^2, 1e-12))
  const float inv = 1.f / sigma;
  const float coef = sdot[0] * inv * inv;
  const float cc = c[0];
  const long total = (long)K * N;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / N), n = (int)(e - (long)k * N);
    const float vh = a[k] * rs_a;
    const float ga = (t[k] - cc * vh) * rs_a;
    dw[e] += g[e] * inv - coef * (vh * b[n] * rs_b + ga * u[n]);
  }
}

extern "C" long sg_spectral_norm_bwd_workspace_floats(int K, int N) { return 2L * K + N + 8; }

// dw [K,N] += gradient w.r.t. w of  w~ = spectral_norm(w, u, power_iteration = 1)  given g = gradient w.r.t. w~
extern "C" int sg_spectral_norm_bwd(const float* w, const float* u, const float* g, float* dw, float* workspace, int K, int N, void* stream) {
  if (!w || !u || !g || !dw || !workspace || K < 1 || N < 1) return SG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* a = workspace;
  float* b = workspace + K;
  float* t = workspace + K + N;
  float* na2 = workspace + 2L * K + N;
  float* nb2 = na2 + 1;
  float* sdot = na2 + 3;
  float* c = na2 + 4;
  (void)hipMemsetAsync(b, 0, sizeof(float) * N, s);
  (void)hipMemsetAsync(na2, 0, sizeof(float) * 8, s);
  SG_KERNEL(k_sn_rows, dim3(sg_cdiv(K, 4)), dim3(256), 0, s, w, u, (const float*)nullptr, a, na2, K, N);
  SG_KERNEL(k_sn_cols, dim3(sg_cdiv(K, 64)), dim3(256), 0, s, w, a, na2, b, K, N, 64);
  SG_KERNEL(k_sn_norm2, dim3(1), dim3(256), 0, s, b, nb2, N);
  SG_KERNEL(k_snb_rows, dim3(sg_cdiv(K, 4)), dim3(256), 0, s, w, g, a, b, na2, nb2, t, c, sdot, K, N);
  SG_KERNEL(k_snb_apply, dim3(sg_grid_for((long)K * N, 256)), dim3(256), 0, s, g, u, a, b, t, na2, nb2, c, sdot, dw, K, N);
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
// loss head
// ------------------------------------------------------------------------------------------
// sums layout (fp64): 0 d_loss 1 d_real 2 d_fake 3 g_loss 4 s_loss 5 s_a 6 s_b 7 r_f 8 r_r 9 g^2 10 r_f^2 11 count
#define LH_NSUM 12

__device__ __forceinline__ float sce(float x, float z) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// per-sample losses; mode 0 = hinge (5th argument ignored), 1 = not_saturating with the call-site
// argument order of data_utils.py:418: (d_real, d_fake, s_style, s_fake, s_real)
__device__ __forceinline__ void losses(int mode, float d_r, float d_f, float s_my, float s_f, float s_r, float& d_lr, float& d_lf,
                                       float& g, float& s_a, float& s_b) {
  if (mode == 0) {
    d_lr = fmaxf(1.f - d_r, 0.f); d_lf = fmaxf(1.f + d_f, 0.f);
    s_a = fmaxf(1.f - s_my, 0.f); s_b = fmaxf(1.f + s_f, 0.f);
    g = -(d_f + s_f);
  } else {
    d_lr = sce(d_r, 1.f); d_lf = sce(d_f, 0.f);
    s_a = sce(s_my, 1.f); s_b = sce(s_f, 0.f);     // s_trainingimgs := S(G(z))
    g = sce(d_f, 1.f) + sce(s_r, 1.f);             // s_fake := S(real images)
  }
}

__global__ __launch_bounds__(256) void k_loss_sums(const float* d_r, const float* d_f, const float* s_my, const float* s_f,
                                                   const float* s_r, const float* r_f, const float* r_r, int B, int mode, double* sums) {
  __shared__ double red[4][LH_NSUM];
  double a[LH_NSUM];
#pragma unroll
  for (int i = 0; i < LH_NSUM; ++i) a[i] = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    float d_lr, d_lf, g, s_a, s_b;
    losses(mode, d_r[b], d_f[b], s_my[b], s_f[b], s_r[b], d_lr, d_lf, g, s_a, s_b);
    a[0] += (double)(d_lr + d_lf); a[1] += d_lr; a[2] += d_lf; a[3] += g; a[4] += (double)(s_a + s_b); a[5] += s_a; a[6] += s_b;
    a[7] += r_f[b]; a[8] += r_r[b]; a[9] += (double)g * g; a[10] += (double)r_f[b] * r_f[b]; a[11] += 1.0;
  }
#pragma unroll
  for (int i = 0; i < LH_NSUM; ++i) {
    a[i] = sg_wave_sum_d(a[i]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = a[i];
  }
  __syncthreads();
  if (threadIdx.x < LH_NSUM) sums[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// scalars[16] in the return order of data_utils.py:470-473; per-sample upstream gradients:
//   gD_r = d sum(d_loss)/d d_r, gD_f = d sum(d_loss)/d d_f, gS_my, gS_f = d sum(s_loss)/d .,
//   gG_d, gG_s, gG_r = d sum(g_final)/d d_f, s_f, r_f
__global__ __launch_bounds__(256) void k_loss_grads(const float* d_r, const float* d_f, const float* s_my, const float* s_f,
                                                    const float* s_r, const float* r_f, int B, int mode, int balance, float alpha,
                                                    const double* sums, float* scalars, float* gD_r, float* gD_f, float* gS_my,
                                                    float* gS_f, float* gG_d, float* gG_s, float* gG_r, float* shD, float* shS) {
  const double n = sums[11];
  const double g_mean = sums[3] / n, r_mean = sums[7] / n;
  const double g_var = fmax(sums[9] / n - g_mean * g_mean, 0.0), r_var = fmax(sums[10] / n - r_mean * r_mean, 0.0);
  const double g_std = sqrt(g_var), r_std = sqrt(r_var);
  const double ratio = g_std / r_std;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double r_bal_mean = alpha * ratio * r_mean;
    const double g_added_mean = g_mean + r_mean, g_bal_mean = g_mean + r_bal_mean;
    scalars[0] = (float)r_mean; scalars[1] = (float)(sums[8] / n); scalars[2] = (float)r_bal_mean; scalars[3] = (float)g_mean;
    scalars[4] = (float)g_added_mean; scalars[5] = (float)g_bal_mean; scalars[6] = (float)(sums[0] / n);
    scalars[7] = (float)(sums[1] / n); scalars[8] = (float)(sums[2] / n);
    scalars[9] = (float)(balance ? g_bal_mean : g_added_mean); scalars[10] = alpha; scalars[11] = (float)r_std;
    scalars[12] = (float)g_std; scalars[13] = (float)(sums[4] / n); scalars[14] = (float)(sums[5] / n); scalars[15] = (float)(sums[6] / n);
  }
  const double r_sum = sums[7];
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    float d_lr, d_lf, g, s_a, s_b;
    losses(mode, d_r[b], d_f[b], s_my[b], s_f[b], s_r[b], d_lr, d_lf, g, s_a, s_b);
    float dg_dd, dg_ds;                       // d g_b / d d_f[b], d g_b / d s_f[b]
    if (mode == 0) {
      gD_r[b] = (1.f - d_r[b] > 0.f) ? -1.f : 0.f;
      gD_f[b] = (1.f + d_f[b] > 0.f) ? 1.f : 0.f;
      gS_my[b] = (1.f - s_my[b] > 0.f) ? -1.f : 0.f;
      gS_f[b] = (1.f + s_f[b] > 0.f) ? 1.f : 0.f;
      dg_dd = -1.f; dg_ds = -1.f;
    } else {
      gD_r[b] = sigm(d_r[b]) - 1.f;
      gD_f[b] = sigm(d_f[b]);
      gS_my[b] = sigm(s_my[b]) - 1.f;
      gS_f[b] = sigm(s_f[b]);
      dg_dd = sigm(d_f[b]) - 1.f; dg_ds = 0.f;
    }
    double wg = 1.0, wr = 1.0;                 // d sum(g_final)/d g_b, /d r_f[b]
    if (balance) {
      // sum_b g_bal = sum g + alpha (g_std / r_std) sum r ; both stds carry gradient (no stop_gradient)
      wg = 1.0 + alpha * (r_sum / r_std) * ((double)g - g_mean) / (n * g_std);
      wr = alpha * ratio - alpha * g_std * r_sum * ((double)r_f[b] - r_mean) / (n * r_std * r_std * r_std);
    }
    gG_d[b] = (float)(wg * dg_dd);
    gG_s[b] = (float)(wg * dg_ds);
    gG_r[b] = (float)wr;
    // Shared backward sweep through D(x_f) / S(x_f): backprop is linear per sample, so ONE sweep with upstream u_b
    // serves both targets -- the weight gradients of sum(d_loss) weight sample b by gD_f/u_b, the image gradient of
    // sum(g_final) scales by gG_d/u_b.  u_b is the larger of the two in magnitude (both factors <= 1).
    if (shD) {
      const float a = gD_f[b], c = gG_d[b];
      const float u = (fabsf(c) >= fabsf(a)) ? c : a;
      const float inv = u != 0.f ? 1.f / u : 0.f;
      shD[b] = u != 0.f ? u : 1.f; shD[B + b] = a * inv; shD[2 * B + b] = c * inv;
    }
    if (shS) {
      const float a = gS_f[b], c = gG_s[b];
      const float u = (fabsf(c) >= fabsf(a)) ? c : a;
      const float inv = u != 0.f ? 1.f / u : 0.f;
      shS[b] = u != 0.f ? u : 1.f; shS[B + b] = a * inv; shS[2 * B + b] = c * inv;
    }
  }
}

extern "C" int sg_loss_sums(const float* d_r, const float* d_f, const float* s_my, const float* s_f, const float* s_r,
                            const float* r_f, const float* r_r, int B, int mode, double* sums, void* stream) {
  if (!d_r || !d_f || !s_my || !s_f || !s_r || !r_f || !r_r || !sums || B < 1) return SG_ERR_ARG;
  SG_KERNEL(k_loss_sums, dim3(1), dim3(256), 0, (hipStream_t)stream, d_r, d_f, s_my, s_f, s_r, r_f, r_r, B, mode, sums);
  return sg_launch_status();
}

extern "C" int sg_loss_grads(const float* d_r, const float* d_f, const float* s_my, const float* s_f, const float* s_r,
                             const float* r_f, int B, int mode, int balance, float alpha, const double* sums, float* scalars,
                             float* gD_r, float* gD_f, float* gS_my, float* gS_f, float* gG_d, float* gG_s, float* gG_r, float* shD,
                             float* shS, void* stream) {
  if (!d_r || !d_f || !s_my || !s_f || !s_r || !r_f || !sums || !scalars || !gD_r || !gD_f || !gS_my || !gS_f || !gG_d || !gG_s || !gG_r)
    return SG_ERR_ARG;
  SG_KERNEL(k_loss_grads, dim3(sg_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, d_r, d_f, s_my, s_f, s_r, r_f, B, mode,
                     balance, alpha, sums, scalars, gD_r, gD_f, gS_my, gS_f, gG_d, gG_s, gG_r, shD, shS);
  return sg_launch_status();
}

// per-sample loss terms for the Python-level `hinge` / `not_saturating` callables: out[7][B] =
// d_loss, d_loss_real, d_loss_fake, g_loss, s_loss, s_a, s_b
__global__ __launch_bounds__(256) void k_loss_terms(const float* d_r, const float* d_f, const float* s_my, const float* s_f,
                                                    const float* s_r, int B, int mode, float* out) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    float d_lr, d_lf, g, s_a, s_b;
    losses(mode, d_r[b], d_f[b], s_my[b], s_f[b], s_r[b], d_lr, d_lf, g, s_a, s_b);
    out[b] = d_lr + d_lf; out[B + b] = d_lr; out[2 * B + b] = d_lf; out[3 * B + b] = g;
    out[4 * B + b] = s_a + s_b; out[5 * B + b] = s_a; out[6 * B + b] = s_b;
  }
}

extern "C" int sg_loss_terms(const float* d_r, const float* d_f, const float* s_my, const float* s_f, const float* s_r, int B,
                             int mode, float* out7, void* stream) {
  if (!d_r || !d_f || !s_my || !s_f || !s_r || !out7 || B < 1) return SG_ERR_ARG;
  SG_KERNEL(k_loss_terms, dim3(sg_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, d_r, d_f, s_my, s_f, s_r, B, mode, out7);
  return sg_launch_status();
}
