// LSTM cell kernels for make_my_recognizer's Bidirectional(LSTM(256, return_sequences=True, dropout=0.5))
// stack (/root/reference/src/bigacgan/net_architecture.py:146-150), plus the LeakyReLU / dropout-mask
// elementwise ops of that model (:102-138,153).
//
// Keras LSTM (implementation 2): z = x_masked W + h_{t-1} U + b, gate order i, f, c~, o;
// i,f,o = sigmoid, c~ = tanh; c = f c_prev + i c~; h = o tanh(c).  The two GEMMs run on the conv /
// dense kernels (x W for all timesteps at once; h U per step into the same z buffer); these kernels
// are the per-step gate math.  Gates overwrite z in place and are what the backward pass reads.
#include "sg_common.h"

__device__ __forceinline__ float sigm_(float x) { return 1.f / (1.f + expf(-x)); }

// z: [B, 4H] rows at stride ldz (pre-activations in, gate activations out); c_prev/c_out: [B,H] contiguous
// (c_prev may be null = zeros); h_out: [B,H] rows at stride ldh; h_copy (nullable): second copy of h, rows at stride ldc
__global__ __launch_bounds__(256) void k_lstm_cell_fwd(float* z, int ldz, const float* c_prev, float* c_out, float* h_out, int ldh,
                                                       float* h_copy, int ldc, int B, int H) {
  const int total = B * H;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int b = e / H, j = e - b * H;
    float* zr = z + (size_t)b * ldz;
    const float i = sigm_(zr[j]), f = sigm_(zr[H + j]), g = tanhf(zr[2 * H + j]), o = sigm_(zr[3 * H + j]);
    const float cp = c_prev ? c_prev[e] : 0.f;
    const float c = f * cp + i * g;
    const float h = o * tanhf(c);
    zr[j] = i; zr[H + j] = f; zr[2 * H + j] = g; zr[3 * H + j] = o;
    c_out[e] = c;
    h_out[(size_t)b * ldh + j] = h;
    if (h_copy) h_copy[(size_t)b * ldc + j] = h;
  }
}

// gates: [B,4H] rows at stride ldz (activations in, d(pre-activation) out); dh = dh_a (rows at stride lda) + dh_b (nullable,
// contiguous [B,H]); dc_next (nullable) contiguous; dc_prev out contiguous
__global__ __launch_bounds__(256) void k_lstm_cell_bwd(float* gates, int ldz, const float* c_prev, const float* c_t, const float* dh_a,
                                                       int lda, const float* dh_b, const float* dc_next, float* dc_prev, int B, int H) {
  const int total = B * H;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int b = e / H, j = e - b * H;
    float* gr = gates + (size_t)b * ldz;
    const float i = gr[j], f = gr[H + j], g = gr[2 * H + j], o = gr[3 * H + j];
    const float tc = tanhf(c_t[e]);
    float dh = dh_a[(size_t)b * lda + j];
    if (dh_b) dh += dh_b[e];
    float dc = dh * o * (1.f - tc * tc);
    if (dc_next) dc += dc_next[e];
    const float cp = c_prev ? c_prev[e] : 0.f;
    gr[j] = dc * g * i * (1.f - i);
    gr[H + j] = dc * cp * f * (1.f - f);
    gr[2 * H + j] = dc * i * (1.f - g * g);
    gr[3 * H + j] = dh * tc * o * (1.f - o);
    dc_prev[e] = dc * f;
  }
}

__global__ __launch_bounds__(256) void k_leaky_fwd(const float* x, float* y, long n, float alpha) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float v = x[e];
    y[e] = v > 0.f ? v : alpha * v;
  }
}
__global__ __launch_bounds__(256) void k_leaky_bwd(const float* dy, const float* x, float* dx, long n, float alpha) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    dx[e] = x[e] > 0.f ? dy[e] : alpha * dy[e];
}
// out[r, c] = x[r, c] * m[(r / rows_per_mask) , c] : rows_per_mask = 1 -> elementwise; = T -> one mask row per sample
__global__ __launch_bounds__(256) void k_mul_mask(const float* x, const float* m, float* out, long rows, int cols, int rows_per_mask) {
  const long n = rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long r = e / cols;
    out[e] = x[e] * m[(r / rows_per_mask) * cols + (e - r * cols)];
  }
}

extern "C" int sg_lstm_cell_fwd(float* z, int ldz, const float* c_prev, float* c_out, float* h_out, int ldh, float* h_copy, int ldc,
                                int B, int H, void* stream) {
  if (!z || !c_out || !h_out || B < 1 || H < 1) return SG_ERR_ARG;
  SG_KERNEL(k_lstm_cell_fwd, dim3(sg_grid_for((long)B * H, 256)), dim3(256), 0, (hipStream_t)stream, z, ldz, c_prev, c_out,
                     h_out, ldh, h_copy, ldc, B, H);
  return sg_launch_status();
}

extern "C" int sg_lstm_cell_bwd(float* gates, int ldz, const float* c_prev, const float* c_t, const float* dh_a, int lda,
                                const float* dh_b, const float* dc_next, float* dc_prev, int B, int H, void* stream) {
  if (!gates || !c_t || !dh_a || !dc_prev || B < 1 || H < 1) return SG_ERR_ARG;
  SG_KERNEL(k_lstm_cell_bwd, dim3(sg_grid_for((long)B * H, 256)), dim3(256), 0, (hipStream_t)stream, gates, ldz, c_prev, c_t,
                     dh_a, lda, dh_b, dc_next, dc_prev, B, H);
  return sg_launch_status();
}

extern "C" int sg_leaky_relu_fwd(const float* x, float* y, long n, float alpha, void* stream) {
  if (!x || !y) return SG_ERR_ARG;
  SG_KERNEL(k_leaky_fwd, dim3(sg_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, alpha);
  return sg_launch_status();
}

extern "C" int sg_leaky_relu_bwd(const float* dy, const float* x, float* dx, long n, float alpha, void* stream) {
  if (!dy || !x || !dx) return SG_ERR_ARG;
  SG_KERNEL(k_leaky_bwd, dim3(sg_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n, alpha);
  return sg_launch_status();
}

extern "C" int sg_mul_mask(const float* x, const float* mask, float* out, long rows, int cols, int rows_per_mask, void* stream) {
  if (!x || !mask || !out || rows_per_mask < 1) return SG_ERR_ARG;
  SG_KERNEL(k_mul_mask, dim3(sg_grid_for(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, x, mask, out, rows, cols, rows_per_mask);
  return sg_launch_status();
}
