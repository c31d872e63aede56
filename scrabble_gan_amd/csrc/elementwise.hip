// HBM-bound elementwise / pooling / reduction kernels of the train step (NHWC fp32).
// All are float4-vectorised along channels (C % 4 == 0 unless noted), grid capped at 2048
// workgroups with a grid-stride loop, one wave = 64 lanes.
#include "sg_common.h"

#define F4(p) (*reinterpret_cast<float4*>(p))
#define CF4(p) (*reinterpret_cast<const float4*>(p))

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// out = mean2x2(a) (+ mean2x2(b)) ; tf.nn.pool(AVG,2x2,SAME,s2) on even dims (resnet_ops.py:106,113)
__global__ __launch_bounds__(256) void k_avgpool2_add(const float* a, const float* b, float* out, int B, int H, int W, int C) {
  const int cq = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo * cq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(e % cq);
    long r = e / cq;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int bb = (int)(r / Ho);
    const size_t i00 = (((size_t)bb * H + 2 * yo) * W + 2 * xo) * C + c;
    const size_t rs = (size_t)W * C;
    float4 s = f4add(f4add(CF4(a + i00), CF4(a + i00 + C)), f4add(CF4(a + i00 + rs), CF4(a + i00 + rs + C)));
    s = f4scale(s, 0.25f);
    if (b) {
      float4 t = f4add(f4add(CF4(b + i00), CF4(b + i00 + C)), f4add(CF4(b + i00 + rs), CF4(b + i00 + rs + C)));
      s = f4add(s, f4scale(t, 0.25f));
    }
    F4(out + e * 4) = s;
  }
}

// dx[b,y,x,c] = 0.25 * dout[b,y/2,x/2,c]   (H,W = input dims)
__global__ __launch_bounds__(256) void k_avgpool2_bwd(const float* dout, float* dx, int B, int H, int W, int C) {
  const int cq = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * H * W * cq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(e % cq);
    long r = e / cq;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int bb = (int)(r / H);
    const float4 v = CF4(dout + (((size_t)bb * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c);
    F4(dx + e * 4) = f4scale(v, 0.25f);
  }
}

// scalar variant for C == 1 tensors (images)
__global__ __launch_bounds__(256) void k_avgpool2_bwd_c1(const float* dout, float* dx, int B, int H, int W) {
  const long total = (long)B * H * W;
  const int Ho = H >> 1, Wo = W >> 1;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W);
    const int y = (int)((e / W) % H);
    const int bb = (int)(e / ((long)W * H));
    dx[e] = 0.25f * dout[((size_t)bb * Ho + (y >> 1)) * Wo + (x >> 1)];
  }
}

__global__ __launch_bounds__(256) void k_add(const float* a, const float* b, float* out, long n4, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride)
    F4(out + 4 * e) = f4add(CF4(a + 4 * e), CF4(b + 4 * e));
  for (long e = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) out[e] = a[e] + b[e];
}

// dx = ref > 0 ? dy : 0
__global__ __launch_bounds__(256) void k_relu_mask(const float* dy, const float* ref, float* dx, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    dx[e] = ref[e] > 0.f ? dy[e] : 0.f;
}

// dx = dy * (1 - y^2)
__global__ __launch_bounds__(256) void k_tanh_bwd(const float* y, const float* dy, float* dx, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float t = y[e];
    dx[e] = dy[e] * (1.f - t * t);
  }
}

// MaxPool2D(pool=(ph,pw)) VALID, stride = pool; idx keeps the window position of the (first) max
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* x, float* y, unsigned char* idx, int B, int H, int W, int C, int ph, int pw) {
  const int cq = C >> 2, Ho = H / ph, Wo = W / pw;
  const long total = (long)B * Ho * Wo * cq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(e % cq);
    long r = e / cq;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int bb = (int)(r / Ho);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 bi = make_uchar4(0, 0, 0, 0);
    for (int iy = 0; iy < ph; ++iy)
      for (int ix = 0; ix < pw; ++ix) {
        const float4 v = CF4(x + (((size_t)bb * H + yo * ph + iy) * W + xo * pw + ix) * C + c);
        const unsigned char k = (unsigned char)(iy * pw + ix);
        if (v.x > best.x) { best.x = v.x; bi.x = k; }
        if (v.y > best.y) { best.y = v.y; bi.y = k; }
        if (v.z > best.z) { best.z = v.z; bi.z = k; }
        if (v.w > best.w) { best.w = v.w; bi.w = k; }
      }
    F4(y + e * 4) = best;
    *reinterpret_cast<uchar4*>(idx + e * 4) = bi;
  }
}

// dx (input shape) = dy routed to the arg-max position, zero elsewhere; optional accumulate
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int ph, int pw, int accum) {
  const int cq = C >> 2, Ho = H / ph, Wo = W / pw;
  const long total = (long)B * Ho * Wo * cq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(e % cq);
    long r = e / cq;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int bb = (int)(r / Ho);
    const float4 g = CF4(dy + e * 4);
    const uchar4 bi = *reinterpret_cast<const uchar4*>(idx + e * 4);
    for (int iy = 0; iy < ph; ++iy)
      for (int ix = 0; ix < pw; ++ix) {
        const unsigned char k = (unsigned char)(iy * pw + ix);
        float4 v = make_float4(bi.x == k ? g.x : 0.f, bi.y == k ? g.y : 0.f, bi.z == k ? g.z : 0.f, bi.w == k ? g.w : 0.f);
        float* p = dx + (((size_t)bb * H + yo * ph + iy) * W + xo * pw + ix) * C + c;
        if (accum) v = f4add(v, CF4(p));
        F4(p) = v;
      }
  }
}

// out[b,c] = mean_hw relu?(x[b,hw,c])   (tf.nn.relu + GlobalAveragePooling2D, net_architecture.py:249-250)
// one workgroup per (b, 64-channel slab): 16 float4 lanes x 16 row lanes
__global__ __launch_bounds__(256) void k_gap_fwd(const float* x, float* out, int HW, int C, int relu) {
  __shared__ float4 red[256];
  const int b = blockIdx.y, c = blockIdx.x * 64 + 4 * (threadIdx.x & 15), rl = threadIdx.x >> 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C)
    for (int r = rl; r < HW; r += 16) {
      float4 v = CF4(x + ((size_t)b * HW + r) * C + c);
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      s = f4add(s, v);
    }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 8; st > 0; st >>= 1) {
    if (rl < st) red[threadIdx.x] = f4add(red[threadIdx.x], red[threadIdx.x + st * 16]);
    __syncthreads();
  }
  if (rl == 0 && c < C) F4(out + (size_t)b * C + c) = f4scale(red[threadIdx.x], 1.f / (float)HW);
}

// dx[b,hw,c] = (relu ? x>0 : 1) * dout[b,c] / HW
__global__ __launch_bounds__(256) void k_gap_bwd(const float* dout, const float* x, float* dx, int B, int HW, int C, int relu) {
  const int cq = C >> 2;
  const long total = (long)B * HW * cq;
  const float inv = 1.f / (float)HW;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(e % cq);
    const int b = (int)(e / ((long)cq * HW));
    float4 g = f4scale(CF4(dout + (size_t)b * C + c), inv);
    if (relu) {
      const float4 v = CF4(x + e * 4);
      if (v.x <= 0.f) g.x = 0.f;
      if (v.y <= 0.f) g.y = 0.f;
      if (v.z <= 0.f) g.z = 0.f;
      if (v.w <= 0.f) g.w = 0.f;
    }
    F4(dx + e * 4) = g;
  }
}

// db[n] += sum_m dy[m,n].  Each workgroup reduces a slab of rows, then one float atomic per column.
// float4 path (N % 4 == 0): CL column lanes x RL row lanes per workgroup, LDS tree over the row lanes
__global__ __launch_bounds__(256) void k_bias_grad_v4(const float* dy, float* db, long M, int N, int rows_per_block) {
  __shared__ float4 red[256];
  const int cqn = N >> 2;
  int CL = 256;
  while (CL > cqn) CL >>= 1;             // largest power of two <= min(cqn, 256)
  const int RL = 256 / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const long m0 = (long)blockIdx.x * rows_per_block;
  const long m1 = min(M, m0 + rows_per_block);
  for (int cg0 = 0; cg0 < cqn; cg0 += CL) {
    const int cg = cg0 + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cg < cqn) {
      // four independent 16-byte loads in flight per thread (a single dependent load per iteration left the sweep
      // latency-bound: 1.9 TB/s measured in round 2)
      float4 s1 = s, s2 = s, s3 = s;
      long m = m0 + rl;
      for (; m + 3L * RL < m1; m += 4L * RL) {
        const float4 a0 = CF4(dy + m * N + 4 * cg), a1 = CF4(dy + (m + RL) * N + 4 * cg);
        const float4 a2 = CF4(dy + (m + 2L * RL) * N + 4 * cg), a3 = CF4(dy + (m + 3L * RL) * N + 4 * cg);
        s = f4add(s, a0); s1 = f4add(s1, a1); s2 = f4add(s2, a2); s3 = f4add(s3, a3);
      }
      for (; m < m1; m += RL) s = f4add(s, CF4(dy + m * N + 4 * cg));
      s = f4add(f4add(s, s1), f4add(s2, s3));
    }
    if (RL > 1) {
      red[threadIdx.x] = s;
      __syncthreads();
      for (int st = RL >> 1; st > 0; st >>= 1) {
        if (rl < st) red[threadIdx.x] = f4add(red[threadIdx.x], red[threadIdx.x + st * CL]);
        __syncthreads();
      }
      s = red[threadIdx.x];
      __syncthreads();
    }
    if (rl == 0 && cg < cqn) {
      float* d = db + 4 * cg;
      atomicAdd(d + 0, s.x); atomicAdd(d + 1, s.y); atomicAdd(d + 2, s.z); atomicAdd(d + 3, s.w);
    }
  }
}

__global__ __launch_bounds__(256) void k_bias_grad(const float* dy, float* db, long M, int N, int rows_per_block) {
  __shared__ float red[256];
  const long m0 = (long)blockIdx.x * rows_per_block;
  const long m1 = min(M, m0 + rows_per_block);
  if (N >= 256 || (256 % N)) {           // generic: columns strided over threads
    for (int n = threadIdx.x; n < N; n += 256) {
      float s = 0.f;
      for (long m = m0; m < m1; ++m) s += dy[m * N + n];
      atomicAdd(db + n, s);
    }
    return;
  }
  const int n = threadIdx.x % N, rl = threadIdx.x / N, lanes = 256 / N;
  float s = 0.f;
  for (long m = m0 + rl; m < m1; m += lanes) s += dy[m * N + n];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < lanes; ++k) s += red[threadIdx.x + k * N];
    atomicAdd(db + n, s);
  }
}

// out = sigma[0] * o + x     (NonLocalBlock residual, arch_ops.py:67)
__global__ __launch_bounds__(256) void k_scale_add(const float* o, const float* x, const float* sigma, float* out, long n4) {
  const float s = sigma[0];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x)
    F4(out + 4 * e) = f4add(f4scale(CF4(o + 4 * e), s), CF4(x + 4 * e));
}

// out = s[0] * a
__global__ __launch_bounds__(256) void k_scale(const float* a, const float* s, float* out, long n4) {
  const float sv = s[0];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x)
    F4(out + 4 * e) = f4scale(CF4(a + 4 * e), sv);
}

// out[0] += sum_i a[i] * b[i]
__global__ __launch_bounds__(256) void k_dot(const float* a, const float* b, float* out, long n4) {
  __shared__ float red[4];
  float s = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    const float4 u = CF4(a + 4 * e), v = CF4(b + 4 * e);
    s += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
  }
  s = sg_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

#define LAUNCH(k, n_items, s, ...) \
  SG_KERNEL(k, dim3(sg_grid_for((n_items), 256)), dim3(256), 0, (hipStream_t)(s), __VA_ARGS__)

extern "C" int sg_avgpool2_add_fwd(const float* a, const float* b, float* out, int B, int H, int W, int C, void* stream) {
  if (!a || !out || (C & 3) || (H & 1) || (W & 1)) return SG_ERR_ARG;
  LAUNCH(k_avgpool2_add, (long)B * (H / 2) * (W / 2) * (C / 4), stream, a, b, out, B, H, W, C);
  return sg_launch_status();
}

extern "C" int sg_avgpool2_bwd(const float* dout, float* dx, int B, int H, int W, int C, void* stream) {
  if (!dout || !dx || (H & 1) || (W & 1)) return SG_ERR_ARG;
  if (C == 1) {
    LAUNCH(k_avgpool2_bwd_c1, (long)B * H * W, stream, dout, dx, B, H, W);
  } else {
    if (C & 3) return SG_ERR_ARG;
    LAUNCH(k_avgpool2_bwd, (long)B * H * W * (C / 4), stream, dout, dx, B, H, W, C);
  }
  return sg_launch_status();
}

extern "C" int sg_add(const float* a, const float* b, float* out, long n, void* stream) {
  if (!a || !b || !out) return SG_ERR_ARG;
  const long n4 = ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0) ? n / 4 : 0;
  LAUNCH(k_add, n4 > 0 ? n4 : n, stream, a, b, out, n4, n);
  return sg_launch_status();
}

extern "C" int sg_relu_mask(const float* dy, const float* ref, float* dx, long n, void* stream) {
  if (!dy || !ref || !dx) return SG_ERR_ARG;
  LAUNCH(k_relu_mask, n, stream, dy, ref, dx, n);
  return sg_launch_status();
}

extern "C" int sg_tanh_bwd(const float* y, const float* dy, float* dx, long n, void* stream) {
  if (!y || !dy || !dx) return SG_ERR_ARG;
  LAUNCH(k_tanh_bwd, n, stream, y, dy, dx, n);
  return sg_launch_status();
}

extern "C" int sg_maxpool_fwd(const float* x, float* y, unsigned char* idx, int B, int H, int W, int C, int ph, int pw, void* stream) {
  if (!x || !y || !idx || (C & 3) || H % ph || W % pw) return SG_ERR_ARG;
  LAUNCH(k_maxpool_fwd, (long)B * (H / ph) * (W / pw) * (C / 4), stream, x, y, idx, B, H, W, C, ph, pw);
  return sg_launch_status();
}

extern "C" int sg_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, int ph, int pw, int accum, void* stream) {
  if (!dy || !dx || !idx || (C & 3) || H % ph || W % pw) return SG_ERR_ARG;
  LAUNCH(k_maxpool_bwd, (long)B * (H / ph) * (W / pw) * (C / 4), stream, dy, idx, dx, B, H, W, C, ph, pw, accum);
  return sg_launch_status();
}

extern "C" int sg_gap_fwd(const float* x, float* out, int B, int HW, int C, int relu, void* stream) {
  if (!x || !out || (C & 3)) return SG_ERR_ARG;
  SG_KERNEL(k_gap_fwd, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, out, HW, C, relu);
  return sg_launch_status();
}

extern "C" int sg_gap_bwd(const float* dout, const float* x, float* dx, int B, int HW, int C, int relu, void* stream) {
  if (!dout || !dx || (C & 3) || (relu && !x)) return SG_ERR_ARG;
  LAUNCH(k_gap_bwd, (long)B * HW * (C / 4), stream, dout, x, dx, B, HW, C, relu);
  return sg_launch_status();
}

extern "C" int sg_bias_grad(const float* dy, float* db, long M, int N, void* stream) {
  if (!dy || !db || N < 1) return SG_ERR_ARG;
  if ((N & 3) == 0 && ((uintptr_t)dy & 15) == 0) {
    // <= 256 workgroups: every workgroup ends in one float atomic per column, and same-address atomics serialise
    // (~90 ns each: 1024 adders per column cost more than the whole sweep)
    long r = (M + 511) / 512;
    if (sg_deterministic()) r = M;          // one workgroup: one adder per column, fixed summation order
    const int rpb = (int)(r < 16 ? 16 : r);
    SG_KERNEL(k_bias_grad_v4, dim3(sg_cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, dy, db, M, N, rpb);
    return sg_launch_status();
  }
  const int rpb = sg_deterministic() ? (int)(M < 1 ? 1 : M) : 1024;
  SG_KERNEL(k_bias_grad, dim3(sg_cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, dy, db, M, N, rpb);
  return sg_launch_status();
}

extern "C" int sg_scale_add(const float* o, const float* x, const float* sigma, float* out, long n, void* stream) {
  if (!o || !x || !sigma || !out || (n & 3)) return SG_ERR_ARG;
  LAUNCH(k_scale_add, n / 4, stream, o, x, sigma, out, n / 4);
  return sg_launch_status();
}

extern "C" int sg_scale(const float* a, const float* s, float* out, long n, void* stream) {
  if (!a || !s || !out || (n & 3)) return SG_ERR_ARG;
  LAUNCH(k_scale, n / 4, stream, a, s, out, n / 4);
  return sg_launch_status();
}

extern "C" int sg_dot_accum(const float* a, const float* b, float* out, long n, void* stream) {
  if (!a || !b || !out || (n & 3)) return SG_ERR_ARG;
  SG_KERNEL(k_dot, dim3(sg_deterministic() ? 1 : sg_grid_for(n / 4, 256 * 8)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4);      // (deterministic: one adder)
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
// host-pipeline helpers
// ------------------------------------------------------------------------------------------
// out[i] = (float(u8[i]) - 127.5) / 127.5   -- the pixel normalisation of load_prepare_data (data_utils.py:82) on the GPU,
// same fp32 operations as the numpy expression (subtract, then divide), so the result is bit-identical
__global__ __launch_bounds__(256) void k_normalize_u8(const uint4* __restrict__ in, float4* __restrict__ out, long n16) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (long)gridDim.x * blockDim.x) {
    const uint4 v = in[e];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 o;
      o.x = ((float)(w[k] & 0xffu) - 127.5f) / 127.5f;
      o.y = ((float)((w[k] >> 8) & 0xffu) - 127.5f) / 127.5f;
      o.z = ((float)((w[k] >> 16) & 0xffu) - 127.5f) / 127.5f;
      o.w = ((float)(w[k] >> 24) - 127.5f) / 127.5f;
      out[4 * e + k] = o;
    }
  }
}

// u8 [n] (device) -> out fp32 [n]; n % 16 == 0
extern "C" int sg_normalize_u8(const unsigned char* u8, float* out, long n, void* stream) {
  if (!u8 || !out || n < 0 || (n & 15)) return SG_ERR_ARG;
  if (n == 0) return SG_OK;
  LAUNCH(k_normalize_u8, n / 16, stream, reinterpret_cast<const uint4*>(u8), reinterpret_cast<float4*>(out), n / 16);
  return sg_launch_status();
}

// y[m, c] += bias[c]   (C % 4 == 0): the bias of a strided Conv2D expressed over the transposed-conv kernels
__global__ __launch_bounds__(256) void k_bias_add(float* y, const float* bias, long n4, int c4) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x)
    F4(y + 4 * e) = f4add(CF4(y + 4 * e), CF4(bias + 4 * (e % c4)));
}

extern "C" int sg_bias_add(float* y, const float* bias, long M, int C, void* stream) {
  if (!y || !bias || M < 0 || C < 4 || (C & 3)) return SG_ERR_ARG;
  if (M == 0) return SG_OK;
  LAUNCH(k_bias_add, M * (C / 4), stream, y, bias, M * (C / 4), C / 4);
  return sg_launch_status();
}

// ---- self-test of SG_KERNEL / sg_launch_status (sg_common.h): a failed launch in the MIDDLE of a sequence must surface
__global__ void k_selftest_fill(float* p, float v) { p[threadIdx.x & 255] = v; }
extern "C" int sg_selftest_launch_status(float* scratch, void* stream) {
  if (!scratch) return SG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  SG_KERNEL(k_selftest_fill, dim3(1), dim3(256), 0, s, scratch, 1.f);
  SG_KERNEL(k_selftest_fill, dim3(1), dim3(2048), 0, s, scratch, 2.f);      // more threads than a workgroup can hold: the launch is refused
  SG_KERNEL(k_selftest_fill, dim3(1), dim3(256), 0, s, scratch, 3.f);
  return sg_launch_status();
}
