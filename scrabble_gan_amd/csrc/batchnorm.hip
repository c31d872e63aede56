// Batch-norm family: batch statistics, (conditional) affine apply + ReLU, and the backward pair.
// Replaces layers.BatchNormalization + the per-sample gamma/beta of ConditionalBatchNorm
// (/root/reference/src/bigacgan/resnet_ops.py:13-28) and the plain BNs of net_architecture.py:42,46,281.
//
// All kernels are HBM-bound sweeps over x[B, HW, C] with float4 channel vectors; the cross-row
// reductions are wavefront/LDS trees inside a workgroup, then a tiny second stage in fp64 so that
// E[x^2]-mean^2 keeps fp32-level accuracy (and so that data-parallel ranks can all-reduce the
// fp64 sums between the two stages: SyncBN).
#include "sg_common.h"

#define F4(p) (*reinterpret_cast<float4*>(p))
#define CF4(p) (*reinterpret_cast<const float4*>(p))

// stage 1: partial[blk][0][c] = sum x, partial[blk][1][c] = sum x^2 over the block's rows
__global__ __launch_bounds__(256) void k_bn_stats_partial(const float* x, float* partial, long M, int C, int rows_per_block) {
  __shared__ float4 r1[256], r2[256];
  const int cqn = C >> 2;
  const int cq = threadIdx.x % cqn, rl = threadIdx.x / cqn;
  int lanes = 1;                                   // row lanes: largest power of two with lanes*cqn <= 256
  while (lanes * 2 * cqn <= 256) lanes *= 2;
  const long m0 = (long)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  for (long m = m0 + rl; m < m1 && rl < lanes; m += lanes) {
    const float4 v = CF4(x + m * C + 4 * cq);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  r1[threadIdx.x] = s; r2[threadIdx.x] = q;
  __syncthreads();
  for (int st = lanes >> 1; st > 0; st >>= 1) {
    if (rl < st) {
      float4 a = r1[threadIdx.x], b = r1[threadIdx.x + st * cqn];
      r1[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
      a = r2[threadIdx.x]; b = r2[threadIdx.x + st * cqn];
      r2[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (rl == 0) {
    F4(partial + ((size_t)blockIdx.x * 2 + 0) * C + 4 * cq) = r1[threadIdx.x];
    F4(partial + ((size_t)blockIdx.x * 2 + 1) * C + 4 * cq) = r2[threadIdx.x];
  }
}

// stage 2: sums[0][c], sums[1][c] (fp64) = sum over blocks.  16 columns x 16 block-lanes per workgroup: each thread
// adds every 16th partial (independent loads, short chain), then an LDS tree over the lanes in fp64.
__global__ __launch_bounds__(256) void k_bn_stats_combine(const float* partial, double* sums, int nblk, int C) {
  __shared__ double red[256];
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0;
  if (c < 2 * C)
    for (int b = bl; b < nblk; b += 16) s += (double)partial[(size_t)b * 2 * C + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 8; st > 0; st >>= 1) {
    if (bl < st) red[threadIdx.x] += red[threadIdx.x + st * 16];
    __syncthreads();
  }
  if (bl == 0 && c < 2 * C) sums[c] = red[threadIdx.x];
}

// stage 3: mean/var (biased) from (possibly all-reduced) fp64 sums and the global row count
__global__ __launch_bounds__(256) void k_bn_stats_finalize(const double* sums, double count, float* mean, float* var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = sums[c] / count;
  double v = sums[C + c] / count - m * m;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)m;
  var[c] = (float)v;
}

// y = ((x - mean) * rsqrt(var + eps)) * gamma[b*gstride + c] + beta[b*gstride + c]  (+ ReLU)
__global__ __launch_bounds__(256) void k_bn_apply(const float* x, const float* mean, const float* var, const float* gamma,
                                                  const float* beta, int gstride, float* y, int B, int HW, int C, float eps, int relu) {
  const int cq = C >> 2;
  const long total = (long)B * HW * cq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(e % cq);
    const int b = (int)(e / ((long)cq * HW));
    const float4 v = CF4(x + e * 4), mu = CF4(mean + c), va = CF4(var + c);
    const float4 g = CF4(gamma + (size_t)b * gstride + c), bt = CF4(beta + (size_t)b * gstride + c);
    float4 o;
    o.x = ((v.x - mu.x) * rsqrtf(va.x + eps)) * g.x + bt.x;
    o.y = ((v.y - mu.y) * rsqrtf(va.y + eps)) * g.y + bt.y;
    o.z = ((v.z - mu.z) * rsqrtf(va.z + eps)) * g.z + bt.z;
    o.w = ((v.w - mu.w) * rsqrtf(va.w + eps)) * g.w + bt.w;
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    F4(y + e * 4) = o;
  }
}

// backward pass 1: per (b,c): dbeta += sum_hw dz, dgamma += sum_hw dz * x_hat, dz = dy * (relu ? y>0 : 1)
// grid = (row chunks, B); float atomics into [B,C] (few adders per address)
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean,
                                                       const float* var, float* dgamma, float* dbeta, int HW, int C,
                                                       float eps, int relu, int rows_per_block) {
  __shared__ float4 r1[256], r2[256];
  const int cqn = C >> 2;
  const int cq = threadIdx.x % cqn, rl = threadIdx.x / cqn;
  int lanes = 1;
  while (lanes * 2 * cqn <= 256) lanes *= 2;
  const int b = blockIdx.y;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(HW, m0 + rows_per_block);
  const float4 mu = CF4(mean + 4 * cq), va = CF4(var + 4 * cq);
  const float4 rs = make_float4(rsqrtf(va.x + eps), rsqrtf(va.y + eps), rsqrtf(va.z + eps), rsqrtf(va.w + eps));
  float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
  for (int m = m0 + rl; m < m1 && rl < lanes; m += lanes) {
    const size_t off = ((size_t)b * HW + m) * C + 4 * cq;
    float4 g = CF4(dy + off);
    if (relu) {
      const float4 o = CF4(y + off);
      if (o.x <= 0.f) g.x = 0.f;
      if (o.y <= 0.f) g.y = 0.f;
      if (o.z <= 0.f) g.z = 0.f;
      if (o.w <= 0.f) g.w = 0.f;
    }
    const float4 v = CF4(x + off);
    sb.x += g.x; sb.y += g.y; sb.z += g.z; sb.w += g.w;
    sg.x += g.x * ((v.x - mu.x) * rs.x); sg.y += g.y * ((v.y - mu.y) * rs.y);
    sg.z += g.z * ((v.z - mu.z) * rs.z); sg.w += g.w * ((v.w - mu.w) * rs.w);
  }
  r1[threadIdx.x] = sb; r2[threadIdx.x] = sg;
  __syncthreads();
  for (int st = lanes >> 1; st > 0; st >>= 1) {
    if (rl < st) {
      float4 a = r1[threadIdx.x], c = r1[threadIdx.x + st * cqn];
      r1[threadIdx.x] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
      a = r2[threadIdx.x]; c = r2[threadIdx.x + st * cqn];
      r2[threadIdx.x] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
    __syncthreads();
  }
  if (rl == 0) {
    const float4 a = r1[threadIdx.x], c = r2[threadIdx.x];
    float* pb = dbeta + (size_t)b * C + 4 * cq;
    float* pg = dgamma + (size_t)b * C + 4 * cq;
    atomicAdd(pb + 0, a.x); atomicAdd(pb + 1, a.y); atomicAdd(pb + 2, a.z); atomicAdd(pb + 3, a.w);
    atomicAdd(pg + 0, c.x); atomicAdd(pg + 1, c.y); atomicAdd(pg + 2, c.z); atomicAdd(pg + 3, c.w);
  }
}

// per channel: s[0][c] = sum_b gamma[b,c]*dbeta[b,c] (= sum dx_hat), s[1][c] = sum_b gamma[b,c]*dgamma[b,c]
// (= sum dx_hat*x_hat), s[2][c] = sum_b dgamma[b,c], s[3][c] = sum_b dbeta[b,c]   (fp64)
__global__ __launch_bounds__(256) void k_bn_bwd_chan(const float* dgamma, const float* dbeta, const float* gamma, int gstride,
                                                     int B, int C, double* s, float* dgamma_c, float* dbeta_c) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int b = 0; b < B; ++b) {
    const double g = gamma[(size_t)b * gstride + c], dg = dgamma[(size_t)b * C + c], db = dbeta[(size_t)b * C + c];
    a0 += g * db; a1 += g * dg; a2 += dg; a3 += db;
  }
  s[c] = a0; s[C + c] = a1; s[2 * C + c] = a2; s[3 * C + c] = a3;
  if (dgamma_c) dgamma_c[c] += (float)a2;     // per-channel affine (plain BatchNormalization): grads accumulate
  if (dbeta_c) dbeta_c[c] += (float)a3;
}

// backward pass 2: dx = rstd * (dz*gamma - s0/n - x_hat * s1/n); use_stats=0 -> inference-mode BN (dx = dz*gamma*rstd)
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean,
                                                      const float* var, const float* gamma, int gstride, const double* s,
                                                      double count, float* dx, int B, int HW, int C, float eps, int relu, int use_stats) {
  const int cq = C >> 2;
  const long total = (long)B * HW * cq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(e % cq);
    const int b = (int)(e / ((long)cq * HW));
    float4 g = CF4(dy + e * 4);
    if (relu) {
      const float4 o = CF4(y + e * 4);
      if (o.x <= 0.f) g.x = 0.f;
      if (o.y <= 0.f) g.y = 0.f;
      if (o.z <= 0.f) g.z = 0.f;
      if (o.w <= 0.f) g.w = 0.f;
    }
    const float4 v = CF4(x + e * 4), mu = CF4(mean + c), va = CF4(var + c), gm = CF4(gamma + (size_t)b * gstride + c);
    const float rs[4] = {rsqrtf(va.x + eps), rsqrtf(va.y + eps), rsqrtf(va.z + eps), rsqrtf(va.w + eps)};
    const float gv[4] = {g.x * gm.x, g.y * gm.y, g.z * gm.z, g.w * gm.w};
    const float xv[4] = {(v.x - mu.x) * rs[0], (v.y - mu.y) * rs[1], (v.z - mu.z) * rs[2], (v.w - mu.w) * rs[3]};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = gv[k];
      if (use_stats) t = t - (float)(s[c + k] / count) - xv[k] * (float)(s[C + c + k] / count);
      o[k] = rs[k] * t;
    }
    F4(dx + e * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// moving <- momentum*moving + (1-momentum)*batch ; variance with Bessel's correction (fused-BN semantics)
__global__ void k_bn_update_moving(float* mm, float* mv, const float* mean, const float* var, double count, float momentum, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float corr = count > 1.0 ? (float)(count / (count - 1.0)) : 1.f;
  mm[c] = momentum * mm[c] + (1.f - momentum) * mean[c];
  mv[c] = momentum * mv[c] + (1.f - momentum) * var[c] * corr;
}

static inline bool chan_ok(int C) { return C >= 4 && (C & 3) == 0 && (C >> 2) <= 256; }

// rows per workgroup of the statistics sweep: ~512 workgroups (the second stage walks them serially per channel),
// 32..2048 rows each -- small per-GPU batches must still spread over the chip
static inline int bn_stats_rpb(long M) {
  long r = (M + 511) / 512;
  return (int)(r < 32 ? 32 : (r > 2048 ? 2048 : r));
}

extern "C" long sg_bn_stats_workspace_floats(long M, int C) { return (long)sg_cdiv(M, bn_stats_rpb(M)) * 2 * C; }

// sums (fp64, [2*C]) = per-channel sum and sum of squares over the M rows of x
extern "C" int sg_bn_stats_sums(const float* x, long M, int C, float* workspace, double* sums, void* stream) {
  if (!x || !workspace || !sums || !chan_ok(C)) return SG_ERR_ARG;
  const int rpb = bn_stats_rpb(M), nblk = sg_cdiv(M, rpb);
  SG_KERNEL(k_bn_stats_partial, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, workspace, M, C, rpb);
  SG_KERNEL(k_bn_stats_combine, dim3(sg_cdiv(2 * C, 16)), dim3(256), 0, (hipStream_t)stream, workspace, sums, nblk, C);
  return sg_launch_status();
}

extern "C" int sg_bn_stats_finalize(const double* sums, double count, float* mean, float* var, int C, void* stream) {
  if (!sums || !mean || !var || count <= 0) return SG_ERR_ARG;
  SG_KERNEL(k_bn_stats_finalize, dim3(sg_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, count, mean, var, C);
  return sg_launch_status();
}

extern "C" int sg_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                           int gstride, float* y, int B, int HW, int C, float eps, int relu, void* stream) {
  if (!x || !mean || !var || !gamma || !beta || !y || (C & 3)) return SG_ERR_ARG;
  SG_KERNEL(k_bn_apply, dim3(sg_grid_for((long)B * HW * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, mean, var,
                     gamma, beta, gstride, y, B, HW, C, eps, relu);
  return sg_launch_status();
}

// dgamma/dbeta are [B,C] and must be zeroed by the caller; chan (fp64 [4*C]) receives the per-channel sums
extern "C" int sg_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* var,
                                const float* gamma, int gstride, float* dgamma, float* dbeta, double* chan, float* dgamma_c,
                                float* dbeta_c, int B, int HW, int C, float eps, int relu, void* stream) {
  if (!dy || !x || !mean || !var || !gamma || !dgamma || !dbeta || !chan || !chan_ok(C) || (relu && !y)) return SG_ERR_ARG;
  long r = ((long)HW * B + 511) / 512;                  // ~512 workgroups; <= HW / 16 float atomics per [b, c] address
  int rpb = (int)(r < 16 ? 16 : (r > 512 ? 512 : r));
  if (sg_deterministic()) rpb = HW;                     // one workgroup per sample: ONE adder per [b, c] address (onto the caller's zeros), fixed order
  SG_KERNEL(k_bn_bwd_reduce, dim3(sg_cdiv(HW, rpb), B), dim3(256), 0, (hipStream_t)stream, dy, y, x, mean, var, dgamma,
                     dbeta, HW, C, eps, relu, rpb);
  SG_KERNEL(k_bn_bwd_chan, dim3(sg_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dgamma, dbeta, gamma, gstride, B, C, chan,
                     dgamma_c, dbeta_c);
  return sg_launch_status();
}

extern "C" int sg_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* var,
                               const float* gamma, int gstride, const double* chan, double count, float* dx, int B, int HW,
                               int C, float eps, int relu, int use_stats, void* stream) {
  if (!dy || !x || !mean || !var || !gamma || !dx || (C & 3) || (relu && !y) || (use_stats && (!chan || count <= 0))) return SG_ERR_ARG;
  SG_KERNEL(k_bn_bwd_apply, dim3(sg_grid_for((long)B * HW * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, dy, y, x,
                     mean, var, gamma, gstride, chan, count, dx, B, HW, C, eps, relu, use_stats);
  return sg_launch_status();
}

extern "C" int sg_bn_update_moving(float* mm, float* mv, const float* mean, const float* var, double count, float momentum,
                                   int C, void* stream) {
  if (!mm || !mv || !mean || !var) return SG_ERR_ARG;
  SG_KERNEL(k_bn_update_moving, dim3(sg_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mm, mv, mean, var, count, momentum, C);
  return sg_launch_status();
}
