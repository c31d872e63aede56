// Per-character filter bank (SpatialEmbedding) fused with the z0 contraction and the seed layout.
// Replaces tf.nn.embedding_lookup + tile + matmul + reshape/reshape/transpose of
// /root/reference/src/bigacgan/arch_ops.py:89-90 and net_architecture.py:262-271 without
// materialising the gathered [B,L,32,8192] tensor:
//   seed[b, r, 4l+pw, q] = sum_k z0[b,k] * E[y[b,l], k, j],   j = pw*2048 + q*4 + r
// One thread owns 4 consecutive j (one float4 of the table row = the 4 seed rows r of one (pw,q)),
// so table reads are 16-byte coalesced and each of the 4 seed stores is coalesced across lanes.
// The 54.5 MB table stays resident in the Infinity Cache across the B*L gathers.
#include "sg_common.h"

#define FB_K 32
#define FB_J 8192

__global__ __launch_bounds__(256) void k_filterbank_fwd(const float* z0, const int* y, const float* table, float* seed, int L, int vocab) {
  __shared__ float zs[FB_K];
  const int bl = blockIdx.y;            // b*L + l
  const int b = bl / L, l = bl - b * L;
  if (threadIdx.x < FB_K) zs[threadIdx.x] = z0[(size_t)b * 128 + threadIdx.x];   // z row stride 128: z0 = z[:, :32]
  __syncthreads();
  int cls = y[bl];
  cls = cls < 0 ? 0 : (cls >= vocab ? vocab - 1 : cls);
  const int j = 4 * (blockIdx.x * 256 + threadIdx.x);
  const float* e = table + (size_t)cls * FB_K * FB_J + j;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int k = 0; k < FB_K; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(e + (size_t)k * FB_J);
    const float z = zs[k];
    acc.x += z * v.x; acc.y += z * v.y; acc.z += z * v.z; acc.w += z * v.w;
  }
  const int pw = j >> 11, q = (j & 2047) >> 2;
  const int W4 = 4 * L;
  const size_t base = ((size_t)b * 4 * W4 + 4 * l + pw) * 512 + q;     // r = 0
  const size_t rs = (size_t)W4 * 512;
  seed[base] = acc.x; seed[base + rs] = acc.y; seed[base + 2 * rs] = acc.z; seed[base + 3 * rs] = acc.w;
}

// dz[b,k] += sum_j E[y[b,l],k,j]*dseed[...]      (one workgroup per (column tile, b*L + l))
__global__ __launch_bounds__(256) void k_filterbank_bwd_dz(const int* y, const float* table, const float* dseed, float* dz, int L, int vocab) {
  __shared__ float red[4][FB_K];
  const int bl = blockIdx.y;
  const int b = bl / L, l = bl - b * L;
  int cls = y[bl];
  cls = cls < 0 ? 0 : (cls >= vocab ? vocab - 1 : cls);
  const int j = 4 * (blockIdx.x * 256 + threadIdx.x);
  const int pw = j >> 11, q = (j & 2047) >> 2;
  const int W4 = 4 * L;
  const size_t base = ((size_t)b * 4 * W4 + 4 * l + pw) * 512 + q;
  const size_t rs = (size_t)W4 * 512;
  const float4 d = make_float4(dseed[base], dseed[base + rs], dseed[base + 2 * rs], dseed[base + 3 * rs]);
  const float* e = table + (size_t)cls * FB_K * FB_J + j;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 8
  for (int k = 0; k < FB_K; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(e + (size_t)k * FB_J);
    float s = v.x * d.x + v.y * d.y + v.z * d.z + v.w * d.w;
    s = sg_wave_sum(s);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < FB_K)
    atomicAdd(dz + (size_t)b * 128 + threadIdx.x,
              red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// The same sums with ONE adder per dz[b, k] (deterministic mode): one workgroup per sample walks its L characters and the 8 column
// tiles in a fixed order, each thread keeps its 32 partial sums in registers, one tree at the end.
__global__ __launch_bounds__(256) void k_filterbank_bwd_dz_det(const int* y, const float* table, const float* dseed, float* dz, int L, int vocab) {
  __shared__ float red[4][FB_K];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W4 = 4 * L;
  const size_t rs = (size_t)W4 * 512;
  float acc[FB_K];
#pragma unroll
  for (int k = 0; k < FB_K; ++k) acc[k] = 0.f;
  for (int l = 0; l < L; ++l) {
    int cls = y[b * L + l];
    cls = cls < 0 ? 0 : (cls >= vocab ? vocab - 1 : cls);
    for (int jt = 0; jt < FB_J / 4 / 256; ++jt) {
      const int j = 4 * (jt * 256 + threadIdx.x);
      const int pw = j >> 11, q = (j & 2047) >> 2;
      const size_t base = ((size_t)b * 4 * W4 + 4 * l + pw) * 512 + q;
      const float4 d = make_float4(dseed[base], dseed[base + rs], dseed[base + 2 * rs], dseed[base + 3 * rs]);
      const float* e = table + (size_t)cls * FB_K * FB_J + j;
#pragma unroll 8
      for (int k = 0; k < FB_K; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(e + (size_t)k * FB_J);
        acc[k] += v.x * d.x + v.y * d.y + v.z * d.z + v.w * d.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < FB_K; ++k) {
    const float s = sg_wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < FB_K) dz[(size_t)b * 128 + threadIdx.x] += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// dE[c,k,j] += sum over the (b,l) with y[b,l] == c of z0[b,k]*dseed[...]
// One workgroup per (column tile, class): it scans the B*L labels (wave ballots, ascending order, so the sum order is
// fixed), keeps the 32 x 4 partial sums of its columns in registers and adds them to the table gradient ONCE, with
// plain stores -- no atomics into the 54.5 MB table, and the rows of classes that do not occur are not touched.
__global__ __launch_bounds__(256) void k_filterbank_bwd_table(const float* z0, const int* y, const float* dseed, float* dtable,
                                                              int BL, int L, int vocab) {
  const int cls = blockIdx.y;
  const int j = 4 * (blockIdx.x * 256 + threadIdx.x);
  const int pw = j >> 11, q = (j & 2047) >> 2;
  const int W4 = 4 * L;
  const size_t rs = (size_t)W4 * 512;
  const int lane = threadIdx.x & 63;
  float4 acc[FB_K];
#pragma unroll
  for (int k = 0; k < FB_K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  bool any = false;
  for (int base0 = 0; base0 < BL; base0 += 64) {
    int yy = -1;
    if (base0 + lane < BL) {
      yy = y[base0 + lane];
      yy = yy < 0 ? 0 : (yy >= vocab ? vocab - 1 : yy);
    }
    unsigned long long m = __ballot(yy == cls);
    while (m) {
      const int i = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int bl = __builtin_amdgcn_readfirstlane(base0 + i);
      const int b = bl / L, l = bl - b * L;
      const size_t base = ((size_t)b * 4 * W4 + 4 * l + pw) * 512 + q;
      const float4 d = make_float4(dseed[base], dseed[base + rs], dseed[base + 2 * rs], dseed[base + 3 * rs]);
      const float* zb = z0 + (size_t)b * 128;          // wave-uniform: scalar loads
#pragma unroll
      for (int k = 0; k < FB_K; ++k) {
        const float z = zb[k];
        acc[k].x += z * d.x; acc[k].y += z * d.y; acc[k].z += z * d.z; acc[k].w += z * d.w;
      }
      any = true;
    }
  }
  if (!any) return;
  float* de = dtable + (size_t)cls * FB_K * FB_J + j;
#pragma unroll
  for (int k = 0; k < FB_K; ++k) {
    float4* dp = reinterpret_cast<float4*>(de + (size_t)k * FB_J);
    float4 v = *dp;
    v.x += acc[k].x; v.y += acc[k].y; v.z += acc[k].z; v.w += acc[k].w;
    *dp = v;
  }
}

// z: [B,128] (only columns 0..31 are read: z0); seed: [B,4,4L,512]
extern "C" int sg_filterbank_fwd(const float* z, const int* y, const float* table, float* seed, int B, int L, int vocab, void* stream) {
  if (!z || !y || !table || !seed || B < 1 || L < 1) return SG_ERR_ARG;
  SG_KERNEL(k_filterbank_fwd, dim3(FB_J / 4 / 256, B * L), dim3(256), 0, (hipStream_t)stream, z, y, table, seed, L, vocab);
  return sg_launch_status();
}

// dtable += ..., dz[:, 0:32] += ...   (dz is the [B,128] gradient of z)
extern "C" int sg_filterbank_bwd(const float* z, const int* y, const float* table, const float* dseed, float* dtable, float* dz,
                                 int B, int L, int vocab, void* stream) {
  if (!z || !y || !table || !dseed || !dtable || !dz || B < 1 || L < 1) return SG_ERR_ARG;
  if (sg_deterministic()) SG_KERNEL(k_filterbank_bwd_dz_det, dim3(B), dim3(256), 0, (hipStream_t)stream, y, table, dseed, dz, L, vocab);
  else SG_KERNEL(k_filterbank_bwd_dz, dim3(FB_J / 4 / 256, B * L), dim3(256), 0, (hipStream_t)stream, y, table, dseed, dz, L, vocab);
  SG_KERNEL(k_filterbank_bwd_table, dim3(FB_J / 4 / 256, vocab), dim3(256), 0, (hipStream_t)stream, z, y, dseed, dtable, B * L, L, vocab);
  return sg_launch_status();
}
