// NonLocalBlock core: softmax(theta . phi^T) . g without materialising the [Nq, Nk] map
// (/root/reference/src/bigacgan/arch_ops.py:51-52,61; un-scaled dot product, d_k = C/8, d_v = C/2).
//
// Head dims here are tiny (d_k = 8, d_v = 32): an MFMA tile would be >75 % padding, so this is a
// VALU flash-style kernel.  One lane owns one query row (q, running max/sum and the 32-wide
// accumulator live in registers); keys/values stream through LDS in tiles and are read as
// wave-uniform broadcasts (no bank conflicts); the online-softmax rescale is amortised over
// groups of 8 keys.  The backward pass recomputes p = exp(s - lse) in two sweeps:
//   dq sweep  : lane = query, streams keys    -> dtheta, delta
//   dkv sweep : lane = key,   streams queries -> dphi, dg      (no atomics, deterministic)
#include "sg_common.h"

#define AT_DK 8
#define AT_DV 32
#define AT_KT 128   // keys (or queries) per LDS tile

__global__ __launch_bounds__(256) void k_attn_fwd(const float* theta, const float* phi, const float* g, float* out, float* lse,
                                                  int Nq, int Nk) {
  __shared__ __attribute__((aligned(16))) float ks[AT_KT * AT_DK];
  __shared__ __attribute__((aligned(16))) float vs[AT_KT * AT_DV];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = qi < Nq;
  float q[AT_DK];
#pragma unroll
  for (int d = 0; d < AT_DK; ++d) q[d] = live ? theta[((size_t)b * Nq + qi) * AT_DK + d] : 0.f;
  float m = -INFINITY, l = 0.f, acc[AT_DV];
#pragma unroll
  for (int c = 0; c < AT_DV; ++c) acc[c] = 0.f;

  for (int k0 = 0; k0 < Nk; k0 += AT_KT) {
    const int kn = min(AT_KT, Nk - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < AT_KT * AT_DK / 4; e += blockDim.x)
      reinterpret_cast<float4*>(ks)[e] = (e * 4 < kn * AT_DK)
          ? reinterpret_cast<const float4*>(phi + ((size_t)b * Nk + k0) * AT_DK)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = threadIdx.x; e < AT_KT * AT_DV / 4; e += blockDim.x)
      reinterpret_cast<float4*>(vs)[e] = (e * 4 < kn * AT_DV)
          ? reinterpret_cast<const float4*>(g + ((size_t)b * Nk + k0) * AT_DV)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int j0 = 0; j0 < kn; j0 += 8) {
      float s[8];
      float mx = m;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float4 ka = reinterpret_cast<const float4*>(ks)[(j0 + jj) * 2];
        const float4 kb = reinterpret_cast<const float4*>(ks)[(j0 + jj) * 2 + 1];
        float t = q[0] * ka.x + q[1] * ka.y + q[2] * ka.z + q[3] * ka.w + q[4] * kb.x + q[5] * kb.y + q[6] * kb.z + q[7] * kb.w;
        if (j0 + jj >= kn) t = -INFINITY;
        s[jj] = t;
        mx = fmaxf(mx, t);
      }
      const float sc = expf(m - mx);      // m = -inf on the first group -> 0
      l *= sc;
#pragma unroll
      for (int c = 0; c < AT_DV; ++c) acc[c] *= sc;
      m = mx;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float p = expf(s[jj] - m);
        l += p;
        const float4* vp = reinterpret_cast<const float4*>(vs + (j0 + jj) * AT_DV);
#pragma unroll
        for (int c4 = 0; c4 < AT_DV / 4; ++c4) {
          const float4 v = vp[c4];
          acc[4 * c4 + 0] += p * v.x; acc[4 * c4 + 1] += p * v.y; acc[4 * c4 + 2] += p * v.z; acc[4 * c4 + 3] += p * v.w;
        }
      }
    }
  }
  if (live) {
    const float inv = 1.f / l;
    float4* op = reinterpret_cast<float4*>(out + ((size_t)b * Nq + qi) * AT_DV);
#pragma unroll
    for (int c4 = 0; c4 < AT_DV / 4; ++c4)
      op[c4] = make_float4(acc[4 * c4] * inv, acc[4 * c4 + 1] * inv, acc[4 * c4 + 2] * inv, acc[4 * c4 + 3] * inv);
    lse[(size_t)b * Nq + qi] = m + logf(l);
  }
}

// dq sweep: dtheta[i] = sum_j p_ij (dP_ij - delta_i) phi_j ; also writes delta_i = dO_i . O_i
__global__ __launch_bounds__(256) void k_attn_bwd_dq(const float* theta, const float* phi, const float* g, const float* out,
                                                     const float* lse, const float* dout, float* dtheta, float* delta, int Nq, int Nk) {
  __shared__ __attribute__((aligned(16))) float ks[AT_KT * AT_DK];
  __shared__ __attribute__((aligned(16))) float vs[AT_KT * AT_DV];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = qi < Nq;
  const size_t row = (size_t)b * Nq + (live ? qi : 0);
  float q[AT_DK], dq[AT_DK], dO[AT_DV];
#pragma unroll
  for (int d = 0; d < AT_DK; ++d) { q[d] = live ? theta[row * AT_DK + d] : 0.f; dq[d] = 0.f; }
  float dl = 0.f;
#pragma unroll
  for (int c4 = 0; c4 < AT_DV / 4; ++c4) {
    const float4 a = reinterpret_cast<const float4*>(dout + row * AT_DV)[c4];
    const float4 o = reinterpret_cast<const float4*>(out + row * AT_DV)[c4];
    dO[4 * c4] = a.x; dO[4 * c4 + 1] = a.y; dO[4 * c4 + 2] = a.z; dO[4 * c4 + 3] = a.w;
    dl += a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
  }
  const float ls = lse[row];
  for (int k0 = 0; k0 < Nk; k0 += AT_KT) {
    const int kn = min(AT_KT, Nk - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < AT_KT * AT_DK / 4; e += blockDim.x)
      reinterpret_cast<float4*>(ks)[e] = (e * 4 < kn * AT_DK)
          ? reinterpret_cast<const float4*>(phi + ((size_t)b * Nk + k0) * AT_DK)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = threadIdx.x; e < AT_KT * AT_DV / 4; e += blockDim.x)
      reinterpret_cast<float4*>(vs)[e] = (e * 4 < kn * AT_DV)
          ? reinterpret_cast<const float4*>(g + ((size_t)b * Nk + k0) * AT_DV)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int j = 0; j < kn; ++j) {
      const float4 ka = reinterpret_cast<const float4*>(ks)[j * 2], kb = reinterpret_cast<const float4*>(ks)[j * 2 + 1];
      const float s = q[0] * ka.x + q[1] * ka.y + q[2] * ka.z + q[3] * ka.w + q[4] * kb.x + q[5] * kb.y + q[6] * kb.z + q[7] * kb.w;
      const float p = expf(s - ls);
      float dp = 0.f;
      const float4* vp = reinterpret_cast<const float4*>(vs + j * AT_DV);
#pragma unroll
      for (int c4 = 0; c4 < AT_DV / 4; ++c4) {
        const float4 v = vp[c4];
        dp += dO[4 * c4] * v.x + dO[4 * c4 + 1] * v.y + dO[4 * c4 + 2] * v.z + dO[4 * c4 + 3] * v.w;
      }
      const float ds = p * (dp - dl);
      dq[0] += ds * ka.x; dq[1] += ds * ka.y; dq[2] += ds * ka.z; dq[3] += ds * ka.w;
      dq[4] += ds * kb.x; dq[5] += ds * kb.y; dq[6] += ds * kb.z; dq[7] += ds * kb.w;
    }
  }
  if (live) {
    float4* dp4 = reinterpret_cast<float4*>(dtheta + row * AT_DK);
    dp4[0] = make_float4(dq[0], dq[1], dq[2], dq[3]);
    dp4[1] = make_float4(dq[4], dq[5], dq[6], dq[7]);
    delta[row] = dl;
  }
}

// dkv sweep: lane = key j: dg_j = sum_i p_ij dO_i ; dphi_j = sum_i p_ij (dP_ij - delta_i) theta_i
#define AT_QREC 44   // per-query LDS record: theta[8], dO[32], lse, delta, pad to 16-byte multiple
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(const float* theta, const float* phi, const float* g, const float* lse,
                                                      const float* dout, const float* delta, float* dphi, float* dg, int Nq, int Nk,
                                                      int q_chunk) {
  // gridDim.z > 1: the query range is split over z (small batches would otherwise leave most CUs idle) and the
  // partial dphi/dg rows are added with float atomics into pre-zeroed outputs
  __shared__ __attribute__((aligned(16))) float qs[AT_KT * AT_QREC];
  const int q_begin = blockIdx.z * q_chunk, q_end = min(Nq, q_begin + q_chunk);
  const int b = blockIdx.y;
  const int kj = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = kj < Nk;
  const size_t krow = (size_t)b * Nk + (live ? kj : 0);
  float k[AT_DK], v[AT_DV], dk[AT_DK], dv[AT_DV];
#pragma unroll
  for (int d = 0; d < AT_DK; ++d) { k[d] = phi[krow * AT_DK + d]; dk[d] = 0.f; }
#pragma unroll
  for (int c = 0; c < AT_DV; ++c) { v[c] = g[krow * AT_DV + c]; dv[c] = 0.f; }
  for (int q0 = q_begin; q0 < q_end; q0 += AT_KT) {
    const int qn = min(AT_KT, q_end - q0);
    __syncthreads();
    for (int e = threadIdx.x; e < AT_KT * 2; e += blockDim.x) {       // theta: 2 float4 per query
      const int i = e >> 1, h = e & 1;
      reinterpret_cast<float4*>(qs + i * AT_QREC)[h] =
          i < qn ? reinterpret_cast<const float4*>(theta + ((size_t)b * Nq + q0 + i) * AT_DK)[h] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int e = threadIdx.x; e < AT_KT * 8; e += blockDim.x) {       // dO: 8 float4 per query
      const int i = e >> 3, h = e & 7;
      reinterpret_cast<float4*>(qs + i * AT_QREC + 8)[h] =
          i < qn ? reinterpret_cast<const float4*>(dout + ((size_t)b * Nq + q0 + i) * AT_DV)[h] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = threadIdx.x; i < AT_KT; i += blockDim.x) {
      qs[i * AT_QREC + 40] = i < qn ? lse[(size_t)b * Nq + q0 + i] : INFINITY;   // p = exp(s - inf) = 0 for padding
      qs[i * AT_QREC + 41] = i < qn ? delta[(size_t)b * Nq + q0 + i] : 0.f;
    }
    __syncthreads();
    for (int i = 0; i < qn; ++i) {
      const float4* r = reinterpret_cast<const float4*>(qs + i * AT_QREC);
      const float4 qa = r[0], qb = r[1];
      const float s = k[0] * qa.x + k[1] * qa.y + k[2] * qa.z + k[3] * qa.w + k[4] * qb.x + k[5] * qb.y + k[6] * qb.z + k[7] * qb.w;
      const float4 tail = r[10];
      const float p = expf(s - tail.x);
      float dp = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < AT_DV / 4; ++c4) {
        const float4 a = r[2 + c4];
        dp += a.x * v[4 * c4] + a.y * v[4 * c4 + 1] + a.z * v[4 * c4 + 2] + a.w * v[4 * c4 + 3];
        dv[4 * c4] += p * a.x; dv[4 * c4 + 1] += p * a.y; dv[4 * c4 + 2] += p * a.z; dv[4 * c4 + 3] += p * a.w;
      }
      const float ds = p * (dp - tail.y);
      dk[0] += ds * qa.x; dk[1] += ds * qa.y; dk[2] += ds * qa.z; dk[3] += ds * qa.w;
      dk[4] += ds * qb.x; dk[5] += ds * qb.y; dk[6] += ds * qb.z; dk[7] += ds * qb.w;
    }
  }
  if (live && gridDim.z > 1) {
#pragma unroll
    for (int d = 0; d < AT_DK; ++d) atomicAdd(dphi + krow * AT_DK + d, dk[d]);
#pragma unroll
    for (int c = 0; c < AT_DV; ++c) atomicAdd(dg + krow * AT_DV + c, dv[c]);
  } else if (live) {
    float4* dp4 = reinterpret_cast<float4*>(dphi + krow * AT_DK);
    dp4[0] = make_float4(dk[0], dk[1], dk[2], dk[3]);
    dp4[1] = make_float4(dk[4], dk[5], dk[6], dk[7]);
    float4* dg4 = reinterpret_cast<float4*>(dg + krow * AT_DV);
#pragma unroll
    for (int c4 = 0; c4 < AT_DV / 4; ++c4) dg4[c4] = make_float4(dv[4 * c4], dv[4 * c4 + 1], dv[4 * c4 + 2], dv[4 * c4 + 3]);
  }
}

// ------------------------------------------------------------------------------------------
// Small-batch variants (per-GPU batches of the 8-way data-parallel shard): 64 queries per workgroup, and the FOUR
// waves of the workgroup split the KEYS of those queries (wave w takes key tiles w, w+4, ...), each with a private
// LDS tile; the partial softmax states are merged through LDS at the end.  4x the parallelism of one wave per 64
// queries, a quarter of the serial key loop.
// ------------------------------------------------------------------------------------------
#define AT_KS 64    // keys per wave-private LDS tile

__global__ __launch_bounds__(256) void k_attn_fwd_ks(const float* theta, const float* phi, const float* g, float* out, float* lse,
                                                     int Nq, int Nk) {
  __shared__ __attribute__((aligned(16))) float sm[4 * AT_KS * (AT_DK + AT_DV)];      // 40 KB: 4 x (K tile + V tile)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ks = sm + wave * AT_KS * (AT_DK + AT_DV);
  float* vs = ks + AT_KS * AT_DK;
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 64 + lane;
  const bool live = qi < Nq;
  float q[AT_DK];
#pragma unroll
  for (int d = 0; d < AT_DK; ++d) q[d] = live ? theta[((size_t)b * Nq + qi) * AT_DK + d] : 0.f;
  float m = -INFINITY, l = 0.f, acc[AT_DV];
#pragma unroll
  for (int c = 0; c < AT_DV; ++c) acc[c] = 0.f;

  for (int k0 = wave * AT_KS; k0 < Nk; k0 += 4 * AT_KS) {        // wave-uniform trip count; no workgroup barrier inside
    const int kn = min(AT_KS, Nk - k0);
    for (int e = lane; e < AT_KS * AT_DK / 4; e += 64)
      reinterpret_cast<float4*>(ks)[e] = (e * 4 < kn * AT_DK)
          ? reinterpret_cast<const float4*>(phi + ((size_t)b * Nk + k0) * AT_DK)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = lane; e < AT_KS * AT_DV / 4; e += 64)
      reinterpret_cast<float4*>(vs)[e] = (e * 4 < kn * AT_DV)
          ? reinterpret_cast<const float4*>(g + ((size_t)b * Nk + k0) * AT_DV)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();
    for (int j0 = 0; j0 < kn; j0 += 8) {
      float s[8];
      float mx = m;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float4 ka = reinterpret_cast<const float4*>(ks)[(j0 + jj) * 2];
        const float4 kb = reinterpret_cast<const float4*>(ks)[(j0 + jj) * 2 + 1];
        float t = q[0] * ka.x + q[1] * ka.y + q[2] * ka.z + q[3] * ka.w + q[4] * kb.x + q[5] * kb.y + q[6] * kb.z + q[7] * kb.w;
        if (j0 + jj >= kn) t = -INFINITY;
        s[jj] = t;
        mx = fmaxf(mx, t);
      }
      const float sc = expf(m - mx);
      l *= sc;
#pragma unroll
      for (int c = 0; c < AT_DV; ++c) acc[c] *= sc;
      m = mx;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float p = expf(s[jj] - m);
        l += p;
        const float4* vp = reinterpret_cast<const float4*>(vs + (j0 + jj) * AT_DV);
#pragma unroll
        for (int c4 = 0; c4 < AT_DV / 4; ++c4) {
          const float4 v = vp[c4];
          acc[4 * c4 + 0] += p * v.x; acc[4 * c4 + 1] += p * v.y; acc[4 * c4 + 2] += p * v.z; acc[4 * c4 + 3] += p * v.w;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // merge the four partial states of each query: sm is re-used as part[w][34][64]
  __syncthreads();
  float* part = sm + wave * (AT_DV + 2) * 64;
  part[0 * 64 + lane] = m;
  part[1 * 64 + lane] = l;
#pragma unroll
  for (int c = 0; c < AT_DV; ++c) part[(2 + c) * 64 + lane] = acc[c];
  __syncthreads();
  if (wave == 0 && live) {
    float ms = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) ms = fmaxf(ms, sm[w * (AT_DV + 2) * 64 + lane]);
    float lt = 0.f, o[AT_DV];
#pragma unroll
    for (int c = 0; c < AT_DV; ++c) o[c] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* pw = sm + w * (AT_DV + 2) * 64;
      const float f = expf(pw[lane] - ms);          // exp(-inf) = 0 for a wave that saw no key
      lt += pw[64 + lane] * f;
#pragma unroll
      for (int c = 0; c < AT_DV; ++c) o[c] += pw[(2 + c) * 64 + lane] * f;
    }
    const float inv = 1.f / lt;
    float4* op = reinterpret_cast<float4*>(out + ((size_t)b * Nq + qi) * AT_DV);
#pragma unroll
    for (int c4 = 0; c4 < AT_DV / 4; ++c4)
      op[c4] = make_float4(o[4 * c4] * inv, o[4 * c4 + 1] * inv, o[4 * c4 + 2] * inv, o[4 * c4 + 3] * inv);
    lse[(size_t)b * Nq + qi] = ms + logf(lt);
  }
}

__global__ __launch_bounds__(256) void k_attn_bwd_dq_ks(const float* theta, const float* phi, const float* g, const float* out,
                                                        const float* lse, const float* dout, float* dtheta, float* delta, int Nq, int Nk) {
  __shared__ __attribute__((aligned(16))) float sm[4 * AT_KS * (AT_DK + AT_DV)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ks = sm + wave * AT_KS * (AT_DK + AT_DV);
  float* vs = ks + AT_KS * AT_DK;
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 64 + lane;
  const bool live = qi < Nq;
  const size_t row = (size_t)b * Nq + (live ? qi : 0);
  float q[AT_DK], dq[AT_DK], dO[AT_DV];
#pragma unroll
  for (int d = 0; d < AT_DK; ++d) { q[d] = live ? theta[row * AT_DK + d] : 0.f; dq[d] = 0.f; }
  float dl = 0.f;
#pragma unroll
  for (int c4 = 0; c4 < AT_DV / 4; ++c4) {
    const float4 a = reinterpret_cast<const float4*>(dout + row * AT_DV)[c4];
    const float4 o = reinterpret_cast<const float4*>(out + row * AT_DV)[c4];
    dO[4 * c4] = a.x; dO[4 * c4 + 1] = a.y; dO[4 * c4 + 2] = a.z; dO[4 * c4 + 3] = a.w;
    dl += a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
  }
  const float ls = lse[row];
  for (int k0 = wave * AT_KS; k0 < Nk; k0 += 4 * AT_KS) {
    const int kn = min(AT_KS, Nk - k0);
    for (int e = lane; e < AT_KS * AT_DK / 4; e += 64)
      reinterpret_cast<float4*>(ks)[e] = (e * 4 < kn * AT_DK)
          ? reinterpret_cast<const float4*>(phi + ((size_t)b * Nk + k0) * AT_DK)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = lane; e < AT_KS * AT_DV / 4; e += 64)
      reinterpret_cast<float4*>(vs)[e] = (e * 4 < kn * AT_DV)
          ? reinterpret_cast<const float4*>(g + ((size_t)b * Nk + k0) * AT_DV)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < kn; ++j) {
      const float4 ka = reinterpret_cast<const float4*>(ks)[j * 2], kb = reinterpret_cast<const float4*>(ks)[j * 2 + 1];
      const float s = q[0] * ka.x + q[1] * ka.y + q[2] * ka.z + q[3] * ka.w + q[4] * kb.x + q[5] * kb.y + q[6] * kb.z + q[7] * kb.w;
      const float p = expf(s - ls);
      float dp = 0.f;
      const float4* vp = reinterpret_cast<const float4*>(vs + j * AT_DV);
#pragma unroll
      for (int c4 = 0; c4 < AT_DV / 4; ++c4) {
        const float4 v = vp[c4];
        dp += dO[4 * c4] * v.x + dO[4 * c4 + 1] * v.y + dO[4 * c4 + 2] * v.z + dO[4 * c4 + 3] * v.w;
      }
      const float ds = p * (dp - dl);
      dq[0] += ds * ka.x; dq[1] += ds * ka.y; dq[2] += ds * ka.z; dq[3] += ds * ka.w;
      dq[4] += ds * kb.x; dq[5] += ds * kb.y; dq[6] += ds * kb.z; dq[7] += ds * kb.w;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();                       // partial dq of the four key slices: plain sum (lse is already final)
  float* part = sm + wave * AT_DK * 64;
#pragma unroll
  for (int d = 0; d < AT_DK; ++d) part[d * 64 + lane] = dq[d];
  __syncthreads();
  if (wave == 0 && live) {
    float r[AT_DK];
#pragma unroll
    for (int d = 0; d < AT_DK; ++d)
      r[d] = sm[d * 64 + lane] + sm[(AT_DK + d) * 64 + lane] + sm[(2 * AT_DK + d) * 64 + lane] + sm[(3 * AT_DK + d) * 64 + lane];
    float4* dp4 = reinterpret_cast<float4*>(dtheta + row * AT_DK);
    dp4[0] = make_float4(r[0], r[1], r[2], r[3]);
    dp4[1] = make_float4(r[4], r[5], r[6], r[7]);
    delta[row] = dl;
  }
}

// one lane per row; 64-lane workgroups when 256-lane ones would leave most of the 256 CUs idle
static inline int at_threads(int rows, int B) { return (long)sg_cdiv(rows, 256) * B >= 1024 ? 256 : 64; }

// theta [B,Nq,8], phi [B,Nk,8], g [B,Nk,32] -> out [B,Nq,32], lse [B,Nq]
extern "C" int sg_attention_fwd(const float* theta, const float* phi, const float* g, float* out, float* lse, int B, int Nq,
                                int Nk, int dk, int dv, void* stream) {
  if (!theta || !phi || !g || !out || !lse || dk != AT_DK || dv != AT_DV || Nk < 1) return SG_ERR_ARG;
  const int tq = at_threads(Nq, B);
  if (tq == 64)      // small batch: 64 queries per workgroup, keys split over its four waves
    hipLaunchKernelGGL(k_attn_fwd_ks, dim3(sg_cdiv(Nq, 64), B), dim3(256), 0, (hipStream_t)stream, theta, phi, g, out, lse, Nq, Nk);
  else
    hipLaunchKernelGGL(k_attn_fwd, dim3(sg_cdiv(Nq, tq), B), dim3(tq), 0, (hipStream_t)stream, theta, phi, g, out, lse, Nq, Nk);
  return sg_launch_status();
}

// delta is a [B,Nq] scratch vector
extern "C" int sg_attention_bwd(const float* theta, const float* phi, const float* g, const float* out, const float* lse,
                                const float* dout, float* dtheta, float* dphi, float* dg, float* delta, int B, int Nq, int Nk,
                                int dk, int dv, void* stream) {
  if (!theta || !phi || !g || !out || !lse || !dout || !dtheta || !dphi || !dg || !delta || dk != AT_DK || dv != AT_DV || Nk < 1)
    return SG_ERR_ARG;
  const int tq = at_threads(Nq, B), tk = at_threads(Nk, B);
  if (tq == 64)
    hipLaunchKernelGGL(k_attn_bwd_dq_ks, dim3(sg_cdiv(Nq, 64), B), dim3(256), 0, (hipStream_t)stream, theta, phi, g, out, lse, dout,
                       dtheta, delta, Nq, Nk);
  else
    hipLaunchKernelGGL(k_attn_bwd_dq, dim3(sg_cdiv(Nq, tq), B), dim3(tq), 0, (hipStream_t)stream, theta, phi, g, out, lse, dout,
                       dtheta, delta, Nq, Nk);
  const long kblocks = (long)sg_cdiv(Nk, tk) * B;
  int zs = kblocks >= 1024 ? 1 : (int)((1024 + kblocks - 1) / kblocks);
  if (zs > 16) zs = 16;
  int q_chunk = sg_cdiv(sg_cdiv(Nq, zs), AT_KT) * AT_KT;        // whole LDS tiles per z slice
  zs = sg_cdiv(Nq, q_chunk);
  if (zs > 1) {
    if (hipMemsetAsync(dphi, 0, sizeof(float) * (size_t)B * Nk * AT_DK, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(dg, 0, sizeof(float) * (size_t)B * Nk * AT_DV, (hipStream_t)stream) != hipSuccess)
      return SG_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k_attn_bwd_dkv, dim3(sg_cdiv(Nk, tk), B, zs), dim3(tk), 0, (hipStream_t)stream, theta, phi, g, lse, dout,
                     delta, dphi, dg, Nq, Nk, q_chunk);
  return sg_launch_status();
}
