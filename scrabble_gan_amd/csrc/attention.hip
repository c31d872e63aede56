// NonLocalBlock core: softmax(theta . phi^T) . g without materialising the [Nq, Nk] map
// (/root/reference/src/bigacgan/arch_ops.py:51-52,61; un-scaled dot product, d_k = C/8 = 8, d_v = C/2 = 32).
//
// Flash-style kernels on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation -- twice
// the rate of plain VALU FMAs, and no per-key LDS broadcast reads, which bound round 1's one-lane-per-query VALU kernels).
// Everything is computed TRANSPOSED so that no intermediate ever changes its register layout:
//   forward, lane = query:  S^T = K Q^T   (A = K block [key][d],  B = Q^T [d][q])      4 MFMAs per 32 keys x 32 queries
//                           the accumulator holds S^T[key(u, h)][q = lane & 31], key(u, h) = (u & 3) + 8 (u >> 2) + 4 h:
//                           the softmax statistics of a query are reductions over the lane's own 16 registers plus one
//                           cross-half shuffle, the rescale of O^T is a per-lane scalar, and register u of P^T IS the B
//                           operand of step u of   O^T += V^T P^T   (A = V^T [c][key pair u])             16 MFMAs
//                           (the order of the reduction index is free as long as both operands agree)
//   backward, two sweeps (as in round 1: no atomics, deterministic):
//     dq sweep,  lane = query: S^T, P^T = exp(S^T - lse), dP^T = V dO^T (16), dS^T = P^T (dP^T - delta),
//                              dQ^T += K^T dS^T (16; rows d >= 8 of the 32-row tile are padding)
//     dkv sweep, lane = key:   S = Q K^T (4), P, dP = dO V^T (16), dS, dV^T += dO^T P (16), dK^T += Q^T dS (16, padded)
// Tiles of the streamed operand (keys for the query sweeps, queries for the key sweep) are staged in LDS with odd row
// strides (9 / 33 floats) so that both the row reads and the column reads of the MFMA operands are conflict-free or 2-way.
#include "sg_common.h"

#define AT_DK 8
#define AT_DV 32
#define AT_KT 128   // keys per LDS tile of the query sweeps
#define AT_QT 64    // queries per LDS tile of the key sweep
#define AT_KS 9     // LDS row stride of theta / phi rows (floats)
#define AT_VS 33    // LDS row stride of g / dO rows where columns are read (dq, dkv sweeps)
#define AT_VF 40    // ... where only rows are read (forward): 4 rows apart = 32 banks apart, conflict-free

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int at_row(int u, int h) { return (u & 3) + 8 * (u >> 2) + 4 * h; }   // accumulator register -> tile row
// exp(x) = 2^(x log2 e) on v_exp_f32, with the rounding error of the product (and of the constant) folded back in: the
// plain product is off by up to |x| * 6e-8 in the exponent, i.e. 2e-6 relative at x = -50; this form stays within ~2 ulp
// (x is clamped at -104: exp is 0 there in fp32, and -inf -- a masked key, the running maximum before the first block -- would
// otherwise give inf - inf in the error term)
__device__ __forceinline__ float at_exp(float xin) {
  const float x = fmaxf(xin, -104.f);
  const float t = x * 1.44269502162933349609375f;                                    // float(log2 e)
  const float e = __builtin_fmaf(x, 1.44269502162933349609375f, -t) + x * 1.925963033500011e-8f;   // + x * (log2 e - float(log2 e))
  const float p = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(p, e * 0.693147180559945309f, p);
}
__device__ __forceinline__ float at_xor32(float v) { return __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------------------------------------------
// forward: WAVES waves x QB blocks of 32 queries per workgroup
template <int WAVES, int QB>
__global__ __launch_bounds__(WAVES * 64) void k_attn_fwd(const float* __restrict__ theta, const float* __restrict__ phi,
                                                         const float* __restrict__ g, float* __restrict__ out, float* __restrict__ lse,
                                                         int Nq, int Nk) {
  __shared__ float ks[AT_KT * AT_KS];
  __shared__ float vs[AT_KT * AT_VF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int b = blockIdx.y;
  const int q_base = (blockIdx.x * WAVES + wave) * (QB * 32);
  float qf[QB][4], m[QB], l[QB];
  f32x16 o[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q_base + qb * 32 + c;
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[qb][t] = q < Nq ? theta[((size_t)b * Nq + q) * AT_DK + 2 * t + h] : 0.f;
    m[qb] = -INFINITY;
    l[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[qb][r] = 0.f;
  }
  for (int k0 = 0; k0 < Nk; k0 += AT_KT) {
    const int kn = min(AT_KT, Nk - k0);
    __syncthreads();
    for (int e = tid; e < AT_KT * AT_DK; e += WAVES * 64) {
      const int key = e >> 3, d = e & 7;
      ks[key * AT_KS + d] = key < kn ? phi[((size_t)b * Nk + k0 + key) * AT_DK + d] : 0.f;
    }
    for (int e = tid; e < AT_KT * AT_DV; e += WAVES * 64) {
      const int key = e >> 5, cc = e & 31;
      vs[key * AT_VF + cc] = key < kn ? g[((size_t)b * Nk + k0 + key) * AT_DV + cc] : 0.f;
    }
    __syncthreads();
    const int nblk = (kn + 31) >> 5;
    for (int kb = 0; kb < nblk; ++kb) {
      float ka[4], va[16];
#pragma unroll
      for (int t = 0; t < 4; ++t) ka[t] = ks[(kb * 32 + c) * AT_KS + 2 * t + h];
#pragma unroll
      for (int u = 0; u < 16; ++u) va[u] = vs[(kb * 32 + at_row(u, h)) * AT_VF + c];
      const bool ragged = kb * 32 + 32 > kn;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], qf[qb][t], s, 0, 0, 0);
        if (ragged) {
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (kb * 32 + at_row(u, h) >= kn) s[u] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int u = 1; u < 16; ++u) mx = fmaxf(mx, s[u]);
        mx = fmaxf(mx, at_xor32(mx));
        const float mn = fmaxf(m[qb], mx);          // finite: key kb * 32 of the block is always a real key
        const float sc = at_exp(m[qb] - mn);        // m = -inf before the first block -> 0
        float rs = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          s[u] = at_exp(s[u] - mn);
          rs += s[u];
        }
        rs += at_xor32(rs);
        l[qb] = l[qb] * sc + rs;
        m[qb] = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][r] *= sc;
#pragma unroll
        for (int u = 0; u < 16; ++u) o[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[u], s[u], o[qb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q_base + qb * 32 + c;
    if (q < Nq) {
      const float inv = 1.f / l[qb];
      float* op = out + ((size_t)b * Nq + q) * AT_DV + 4 * h;        // O^T register 4 g + j <-> channel 8 g + 4 h + j
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(op + 8 * gq) =
            make_float4(o[qb][4 * gq] * inv, o[qb][4 * gq + 1] * inv, o[qb][4 * gq + 2] * inv, o[qb][4 * gq + 3] * inv);
      if (h == 0) lse[(size_t)b * Nq + q] = m[qb] + logf(l[qb]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dq sweep: dtheta[i] = sum_j p_ij (dP_ij - delta_i) phi_j ; also writes delta_i = dO_i . O_i
template <int WAVES, int QB>
__global__ __launch_bounds__(WAVES * 64) void k_attn_bwd_dq(const float* __restrict__ theta, const float* __restrict__ phi,
                                                            const float* __restrict__ g, const float* __restrict__ out,
                                                            const float* __restrict__ lse, const float* __restrict__ dout,
                                                            float* __restrict__ dtheta, float* __restrict__ delta, int Nq, int Nk) {
  __shared__ float ks[AT_KT * AT_KS];
  __shared__ float vs[AT_KT * AT_VS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int b = blockIdx.y;
  const int q_base = (blockIdx.x * WAVES + wave) * (QB * 32);
  float qf[QB][4], dof[QB][16], ls[QB], dl[QB];
  f32x16 dq[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q_base + qb * 32 + c;
    const bool live = q < Nq;
    const size_t row = (size_t)b * Nq + (live ? q : 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[qb][t] = live ? theta[row * AT_DK + 2 * t + h] : 0.f;
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      dof[qb][t] = live ? dout[row * AT_DV + 2 * t + h] : 0.f;
      part += dof[qb][t] * (live ? out[row * AT_DV + 2 * t + h] : 0.f);
    }
    dl[qb] = part + at_xor32(part);
    ls[qb] = live ? lse[row] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[qb][r] = 0.f;
  }
  for (int k0 = 0; k0 < Nk; k0 += AT_KT) {
    const int kn = min(AT_KT, Nk - k0);
    __syncthreads();
    for (int e = tid; e < AT_KT * AT_DK; e += WAVES * 64) {
      const int key = e >> 3, d = e & 7;
      ks[key * AT_KS + d] = key < kn ? phi[((size_t)b * Nk + k0 + key) * AT_DK + d] : 0.f;
    }
    for (int e = tid; e < AT_KT * AT_DV; e += WAVES * 64) {
      const int key = e >> 5, cc = e & 31;
      vs[key * AT_VS + cc] = key < kn ? g[((size_t)b * Nk + k0 + key) * AT_DV + cc] : 0.f;
    }
    __syncthreads();
    const int nblk = (kn + 31) >> 5;
    for (int kb = 0; kb < nblk; ++kb) {
      float ka[4], vc[16], kt[16];
#pragma unroll
      for (int t = 0; t < 4; ++t) ka[t] = ks[(kb * 32 + c) * AT_KS + 2 * t + h];
#pragma unroll
      for (int t = 0; t < 16; ++t) vc[t] = vs[(kb * 32 + c) * AT_VS + 2 * t + h];          // A of dP^T = V dO^T: [key][c pair t]
#pragma unroll
      for (int u = 0; u < 16; ++u) kt[u] = ks[(kb * 32 + at_row(u, h)) * AT_KS + (c & 7)];  // A of dQ^T = K^T dS^T: [d][key pair u]
      const bool ragged = kb * 32 + 32 > kn;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 4; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], qf[qb][t], s, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[t], dof[qb][t], dp, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          float p = at_exp(s[u] - ls[qb]);
          if (ragged && kb * 32 + at_row(u, h) >= kn) p = 0.f;
          s[u] = p * (dp[u] - dl[qb]);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) dq[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[u], s[u], dq[qb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q_base + qb * 32 + c;
    if (q < Nq) {
      const size_t row = (size_t)b * Nq + q;      // dQ^T register r (< 4) of half h <-> d = 4 h + r
      *reinterpret_cast<float4*>(dtheta + row * AT_DK + 4 * h) = make_float4(dq[qb][0], dq[qb][1], dq[qb][2], dq[qb][3]);
      if (h == 0) delta[row] = dl[qb];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dkv sweep, lane = key: dg_j = sum_i p_ij dO_i ; dphi_j = sum_i p_ij (dP_ij - delta_i) theta_i.
// gridDim.z > 1: the query range is split over z (small batches would otherwise leave most CUs idle) and the partial
// dphi / dg rows are added with float atomics into pre-zeroed outputs.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_attn_bwd_dkv(const float* __restrict__ theta, const float* __restrict__ phi,
                                                             const float* __restrict__ g, const float* __restrict__ lse,
                                                             const float* __restrict__ dout, const float* __restrict__ delta,
                                                             float* __restrict__ dphi, float* __restrict__ dg, int Nq, int Nk, int q_chunk) {
  __shared__ float qs[AT_QT * AT_KS];
  __shared__ float dos[AT_QT * AT_VS];
  __shared__ __attribute__((aligned(16))) float lss[AT_QT];
  __shared__ __attribute__((aligned(16))) float des[AT_QT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
  const int q_begin = blockIdx.z * q_chunk, q_end = min(Nq, q_begin + q_chunk);
  const int b = blockIdx.y;
  const int key = (blockIdx.x * WAVES + wave) * 32 + c;
  const bool live = key < Nk;
  const size_t krow = (size_t)b * Nk + (live ? key : 0);
  float kf[4], vf[16];
#pragma unroll
  for (int t = 0; t < 4; ++t) kf[t] = live ? phi[krow * AT_DK + 2 * t + h] : 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) vf[t] = live ? g[krow * AT_DV + 2 * t + h] : 0.f;
  f32x16 dv, dk;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
  for (int q0 = q_begin; q0 < q_end; q0 += AT_QT) {
    const int qn = min(AT_QT, q_end - q0);
    __syncthreads();
    for (int e = tid; e < AT_QT * AT_DK; e += WAVES * 64) {
      const int i = e >> 3, d = e & 7;
      qs[i * AT_KS + d] = i < qn ? theta[((size_t)b * Nq + q0 + i) * AT_DK + d] : 0.f;
    }
    for (int e = tid; e < AT_QT * AT_DV; e += WAVES * 64) {
      const int i = e >> 5, cc = e & 31;
      dos[i * AT_VS + cc] = i < qn ? dout[((size_t)b * Nq + q0 + i) * AT_DV + cc] : 0.f;
    }
    for (int i = tid; i < AT_QT; i += WAVES * 64) {
      lss[i] = i < qn ? lse[(size_t)b * Nq + q0 + i] : INFINITY;      // p = exp(s - inf) = 0 for padding
      des[i] = i < qn ? delta[(size_t)b * Nq + q0 + i] : 0.f;
    }
    __syncthreads();
    const int nblk = (qn + 31) >> 5;
    for (int qb = 0; qb < nblk; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[(qb * 32 + c) * AT_KS + 2 * t + h], kf[t], s, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 16; ++t) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dos[(qb * 32 + c) * AT_VS + 2 * t + h], vf[t], dp, 0, 0, 0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {           // registers 4 g .. 4 g + 3 <-> queries 8 g + 4 h + 0 .. 3
        const float4 l4 = *reinterpret_cast<const float4*>(lss + qb * 32 + 8 * gq + 4 * h);
        const float4 d4 = *reinterpret_cast<const float4*>(des + qb * 32 + 8 * gq + 4 * h);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = at_exp(s[4 * gq + j] - lv[j]);
          s[4 * gq + j] = p;
          dp[4 * gq + j] = p * (dp[4 * gq + j] - dvv[j]);
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        dv = __builtin_amdgcn_mfma_f32_32x32x2f32(dos[(qb * 32 + at_row(u, h)) * AT_VS + c], s[u], dv, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 16; ++u)
        dk = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[(qb * 32 + at_row(u, h)) * AT_KS + (c & 7)], dp[u], dk, 0, 0, 0);
    }
  }
  if (live) {
    float* gp = dg + krow * AT_DV + 4 * h;        // dV^T register 4 g + j <-> channel 8 g + 4 h + j
    float* kp = dphi + krow * AT_DK + 4 * h;      // dK^T register r (< 4) <-> d = 4 h + r
    if (gridDim.z > 1) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(gp + 8 * gq + j, dv[4 * gq + j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(kp + j, dk[j]);
    } else {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(gp + 8 * gq) = make_float4(dv[4 * gq], dv[4 * gq + 1], dv[4 * gq + 2], dv[4 * gq + 3]);
      *reinterpret_cast<float4*>(kp) = make_float4(dk[0], dk[1], dk[2], dk[3]);
    }
  }
}

// Queries per workgroup: the widest configuration (the key tiles are staged once per workgroup) that still gives every CU work
static inline int at_q_cfg(int Nq, int B) {
  if ((long)sg_cdiv(Nq, 256) * B >= 512) return 256;
  if ((long)sg_cdiv(Nq, 128) * B >= 320) return 128;
  if ((long)sg_cdiv(Nq, 64) * B >= 320) return 64;
  return 32;
}

// theta [B,Nq,8], phi [B,Nk,8], g [B,Nk,32] -> out [B,Nq,32], lse [B,Nq]
extern "C" int sg_attention_fwd(const float* theta, const float* phi, const float* g, float* out, float* lse, int B, int Nq,
                                int Nk, int dk, int dv, void* stream) {
  if (!theta || !phi || !g || !out || !lse || dk != AT_DK || dv != AT_DV || Nk < 1) return SG_ERR_ARG;
  if (B < 1 || Nq < 1) return SG_OK;
  const int cfg = at_q_cfg(Nq, B);
  hipStream_t s = (hipStream_t)stream;
  if (cfg == 256) SG_KERNEL((k_attn_fwd<4, 2>), dim3(sg_cdiv(Nq, 256), B), dim3(256), 0, s, theta, phi, g, out, lse, Nq, Nk);
  else if (cfg == 128) SG_KERNEL((k_attn_fwd<4, 1>), dim3(sg_cdiv(Nq, 128), B), dim3(256), 0, s, theta, phi, g, out, lse, Nq, Nk);
  else if (cfg == 64) SG_KERNEL((k_attn_fwd<2, 1>), dim3(sg_cdiv(Nq, 64), B), dim3(128), 0, s, theta, phi, g, out, lse, Nq, Nk);
  else SG_KERNEL((k_attn_fwd<1, 1>), dim3(sg_cdiv(Nq, 32), B), dim3(64), 0, s, theta, phi, g, out, lse, Nq, Nk);
  return sg_launch_status();
}

// delta is a [B,Nq] scratch vector
extern "C" int sg_attention_bwd(const float* theta, const float* phi, const float* g, const float* out, const float* lse,
                                const float* dout, float* dtheta, float* dphi, float* dg, float* delta, int B, int Nq, int Nk,
                                int dk, int dv, void* stream) {
  if (!theta || !phi || !g || !out || !lse || !dout || !dtheta || !dphi || !dg || !delta || dk != AT_DK || dv != AT_DV || Nk < 1)
    return SG_ERR_ARG;
  if (B < 1 || Nq < 1) return SG_OK;
  hipStream_t s = (hipStream_t)stream;
  const int cfg = at_q_cfg(Nq, B);
  if (cfg == 256)
    SG_KERNEL((k_attn_bwd_dq<4, 2>), dim3(sg_cdiv(Nq, 256), B), dim3(256), 0, s, theta, phi, g, out, lse, dout, dtheta, delta, Nq, Nk);
  else if (cfg == 128)
    SG_KERNEL((k_attn_bwd_dq<4, 1>), dim3(sg_cdiv(Nq, 128), B), dim3(256), 0, s, theta, phi, g, out, lse, dout, dtheta, delta, Nq, Nk);
  else if (cfg == 64)
    SG_KERNEL((k_attn_bwd_dq<2, 1>), dim3(sg_cdiv(Nq, 64), B), dim3(128), 0, s, theta, phi, g, out, lse, dout, dtheta, delta, Nq, Nk);
  else
    SG_KERNEL((k_attn_bwd_dq<1, 1>), dim3(sg_cdiv(Nq, 32), B), dim3(64), 0, s, theta, phi, g, out, lse, dout, dtheta, delta, Nq, Nk);
  // key sweep: every workgroup stages ALL queries of its z slice, so the more keys it owns the better that is amortised:
  // the widest of 128 / 64 / 32 keys per workgroup that still gives two rounds of workgroups and wastes the fewest padded
  // key lanes; then split the query range until the grid fills the chip
  int kw = 1;
  long best_pad = -1;
  for (int w = 4; w >= 1; w >>= 1) {
    const long blocks = (long)sg_cdiv(Nk, 32 * w) * B;
    const long pad = (long)sg_cdiv(Nk, 32 * w) * 32 * w - Nk;
    if (blocks >= 320 && (best_pad < 0 || pad < best_pad)) { kw = w; best_pad = pad; }
  }
  const long kblocks = (long)sg_cdiv(Nk, 32 * kw) * B;
  int zs = kblocks >= 320 ? 1 : (int)((512 + kblocks - 1) / kblocks);
  // ... and until every SIMD has two waves (1 024 SIMDs): at the 8-way shard batch the generator site (16 x 1 280 keys) gave
  // 320 workgroups of 2 waves -- 0.6 waves per SIMD, nothing to hide the LDS staging and the barriers behind
  const long waves = kblocks * kw;
  if (waves * zs < 2048 && (long)Nq * Nk >= (1L << 22)) zs = (int)((2048 + waves - 1) / waves);      // (large maps only)
  if (zs > 16) zs = 16;
  if (sg_deterministic()) zs = 1;                       // (no query split: every dphi / dg row has one writer, plain stores)
  int q_chunk = sg_cdiv(sg_cdiv(Nq, zs), AT_QT) * AT_QT;        // whole LDS tiles per z slice
  zs = sg_cdiv(Nq, q_chunk);
  if (zs > 1) {
    if (hipMemsetAsync(dphi, 0, sizeof(float) * (size_t)B * Nk * AT_DK, s) != hipSuccess ||
        hipMemsetAsync(dg, 0, sizeof(float) * (size_t)B * Nk * AT_DV, s) != hipSuccess)
      return SG_ERR_LAUNCH;
  }
  const dim3 grid(sg_cdiv(Nk, 32 * kw), B, zs);
  if (kw == 4) SG_KERNEL((k_attn_bwd_dkv<4>), grid, dim3(256), 0, s, theta, phi, g, lse, dout, delta, dphi, dg, Nq, Nk, q_chunk);
  else if (kw == 2) SG_KERNEL((k_attn_bwd_dkv<2>), grid, dim3(128), 0, s, theta, phi, g, lse, dout, delta, dphi, dg, Nq, Nk, q_chunk);
  else SG_KERNEL((k_attn_bwd_dkv<1>), grid, dim3(64), 0, s, theta, phi, g, lse, dout, delta, dphi, dg, Nq, Nk, q_chunk);
  return sg_launch_status();
}
