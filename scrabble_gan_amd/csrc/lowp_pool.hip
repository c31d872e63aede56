// Low-precision operand copies written by the kernels that already stream the tensor (configs c3 / c5):
// the backward of the 2x2 average pool of a ResNetBlockDown (resnet_ops.py:105-106) feeds ONLY convolution launches --
// conv2's weight-grad and data-grad -- which in bf16 / fp8 mode read bf16 / fp8 operand copies.  d_c2[b,y,x,c] =
// 0.25 * dout[b,y/2,x/2,c] is a 4x replication, so instead of writing it in fp32 (4 B/element), re-reading it for an
// amax pass (fp8) and again for the conversion pass (4 + 4 B/element), these kernels read dout (a quarter of the pixels)
// and write the operand copies directly: the plain copy (data-grad operand) and the copy with the per-sample factors
// of the shared backward sweep folded in (weight-grad operand).  Scaling by 0.25 is exact, so the copies are BIT-IDENTICAL
// to what sg_avgpool2_bwd + sg_cvt_bf16 / sg_amax2_f32 + sg_cvt_fp8_grad produce (tests compare them).
#include "sg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ unsigned lp_pack4_e5m2(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(a, -57344.f), 57344.f), fminf(fmaxf(b, -57344.f), 57344.f), w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(c, -57344.f), 57344.f), fminf(fmaxf(d, -57344.f), 57344.f), w, true);
  return (unsigned)w;
}

__device__ __forceinline__ unsigned lp_pack4_e4m3(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -448.f), 448.f), fminf(fmaxf(b, -448.f), 448.f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -448.f), 448.f), fminf(fmaxf(d, -448.f), 448.f), w, true);
  return (unsigned)w;
}

// FP8 = false: out_a = bf16(0.25 dout) (nullable), out_b = bf16(rowscale[b] * 0.25 dout) (nullable)
// FP8 = true : out_a = e4m3(0.25 dout * 448 / (0.25 amax_in[0])) (nullable), out_b = e5m2(rowscale[b] * 0.25 dout * 57344 / (0.25 amax_in[1]));
//              amax_out[0..1] = 0.25 * amax_in[0..1] (the amax scalars of the two copies)
template <bool FP8>
__global__ __launch_bounds__(256) void k_avgpool2_bwd_lowp(const float* __restrict__ dout, void* __restrict__ out_a, void* __restrict__ out_b,
                                                           const float* __restrict__ rowscale, const float* __restrict__ amax_in,
                                                           float* __restrict__ amax_out, int B, int H, int W, int C) {
  const int c8n = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * H * W * c8n;
  float s4 = 1.f, s5 = 1.f;
  if constexpr (FP8) {
    const float a0 = 0.25f * amax_in[0], a1 = 0.25f * amax_in[1];
    s4 = a0 > 0.f ? 448.f / a0 : 1.f;
    s5 = a1 > 0.f ? 57344.f / a1 : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) { amax_out[0] = a0; amax_out[1] = a1; }
  }
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = 8 * (int)(e % c8n);
    long r = e / c8n;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int bb = (int)(r / H);
    const float4* src = reinterpret_cast<const float4*>(dout + (((size_t)bb * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c);
    float4 v0 = src[0], v1 = src[1];
    v0.x *= 0.25f; v0.y *= 0.25f; v0.z *= 0.25f; v0.w *= 0.25f; v1.x *= 0.25f; v1.y *= 0.25f; v1.z *= 0.25f; v1.w *= 0.25f;
    if (out_a) {
      if constexpr (FP8) {
        reinterpret_cast<uint2*>(out_a)[e] = make_uint2(lp_pack4_e4m3(v0.x * s4, v0.y * s4, v0.z * s4, v0.w * s4),
                                                        lp_pack4_e4m3(v1.x * s4, v1.y * s4, v1.z * s4, v1.w * s4));
      } else {
        bf16x8 h;
        h[0] = (__bf16)v0.x; h[1] = (__bf16)v0.y; h[2] = (__bf16)v0.z; h[3] = (__bf16)v0.w;
        h[4] = (__bf16)v1.x; h[5] = (__bf16)v1.y; h[6] = (__bf16)v1.z; h[7] = (__bf16)v1.w;
        reinterpret_cast<bf16x8*>(out_a)[e] = h;
      }
    }
    if (out_b) {
      if (rowscale) {
        const float f = rowscale[bb];
        v0.x *= f; v0.y *= f; v0.z *= f; v0.w *= f; v1.x *= f; v1.y *= f; v1.z *= f; v1.w *= f;
      }
      if constexpr (FP8) {
        reinterpret_cast<uint2*>(out_b)[e] = make_uint2(lp_pack4_e5m2(v0.x * s5, v0.y * s5, v0.z * s5, v0.w * s5),
                                                        lp_pack4_e5m2(v1.x * s5, v1.y * s5, v1.z * s5, v1.w * s5));
      } else {
        bf16x8 h;
        h[0] = (__bf16)v0.x; h[1] = (__bf16)v0.y; h[2] = (__bf16)v0.z; h[3] = (__bf16)v0.w;
        h[4] = (__bf16)v1.x; h[5] = (__bf16)v1.y; h[6] = (__bf16)v1.z; h[7] = (__bf16)v1.w;
        reinterpret_cast<bf16x8*>(out_b)[e] = h;
      }
    }
  }
}

// dx16 [B,H,W,C] bf16 = bf16(0.25 * dout[b,y/2,x/2,c]) (nullable), dx16_scaled = bf16(rowscale[b] * 0.25 * dout[...]) (nullable;
// rowscale nullable = no factor).  H, W = dims of dx; C % 8 == 0.
extern "C" int sg_avgpool2_bwd_bf16(const float* dout, void* dx16, void* dx16_scaled, const float* rowscale, int B, int H, int W, int C,
                                    void* stream) {
  if (!dout || (!dx16 && !dx16_scaled) || (H & 1) || (W & 1) || (C & 7)) return SG_ERR_ARG;
  const long n = (long)B * H * W * (C / 8);
  if (n == 0) return SG_OK;
  SG_KERNEL(k_avgpool2_bwd_lowp<false>, dim3(sg_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dout, dx16, dx16_scaled, rowscale,
                     (const float*)nullptr, (float*)nullptr, B, H, W, C);
  return sg_launch_status();
}

// dx_e4m3 (nullable) / dx_e5m2 [B,H,W,C]: the fp8 operand copies of 0.25 * dout[b,y/2,x/2,c] (dx_e5m2 with rowscale[b] folded in);
// amax_dout = {max |dout|, max |rowscale dout|} (sg_amax2_f32 on dout); amax_dx (2 floats) receives the amax scalars of the copies.
extern "C" int sg_avgpool2_bwd_fp8(const float* dout, void* dx_e4m3, void* dx_e5m2, const float* rowscale, const float* amax_dout,
                                   float* amax_dx, int B, int H, int W, int C, void* stream) {
  if (!dout || (!dx_e4m3 && !dx_e5m2) || !amax_dout || !amax_dx || (H & 1) || (W & 1) || (C & 7)) return SG_ERR_ARG;
  const long n = (long)B * H * W * (C / 8);
  if (n == 0) return SG_OK;
  SG_KERNEL(k_avgpool2_bwd_lowp<true>, dim3(sg_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dout, dx_e4m3, dx_e5m2, rowscale,
                     amax_dout, amax_dx, B, H, W, C);
  return sg_launch_status();
}
