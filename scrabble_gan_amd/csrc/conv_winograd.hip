// Winograd-domain 3x3 convolutions, fp32 (config c2 of BASELINE.json: the fp32 matrix rate bounds the step, so the lever left is
// the NUMBER of products).  A stride-1 SAME 3x3 convolution over m x m output tiles
//     Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray 2015; d = the (m+2) x (m+2) input patch of the tile, g = the filter)
// needs (m+2)^2 products per tile, channel pair and m^2 outputs where the direct form needs 9 m^2: F(2x2, 3x3) 16 per 4 outputs
// (first half of this file), F(4x4, 3x3) 36 per 16 outputs (second half; transforms generated into wino_f43.h).  The (m+2)^2
// "frequencies" are independent [tiles x Cin] x [Cin x Cout] matrix products.  Three steps:
//   1. input transform   x [B,H,W,C]           -> V [P][Tp][C]       (k_wino_in / k_w43_in; the operand ReLU of the pre-activation
//                                                                     blocks is applied to the loaded values)
//   2. ONE grouped launch of the DMA-fed implicit-GEMM kernel (conv_bf16v2.hip, fp32 operands, v_mfma_f32_32x32x2_f32): group f
//      multiplies V[f] with the transformed filter U[f] = [N][K] into Mt[f] -- the same loop and epilogue as the direct fp32
//      launch, as a 1x1 convolution over P Tp "pixels" on 128 x 128 tiles;
//   3. output transform  Mt [P][Tp][N]         -> y [B,H,W,N]        (k_wino_out / k_w43_out: then the conv entry points' epilogue --
//                                                                     bias, ReLU mask of the data-grad, accumulate, output ReLU).
// The data-grad of such a convolution is the same convolution of dy with the spatially flipped filter and the channel roles
// swapped, so it runs through the same three steps with U built from w [kh,kw,Cin,Cout] as it lies (flip = 1).
// The transforms are HBM-bound sweeps (V and Mt are 4x the activation each for m = 2, 2.25x for m = 4).
// Numerics: exact fp32 products and fp32 accumulation like the direct kernels; the rounding inside the transforms is what differs
// (measured against the fp64 oracle: F(2x2) <= 1e-6 of max |ref|, F(4x4) <= 1.3e-5; tests/test_winograd_gpu.py holds both to the
// direct kernels' 2e-5).
// Tiles: t = (b * H/m + ty) * W/m + tx, T = B H/m W/m of them, planes padded to Tp = T rounded up to 128 rows (the tile height of
// the grouped product: a tile never straddles two frequencies; 128 rather than 256 rows because the 8-way shard batch has
// T = 320 on the 4x20 layers); the pad rows are never written nor read back.
#include "sg_conv2.h"
#include "wino_f43.h"
#include <stdlib.h>

#define WINO_F 16

static inline long wino_tp(long T) { return (T + 127) / 128 * 128; }      // (the grouped product runs on 128-row tiles)


// ---- filter transform: in [3][3][N][K] (fp32) -> U [16][N][K];  flip != 0 reads tap (2 - a, 2 - b) for (a, b)
__global__ __launch_bounds__(256) void k_wino_filter(const float* __restrict__ in, float* __restrict__ U, long NK, int flip) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < NK; e += (long)gridDim.x * 256) {
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = in[(size_t)((flip ? 2 - a : a) * 3 + (flip ? 2 - b : b)) * NK + e];
    float h[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      h[0][b] = g[0][b];
      h[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      h[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      h[3][b] = g[2][b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      U[(size_t)(i * 4 + 0) * NK + e] = h[i][0];
      U[(size_t)(i * 4 + 1) * NK + e] = 0.5f * (h[i][0] + h[i][1] + h[i][2]);
      U[(size_t)(i * 4 + 2) * NK + e] = 0.5f * (h[i][0] - h[i][1] + h[i][2]);
      U[(size_t)(i * 4 + 3) * NK + e] = h[i][2];
    }
  }
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// ---- input transform: one thread = one tile x four channels (a wave = 256 consecutive channels of a tile: 1 KB per access)
template <bool RELU>
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ x, float* __restrict__ V, int H, int W, int C, long T, long Tp) {
  const int C4 = C >> 2, H2 = H >> 1, W2 = W >> 1;
  const long items = T * C4;
  const size_t plane = (size_t)Tp * C;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const long t = it / C4;
    const int c = (int)(it - t * C4) * 4;
    const int tx = (int)(t % W2);
    const long r = t / W2;
    const int ty = (int)(r % H2);
    const long b = r / H2;
    float4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = 2 * ty - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ix = 2 * tx - 1 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4*>(x + ((size_t)(b * H + iy) * W + ix) * C + c);
        if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        d[i][j] = v;
      }
    }
    // B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    float4 u[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u[0][j] = f4_sub(d[0][j], d[2][j]);
      u[1][j] = f4_add(d[1][j], d[2][j]);
      u[2][j] = f4_sub(d[2][j], d[1][j]);
      u[3][j] = f4_sub(d[1][j], d[3][j]);
    }
    float* out = V + (size_t)t * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 0) * plane) = f4_sub(u[i][0], u[i][2]);
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 1) * plane) = f4_add(u[i][1], u[i][2]);
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 2) * plane) = f4_sub(u[i][2], u[i][1]);
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 3) * plane) = f4_sub(u[i][1], u[i][3]);
    }
  }
}

// ---- output transform + the conv entry points' epilogue: one thread = one tile x four output channels
__global__ __launch_bounds__(256) void k_wino_out(const float* __restrict__ Mt, float* __restrict__ y, const float* __restrict__ bias,
                                                  const float* __restrict__ bias2, const float* __restrict__ mask, int H, int W, int N, long T,
                                                  long Tp, int flags) {
  const int N4 = N >> 2, H2 = H >> 1, W2 = W >> 1;
  const long items = T * N4;
  const size_t plane = (size_t)Tp * N;
  const bool accum = (flags & SG_ACCUM) != 0, relu_out = (flags & SG_RELU_OUT) != 0;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const long t = it / N4;
    const int n = (int)(it - t * N4) * 4;
    const int tx = (int)(t % W2);
    const long r = t / W2;
    const int ty = (int)(r % H2);
    const long b = r / H2;
    const float* src = Mt + (size_t)t * N + n;
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const float4*>(src + (size_t)(i * 4 + j) * plane);
    // A^T m A with A^T = [1 1 1 0; 0 1 -1 -1]
    float4 q[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q[0][j] = f4_add(f4_add(m[0][j], m[1][j]), m[2][j]);
      q[1][j] = f4_sub(f4_sub(m[1][j], m[2][j]), m[3][j]);
    }
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bs = f4_add(bs, *reinterpret_cast<const float4*>(bias + n));
    if (bias2) bs = f4_add(bs, *reinterpret_cast<const float4*>(bias2 + n));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 v = j == 0 ? f4_add(f4_add(q[i][0], q[i][1]), q[i][2]) : f4_sub(f4_sub(q[i][1], q[i][2]), q[i][3]);
        v = f4_add(v, bs);
        const size_t idx = ((size_t)(b * H + 2 * ty + i) * W + 2 * tx + j) * N + n;
        if (mask) {
          const float4 k = *reinterpret_cast<const float4*>(mask + idx);
          if (k.x <= 0.f) v.x = 0.f;
          if (k.y <= 0.f) v.y = 0.f;
          if (k.z <= 0.f) v.z = 0.f;
          if (k.w <= 0.f) v.w = 0.f;
        }
        if (accum) v = f4_add(v, *reinterpret_cast<const float4*>(y + idx));
        if (relu_out) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *reinterpret_cast<float4*>(y + idx) = v;
      }
  }
}

// ---- weight gradient.  With M = sum_c U (.) V and Y = A^T M A:  dM = A dY A^T (4x4 from the tile's 2x2 output gradients),
//      dU[f][c][n] = sum_tiles V[f][t][c] dM[f][t][n] (sixteen [K x T] x [T x N] products: the fp32 weight-grad kernel of
//      conv_wgrad.hip in its grouped form), dW = G^T dU G.
// gradient transform: dy [B,H,W,N] -> Qt [16][Tp][N], rows scaled by sample_scale[b] (nullable); db (nullable) += column sums of
// the scaled dy.  One thread = one tile x four channels; A = [1 0; 1 1; 1 -1; 0 -1].
// db_part (nullable): WINO_DB_PARTS x N scratch rows, zeroed by the caller -- block b adds its column sums to row b % WINO_DB_PARTS and
// k_wino_db_finish folds the rows into db: chains of gridDim / 64 adders per address instead of gridDim (2 048 same-address float
// atomics cost 0.15 ms per launch: the gradient transform ran 2.7x slower than the input transform of the same tensor)
#define WINO_DB_PARTS 64
__global__ __launch_bounds__(256) void k_wino_db_finish(const float* __restrict__ db_part, float* __restrict__ db, int N) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int r = 0; r < WINO_DB_PARTS; ++r) acc += db_part[(size_t)r * N + n];
  db[n] += acc;
}

__global__ __launch_bounds__(256) void k_wino_dy(const float* __restrict__ dy, float* __restrict__ Qt, const float* __restrict__ sample_scale,
                                                 float* __restrict__ db_in, float* __restrict__ db_part, int H, int W, int N, long T, long Tp) {
  float* const db = db_in ? (db_part ? db_part + (size_t)(blockIdx.x % WINO_DB_PARTS) * N : db_in) : nullptr;
  __shared__ float4 red[256];
  const int N4 = N >> 2, H2 = H >> 1, W2 = W >> 1;
  const long items = T * N4;
  const size_t plane = (size_t)Tp * N;
  // a thread keeps ONE channel group over its whole grid-stride loop when the stride is a multiple of N4 (every power-of-two
  // channel count): threads tid, tid + N4, ... of a block then share it; otherwise the column sums go out per item
  const bool fixed_col = ((long)gridDim.x * 256) % N4 == 0;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const long t = it / N4;
    const int n = (int)(it - t * N4) * 4;
    const int tx = (int)(t % W2);
    const long r = t / W2;
    const int ty = (int)(r % H2);
    const long b = r / H2;
    const float sc = sample_scale ? sample_scale[b] : 1.f;
    float4 d[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 v = *reinterpret_cast<const float4*>(dy + ((size_t)(b * H + 2 * ty + i) * W + 2 * tx + j) * N + n);
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        d[i][j] = v;
      }
    if (db) {
      const float4 s4 = f4_add(f4_add(d[0][0], d[0][1]), f4_add(d[1][0], d[1][1]));
      if (fixed_col) bsum = f4_add(bsum, s4);
      else { atomicAdd(db + n, s4.x); atomicAdd(db + n + 1, s4.y); atomicAdd(db + n + 2, s4.z); atomicAdd(db + n + 3, s4.w); }
    }
    float4 rr[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      rr[0][j] = d[0][j];
      rr[1][j] = f4_add(d[0][j], d[1][j]);
      rr[2][j] = f4_sub(d[0][j], d[1][j]);
      rr[3][j] = make_float4(-d[1][j].x, -d[1][j].y, -d[1][j].z, -d[1][j].w);
    }
    float* out = Qt + (size_t)t * N + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 0) * plane) = rr[i][0];
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 1) * plane) = f4_add(rr[i][0], rr[i][1]);
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 2) * plane) = f4_sub(rr[i][0], rr[i][1]);
      *reinterpret_cast<float4*>(out + (size_t)(i * 4 + 3) * plane) = make_float4(-rr[i][1].x, -rr[i][1].y, -rr[i][1].z, -rr[i][1].w);
    }
  }
  if (db && fixed_col) {          // (block-uniform) threads tid, tid + N4, ... of the block hold the same channel group when N4 < 256
    red[threadIdx.x] = bsum;
    __syncthreads();
    const int per = N4 < 256 ? N4 : 256;
    if ((int)threadIdx.x < per) {
      float4 s4 = red[threadIdx.x];
      for (int k = threadIdx.x + per; k < 256; k += per) s4 = f4_add(s4, red[k]);
      const int n = (int)((((long)blockIdx.x * 256 + threadIdx.x) % N4) * 4);
      atomicAdd(db + n, s4.x); atomicAdd(db + n + 1, s4.y); atomicAdd(db + n + 2, s4.z); atomicAdd(db + n + 3, s4.w);
    }
  }
}

// filter-gradient transform: dw [3][3][K][N] += G^T dU G, dU [16][K][N]
__global__ __launch_bounds__(256) void k_wino_dw(const float* __restrict__ dU, float* __restrict__ dw, long KN) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < KN; e += (long)gridDim.x * 256) {
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) u[i][j] = dU[(size_t)(i * 4 + j) * KN + e];
    float h[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
      h[1][j] = 0.5f * (u[1][j] - u[2][j]);
      h[2][j] = u[3][j] + 0.5f * (u[1][j] + u[2][j]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      dw[(size_t)(a * 3 + 0) * KN + e] += h[a][0] + 0.5f * (h[a][1] + h[a][2]);
      dw[(size_t)(a * 3 + 1) * KN + e] += 0.5f * (h[a][1] - h[a][2]);
      dw[(size_t)(a * 3 + 2) * KN + e] += h[a][3] + 0.5f * (h[a][1] + h[a][2]);
    }
  }
}

// ==========================================================================================================
// F(4x4, 3x3): 36 products per 4x4 output tile (2.25 per output against F(2x2)'s 4 and the direct form's 9), 6x6 input patches,
// V / Mt are 2.25x the activation (F(2x2): 4x).  1-D transforms from wino_f43.h (generated: Cook-Toom over {0, +-1/2, +-3/2, inf},
// the point set with the smallest fp32 error in an emulation of this pipeline -- tools/gen_winograd_f43.py), applied along the
// columns, then along the rows.  Same kernels-and-planes structure as above with 36 planes; tiles t = (b * H/4 + ty) * W/4 + tx.
// ==========================================================================================================
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// VT = the channels one thread carries through a tile's transform: v4f (16-byte accesses, 144 live registers for the 6x6 patch: two
// waves per SIMD) or v2f (8-byte accesses, half the registers, twice the waves) -- SG_WINO_VEC picks, measured in DESIGN 3e
template <typename VT> __device__ __forceinline__ VT ldv(const float* p) { return *reinterpret_cast<const VT*>(p); }
template <typename VT> __device__ __forceinline__ void stv(float* p, VT v) { *reinterpret_cast<VT*>(p) = v; }
template <typename VT> __device__ __forceinline__ VT zerov() { VT z; for (int i = 0; i < (int)(sizeof(VT) / 4); ++i) z[i] = 0.f; return z; }
template <typename VT> __device__ __forceinline__ VT reluv(VT v) { return __builtin_elementwise_max(v, zerov<VT>()); }

__global__ __launch_bounds__(256) void k_w43_filter(const float* __restrict__ in, float* __restrict__ U, long NK, int flip) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < NK; e += (long)gridDim.x * 256) {
    float g[3][3], h[6][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = in[(size_t)((flip ? 2 - a : a) * 3 + (flip ? 2 - b : b)) * NK + e];
#pragma unroll
    for (int b = 0; b < 3; ++b) w43_g(g[0][b], g[1][b], g[2][b], h[0][b], h[1][b], h[2][b], h[3][b], h[4][b], h[5][b]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float u0, u1, u2, u3, u4, u5;
      w43_g(h[i][0], h[i][1], h[i][2], u0, u1, u2, u3, u4, u5);
      float* o = U + (size_t)(i * 6) * NK + e;
      o[0] = u0; o[(size_t)NK] = u1; o[(size_t)2 * NK] = u2; o[(size_t)3 * NK] = u3; o[(size_t)4 * NK] = u4; o[(size_t)5 * NK] = u5;
    }
  }
}

// UPS (round 4): x is the HALF-resolution tensor [B, H/2, W/2, C] and the patch is taken from 0.25 * upsample2x2(x) -- the backward of a 2x2
// AVG pool folded into the transform of the gradient it produces (16 loads per tile instead of 36, no full-resolution gradient in HBM)
template <bool RELU, typename VT, bool UPS = false>
__global__ __launch_bounds__(256) void k_w43_in(const float* __restrict__ x, float* __restrict__ V, int H, int W, int C, long T, long Tp) {
  constexpr int VW = sizeof(VT) / 4;
  const int CV = C / VW, H4 = H >> 2, W4 = W >> 2;
  const long items = T * CV;
  const size_t plane = (size_t)Tp * C;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const long t = it / CV;
    const int c = (int)(it - t * CV) * VW;
    const int tx = (int)(t % W4);
    const long r = t / W4;
    const int ty = (int)(r % H4);
    const long b = r / H4;
    VT u[6][6];                                    // columns transformed: u[i][j] = sum_a BT[i][a] d[a][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int ix = 4 * tx - 1 + j;
      VT d[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int iy = 4 * ty - 1 + a;
        VT v = zerov<VT>();
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          if constexpr (UPS) v = ldv<VT>(x + ((size_t)(b * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1)) * C + c) * 0.25f;
          else v = ldv<VT>(x + ((size_t)(b * H + iy) * W + ix) * C + c);
        }
        d[a] = RELU ? reluv<VT>(v) : v;
      }
      w43_bt(d[0], d[1], d[2], d[3], d[4], d[5], u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j]);
    }
    float* out = V + (size_t)t * C + c;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      VT o0, o1, o2, o3, o4, o5;
      w43_bt(u[i][0], u[i][1], u[i][2], u[i][3], u[i][4], u[i][5], o0, o1, o2, o3, o4, o5);
      float* q = out + (size_t)(i * 6) * plane;
      stv<VT>(q, o0); stv<VT>(q + plane, o1); stv<VT>(q + 2 * plane, o2); stv<VT>(q + 3 * plane, o3); stv<VT>(q + 4 * plane, o4); stv<VT>(q + 5 * plane, o5);
    }
  }
}

template <typename VT, bool POOL = false>
__global__ __launch_bounds__(256) void k_w43_out(const float* __restrict__ Mt, float* __restrict__ y, const float* __restrict__ bias,
                                                 const float* __restrict__ bias2, const float* __restrict__ mask, int H, int W, int N, long T,
                                                 long Tp, int flags) {
  constexpr int VW = sizeof(VT) / 4;
  const int N4 = N / VW, H4 = H >> 2, W4 = W >> 2;
  const long items = T * N4;
  const size_t plane = (size_t)Tp * N;
  const bool accum = (flags & SG_ACCUM) != 0, relu_out = (flags & SG_RELU_OUT) != 0;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const long t = it / N4;
    const int n = (int)(it - t * N4) * VW;
    const int tx = (int)(t % W4);
    const long r = t / W4;
    const int ty = (int)(r % H4);
    const long b = r / H4;
    const float* src = Mt + (size_t)t * N + n;
    VT q[4][6];                                   // columns reduced: q[i][j] = sum_a AT[i][a] m[a][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      VT m[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) m[a] = ldv<VT>(src + (size_t)(a * 6 + j) * plane);
      w43_at(m[0], m[1], m[2], m[3], m[4], m[5], q[0][j], q[1][j], q[2][j], q[3][j]);
    }
    VT bs = zerov<VT>();
    if (bias) bs += ldv<VT>(bias + n);
    if (bias2) bs += ldv<VT>(bias2 + n);
    if constexpr (POOL) {
      // SG_POOL2_OUT: the 4 x 4 tile's 2 x 2 means (+ bias: a mean of conv + b is the mean of conv, + b) go to the pooled tensor
      // [B, H/2, W/2, N]; no mask / output ReLU in this form
      const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        VT o0[4], o1[4];
        w43_at(q[2 * pi][0], q[2 * pi][1], q[2 * pi][2], q[2 * pi][3], q[2 * pi][4], q[2 * pi][5], o0[0], o0[1], o0[2], o0[3]);
        w43_at(q[2 * pi + 1][0], q[2 * pi + 1][1], q[2 * pi + 1][2], q[2 * pi + 1][3], q[2 * pi + 1][4], q[2 * pi + 1][5], o1[0], o1[1], o1[2], o1[3]);
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
          VT v = ((o0[2 * pj] + o0[2 * pj + 1]) + (o1[2 * pj] + o1[2 * pj + 1])) * 0.25f + bs;
          const size_t idx = ((size_t)(b * Hp + 2 * ty + pi) * Wp + 2 * tx + pj) * N + n;
          if (accum) v += ldv<VT>(y + idx);
          stv<VT>(y + idx, v);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        VT o[4];
        w43_at(q[i][0], q[i][1], q[i][2], q[i][3], q[i][4], q[i][5], o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          VT v = o[j] + bs;
          const size_t idx = ((size_t)(b * H + 4 * ty + i) * W + 4 * tx + j) * N + n;
          if (mask) {
            const VT k = ldv<VT>(mask + idx);
#pragma unroll
            for (int e = 0; e < VW; ++e)
              if (k[e] <= 0.f) v[e] = 0.f;
          }
          if (accum) v += ldv<VT>(y + idx);
          if (relu_out) v = reluv<VT>(v);
          stv<VT>(y + idx, v);
        }
      }
    }
  }
}

template <typename VT, bool UPS = false>
__global__ __launch_bounds__(256) void k_w43_dy(const float* __restrict__ dy, float* __restrict__ Qt, const float* __restrict__ sample_scale,
                                                float* __restrict__ db_in, float* __restrict__ db_part, int H, int W, int N, long T, long Tp) {
  float* const db = db_in ? (db_part ? db_part + (size_t)(blockIdx.x % WINO_DB_PARTS) * N : db_in) : nullptr;
  constexpr int VW = sizeof(VT) / 4;
  __shared__ float red[256 * VW];
  const int N4 = N / VW, H4 = H >> 2, W4 = W >> 2;
  const long items = T * N4;
  const size_t plane = (size_t)Tp * N;
  const bool fixed_col = ((long)gridDim.x * 256) % N4 == 0;          // (see k_wino_dy)
  VT bsum = zerov<VT>();
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const long t = it / N4;
    const int n = (int)(it - t * N4) * VW;
    const int tx = (int)(t % W4);
    const long r = t / W4;
    const int ty = (int)(r % H4);
    const long b = r / H4;
    const float sc = sample_scale ? sample_scale[b] : 1.f;
    VT rr[6][4];                                  // columns expanded: rr[i][j] = sum_a A[i][a] d[a][j]
    VT s4 = zerov<VT>();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      VT d[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if constexpr (UPS) d[a] = ldv<VT>(dy + ((size_t)(b * (H >> 1) + 2 * ty + (a >> 1)) * (W >> 1) + 2 * tx + (j >> 1)) * N + n) * (0.25f * sc);
        else d[a] = ldv<VT>(dy + ((size_t)(b * H + 4 * ty + a) * W + 4 * tx + j) * N + n) * sc;
        s4 += d[a];
      }
      w43_a(d[0], d[1], d[2], d[3], rr[0][j], rr[1][j], rr[2][j], rr[3][j], rr[4][j], rr[5][j]);
    }
    if (db) {
      if (fixed_col) bsum += s4;
      else {
#pragma unroll
        for (int e = 0; e < VW; ++e) atomicAdd(db + n + e, s4[e]);
      }
    }
    float* out = Qt + (size_t)t * N + n;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      VT o0, o1, o2, o3, o4, o5;
      w43_a(rr[i][0], rr[i][1], rr[i][2], rr[i][3], o0, o1, o2, o3, o4, o5);
      float* q = out + (size_t)(i * 6) * plane;
      stv<VT>(q, o0); stv<VT>(q + plane, o1); stv<VT>(q + 2 * plane, o2); stv<VT>(q + 3 * plane, o3); stv<VT>(q + 4 * plane, o4); stv<VT>(q + 5 * plane, o5);
    }
  }
  if (db && fixed_col) {
#pragma unroll
    for (int e = 0; e < VW; ++e) red[threadIdx.x * VW + e] = bsum[e];
    __syncthreads();
    const int per = N4 < 256 ? N4 : 256;
    if ((int)threadIdx.x < per) {
      const int n = (int)((((long)blockIdx.x * 256 + threadIdx.x) % N4) * VW);
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        float acc = 0.f;
        for (int k = threadIdx.x; k < 256; k += per) acc += red[k * VW + e];
        atomicAdd(db + n + e, acc);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_w43_dw(const float* __restrict__ dU, float* __restrict__ dw, long KN) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < KN; e += (long)gridDim.x * 256) {
    float h[3][6];                                 // columns reduced: h[a][j] = sum_i GT[a][i] u[i][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float u[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) u[i] = dU[(size_t)(i * 6 + j) * KN + e];
      w43_gt(u[0], u[1], u[2], u[3], u[4], u[5], h[0][j], h[1][j], h[2][j]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float g0, g1, g2;
      w43_gt(h[a][0], h[a][1], h[a][2], h[a][3], h[a][4], h[a][5], g0, g1, g2);
      dw[(size_t)(a * 3 + 0) * KN + e] += g0;
      dw[(size_t)(a * 3 + 1) * KN + e] += g1;
      dw[(size_t)(a * 3 + 2) * KN + e] += g2;
    }
  }
}

// ------------------------------------------------------------------------------------------
// C-ABI (include/scrabble_hip.h).  tile = 2: F(2x2, 3x3), 16 planes, H and W even;  tile = 4: F(4x4, 3x3), 36 planes, H % 4 == W % 4 == 0.
// ------------------------------------------------------------------------------------------
static inline int wino_planes(int tile) { return (tile + 2) * (tile + 2); }
static int wino_vec() {
  static const int v = getenv("SG_WINO_VEC") ? atoi(getenv("SG_WINO_VEC")) : 4;
  return v == 2 ? 2 : 4;
}
static bool wino_geom_ok(int B, int H, int W, int tile) {
  return (tile == 2 || tile == 4) && B > 0 && H > 0 && W > 0 && !(H % tile) && !(W % tile);
}
static inline long wino_tiles(int B, int H, int W, int tile) { return (long)B * (H / tile) * (W / tile); }
static bool wino_shape_ok(int B, int H, int W, int K, int N, int tile) {
  return wino_geom_ok(B, H, W, tile) && K > 0 && N > 0 && !(K % 32) && !(N % 64);
}

extern "C" long sg_wino_plane_rows(int B, int H, int W, int tile) {
  return wino_geom_ok(B, H, W, tile) ? wino_tp(wino_tiles(B, H, W, tile)) : 0;
}

extern "C" long sg_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout, int tile) {
  if (!wino_geom_ok(B, H, W, tile) || Cin <= 0 || Cout <= 0) return 0;
  return (long)sizeof(float) * wino_planes(tile) * wino_tp(wino_tiles(B, H, W, tile)) * ((long)Cin + Cout);
}

extern "C" int sg_wino_filter(const float* w_nk, float* u, int N, int K, int flip, int tile, void* stream) {
  if (!w_nk || !u || N <= 0 || K <= 0 || (tile != 2 && tile != 4)) return SG_ERR_ARG;
  const long NK = (long)N * K;
  if (tile == 2) SG_KERNEL(k_wino_filter, dim3(sg_grid_for(NK, 256)), dim3(256), 0, (hipStream_t)stream, w_nk, u, NK, flip);
  else SG_KERNEL(k_w43_filter, dim3(sg_grid_for(NK, 256)), dim3(256), 0, (hipStream_t)stream, w_nk, u, NK, flip);
  return sg_launch_status();
}

// ---- the three steps as separate entry points (the host times them apart: two HBM-bound sweeps around one matrix-bound launch)
static int wino_input_impl(const float* x, float* V, int B, int H, int W, int C, int relu, int tile, int ups, void* stream);
extern "C" int sg_wino_input(const float* x, float* V, int B, int H, int W, int C, int relu, int tile, void* stream) {
  return wino_input_impl(x, V, B, H, W, C, relu, tile, 0, stream);
}
// x_half [B, H/2, W/2, C] -> V of 0.25 * upsample2x2(x_half) (tile = 4 only; no operand ReLU): the input transform of a data-grad whose
// gradient operand is the backward of a 2x2 AVG pool (resnet_ops.py:105-106), without the full-resolution gradient
extern "C" int sg_wino_input_ups(const float* x_half, float* V, int B, int H, int W, int C, int tile, void* stream) {
  if (tile != 4) return SG_ERR_UNSUPPORTED;
  return wino_input_impl(x_half, V, B, H, W, C, 0, tile, 1, stream);
}
static int wino_input_impl(const float* x, float* V, int B, int H, int W, int C, int relu, int tile, int ups, void* stream) {
  if (!x || !V) return SG_ERR_ARG;
  if (!wino_shape_ok(B, H, W, C, 64, tile)) return SG_ERR_UNSUPPORTED;
  if (ups) {
    const long Tu = wino_tiles(B, H, W, 4);
    SG_KERNEL((k_w43_in<false, v4f, true>), dim3(sg_grid_for(Tu * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, V, H, W, C, Tu, wino_tp(Tu));
    return sg_launch_status();
  }
  const long T = wino_tiles(B, H, W, tile), Tp = wino_tp(T);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(sg_grid_for(T * (C / 4), 256)), block(256);
  if (tile == 2) {
    if (relu) SG_KERNEL(k_wino_in<true>, grid, block, 0, s, x, V, H, W, C, T, Tp);
    else SG_KERNEL(k_wino_in<false>, grid, block, 0, s, x, V, H, W, C, T, Tp);
  } else {
    if (wino_vec() == 2) {
      const dim3 grid2(sg_grid_for(T * (C / 2), 256));
      if (relu) SG_KERNEL((k_w43_in<true, v2f>), grid2, block, 0, s, x, V, H, W, C, T, Tp);
      else SG_KERNEL((k_w43_in<false, v2f>), grid2, block, 0, s, x, V, H, W, C, T, Tp);
    } else {
      if (relu) SG_KERNEL((k_w43_in<true, v4f>), grid, block, 0, s, x, V, H, W, C, T, Tp);
      else SG_KERNEL((k_w43_in<false, v4f>), grid, block, 0, s, x, V, H, W, C, T, Tp);
    }
  }
  return sg_launch_status();
}

extern "C" int sg_wino_gemm(const float* V, const float* u, float* Mt, int B, int H, int W, int K, int N, int tile, void* stream) {
  if (!V || !u || !Mt) return SG_ERR_ARG;
  if (!wino_shape_ok(B, H, W, K, N, tile)) return SG_ERR_UNSUPPORTED;          // (N % 64 == 0: 128-wide tiles, or 64-wide for the 64-filter layers)
  const long Tp = wino_tp(wino_tiles(B, H, W, tile));
  const int F = wino_planes(tile);
  if (F * Tp >= (1L << 31) - 256) return SG_ERR_UNSUPPORTED;
  SgIgemm2Args g{};
  g.a = reinterpret_cast<const u16*>(V);
  g.w = reinterpret_cast<const u16*>(u);
  g.out = Mt;
  g.Bn = (int)(F * Tp); g.Ha = 1; g.Wa = 1; g.Ca = K; g.Hg = 1; g.Wg = 1; g.a_sy = 1; g.a_sx = 1;
  g.Ho = 1; g.Wo = 1; g.N = N; g.o_sy = 1; g.o_sx = 1; g.o_oy = 0; g.o_ox = 0;
  g.ntaps = 1; g.flags = 0;
  g.taps[0] = SgTap{0, 0, 0};
  g.group_rows = (int)Tp;
  g.a_group_bytes = (long)sizeof(float) * Tp * K;
  g.w_group_bytes = (long)sizeof(float) * N * K;
  return sg_launch_igemm_bf16v2(g, (hipStream_t)stream, nullptr, 4);
}

extern "C" int sg_wino_output(const float* Mt, float* y, const float* bias, const float* bias2, const float* mask, int B, int H, int W, int N,
                              int flags, int tile, void* stream) {
  if (!Mt || !y) return SG_ERR_ARG;
  if (!wino_shape_ok(B, H, W, 32, N, tile) || (flags & (SG_TANH_OUT | SG_RELU_IN))) return SG_ERR_UNSUPPORTED;
  if ((flags & SG_POOL2_OUT) && (tile != 4 || mask || (flags & SG_RELU_OUT))) return SG_ERR_UNSUPPORTED;
  const long T = wino_tiles(B, H, W, tile), Tp = wino_tp(T);
  const dim3 grid(sg_grid_for(T * (N / 4), 256)), block(256);
  if (flags & SG_POOL2_OUT) {
    SG_KERNEL((k_w43_out<v4f, true>), grid, block, 0, (hipStream_t)stream, Mt, y, bias, bias2, mask, H, W, N, T, Tp, flags);
    return sg_launch_status();
  }
  if (tile == 2) SG_KERNEL(k_wino_out, grid, block, 0, (hipStream_t)stream, Mt, y, bias, bias2, mask, H, W, N, T, Tp, flags);
  else if (wino_vec() == 2) SG_KERNEL(k_w43_out<v2f>, dim3(sg_grid_for(T * (N / 2), 256)), block, 0, (hipStream_t)stream, Mt, y, bias, bias2, mask, H, W, N, T, Tp, flags);
  else SG_KERNEL(k_w43_out<v4f>, grid, block, 0, (hipStream_t)stream, Mt, y, bias, bias2, mask, H, W, N, T, Tp, flags);
  return sg_launch_status();
}

// a [B,H,W,K] -> out [B,H,W,N] through the three steps; u [planes][N][K]
static int wino_conv(const float* a, const float* u, const float* bias, const float* bias2, const float* mask, float* out, int B, int H, int W,
                     int K, int N, int flags, int tile, void* workspace, long workspace_bytes, hipStream_t s) {
  if (!a || !u || !out || !workspace) return SG_ERR_ARG;
  if (!wino_shape_ok(B, H, W, K, N, tile) || (flags & SG_TANH_OUT)) return SG_ERR_UNSUPPORTED;
  if (workspace_bytes < sg_wino_workspace_bytes(B, H, W, K, N, tile)) return SG_ERR_ARG;
  const long Tp = wino_tp(wino_tiles(B, H, W, tile));
  float* V = reinterpret_cast<float*>(workspace);
  float* Mt = V + (size_t)wino_planes(tile) * Tp * K;
  int rc = wino_input_impl(a, V, B, H, W, K, (flags & SG_RELU_IN) != 0, tile, (flags & SG_UPS2_IN) != 0, s);
  if (rc != SG_OK) return rc;
  rc = sg_wino_gemm(V, u, Mt, B, H, W, K, N, tile, s);
  if (rc != SG_OK) return rc;
  return sg_wino_output(Mt, out, bias, bias2, mask, B, H, W, N, flags & ~(SG_RELU_IN | SG_UPS2_IN), tile, s);
}

extern "C" int sg_conv2d_fwd_wino(const float* x, const float* u_fwd, const float* bias, const float* bias2, float* y, int B, int H, int W,
                                  int Cin, int Cout, int flags, int tile, void* workspace, long workspace_bytes, void* stream) {
  return wino_conv(x, u_fwd, bias, bias2, nullptr, y, B, H, W, Cin, Cout, flags, tile, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int sg_conv2d_bwd_data_wino(const float* dy, const float* u_bwd, const float* mask, float* dx, int B, int H, int W, int Cin,
                                       int Cout, int flags, int tile, void* workspace, long workspace_bytes, void* stream) {
  if (flags & (SG_RELU_IN | SG_RELU_OUT | SG_POOL2_OUT)) return SG_ERR_ARG;
  if ((flags & SG_UPS2_IN) && tile != 4) return SG_ERR_UNSUPPORTED;
  return wino_conv(dy, u_bwd, nullptr, nullptr, mask, dx, B, H, W, Cout, Cin, flags, tile, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- weight gradient: workspace = V [planes][Tp][Cin] | Qt [planes][Tp][Cout] | dU [planes][Cin][Cout] | bias-gradient partial rows [64][Cout]
extern "C" long sg_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int tile) {
  if (!wino_geom_ok(B, H, W, tile) || Cin <= 0 || Cout <= 0) return 0;
  const long Tp = wino_tp(wino_tiles(B, H, W, tile));
  return (long)sizeof(float) * (wino_planes(tile) * (Tp * ((long)Cin + Cout) + (long)Cin * Cout) + (long)WINO_DB_PARTS * Cout);
}

static int wino_grad_input_impl(const float* dy, float* Qt, const float* sample_scale, float* db, float* db_scratch, int B, int H, int W,
                                int N, int tile, int ups, void* stream);
extern "C" int sg_wino_grad_input(const float* dy, float* Qt, const float* sample_scale, float* db, float* db_scratch, int B, int H, int W,
                                  int N, int tile, void* stream) {
  return wino_grad_input_impl(dy, Qt, sample_scale, db, db_scratch, B, H, W, N, tile, 0, stream);
}
// dy_half [B, H/2, W/2, N]: the gradient transform (and bias gradient) of 0.25 * upsample2x2(dy_half); tile = 4 only
extern "C" int sg_wino_grad_input_ups(const float* dy_half, float* Qt, const float* sample_scale, float* db, float* db_scratch, int B, int H,
                                      int W, int N, int tile, void* stream) {
  if (tile != 4) return SG_ERR_UNSUPPORTED;
  return wino_grad_input_impl(dy_half, Qt, sample_scale, db, db_scratch, B, H, W, N, tile, 1, stream);
}
static int wino_grad_input_impl(const float* dy, float* Qt, const float* sample_scale, float* db, float* db_scratch, int B, int H, int W,
                                int N, int tile, int ups, void* stream) {
  if (!dy || !Qt) return SG_ERR_ARG;
  if (!wino_geom_ok(B, H, W, tile) || (N & 3) || N <= 0) return SG_ERR_UNSUPPORTED;
  const long T = wino_tiles(B, H, W, tile), Tp = wino_tp(T);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(sg_grid_for(T * (N / 4), 256)), block(256);
  float* part = db ? db_scratch : nullptr;       // (db_scratch: WINO_DB_PARTS x N floats; null -> every block adds into db itself)
  if (part && hipMemsetAsync(part, 0, sizeof(float) * (size_t)WINO_DB_PARTS * N, s) != hipSuccess) return SG_ERR_LAUNCH;
  if (ups) SG_KERNEL((k_w43_dy<v4f, true>), grid, block, 0, s, dy, Qt, sample_scale, db, part, H, W, N, T, Tp);
  else if (tile == 2) SG_KERNEL(k_wino_dy, grid, block, 0, s, dy, Qt, sample_scale, db, part, H, W, N, T, Tp);
  else if (wino_vec() == 2) SG_KERNEL(k_w43_dy<v2f>, dim3(sg_grid_for(T * (N / 2), 256)), block, 0, s, dy, Qt, sample_scale, db, part, H, W, N, T, Tp);
  else SG_KERNEL(k_w43_dy<v4f>, grid, block, 0, s, dy, Qt, sample_scale, db, part, H, W, N, T, Tp);
  if (part) SG_KERNEL(k_wino_db_finish, dim3(sg_cdiv(N, 256)), dim3(256), 0, s, part, db, N);
  return sg_launch_status();
}

// dU [planes][K][N] = sum over the T tiles of V[f]^T Qt[f] (dU is overwritten)
// v_plane_rows: rows between two planes of V (0 = Tp of this batch; larger when V is a batch slice of a transform made for a bigger batch)
extern "C" int sg_wino_wgrad_gemm(const float* V, const float* Qt, float* dU, int B, int H, int W, int K, int N, int tile, long v_plane_rows,
                                  void* stream) {
  if (!V || !Qt || !dU) return SG_ERR_ARG;
  if (!wino_geom_ok(B, H, W, tile) || (K & 3) || (N & 3) || K <= 0 || N <= 0) return SG_ERR_UNSUPPORTED;
  const long T = wino_tiles(B, H, W, tile), Tp = wino_tp(T);
  if (v_plane_rows != 0 && v_plane_rows < T) return SG_ERR_ARG;
  if (T >= (1L << 31) - 256) return SG_ERR_UNSUPPORTED;
  const int F = wino_planes(tile);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dU, 0, sizeof(float) * (size_t)F * K * N, s) != hipSuccess) return SG_ERR_LAUNCH;
  SgWgradArgs a{};
  a.p = V; a.q = Qt; a.dw = dU;
  a.Bn = (int)T; a.Hp = 1; a.Wp = 1; a.Cp = K; a.p_sy = 1; a.p_sx = 1;
  a.Hq = 1; a.Wq = 1; a.Cq = N; a.q_sy = 1; a.q_sx = 1; a.Hg = 1; a.Wg = 1;
  a.ntaps = F; a.flags = 0;
  a.p_plane = (v_plane_rows ? v_plane_rows : Tp) * K;
  a.q_plane = Tp * N;
  return sg_launch_wgrad(a, s);
}

extern "C" int sg_wino_filter_grad(const float* dU, float* dw, int K, int N, int tile, void* stream) {
  if (!dU || !dw || K <= 0 || N <= 0 || (tile != 2 && tile != 4)) return SG_ERR_ARG;
  const long KN = (long)K * N;
  if (tile == 2) SG_KERNEL(k_wino_dw, dim3(sg_grid_for(KN, 256)), dim3(256), 0, (hipStream_t)stream, dU, dw, KN);
  else SG_KERNEL(k_w43_dw, dim3(sg_grid_for(KN, 256)), dim3(256), 0, (hipStream_t)stream, dU, dw, KN);
  return sg_launch_status();
}

extern "C" int sg_conv2d_bwd_weight_wino(const float* x, const float* dy, float* dw, float* db, const float* sample_scale, int B, int H, int W,
                                         int Cin, int Cout, int flags, int tile, void* workspace, long workspace_bytes, void* stream) {
  if (!x || !dy || !dw || !workspace) return SG_ERR_ARG;
  if (!wino_shape_ok(B, H, W, Cin, 64, tile) || (Cout & 3)) return SG_ERR_UNSUPPORTED;
  if (workspace_bytes < sg_wino_wgrad_workspace_bytes(B, H, W, Cin, Cout, tile)) return SG_ERR_ARG;
  const long Tp = wino_tp(wino_tiles(B, H, W, tile));
  const int F = wino_planes(tile);
  float* V = reinterpret_cast<float*>(workspace);
  float* Qt = V + (size_t)F * Tp * Cin;
  float* dU = Qt + (size_t)F * Tp * Cout;
  int rc = sg_wino_input(x, V, B, H, W, Cin, (flags & SG_RELU_IN) != 0, tile, stream);
  if (rc != SG_OK) return rc;
  if ((flags & SG_UPS2_IN) && tile != 4) return SG_ERR_UNSUPPORTED;
  rc = wino_grad_input_impl(dy, Qt, sample_scale, db, dU + (size_t)F * Cin * Cout, B, H, W, Cout, tile, (flags & SG_UPS2_IN) != 0, stream);
  if (rc != SG_OK) return rc;
  rc = sg_wino_wgrad_gemm(V, Qt, dU, B, H, W, Cin, Cout, tile, 0, stream);
  if (rc != SG_OK) return rc;
  return sg_wino_filter_grad(dU, dw, Cin, Cout, tile, stream);
}
