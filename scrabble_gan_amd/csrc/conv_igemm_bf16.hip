// bf16 variant of the implicit-GEMM convolution (config c3 of BASELINE.json: "bf16 MFMA convs"), on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// Activations stay fp32 NHWC in HBM (BN / attention / losses are fp32 kernels); the operand tiles are rounded to bf16
// (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into LDS.  Filters come PACKED: sg_pack_filter_bf16 writes
// a bf16 copy [tap][N][K] (reduction index contiguous) once per optimizer step, so the B tile is a straight 16-byte
// copy and costs half the bytes of the fp32 filter.
//   LDS layout : As[m][k], Bs[n][k], bf16, row stride 40 elements (80 B): the 16-byte fragment reads of 8 lanes
//                fall into disjoint bank groups
//   MFMA       : 32x32x16 bf16, lane l supplies A[i = l&31][k = 8*(l>>5) .. +7] and B[k = same][j = l&31] as one
//                ds_read_b128 each; same accumulator layout as the fp32 kernel, so the epilogue is shared in form
//   pipeline   : BK = 32 (two k-steps of 4 MFMAs per wave).  The matrix cores are 16x faster than in fp32, so the loop
//                is fed three levels deep: global loads one k-tile ahead in registers, LDS double buffer, fragment
//                reads one k-step ahead (sched_barrier-pinned); the workgroup barrier sits in the middle of a tile.
//   work split : same CU-quantum decomposition as sg_igemm_kernel (tail tiles cut along the reduction)
#include "sg_conv.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define SG_IDENT_OUT 32

__device__ __forceinline__ int sgb_xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// ------------------------------------------------------------------------------------------
// filter packing: fp32 [taps][K][N] (transpose = 1: Conv2D forward layout seen as K x N) or fp32 [taps][N][K]
// (transpose = 0) -> bf16 [taps][N][K]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_filter_bf16(const float* w, __bf16* out, int K, int N, int transpose) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const float* wt = w + (size_t)t * K * N;
  __bf16* ot = out + (size_t)t * K * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  if (transpose) {
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int k = k0 + r, n = n0 + tx;
      tile[r][tx] = (k < K && n < N) ? wt[(size_t)k * N + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int n = n0 + r, k = k0 + tx;
      if (n < N && k < K) ot[(size_t)n * K + k] = (__bf16)tile[tx][r];
    }
  } else {
    const size_t total = (size_t)K * N;
    for (size_t e = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * gridDim.y * 256)
      ot[e] = (__bf16)wt[e];
  }
}

extern "C" int sg_pack_filter_bf16(const float* w, void* out, int taps, int K, int N, int transpose, void* stream) {
  if (!w || !out || taps < 1 || K < 1 || N < 1) return SG_ERR_ARG;
  const dim3 grid(sg_cdiv(N, 32), sg_cdiv(K, 32), taps);
  SG_KERNEL(k_pack_filter_bf16, grid, dim3(256), 0, (hipStream_t)stream, w, (__bf16*)out, K, N, transpose);
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int OCC>
__global__ __launch_bounds__(WM* WN * 64, OCC) void sg_igemm_bf16_kernel(const SgIgemmArgs p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BK = 32, LDK = 40;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_P = BM * 8 / NT;          // float4 chunks (4 k) per thread per k-tile
  constexpr int B_P = BN * 4 / NT;          // 16-byte chunks (8 k, bf16) per thread per k-tile
  static_assert(A_P >= 1 && B_P >= 1, "tile/thread mismatch");

  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BM * LDK + 2 * BN * LDK];
  unsigned short* As = smem;
  unsigned short* Bs = smem + 2 * BM * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int M = p.Bn * p.Hg * p.Wg;
  const int HW = p.Hg * p.Wg;
  const int n_tiles = (p.N + BN - 1) / BN;
  int wg, split, nsplit;
  if ((int)blockIdx.x < p.full_tiles) {
    wg = sgb_xcd_remap(blockIdx.x, p.full_tiles);
    split = 0;
    nsplit = 1;
  } else {
    const int tail_tiles = p.n_tiles_total - p.full_tiles;
    const int u = sgb_xcd_remap(blockIdx.x - p.full_tiles, tail_tiles * p.tail_split);
    nsplit = p.tail_split;
    split = u / tail_tiles;
    wg = p.full_tiles + (u - split * tail_tiles);
  }
  const int m0 = (wg / n_tiles) * BM;
  const int n0 = (wg % n_tiles) * BN;

  const int kchunks = (p.Ca + BK - 1) / BK;
  const int KT_all = p.ntaps * kchunks;
  const int kt_begin = (int)(((long)KT_all * split) / nsplit);
  const int KT = (int)(((long)KT_all * (split + 1)) / nsplit) - kt_begin;
  const bool relu_in = (p.flags & SG_RELU_IN) != 0;

  constexpr unsigned OOB = 0xFFFFFFE0u;
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  auto bload = [](decltype(rsrc_a) r, unsigned voff) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0); };

  // A: thread -> (row = tid / 8 + 32 i, channels 4*(tid % 8) ..+3 of the k-tile)
  const int kc = tid & 7;
  const int arow0 = tid >> 3;
  unsigned a_base[A_P], a_mask[A_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    const int m = m0 + arow0 + i * (NT / 8);
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int b = mm / HW;
    const int rem = mm - b * HW;
    const int yg = rem / p.Wg;
    const int xg = rem - yg * p.Wg;
    const int y = yg * p.a_sy, x = xg * p.a_sx;
    a_base[i] = 4u * (unsigned)(((b * p.Ha + y) * p.Wa + x) * p.Ca + 4 * kc);
    unsigned mk = 0;
    for (int t = 0; t < p.ntaps; ++t) {
      const int iy = y + p.taps[t].dy, ix = x + p.taps[t].dx;
      if (ok && iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) mk |= 1u << t;
    }
    a_mask[i] = mk;
  }
  // B: thread -> (row n = tid / 4 + 64 i, k = 8*(tid % 4) ..+7 of the k-tile), packed bf16 [tap][N][K]
  const int kc8 = tid & 3;
  const int brow0 = tid >> 2;
  unsigned b_off[B_P];
  bool b_ok[B_P];
#pragma unroll
  for (int i = 0; i < B_P; ++i) {
    const int n = n0 + brow0 + i * (NT / 4);
    b_ok[i] = n < p.N;
    b_off[i] = 2u * (unsigned)(n * p.ldw + 8 * kc8);
  }

  using v4u = decltype(bload(rsrc_a, 0u));
  v4u a_reg[A_P], b_reg[B_P];
  // channel-chunk-major reduction order, taps innermost, tap constants in an LDS table: see sg_igemm_kernel
  __shared__ int tap_tab[2 * SG_MAX_TAPS];
  if (tid < p.ntaps) {
    tap_tab[2 * tid] = 4 * (p.taps[tid].dy * p.Wa + p.taps[tid].dx) * p.Ca;
    tap_tab[2 * tid + 1] = 2 * p.taps[tid].w_off;
  }
  __syncthreads();
  const int nt1 = p.ntaps > 0 ? p.ntaps : 1;
  int lt = kt_begin % nt1, lc0 = (kt_begin / nt1) * BK;
  int tap_off = 0, w_tap = 0;
  auto set_tap = [&]() {
    tap_off = tap_tab[2 * lt];
    w_tap = tap_tab[2 * lt + 1];
  };
  set_tap();
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const short rfloor = relu_in ? (short)0 : (short)0x8000;      // 0x8000 = most negative int16: ReLU off
  const s16x4 rfloor4 = {rfloor, rfloor, rfloor, rfloor};

  auto load_tile = [&]() {        // fetch the k-tile at the cursor into registers, then advance the cursor
    const unsigned toff = (unsigned)(tap_off + 4 * lc0);
    const bool cok = lc0 + 4 * kc < p.Ca;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const bool ok = ((a_mask[i] >> lt) & 1u) && cok;
      a_reg[i] = bload(rsrc_a, ok ? a_base[i] + toff : OOB);
    }
    const unsigned woff = (unsigned)(w_tap + 2 * lc0);
    const bool kok = lc0 + 8 * kc8 < p.Ca;
#pragma unroll
    for (int i = 0; i < B_P; ++i) b_reg[i] = bload(rsrc_w, (b_ok[i] && kok) ? b_off[i] + woff : OOB);
    ++lt;
    const bool wrap = lt >= p.ntaps;
    lt = wrap ? 0 : lt;
    lc0 += wrap ? BK : 0;
    set_tap();
  };
  auto store_tile = [&](int buf) {
    unsigned short* as = As + buf * BM * LDK;
    unsigned short* bs = Bs + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(&a_reg[i]);
      bf16x4 h;
      h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
      // ReLU on the rounded operand: a bf16 is negative iff it is negative as an int16 -> packed 16-bit max
      const s16x4 hr = __builtin_elementwise_max(__builtin_bit_cast(s16x4, h), rfloor4);
      *reinterpret_cast<s16x4*>(as + (arow0 + i * (NT / 8)) * LDK + 4 * kc) = hr;
    }
#pragma unroll
    for (int i = 0; i < B_P; ++i)
      *reinterpret_cast<v4u*>(bs + (brow0 + i * (NT / 4)) * LDK + 8 * kc8) = b_reg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int khalf = lane >> 5;
  const int a_row = wm * (BM / WM) + (lane & 31);
  const int b_row = wn * (BN / WN) + (lane & 31);

  if (KT > 0) {
    load_tile();
    store_tile(0);
    if (KT > 1) load_tile();
  }
  __syncthreads();

  bf16x8 af[2][TM], bf[2][TN];
  auto read_frags = [&](int buf, int step, int slot) {
    const unsigned short* as = As + buf * BM * LDK + a_row * LDK + 16 * step + 8 * khalf;
    const unsigned short* bs = Bs + buf * BN * LDK + b_row * LDK + 16 * step + 8 * khalf;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * LDK);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * LDK);
  };
  auto mma = [&](int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][i], bf[slot][j], acc[i][j], 0, 0, 0);
  };
  read_frags(0, 0, 0);

  // tile t: registers hold tile t+1 (if any); `next` = a tile t+1 exists, `next2` = a tile t+2 exists
  auto k_tile = [&](int buf, auto next_tag, auto next2_tag) {
    constexpr bool next = decltype(next_tag)::value, next2 = decltype(next2_tag)::value;
    read_frags(buf, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    if constexpr (next) {
      store_tile(buf ^ 1);
      if constexpr (next2) load_tile();
      __syncthreads();
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (next) read_frags(buf ^ 1, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
  };
  int kt = 0;
  for (; kt + 2 < KT; ++kt) k_tile(kt & 1, std::true_type{}, std::true_type{});
  if (kt + 1 < KT) { k_tile(kt & 1, std::true_type{}, std::false_type{}); ++kt; }
  if (kt < KT) k_tile(kt & 1, std::false_type{}, std::false_type{});

  // ---- epilogue (fp32, same form as sg_igemm_kernel) ----
  const bool accum = (p.flags & SG_ACCUM) != 0;
  const bool relu_out = (p.flags & SG_RELU_OUT) != 0;
  const bool ident = (p.flags & SG_IDENT_OUT) != 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
    if (n >= p.N) continue;
    float bsum = 0.f;
    if (p.bias && split == 0) bsum += p.bias[n];
    if (p.bias2 && split == 0) bsum += p.bias2[n];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const int m = m0 + row;
        if (m >= M) continue;
        size_t idx;
        if (ident) {
          idx = (size_t)m * p.N + n;
        } else {
          const int b = m / HW;
          const int rem = m - b * HW;
          const int yg = rem / p.Wg;
          const int xg = rem - yg * p.Wg;
          idx = ((size_t)(b * p.Ho + yg * p.o_sy + p.o_oy) * p.Wo + xg * p.o_sx + p.o_ox) * p.N + n;
        }
        float v = acc[i][j][r] + bsum;
        if (p.mask && p.mask[idx] <= 0.f) v = 0.f;
        if (nsplit > 1) {
          atomicAdd(p.out + idx, v);
          continue;
        }
        if (accum) v += p.out[idx];
        if (relu_out) v = fmaxf(v, 0.f);
        p.out[idx] = v;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int OCC>
static int launch_bf16_cfg(const SgIgemmArgs& a_in, hipStream_t s) {
  SgIgemmArgs a = a_in;
  if (a.o_sy == 1 && a.o_sx == 1 && a.o_oy == 0 && a.o_ox == 0 && a.Ho == a.Hg && a.Wo == a.Wg) a.flags |= SG_IDENT_OUT;
  const long M = (long)a.Bn * a.Hg * a.Wg;
  const int n_tiles = sg_cdiv(a.N, BN);
  const int tiles = sg_cdiv(M, BM) * n_tiles;
  if (tiles <= 0) return SG_OK;
  const int KT_all = a.ntaps * sg_cdiv(a.Ca, 32);
  const bool can_split = !(a.flags & SG_RELU_OUT) && ((a.flags & SG_ACCUM) || (a.flags & SG_IDENT_OUT));
  constexpr int CUS = 256;
  int full = tiles, nsplit = 1;
  if (can_split && KT_all >= 16) {        // same balance model as launch_cfg in conv_igemm.hip
    const int rem = tiles % CUS;
    if (rem > 0) {
      const int full_c = tiles < 3 * CUS ? 0 : (tiles - rem) / n_tiles * n_tiles;
      const int tail = tiles - full_c;
      auto cost = [&](int sp) {
        const int per_cu = (tail * sp + CUS - 1) / CUS;
        const int resident = full_c > 0 ? OCC : (per_cu < OCC ? per_cu : OCC);
        const double eff = resident >= 3 ? 1.0 : (resident == 2 ? 0.9 : 0.7);
        return (double)per_cu / sp * (1.0 + 0.02 * (sp - 1)) / eff;
      };
      int best = 1;
      double best_cost = cost(1);
      const int sp_max = KT_all / 8 < 16 ? KT_all / 8 : 16;
      for (int sp = 2; sp <= sp_max; ++sp) {
        const double c = cost(sp);
        if (c < 0.97 * best_cost) { best = sp; best_cost = c; }
      }
      if (best > 1) { full = full_c; nsplit = best; }
    }
  }
  if (nsplit > 1 && !(a.flags & SG_ACCUM)) {
    const size_t row0 = (size_t)(full / n_tiles) * BM;
    if (hipMemsetAsync(a.out + row0 * a.N, 0, sizeof(float) * ((size_t)M - row0) * a.N, s) != hipSuccess) return SG_ERR_LAUNCH;
  }
  a.full_tiles = full;
  a.tail_split = nsplit;
  a.n_tiles_total = tiles;
  SG_KERNEL((sg_igemm_bf16_kernel<BM, BN, WM, WN, OCC>), dim3(full + (tiles - full) * nsplit), dim3(WM * WN * 64), 0, s, a);
  return sg_launch_status();
}

// a.w = packed bf16 filter [tap][N][K = Ca] (sg_pack_filter_bf16), taps[t].w_off in elements, ldw = Ca
static int sg_launch_igemm_bf16(const SgIgemmArgs& a_in, hipStream_t s) {
  SgIgemmArgs a = a_in;
  if ((a.Ca & 7) || a.ldw != a.Ca || a.N <= 32) return SG_ERR_UNSUPPORTED;
  if (a.ntaps < 1 || a.ntaps > SG_MAX_TAPS) return SG_ERR_ARG;
  const long a_elems = (long)a.Bn * a.Ha * a.Wa * a.Ca;
  long w_elems = 0;
  for (int t = 0; t < a.ntaps; ++t) w_elems = a.taps[t].w_off > w_elems ? a.taps[t].w_off : w_elems;
  w_elems += (long)a.N * a.ldw;
  if (a_elems >= (1L << 30) - 8 || w_elems >= (1L << 31) - 16 || (long)a.Bn * a.Ho * a.Wo * a.N >= (1L << 31)) return SG_ERR_ARG;
  a.a_bytes = (unsigned)(4 * a_elems);
  a.w_bytes = (unsigned)(2 * w_elems);
  if (a.N > 64) return launch_bf16_cfg<128, 128, 2, 2, 3>(a, s);
  return launch_bf16_cfg<128, 64, 2, 2, 3>(a, s);
}

// ------------------------------------------------------------------------------------------
// C-ABI (include/scrabble_hip.h): same contracts as sg_conv2d_fwd / sg_conv2d_bwd_data, with the filter given as
// the packed bf16 copy.  wp_fwd = pack(w [kh,kw,Cin,Cout], K = Cin, N = Cout, transpose = 1): [tap][Cout][Cin]
//                        wp_bwd = pack(w [kh,kw,Cin,Cout], K = Cout, N = Cin, transpose = 0): [tap][Cin][Cout]
// ------------------------------------------------------------------------------------------
extern "C" int sg_conv2d_fwd_bf16(const float* x, const void* wp_fwd, const float* bias, const float* bias2, float* y, int B, int H,
                                  int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream) {
  if (!x || !wp_fwd || !y || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemmArgs a{};
  a.a = x; a.w = (const float*)wp_fwd; a.out = y; a.bias = bias; a.bias2 = bias2; a.mask = nullptr;
  a.Bn = B; a.Ha = H; a.Wa = W; a.Ca = Cin; a.Hg = Ho; a.Wg = Wo; a.a_sy = 1; a.a_sx = 1;
  a.Ho = Ho; a.Wo = Wo; a.N = Cout; a.o_sy = 1; a.o_sx = 1; a.o_oy = 0; a.o_ox = 0;
  a.ntaps = kh * kw; a.ldw = Cin; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm_bf16(a, (hipStream_t)stream);
}

extern "C" int sg_conv2d_bwd_data_bf16(const float* dy, const void* wp_bwd, const float* mask, float* dx, int B, int H, int W,
                                       int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream) {
  if (!dy || !wp_bwd || !dx || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemmArgs a{};
  a.a = dy; a.w = (const float*)wp_bwd; a.out = dx; a.mask = mask;
  a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.ldw = Cout; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm_bf16(a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Conv2DTranspose(padding='same', strides (sh, sw)) with bf16 matrix-core operands: the same parity-class / tap-list
// launches as the fp32 entry points (conv_igemm.hip), filter w [kh,kw,Cout,Cin] given as packed bf16 copies:
//   forward  : wp = pack(w, taps, K = Cin,  N = Cout, transpose = 0)   (each tap is already [N = Cout][K = Cin])
//   data-grad: wp = pack(w, taps, K = Cout, N = Cin,  transpose = 1)
// ------------------------------------------------------------------------------------------
static inline int floordiv_b(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

extern "C" int sg_conv2d_transpose_fwd_bf16(const float* x, const void* wp, const float* bias, const float* bias2, float* y, int B,
                                            int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, int flags, void* stream) {
  if (!x || !wp || !y || kh * kw > SG_MAX_TAPS || sh < 1 || sw < 1) return SG_ERR_ARG;
  const int pbh = (sh == 1) ? kh / 2 : 0, pbw = (sw == 1) ? kw / 2 : 0;
  for (int py = 0; py < sh; ++py)
    for (int px = 0; px < sw; ++px) {
      SgIgemmArgs a{};
      a.a = x; a.w = (const float*)wp; a.out = y; a.bias = bias; a.bias2 = bias2;
      a.Bn = B; a.Ha = H; a.Wa = W; a.Ca = Cin; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
      a.Ho = sh * H; a.Wo = sw * W; a.N = Cout; a.o_sy = sh; a.o_sx = sw; a.o_oy = py; a.o_ox = px;
      a.ldw = Cin; a.flags = flags; a.ntaps = 0;
      for (int ky = 0; ky < kh; ++ky) {
        if ((py + pbh - ky) % sh) continue;
        for (int kx = 0; kx < kw; ++kx) {
          if ((px + pbw - kx) % sw) continue;
          a.taps[a.ntaps++] = SgTap{floordiv_b(py + pbh - ky, sh), floordiv_b(px + pbw - kx, sw), (ky * kw + kx) * Cin * Cout};
        }
      }
      if (a.ntaps == 0) return SG_ERR_UNSUPPORTED;      // a class that only writes its bias: the fp32 entry point handles it
      const int rc = sg_launch_igemm_bf16(a, (hipStream_t)stream);
      if (rc != SG_OK) return rc;
    }
  return SG_OK;
}

extern "C" int sg_conv2d_transpose_bwd_data_bf16(const float* dy, const void* wp, const float* mask, float* dx, int B, int H, int W,
                                                 int Cin, int Cout, int kh, int kw, int sh, int sw, int flags, void* stream) {
  if (!dy || !wp || !dx || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int pbh = (sh == 1) ? kh / 2 : 0, pbw = (sw == 1) ? kw / 2 : 0;
  SgIgemmArgs a{};
  a.a = dy; a.w = (const float*)wp; a.out = dx; a.mask = mask;
  a.Bn = B; a.Ha = sh * H; a.Wa = sw * W; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = sh; a.a_sx = sw;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.ldw = Cout; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - pbh, kx - pbw, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm_bf16(a, (hipStream_t)stream);
}
