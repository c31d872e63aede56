// bf16 implicit-GEMM convolution, second generation (config c3 of BASELINE.json): bf16 ACTIVATIONS IN HBM, operand
// tiles moved global -> LDS by the DMA path (global_load_lds_dwordx4), 256x256 tiles, 8 waves with 128x64 wave tiles.
//
// Why (round-1 measurement): the first bf16 kernel reads fp32 activations, rounds them in
// registers and writes the tiles with ds_write -- per k-tile the three resident workgroups of a CU push 48 KB through the
// VGPR->LDS store path (~80 B/clk peak) and read 96 KB of fragments with 64x64 wave tiles: the LDS pipe, not the matrix
// core, set the rate (22 % of the dense bf16 peak).  Here
//   * the activation operand is a bf16 NHWC tensor (written by the producing kernel's epilogue or by sg_cvt_bf16), so an
//     operand row of one k-tile is ONE 128-byte line = 64 channels of one pixel at one tap;
//   * both tiles go global -> LDS without touching a VGPR: 8 x global_load_lds_dwordx4 per thread and k-tile.  The LDS
//     image of a wave instruction is lane-linear (8 rows x 128 B), so the bank swizzle sits on the SOURCE side: the lane
//     that owns 16-byte slot p of row r fetches k-chunk p ^ ((r >> 1) & 7); a fragment read of chunk c goes to slot
//     c ^ ((r >> 1) & 7) -- conflict-free for ds_read_b128's 16-lane groups;
//   * a row that must contribute zeros (padding tap, pixel past the end of the batch) fetches from a 256-byte zero page
//     in device memory instead of its pixel -- no select after the load, no dependence on out-of-range semantics;
//   * ReLU on the operand is a packed signed 16-bit max on the FRAGMENT registers (bf16 is negative iff its int16 is);
//   * 128x64 wave tiles: 6 ds_read_b128 per 8 MFMAs (0.75 KB per MFMA instead of 1 KB), fragment reads one k-step ahead;
//   * two 64 KB stages (A 256x64 + B 256x64, bf16).  One workgroup barrier per k-tile, placed after the fragment reads of
//     the tile's LAST k-step have returned: behind it the other stage's data has landed (each wave waited for its own DMAs)
//     and this stage is free, so the DMAs of tile t+2 are issued there and spread over the following k-steps, while the
//     last k-step's MFMAs already run on prefetched fragments of tile t+1's first step.
// Reduction order: channel-chunk-major, taps innermost (the 3x3 neighbourhood of a pixel row stays in L2).
#include "sg_conv2.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __attribute__((aligned(256))) unsigned int sg2_zero_page[64];      // 256 bytes of zeros (never written)

__device__ __forceinline__ int sg2_xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// ------------------------------------------------------------------------------------------
// fp32 -> bf16 (round to nearest even), optional ReLU, optional per-row factor (row = `rowlen` consecutive elements:
// the per-sample factors of the shared backward sweep applied while the weight-grad operand is converted)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cvt_bf16(const float* __restrict__ x, u16* __restrict__ out, long n8, int relu,
                                                  const float* __restrict__ rowscale, long rowlen) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (long)gridDim.x * blockDim.x) {
    float4 v0 = reinterpret_cast<const float4*>(x)[2 * e], v1 = reinterpret_cast<const float4*>(x)[2 * e + 1];
    if (rowscale) {
      const float s = rowscale[(8 * e) / rowlen];
      v0.x *= s; v0.y *= s; v0.z *= s; v0.w *= s; v1.x *= s; v1.y *= s; v1.z *= s; v1.w *= s;
    }
    if (relu) {
      v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
      v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
    }
    bf16x8 h;
    h[0] = (__bf16)v0.x; h[1] = (__bf16)v0.y; h[2] = (__bf16)v0.z; h[3] = (__bf16)v0.w;
    h[4] = (__bf16)v1.x; h[5] = (__bf16)v1.y; h[6] = (__bf16)v1.z; h[7] = (__bf16)v1.w;
    reinterpret_cast<bf16x8*>(out)[e] = h;
  }
}

// x fp32 [n] -> out bf16 [n], n % 8 == 0; rowscale (nullable) [n / rowlen], rowlen % 8 == 0
extern "C" int sg_cvt_bf16(const float* x, void* out, long n, int relu, const float* rowscale, long rowlen, void* stream) {
  if (!x || !out || n < 0 || (n & 7) || (rowscale && (rowlen <= 0 || (rowlen & 7)))) return SG_ERR_ARG;
  if (n == 0) return SG_OK;
  SG_KERNEL(k_cvt_bf16, dim3(sg_grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (u16*)out, n / 8, relu, rowscale, rowlen);
  return sg_launch_status();
}

// Same conversion of a [M, C] gradient (rows = pixels) fused with its bias gradient: dbias[c] += sum_m rowscale[sample(m)] * x[m][c]
// (fp32 sums of the scaled fp32 values, before the bf16 rounding) -- ONE sweep over dy instead of three (convert,
// scaled fp32 copy, column sums).  CL column lanes of 8 channels x RL row lanes per workgroup, LDS tree over the row lanes.
__global__ __launch_bounds__(256) void k_cvt_bf16_bias(const float* __restrict__ x, u16* __restrict__ out, u16* __restrict__ out_plain, long M,
                                                       int C, const float* __restrict__ rowscale, long rows_per_sample,
                                                       float* __restrict__ dbias, int rows_per_block) {
  __shared__ float red[256 * 8];
  const int c8 = C >> 3;
  int CL = 256;
  while (CL > c8) CL >>= 1;
  const int RL = 256 / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const long m0 = (long)blockIdx.x * rows_per_block;
  const long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  for (int cg0 = 0; cg0 < c8; cg0 += CL) {
    const int cg = cg0 + cl;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cg < c8) {
      for (long m = m0 + rl; m < m1; m += RL) {
        const float4* src = reinterpret_cast<const float4*>(x + m * C + 8 * cg);
        float4 v0 = src[0], v1 = src[1];
        if (out_plain) {
          bf16x8 hp;
          hp[0] = (__bf16)v0.x; hp[1] = (__bf16)v0.y; hp[2] = (__bf16)v0.z; hp[3] = (__bf16)v0.w;
          hp[4] = (__bf16)v1.x; hp[5] = (__bf16)v1.y; hp[6] = (__bf16)v1.z; hp[7] = (__bf16)v1.w;
          *reinterpret_cast<bf16x8*>(out_plain + m * C + 8 * cg) = hp;
        }
        if (rowscale) {
          const float f = rowscale[m / rows_per_sample];
          v0.x *= f; v0.y *= f; v0.z *= f; v0.w *= f; v1.x *= f; v1.y *= f; v1.z *= f; v1.w *= f;
        }
        bf16x8 hh;
        hh[0] = (__bf16)v0.x; hh[1] = (__bf16)v0.y; hh[2] = (__bf16)v0.z; hh[3] = (__bf16)v0.w;
        hh[4] = (__bf16)v1.x; hh[5] = (__bf16)v1.y; hh[6] = (__bf16)v1.z; hh[7] = (__bf16)v1.w;
        *reinterpret_cast<bf16x8*>(out + m * C + 8 * cg) = hh;
        s[0] += v0.x; s[1] += v0.y; s[2] += v0.z; s[3] += v0.w; s[4] += v1.x; s[5] += v1.y; s[6] += v1.z; s[7] += v1.w;
      }
    }
    if (RL > 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[j * 256 + threadIdx.x] = s[j];
      __syncthreads();
      for (int st = RL >> 1; st > 0; st >>= 1) {
        if (rl < st) {
#pragma unroll
          for (int j = 0; j < 8; ++j) red[j * 256 + threadIdx.x] += red[j * 256 + threadIdx.x + st * CL];
        }
        __syncthreads();
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = red[j * 256 + threadIdx.x];
      __syncthreads();
    }
    if (rl == 0 && cg < c8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(dbias + 8 * cg + j, s[j]);
    }
  }
}

// x fp32 [M, C] -> out bf16 [M, C] (times rowscale[m / rows_per_sample] when given), out_plain (nullable) bf16 [M, C] = the
// unscaled copy, dbias [C] += column sums of the scaled fp32 values; C % 8 == 0.
extern "C" int sg_cvt_bf16_bias(const float* x, void* out, void* out_plain, long M, int C, const float* rowscale, long rows_per_sample,
                                float* dbias, void* stream) {
  if (!x || !out || !dbias || M < 0 || C <= 0 || (C & 7) || (rowscale && rows_per_sample <= 0)) return SG_ERR_ARG;
  if (M == 0) return SG_OK;
  long r = (M + 1023) / 1024;                 // ~1024 workgroups: enough to fill the chip, few enough atomics per column
  if (sg_deterministic()) r = M;              // one workgroup, one adder per column (slow: a reproducibility mode)
  const int rpb = (int)(r < 32 ? 32 : r);
  SG_KERNEL(k_cvt_bf16_bias, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, x, (u16*)out, (u16*)out_plain, M, C,
                     rowscale, rows_per_sample, dbias, rpb);
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
constexpr int SG2_BM = 256, SG2_BK = 64;
constexpr int SG2_TILE = SG2_BM * SG2_BK * 2;                 // bytes of a 256-row operand tile (32 KB)
constexpr int SG2_LDS = 4 * SG2_TILE;                         // A stage 0 | A stage 1 | B stage 0 | B stage 1 (BN = 256)

// BN = 256: 2 x 4 waves, wave tile 128 x 64 (the large layers);  BN = 128: 4 x 2 waves, 64 x 64;  BN = 64: 8 x 1 waves,
// 32 x 64 (the 64-filter layers, which are bandwidth-bound anyway).  The A tile is always 256 rows.
// ES = bytes per operand element: 2 = bf16 (k-tile = 64 channels, four 32x32x16 MFMA steps), 1 = fp8 e4m3 (k-tile = 128
// channels, two v_mfma_scale_f32_32x32x64_f8f6f4 steps with unit block scales: 2x the bf16 rate per clock).  The byte
// geometry of the tiles (128-byte rows, 16-byte chunks, swizzle, DMA roles) is the same for both.
// RELU: max(a, 0) on the A fragments in registers (SG_RELU_IN); compiled out otherwise -- a VALU instruction in front of an
// MFMA pair is NOT free on the fp32 stream (tools/mfma_peak.hip: 154 -> 141 TFLOP/s with two per pair), so launches without
// the flag (every data-grad, convs behind a fused BN+ReLU) run a loop without any.
// BMT = rows of the A tile: 256 (8 waves, one workgroup per CU) or 128 with BN = 128 (round 3: 4 waves as 2 x 2 of 64 x 64 wave
// tiles, 64 KB of LDS -> TWO workgroups per CU: the HBM-bound epilogue of one overlaps the k-loop of the other; the 64-filter
// configuration, which sits at two per CU already, loses 35-45 % when forced down to one: profiles/r03_probe_occupancy64.txt)
// NWO (round 4) = waves per workgroup when it is not the tile's default (0): the fp32 128 x 128 tile with EIGHT waves of 32 x 64
// (four per SIMD at two workgroups per CU, 128 VGPRs each): a lone wave cannot keep the fp32 matrix pipe full through its own
// fragment waits and the k-tile barrier, so while one workgroup is in its prologue / epilogue the other's waves left the pipe
// 20-25 % idle (MFMA busy 78.5 % at K = 512 against 84.8 % at K = 1024, profiles/r04_probe_clock_bs128.txt).
template <int BN, int ES, bool RELU, int BMT = 256, int NWO = 0>
__global__ __launch_bounds__((NWO ? NWO : (BMT == 256 ? 8 : 4)) * 64, 2) void sg_igemm_bf16v2_kernel(const SgIgemm2Args p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int BM = BMT, BK = 128 / ES;               // channels per k-tile
  constexpr int NW = NWO ? NWO : (BM == 256 ? 8 : 4);   // waves per workgroup
  constexpr int ATILE = BM * 128;                       // bytes of an A stage
  constexpr int WN = BN / 64, WM = NW / WN;             // waves along n / m
  constexpr int TM = BM / WM / 32, TN = 2;              // 32 x 32 MFMA tiles per wave
  constexpr int BQ = BN / (8 * NW);                     // B-tile DMA instructions per thread
  constexpr int AQ = BM / (8 * NW);                     // A-tile DMA instructions per thread (4, or 2 with eight waves on 128 rows)
  static_assert(BM == 256 || (BM == 128 && (BN == 128 || (BN == 64 && ES == 4))), "tile configurations");      // (128 x 64: the grouped fp32 products of 64-filter layers, round 4)
  static_assert(NWO == 0 || (NWO == 8 && BM == 128 && ES == 4), "eight waves: the fp32 128 x 128 tile only");
  constexpr int BTILE = BN * 128;                       // bytes of the B tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int M = p.Bn * p.Hg * p.Wg;
  const int HW = p.Hg * p.Wg;
  const int n_tiles = p.N / BN;
  int wg, split, nsplit;
  if ((int)blockIdx.x < p.full_tiles) {
    wg = sg2_xcd_remap(blockIdx.x, p.full_tiles);
    split = 0;
    nsplit = 1;
  } else {
    const int tail_tiles = p.n_tiles_total - p.full_tiles;
    const int u = sg2_xcd_remap(blockIdx.x - p.full_tiles, tail_tiles * p.tail_split);
    nsplit = p.tail_split;
    split = u / tail_tiles;
    wg = p.full_tiles + (u - split * tail_tiles);
  }
  const int m0 = (wg / n_tiles) * BM;
  const int n0 = (wg % n_tiles) * BN;
  const int grp = p.group_rows ? m0 / p.group_rows : 0;      // (uniform: a tile lies inside one group)
  const int grp_row0 = grp * p.group_rows;
  const int kchunks = p.Ca / BK;
  const int KT_all = p.ntaps * kchunks;
  const int kt_begin = (int)(((long)KT_all * split) / nsplit);
  const int KT = (int)(((long)KT_all * (split + 1)) / nsplit) - kt_begin;

  // ---- DMA lane roles: instruction i of wave w covers tile rows (8 i + w) * 8 .. + 7; lane -> row (l >> 3), slot (l & 7)
  const int lrow = lane >> 3, lslot = lane & 7;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(sg2_zero_page) + 16 * lslot;
  unsigned a_off[4], a_msk[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (NW * i + wave) * 8 + lrow;
    const int chunk = lslot ^ ((r >> 1) & 7);
    const int m = m0 + r;
    const bool ok = m < M;
    const int mm = ok ? m - grp_row0 : 0;
    const int b = mm / HW;
    const int rem = mm - b * HW;
    const int yg = rem / p.Wg;
    const int xg = rem - yg * p.Wg;
    const int y = yg * p.a_sy, x = xg * p.a_sx;
    a_off[i] = (unsigned)ES * (unsigned)(((b * p.Ha + y) * p.Wa + x) * p.Ca) + 16u * chunk;
    unsigned mk = 0;
    for (int t = 0; t < p.ntaps; ++t) {
      const int iy = y + p.taps[t].dy, ix = x + p.taps[t].dx;
      if (ok && iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) mk |= 1u << t;
    }
    a_msk[i] = mk;
    b_off[i] = (unsigned)ES * (unsigned)((n0 + (r % BN)) * p.Ca) + 16u * chunk;       // (instructions i >= BN / 64 are not issued)
  }
  // tap constants live in the lanes of two VGPRs (lane t = tap t): {byte offset of the tap in the activation, byte offset
  // of the tap's slab in the packed filter}; v_readlane with the (uniform) tap cursor fetches them without a memory access
  int tab_a = 0, tab_w = 0;
  if (lane < p.ntaps) {
    tab_a = ES * (p.taps[lane].dy * p.Wa + p.taps[lane].dx) * p.Ca;
    tab_w = ES * p.taps[lane].w_off;
  }
  int lt = kt_begin % p.ntaps, lc0 = (kt_begin / p.ntaps) * BK;         // cursor of the tile being loaded
  int lidx = 0;                                                          // its index in this workgroup's reduction range
  int cur_a = 0, cur_w = 0;                                              // its tap / channel-chunk byte offsets (uniform)
  unsigned cur_bit = 0;
  bool cur_live = true;
  // Tiles past the end of the range ("phantom" tiles) are loaded like any other, from the zero page: the k-loop below
  // has no tile-count dependent branch (a branch makes hipcc fall back from counted lgkmcnt waits to lgkmcnt(0), which
  // serialises the fragment prefetch), and a phantom tile's products are zeros.
  auto set_cursor = [&]() {
    cur_live = lidx < KT;
    cur_a = __builtin_amdgcn_readlane(tab_a, lt) + ES * lc0;
    cur_w = __builtin_amdgcn_readlane(tab_w, lt) + ES * lc0;
    cur_bit = cur_live ? (1u << lt) : 0u;
  };
  set_cursor();
  auto advance = [&]() {
    ++lidx;
    ++lt;
    const bool wrap = lt >= p.ntaps;
    lt = wrap ? 0 : lt;
    lc0 += wrap ? BK : 0;
    set_cursor();
  };
  const unsigned long long a_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(p.a) + (unsigned long long)grp * (unsigned long long)p.a_group_bytes;
  const unsigned long long z_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(zero);
  const unsigned long long w_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(p.w) + (unsigned long long)grp * (unsigned long long)p.w_group_bytes;
  // part q (0..3) of the tile at the cursor into stage `st`: A rows of instruction q and B rows of instruction q
  auto issue_part = [&](int st, int q) {
    const bool ok = (a_msk[q] & cur_bit) != 0;
    const unsigned long long pa = a_base64 + (unsigned long long)(a_off[q] + (unsigned)cur_a);
    const unsigned lo = ok ? (unsigned)pa : (unsigned)z_base64;
    const unsigned hi = ok ? (unsigned)(pa >> 32) : (unsigned)(z_base64 >> 32);
    const unsigned char* src_a = reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
    const unsigned long long pb = cur_live ? w_base64 + (unsigned long long)(b_off[q] + (unsigned)cur_w) : z_base64;
    const unsigned char* src_b = reinterpret_cast<const unsigned char*>((uintptr_t)pb);
    unsigned char* dst_a = smem + st * ATILE + (NW * q + wave) * 1024;
    unsigned char* dst_b = smem + 2 * ATILE + st * BTILE + (NW * q + wave) * 1024;
    if (q < AQ)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_a, (__attribute__((address_space(3))) void*)dst_a, 16, 0, 0);
    if (q < BQ)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_b, (__attribute__((address_space(3))) void*)dst_b, 16, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addressing: row (l & 31) of a 32-row group, k-chunk 2 s + (l >> 5) of step s, swizzled slot
  const int frow = lane & 31, khalf = lane >> 5, swz = (frow >> 1) & 7;
  // Fragment reads are inline asm with hand-counted s_waitcnt: hipcc treats global_load_lds as a FLAT access that may touch
  // LDS, and from the first one on turns every counted lgkmcnt(N) into lgkmcnt(0) -- which would wait for the NEXT step's
  // fragment reads before the current step's MFMAs (seen in the ISA).  The asm reads are invisible to that pass; the
  // sched_barriers keep the (register-only) MFMAs behind the waits (guide section 5.4 rule 18).
  // Per-lane LDS byte addresses of the four k-steps' chunks in row `frow` of the wave's first 32-row group; the 32-row
  // group (i, j) and the stage are immediates of the ds_read.
  typedef int v4i __attribute__((ext_vector_type(4)));
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned a_addr[4], b_addr[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned ko = 16u * (unsigned)((2 * s + khalf) ^ swz);
    a_addr[s] = lds0 + (unsigned)((wm * (TM * 32) + frow) * 128) + ko;
    b_addr[s] = lds0 + (unsigned)(2 * ATILE + (wn * 64 + frow) * 128) + ko;
  }
  const s16x8 rfloor8 = {0, 0, 0, 0, 0, 0, 0, 0};      // (bf16 compares like a sign-magnitude integer: max as int16 against +0)

  v4i af[2][4], bfr[2][2];
#define SG2_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // (macros, not lambdas: the stage / step / slot must reach the asm as literal constants)
#define SG2_READ_FRAGS(st, s, slot)                                                   \
  do {                                                                                \
    SG2_DSR(af[slot][0], a_addr[s], (st) * ATILE + 0 * 4096);                      \
    if constexpr (TM > 1) SG2_DSR(af[slot][1], a_addr[s], (st) * ATILE + 1 * 4096); \
    if constexpr (TM > 2) SG2_DSR(af[slot][2], a_addr[s], (st) * ATILE + 2 * 4096); \
    if constexpr (TM > 2) SG2_DSR(af[slot][3], a_addr[s], (st) * ATILE + 3 * 4096); \
    SG2_DSR(bfr[slot][0], b_addr[s], (st) * BTILE + 0 * 4096);                        \
    SG2_DSR(bfr[slot][1], b_addr[s], (st) * BTILE + 1 * 4096);                        \
  } while (0)
  auto mma = [&](int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      bf16x8 a = __builtin_bit_cast(bf16x8, af[slot][i]);
      if constexpr (RELU) a = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, af[slot][i]), rfloor8));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bfr[slot][j]), acc[i][j], 0, 0, 0);
    }
  };
  // the MFMAs of a step with the four DMA parts of the NEXT tile for stage `st` between the MFMA pairs (address math and
  // issue in the shadow of the matrix pipe)
  auto mma_dma = [&](int slot, int st) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      bf16x8 a = __builtin_bit_cast(bf16x8, af[slot][i]);
      if constexpr (RELU) a = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, af[slot][i]), rfloor8));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bfr[slot][j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = i * 4 / TM; q < (i + 1) * 4 / TM; ++q) issue_part(st, q);
      __builtin_amdgcn_sched_barrier(0);
    }
    advance();
  };
  // wait until only the TM + TN fragment reads just issued are still in flight
#define SG2_WAIT_FRAGS()                                                        \
  do {                                                                          \
    if constexpr (TM + TN == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); \
    else if constexpr (TM + TN == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); \
    else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                          \
  } while (0)

  // tile t in stage t & 1; on entry: fragments of (t, step 0) in flight into slot 0, ALL of tile t+1 issued (during the last
  // step of tile t-1: a whole tile of lead -- with the parts spread over the tile's own first steps the last ones had
  // half a tile, less than an L2 round trip under load, and the vmcnt(0) below stalled every tile).
  // After step 2: tile t+1 has landed (this wave's DMAs; the barrier extends that to every wave's) and every wave has its
  // last fragments of stage `st` in registers, so stage `st` is free for tile t+2.
#define SG2_K_TILE(st, sn)                                                                                   \
  do {                                                                                                       \
    SG2_READ_FRAGS(st, 1, 1);                                                                                \
    SG2_WAIT_FRAGS(); /* slot 0 is in; the reads of step 1 stay in flight under the MFMAs */                 \
    mma(0);                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    SG2_READ_FRAGS(st, 2, 0);                                                                                \
    SG2_WAIT_FRAGS();                                                                                        \
    mma(1);                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    SG2_READ_FRAGS(st, 3, 1);                                                                                \
    SG2_WAIT_FRAGS();                                                                                        \
    mma(0);                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                              \
    __builtin_amdgcn_s_barrier();                                                                            \
    SG2_READ_FRAGS(sn, 0, 0);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    mma_dma(1, st);                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)
  // ---------------- fp8: two k-steps of 64 channels per tile; a fragment = 32 bytes = two swizzled 16-byte chunks
  typedef int v8i __attribute__((ext_vector_type(8)));
  unsigned a8[2][2], b8[2][2];                       // [step][chunk half] byte addresses of row `frow`
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned ko = 16u * (unsigned)((4 * s + 2 * khalf + e) ^ swz);
      a8[s][e] = lds0 + (unsigned)((wm * (TM * 32) + frow) * 128) + ko;
      b8[s][e] = lds0 + (unsigned)(2 * ATILE + (wn * 64 + frow) * 128) + ko;
    }
  // ONE fragment set (48 registers; a second one does not fit beside the 128 accumulators): the reads of the next step are
  // issued right behind the MFMAs of the current one (an MFMA takes its operands at issue), the partner wave of the SIMD
  // covers most of the read latency
  v4i afl[1][4], afh[1][4], bfl[1][2], bfh[1][2];    // [slot][group], low / high 16 bytes of a fragment
#define SG8_READ_FRAGS(st, s, slot)                                                          \
  do {                                                                                       \
    SG2_DSR(afl[slot][0], a8[s][0], (st) * ATILE + 0 * 4096);                             \
    SG2_DSR(afh[slot][0], a8[s][1], (st) * ATILE + 0 * 4096);                             \
    if constexpr (TM > 1) SG2_DSR(afl[slot][1], a8[s][0], (st) * ATILE + 1 * 4096);       \
    if constexpr (TM > 1) SG2_DSR(afh[slot][1], a8[s][1], (st) * ATILE + 1 * 4096);       \
    if constexpr (TM > 2) SG2_DSR(afl[slot][2], a8[s][0], (st) * ATILE + 2 * 4096);       \
    if constexpr (TM > 2) SG2_DSR(afh[slot][2], a8[s][1], (st) * ATILE + 2 * 4096);       \
    if constexpr (TM > 2) SG2_DSR(afl[slot][3], a8[s][0], (st) * ATILE + 3 * 4096);       \
    if constexpr (TM > 2) SG2_DSR(afh[slot][3], a8[s][1], (st) * ATILE + 3 * 4096);       \
    SG2_DSR(bfl[slot][0], b8[s][0], (st) * BTILE + 0 * 4096);                                \
    SG2_DSR(bfh[slot][0], b8[s][1], (st) * BTILE + 0 * 4096);                                \
    SG2_DSR(bfl[slot][1], b8[s][0], (st) * BTILE + 1 * 4096);                                \
    SG2_DSR(bfh[slot][1], b8[s][1], (st) * BTILE + 1 * 4096);                                \
  } while (0)
  auto mma8 = [&](int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const v8i a = __builtin_shufflevector(afl[slot][i], afh[slot][i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const v8i b = __builtin_shufflevector(bfl[slot][j], bfh[slot][j], 0, 1, 2, 3, 4, 5, 6, 7);
        // cbsz = blgp = 0: both operands fp8 e4m3; E8M0 scale 127 = 2^0 on both sides
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i][j], 0, 0, 0, 127, 0, 127);
      }
    }
  };
#define SG8_WAIT_ALL() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SG8_K_TILE(st, sn)                                                             \
  do {                                                                                 \
    SG8_WAIT_ALL();                                                                    \
    mma8(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    SG8_READ_FRAGS(st, 1, 0);                                                          \
    SG8_WAIT_ALL();                                                                    \
    mma8(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
    __builtin_amdgcn_s_barrier();                                                      \
    SG8_READ_FRAGS(sn, 0, 0);                                                          \
    issue_part(st, 0);                                                                 \
    issue_part(st, 1);                                                                 \
    issue_part(st, 2);                                                                 \
    issue_part(st, 3);                                                                 \
    advance();                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
  // ---------------- fp32 (ES = 4): a k-tile is 32 channels = eight 16-byte chunks of 4 floats.  v_mfma_f32_32x32x2_f32
  // takes ONE float per lane and operand: lanes 0-31 supply k = 0, lanes 32-63 k = 1.  The order of the reduction index is
  // free as long as both operands agree, so lane half h reads the 8 bytes {4 c + 2 h, 4 c + 2 h + 1} of chunk c with ONE
  // ds_read_b64 and feeds two MFMAs (element 0, then element 1): 6 reads per 16 MFMAs (1 024 matrix cycles) -- LDS and
  // issue slots are almost idle, the loop is matrix-pipe bound.  Exact fp32 products and fp32 accumulation as in
  // sg_igemm_kernel (only the summation order differs).
  typedef float v2f __attribute__((ext_vector_type(2)));
  unsigned a4[8], b4[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const unsigned ko = 16u * (unsigned)(s ^ swz) + 8u * (unsigned)khalf;
    a4[s] = lds0 + (unsigned)((wm * (TM * 32) + frow) * 128) + ko;
    b4[s] = lds0 + (unsigned)(2 * ATILE + (wn * 64 + frow) * 128) + ko;
  }
  v2f af4[2][4], bf4[2][2];
#define SG4_DSR(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define SG4_READ_FRAGS(st, s, slot)                                                          \
  do {                                                                                       \
    SG4_DSR(af4[slot][0], a4[s], (st) * ATILE + 0 * 4096);                                \
    if constexpr (TM > 1) SG4_DSR(af4[slot][1], a4[s], (st) * ATILE + 1 * 4096);          \
    if constexpr (TM > 2) SG4_DSR(af4[slot][2], a4[s], (st) * ATILE + 2 * 4096);          \
    if constexpr (TM > 2) SG4_DSR(af4[slot][3], a4[s], (st) * ATILE + 3 * 4096);          \
    SG4_DSR(bf4[slot][0], b4[s], (st) * BTILE + 0 * 4096);                                   \
    SG4_DSR(bf4[slot][1], b4[s], (st) * BTILE + 1 * 4096);                                   \
  } while (0)
  // ReLU: ONE v_max_i32 per element on the bit pattern (negative floats are negative integers; fmaxf / v_med3 come with a
  // canonicalising pre-pass, inline asm would hide the VALU -> MFMA hazard from hipcc: no s_nop, NaNs -- seen), applied IN
  // PLACE to the whole fragment set in front of the step's MFMAs: a max in front of every MFMA pair costs 9 % of the fp32
  // matrix rate (dependent VALU + s_nop in every gap; tools/mfma_peak.hip: 154 -> 141 TFLOP/s), batched it costs 2 %
  auto relu4 = [&](int slot) {
    if constexpr (RELU) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float x = af4[slot][i][e];      // (a scalar copy first: __builtin_bit_cast applied to a vector ELEMENT reads element 0)
          af4[slot][i][e] = __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, x), 0));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto mma4 = [&](int slot) {
    relu4(slot);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af4[slot][i][e], bf4[slot][j][e], acc[i][j], 0, 0, 0);
  };
  auto mma4_dma = [&](int slot, int st) {
    relu4(slot);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af4[slot][i][e], bf4[slot][j][e], acc[i][j], 0, 0, 0);
        if (e == 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = i * 4 / TM; q < (i + 1) * 4 / TM; ++q) issue_part(st, q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    advance();
  };
#define SG4_STEP(st, s_next, slot_next, slot_cur)                                            \
  do {                                                                                       \
    SG4_READ_FRAGS(st, s_next, slot_next);                                                   \
    SG2_WAIT_FRAGS();                                                                        \
    mma4(slot_cur);                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#define SG4_K_TILE(st, sn)                                                                   \
  do {                                                                                       \
    SG4_STEP(st, 1, 1, 0);                                                                   \
    SG4_STEP(st, 2, 0, 1);                                                                   \
    SG4_STEP(st, 3, 1, 0);                                                                   \
    SG4_STEP(st, 4, 0, 1);                                                                   \
    SG4_STEP(st, 5, 1, 0);                                                                   \
    SG4_STEP(st, 6, 0, 1);                                                                   \
    SG4_STEP(st, 7, 1, 0);                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
    __builtin_amdgcn_s_barrier();                                                            \
    SG4_READ_FRAGS(sn, 0, 0);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    mma4_dma(1, st);                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
  {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_part(0, q);
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_part(1, q);
    advance();
    if constexpr (ES == 2) SG2_READ_FRAGS(0, 0, 0);
    else if constexpr (ES == 4) SG4_READ_FRAGS(0, 0, 0);
    else SG8_READ_FRAGS(0, 0, 0);
  }
  for (int kt = 0; kt < KT; kt += 2) {          // two tiles per iteration: the stage index is a compile-time constant;
    if constexpr (ES == 2) {                    // with KT odd the last tile of the last iteration is a phantom tile
      SG2_K_TILE(0, 1);
      SG2_K_TILE(1, 0);
    } else if constexpr (ES == 4) {
      SG4_K_TILE(0, 1);
      SG4_K_TILE(1, 0);
    } else {
      SG8_K_TILE(0, 1);
      SG8_K_TILE(1, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the phantom tiles' DMAs and fragment reads
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue, staged through LDS (free now) so that every global access is 16 bytes per lane.
  // The accumulator layout (row = pixel in the registers, lane = channel) would give 4-byte stores and 4-byte mask loads
  // in 128-byte half-wave segments: 64 store + 64 twin-store + 64 mask-load instructions per 32x64 sub-tile, and the
  // 64-filter layers / the data-grads were epilogue-bound (round-2 probe: the bf16 copy alone cost 16 %, the mask 26 %).
  // Each wave owns 8 KB of LDS: it writes one 32-row x 64-column fp32 sub-tile (32 ds_write_b32), reads it back as rows
  // (8 ds_read_b128: lane -> row 4 it + (l >> 4), columns 4 (l & 15) .. + 3, conflict-free for ds_read_b128's lane
  // groups), and applies scale / bias / mask / accumulate / ReLU on float4s: 8 x 16-byte stores (+ 8 x 8-byte twin stores).
  __builtin_amdgcn_s_barrier();                      // every wave is done with the operand stages
  const bool accum = (p.flags & SG_ACCUM) != 0;
  const bool relu_out = (p.flags & SG_RELU_OUT) != 0;
  const bool ident = (p.flags & SG2_IDENT_OUT) != 0;
  float oscale = 1.f;
  if constexpr (ES == 1) oscale = p.amax_a[0] * p.amax_w[0] * (1.f / (448.f * 448.f));
  float* stage = reinterpret_cast<float*>(smem + wave * 8192);
  const int c4 = lane & 15, rsub = lane >> 4;
  const int ncol = n0 + wn * 64 + 4 * c4;
  float amx = 0.f, amx_s = 0.f;          // running max |v| (and max |rowscale v|) of the values this lane stores
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (split == 0) {
    if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + ncol); bsum.x += t.x; bsum.y += t.y; bsum.z += t.z; bsum.w += t.w; }
    if (p.bias2) { const float4 t = *reinterpret_cast<const float4*>(p.bias2 + ncol); bsum.x += t.x; bsum.y += t.y; bsum.z += t.z; bsum.w += t.w; }
  }
  if (nsplit > 1) {
    // reduction-split tile: partial sums meet through float atomics, straight from the accumulator layout (one 128-byte row
    // segment per half-wave: the full atomic rate; the row-major float4 layout below would scatter every wave instruction
    // over 4 rows x 16 separate dwords -- measured 340 vs 630 TF/s on the 4x20 layer)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 64 + j * 32 + (lane & 31);
      float bs = 0.f;
      if (p.bias && split == 0) bs += p.bias[n];
      if (p.bias2 && split == 0) bs += p.bias2[n];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
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
          float v = acc[i][j][r] * oscale + bs;
          if (p.mask16) {
            if ((short)p.mask16[idx] <= 0) v = 0.f;
          } else if (p.mask && p.mask[idx] <= 0.f) {
            v = 0.f;
          }
          atomicAdd(p.out + idx, v);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // row indices of this pass and the global reads (ReLU mask, accumulate operand) issued up front: eight independent
    // 16-byte loads in flight instead of one dependent load per row group
    size_t idx[8];
    bool live[8];
    float4 mk32[8], prev[8];
    uint2 mk16[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int m = m0 + wm * (TM * 32) + i * 32 + 4 * it + rsub;
      live[it] = m < M;
      const int mm = live[it] ? m : 0;
      if (ident) {
        idx[it] = (size_t)mm * p.N + ncol;
      } else {
        const int b = mm / HW;
        const int rem = mm - b * HW;
        const int yg = rem / p.Wg;
        const int xg = rem - yg * p.Wg;
        idx[it] = ((size_t)(b * p.Ho + yg * p.o_sy + p.o_oy) * p.Wo + xg * p.o_sx + p.o_ox) * p.N + ncol;
      }
      if (p.mask16) mk16[it] = *reinterpret_cast<const uint2*>(p.mask16 + idx[it]);
      else if (p.mask) mk32[it] = *reinterpret_cast<const float4*>(p.mask + idx[it]);
      if (accum) prev[it] = *reinterpret_cast<const float4*>(p.out + idx[it]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * khalf) * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
    // (same wave: LDS operations of one wave complete in order, the compiler orders the dependent reads)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      float4 v = *reinterpret_cast<const float4*>(stage + (4 * it + rsub) * 64 + 4 * c4);
      if (!live[it]) continue;
      v.x = v.x * oscale + bsum.x; v.y = v.y * oscale + bsum.y; v.z = v.z * oscale + bsum.z; v.w = v.w * oscale + bsum.w;
      if (p.mask16) {          // bf16 <= 0  <=>  its int16 is negative or +0 (-0 = 0x8000 is negative as int16)
        if ((short)(mk16[it].x & 0xffffu) <= 0) v.x = 0.f;
        if ((short)(mk16[it].x >> 16) <= 0) v.y = 0.f;
        if ((short)(mk16[it].y & 0xffffu) <= 0) v.z = 0.f;
        if ((short)(mk16[it].y >> 16) <= 0) v.w = 0.f;
      } else if (p.mask) {
        if (mk32[it].x <= 0.f) v.x = 0.f;
        if (mk32[it].y <= 0.f) v.y = 0.f;
        if (mk32[it].z <= 0.f) v.z = 0.f;
        if (mk32[it].w <= 0.f) v.w = 0.f;
      }
      if (accum) { v.x += prev[it].x; v.y += prev[it].y; v.z += prev[it].z; v.w += prev[it].w; }
      if (relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
      bf16x4v h;
      if (p.out16) { h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w; }
      if (p.amax_out) {
        // (operand-only results -- no fp32 tensor: the next launch reads the bf16 values, so THEIR amax is recorded)
        const float4 q = p.out ? v : make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        const float a4 = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)));
        amx = fmaxf(amx, a4);
        if (p.amax_rowscale) amx_s = fmaxf(amx_s, a4 * fabsf(p.amax_rowscale[(m0 + wm * (TM * 32) + i * 32 + 4 * it + rsub) / HW]));
      }
      if (p.out) *reinterpret_cast<float4*>(p.out + idx[it]) = v;
      if (p.out16) *reinterpret_cast<bf16x4v*>(p.out16 + idx[it]) = h;
    }
  }
  if (p.amax_out) {        // (wave-uniform; non-negative floats order like their bit patterns)
    amx = sg_wave_max(amx);
    amx_s = p.amax_rowscale ? sg_wave_max(amx_s) : amx;
    if (lane == 0) {
      sg_atomic_max_nonneg(p.amax_out, amx);
      sg_atomic_max_nonneg(p.amax_out + 1, amx_s);
    }
  }
}

// amax of the rows a launch summed with float atomics (reduction-split tail tiles: their final values exist only after the
// kernel): amax2[0] = max(., max |x|), amax2[1] = max(., max |rowscale[row / rows_per_sample] x|); x [rows, N], N % 4 == 0
__global__ __launch_bounds__(256) void k_amax_rows(const float* __restrict__ x, long rows, int N, long row0, const float* __restrict__ rowscale,
                                                   long rows_per_sample, unsigned* amax_bits) {
  const int n4 = N >> 2;
  const long total = rows * n4;
  float m0 = 0.f, m1 = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[e];
    const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    m0 = fmaxf(m0, a);
    m1 = fmaxf(m1, rowscale ? a * fabsf(rowscale[(row0 + e / n4) / rows_per_sample]) : a);
  }
  m0 = sg_wave_max(m0);
  m1 = sg_wave_max(m1);
  if ((threadIdx.x & 63) == 0) {
    sg_atomic_max_nonneg(reinterpret_cast<float*>(amax_bits), m0);
    sg_atomic_max_nonneg(reinterpret_cast<float*>(amax_bits) + 1, m1);
  }
}

static int g2_split_override = -1;
extern "C" void sg_debug_set_splitk_v2(int n) { g2_split_override = n; }

template <int BN, int ES, bool RELU, int BMT = 256, int NWO = 0>
static int sg2_launch_bn_r(SgIgemm2Args a, hipStream_t s, long* twin_rows_done) {
  const long M = (long)a.Bn * a.Hg * a.Wg;
  const int n_tiles = a.N / BN;
  const int tiles = sg_cdiv(M, BMT) * n_tiles;
  if (tiles <= 0) return SG_OK;
  const int KT_all = a.ntaps * (a.Ca / (128 / ES));
  // one workgroup per CU: a launch of T equal tiles takes ceil(T / 256) tile-times; the tiles beyond the last multiple of
  // 256 (all of them when T < 512) are cut along the reduction and summed with float atomics (model of launch_cfg in
  // conv_igemm.hip at OCC = 1)
  const bool can_split = a.out && !(a.flags & SG_RELU_OUT) && ((a.flags & SG_ACCUM) || (a.flags & SG2_IDENT_OUT));   // (partial tiles meet in the fp32 result)
  constexpr int CUS = BMT == 256 ? 256 : 512;          // workgroup slots of the chip for this configuration
  int full = tiles, nsplit = 1;
  if (can_split && g2_split_override != 1 && KT_all >= 8) {
    const int rem = tiles % CUS;
    if (g2_split_override > 1) {
      full = 0;
      nsplit = g2_split_override < KT_all ? g2_split_override : 1;
    } else if (rem > 0) {
      const int full_c = tiles < 2 * CUS ? 0 : (tiles - rem) / n_tiles * n_tiles;
      const int tail = tiles - full_c;
      auto cost = [&](int sp) { return (double)((tail * sp + CUS - 1) / CUS) / sp * (1.0 + 0.03 * (sp - 1)); };
      int best = 1;
      double best_cost = cost(1);
      const int sp_max = KT_all / 4 < 16 ? KT_all / 4 : 16;
      for (int sp = 2; sp <= sp_max; ++sp) {
        const double c = cost(sp);
        if (c < 0.97 * best_cost) { best = sp; best_cost = c; }
      }
      if (best > 1) { full = full_c; nsplit = best; }
    }
  }
  const long row0 = nsplit > 1 ? (long)(full / n_tiles) * BMT : M;
  if (nsplit > 1 && !(a.flags & SG_ACCUM)) {
    if (hipMemsetAsync(a.out + (size_t)row0 * a.N, 0, sizeof(float) * (size_t)(M - row0) * a.N, s) != hipSuccess) return SG_ERR_LAUNCH;
  }
  a.full_tiles = full;
  a.tail_split = nsplit;
  a.n_tiles_total = tiles;
  static bool attr_done = false;
  static const int lds_pad = getenv("SG2_LDS_PAD") ? atoi(getenv("SG2_LDS_PAD")) : 0;      // (occupancy probe: extra bytes requested, never touched)
  const int LDS_BYTES = 2 * BMT * 128 + 2 * BN * 128 + lds_pad;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sg_igemm_bf16v2_kernel<BN, ES, RELU, BMT, NWO>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return SG_ERR_UNSUPPORTED;
    }
    attr_done = true;
  }
  SG_KERNEL((sg_igemm_bf16v2_kernel<BN, ES, RELU, BMT, NWO>), dim3(full + (tiles - full) * nsplit), dim3((NWO ? NWO : (BMT == 256 ? 8 : 4)) * 64), LDS_BYTES, s, a);
  if (twin_rows_done) *twin_rows_done = (a.flags & SG2_IDENT_OUT) ? row0 : (nsplit > 1 ? 0 : M);
  if (a.amax_out && nsplit > 1) {
    // the split tiles' values are final only now: their amax comes from a sweep over those rows (identity layouts: the rows
    // behind row0; otherwise over the whole result -- the full tiles' contribution is then counted twice, harmlessly)
    const bool ident = (a.flags & SG2_IDENT_OUT) != 0;
    const long r0 = ident ? row0 : 0, nrows = ident ? M - row0 : (long)a.Bn * a.Ho * a.Wo;
    const long per_sample = ident ? (long)a.Hg * a.Wg : (long)a.Ho * a.Wo;
    SG_KERNEL(k_amax_rows, dim3(sg_grid_for(nrows * (a.N / 4), 256)), dim3(256), 0, s, a.out + (size_t)r0 * a.N, nrows, a.N, r0,
                       a.amax_rowscale, per_sample, reinterpret_cast<unsigned*>(a.amax_out));
  }
  return sg_launch_status();
}

template <int BN, int ES, int BMT = 256, int NWO = 0>
static int sg2_launch_bn(const SgIgemm2Args& a, hipStream_t s, long* twin_rows_done) {
  if constexpr (ES == 1) return sg2_launch_bn_r<BN, ES, false, BMT>(a, s, twin_rows_done);     // (fp8: the ReLU is applied by the fp8 convert)
  else return (a.flags & SG_RELU_IN) ? sg2_launch_bn_r<BN, ES, true, BMT, NWO>(a, s, twin_rows_done) : sg2_launch_bn_r<BN, ES, false, BMT, NWO>(a, s, twin_rows_done);
}

// SG2_TILE128 / sg_debug_set_tile128: 0 = never (default), 1 = the rule below, 2 = whenever the filter count is a multiple of 128
static int g2_tile128 = getenv("SG2_TILE128") ? atoi(getenv("SG2_TILE128")) : 0;
extern "C" void sg_debug_set_tile128(int mode) { g2_tile128 = mode; }
// where the 128 x 128 configuration measured ahead: 1x1 convolutions over >= 1024 channels on few pixels (the epilogue is most of the launch)
static bool sg2_prefer_128(const SgIgemm2Args& a, long m_tiles) {
  return a.ntaps == 1 && a.Ca >= 1024 && m_tiles <= 1024;
}

// -> SG_OK, and *twin_rows_done = number of leading output rows (pixels) whose bf16 copy the kernel wrote itself (the
// rows of reduction-split tiles are summed by atomics: their bf16 copy needs a convert pass afterwards)
int sg_launch_igemm_bf16v2(const SgIgemm2Args& a_in, hipStream_t s, long* twin_rows_done, int es) {
  SgIgemm2Args a = a_in;
  if ((a.Ca % (128 / es)) || (a.N % 64) || a.ntaps < 1 || a.ntaps > SG_MAX_TAPS) return SG_ERR_UNSUPPORTED;
  if (es == 1 && ((a.N % 256) || !a.amax_a || !a.amax_w)) return SG_ERR_UNSUPPORTED;
  if (a.group_rows && ((a.group_rows % 128) || a.ntaps != 1 || a.Ha != 1 || a.Wa != 1 || a.Hg != 1 || a.Wg != 1 || a.Ho != 1 || a.Wo != 1 || a.out16 ||
                       a.mask || a.mask16 || a.amax_out || (long)a.Bn % a.group_rows))
    return SG_ERR_ARG;      // (grouped launches: plain [rows, Ca] x [N, Ca]^T products per group)
  const long a_bytes = (long)es * (a.group_rows ? a.group_rows : a.Bn) * a.Ha * a.Wa * a.Ca;
  long w_elems = 0;
  for (int t = 0; t < a.ntaps; ++t) w_elems = a.taps[t].w_off > w_elems ? a.taps[t].w_off : w_elems;
  w_elems += (long)a.N * a.Ca;
  if (a_bytes >= (1L << 32) - 64 || es * w_elems >= (1L << 32) - 64 || (!a.group_rows && (long)a.Bn * a.Ho * a.Wo * a.N >= (1L << 31)) ||
      (long)a.Bn * a.Hg * a.Wg >= (1L << 31) - 256)
    return SG_ERR_UNSUPPORTED;      // (grouped launches index the result with 64-bit row offsets only: identity layout)
  if (a.o_sy == 1 && a.o_sx == 1 && a.o_oy == 0 && a.o_ox == 0 && a.Ho == a.Hg && a.Wo == a.Wg) a.flags |= SG2_IDENT_OUT;
  // Tile width: the widest one that divides N, unless the launch cannot be cut along the reduction (a fused output ReLU
  // rules the float-atomic tail split out) and a narrower tile fills the 256 CUs' rounds better.  Relative tile times:
  // the bf16 loop is bound by the L2 -> LDS feed ((256 + BN) x 128 bytes per tile), the fp32 loop by the matrix pipe.
  const long m_tiles = sg_cdiv((long)a.Bn * a.Hg * a.Wg, SG2_BM);
  const bool can_split = a.out && !(a.flags & SG_RELU_OUT) && ((a.flags & SG_ACCUM) || (a.flags & SG2_IDENT_OUT)) && a.ntaps * (a.Ca / (128 / es)) >= 8;
  int bn = 0;
  double best = 1e30;
  for (int c = 256; c >= 64; c >>= 1) {
    if (a.N % c) continue;
    const long tiles = m_tiles * (a.N / c);
    const double rounds = can_split && tiles % 256 ? (tiles < 256 ? 1.0 : (double)tiles / 256.0) : (double)((tiles + 255) / 256);
    // (fp8: the loop moves the same bytes per tile as bf16 in half the matrix time -- the narrow tiles lose more)
    const double t_tile = es == 4 ? (c == 256 ? 1.0 : c == 128 ? 0.525 : 0.29) : es == 1 ? (c == 256 ? 1.0 : c == 128 ? 0.8 : 0.7)
                                                                                        : (c == 256 ? 1.0 : c == 128 ? 0.75 : 0.625);
    if (rounds * t_tile < 0.97 * best) { best = rounds * t_tile; bn = c; }
  }
  static const int bn_env = getenv("SG2_FORCE_BN") ? atoi(getenv("SG2_FORCE_BN")) : 0;      // (debugging aid)
  if (bn_env > 0 && a.N % bn_env == 0) bn = bn_env;
  if (es == 1) {          // (narrow tiles: the small grids of the 8-way shard batch -- 160 tiles of 256 x 256 on the 4x20 layers at B = 128)
    if (bn == 256) return sg2_launch_bn<256, 1>(a, s, twin_rows_done);
    if (bn == 128) return sg2_launch_bn<128, 1>(a, s, twin_rows_done);
    return sg2_launch_bn<64, 1>(a, s, twin_rows_done);
  }
  if (es == 4) {
    // grouped launches (Winograd-domain products): 128 x 128 tiles, two workgroups per CU -- level with the 256-row tiles on the
    // large launches (profiles/r03_probe_winograd.txt: 10.49 vs 10.58 ms, 8.67 vs 8.65 ms) and half the plane padding on the small ones
    // (round 4) eight waves per 128 x 128 tile (NWO = 8) on launches of fewer than eight rounds of the chip's 512 workgroup slots -- the
    // shard batch: +8-20 % on its small planes, 34.1 -> 33.3 ms per step (profiles/r04_probe_w8.txt).  On the headline batch the variant is
    // +3 % in isolation but -0.7 % in the two-stream step (its four waves per SIMD hold 492 of the 512 VGPRs: no room for the other
    // network's HBM-bound waves beside it), so large launches keep four waves.  SG2_W8 = 0: never, 2: always.
    static const int w8 = getenv("SG2_W8") ? atoi(getenv("SG2_W8")) : 1;
    const long g_tiles = (long)sg_cdiv((long)a.Bn * a.Hg * a.Wg, 128) * (a.N / 128);
    if (a.group_rows && a.N % 128 == 0 && (w8 == 2 || (w8 == 1 && g_tiles < 8 * 512))) return sg2_launch_bn<128, 4, 128, 8>(a, s, twin_rows_done);
    if (a.group_rows && a.N % 128) return a.N % 64 == 0 ? sg2_launch_bn<64, 4, 128>(a, s, twin_rows_done) : SG_ERR_UNSUPPORTED;      // (64-filter layers: 128 x 64 tiles, three workgroups per CU)
    if (a.group_rows) return sg2_launch_bn<128, 4, 128>(a, s, twin_rows_done);
    if (bn == 256) return sg2_launch_bn<256, 4>(a, s, twin_rows_done);
    if (bn == 128) return sg2_launch_bn<128, 4>(a, s, twin_rows_done);
    return sg2_launch_bn<64, 4>(a, s, twin_rows_done);
  }
  // bf16: the 128 x 128 / two-workgroups-per-CU configuration.  MEASURED SLOWER than the wide tiles on every layer of the step but
  // the 4x20 1x1 shortcut (profiles/r03_probe_tile128.txt: 512 -> 512 16x80 forward 1 116 -> 892 TF/s, data-grad 976 -> 863): twice the
  // L2 -> LDS bytes per FLOP and a third more LDS reads per MFMA cost more than the overlapped epilogue returns.  Kept (parity-green
  // when forced: sg_debug_set_tile128(2) / SG2_TILE128=2) as the starting point for a 256 x 128 two-workgroup variant; off by default.
  if (a.N % 128 == 0 && bn >= 128 && (g2_tile128 == 2 || (g2_tile128 == 1 && sg2_prefer_128(a, m_tiles)))) return sg2_launch_bn<128, 2, 128>(a, s, twin_rows_done);
  if (bn == 256) return sg2_launch_bn<256, 2>(a, s, twin_rows_done);
  if (bn == 128) return sg2_launch_bn<128, 2>(a, s, twin_rows_done);
  return sg2_launch_bn<64, 2>(a, s, twin_rows_done);
}

// ------------------------------------------------------------------------------------------
// C-ABI (include/scrabble_hip.h).  Contracts of sg_conv2d_fwd / sg_conv2d_bwd_data with
//   x16 / dy16 : the activation operand as a bf16 NHWC tensor (sg_cvt_bf16 or a previous launch's y16 / dx16);
//   wp         : the packed bf16 filter of sg_pack_filter_bf16 (forward: [tap][Cout][Cin], data-grad: [tap][Cin][Cout]);
//   y16 / dx16 : nullable; receives the bf16 copy of the fp32 result.
// Shapes that do not qualify (reduction channels % 64, output channels % 64) return SG_ERR_UNSUPPORTED: the caller
// falls back to sg_conv2d_fwd_bf16 / sg_conv2d_bwd_data_bf16 on the fp32 tensor.
// ------------------------------------------------------------------------------------------
static int finish_twin(const SgIgemm2Args& a, long rows_done, hipStream_t s) {
  if (!a.out16) return SG_OK;
  const long M = (long)a.Bn * a.Ho * a.Wo;
  if (rows_done >= M) return SG_OK;
  const long n = (M - rows_done) * a.N;
  SG_KERNEL(k_cvt_bf16, dim3(sg_grid_for(n / 8, 256)), dim3(256), 0, s, a.out + (size_t)rows_done * a.N,
                     a.out16 + (size_t)rows_done * a.N, n / 8, 0, (const float*)nullptr, 8L);
  return sg_launch_status();
}

extern "C" int sg_conv2d_fwd_bf16v2(const void* x16, const void* wp_fwd, const float* bias, const float* bias2, float* y, void* y16,
                                    int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, float* amax_y,
                                    void* stream) {
  if (!x16 || !wp_fwd || (!y && !y16) || (!y && (flags & SG_ACCUM)) || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;      // (y null: operand-only result, y16)
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemm2Args a{};
  a.a = (const u16*)x16; a.w = (const u16*)wp_fwd; a.out = y; a.out16 = (u16*)y16; a.bias = bias; a.bias2 = bias2; a.mask = nullptr;
  a.amax_out = amax_y;
  a.Bn = B; a.Ha = H; a.Wa = W; a.Ca = Cin; a.Hg = Ho; a.Wg = Wo; a.a_sy = 1; a.a_sx = 1;
  a.Ho = Ho; a.Wo = Wo; a.N = Cout; a.o_sy = 1; a.o_sx = 1; a.o_oy = 0; a.o_ox = 0;
  a.ntaps = kh * kw; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  long done = 0;
  const int rc = sg_launch_igemm_bf16v2(a, (hipStream_t)stream, &done, 2);
  return rc != SG_OK ? rc : finish_twin(a, done, (hipStream_t)stream);
}

extern "C" int sg_conv2d_bwd_data_bf16v2(const void* dy16, const void* wp_bwd, const float* mask, const void* mask16, float* dx, void* dx16,
                                         int B, int H, int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, float* amax_dx,
                                         const float* amax_rowscale, void* stream) {
  if (!dy16 || !wp_bwd || !dx || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemm2Args a{};
  a.a = (const u16*)dy16; a.w = (const u16*)wp_bwd; a.out = dx; a.out16 = (u16*)dx16; a.mask = mask; a.mask16 = (const u16*)mask16;
  a.amax_out = amax_dx; a.amax_rowscale = amax_rowscale;
  a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * Cin * Cout};
  long done = 0;
  const int rc = sg_launch_igemm_bf16v2(a, (hipStream_t)stream, &done, 2);
  return rc != SG_OK ? rc : finish_twin(a, done, (hipStream_t)stream);
}

// fp32 operands on the same kernel (ES = 4): x / dy are the fp32 NHWC tensors themselves; wt_fwd = the filter transposed to
// [tap][Cout][Cin] (sg_transpose_filter(w, taps, K = Cin, N = Cout), once per optimizer step); the data-grad reads the
// filter w [kh,kw,Cin,Cout] as it is (per tap [N = Cin][K = Cout]).  Contracts of sg_conv2d_fwd / sg_conv2d_bwd_data;
// SG_ERR_UNSUPPORTED unless reduction channels % 32 == 0 and output channels % 64 == 0 (caller: the first-generation entry).
extern "C" int sg_conv2d_fwd_v2(const float* x, const float* wt_fwd, const float* bias, const float* bias2, float* y, int B, int H,
                                int W, int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream) {
  if (!x || !wt_fwd || !y || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemm2Args a{};
  a.a = (const u16*)x; a.w = (const u16*)wt_fwd; a.out = y; a.bias = bias; a.bias2 = bias2;
  a.Bn = B; a.Ha = H; a.Wa = W; a.Ca = Cin; a.Hg = Ho; a.Wg = Wo; a.a_sy = 1; a.a_sx = 1;
  a.Ho = Ho; a.Wo = Wo; a.N = Cout; a.o_sy = 1; a.o_sx = 1; a.o_oy = 0; a.o_ox = 0;
  a.ntaps = kh * kw; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm_bf16v2(a, (hipStream_t)stream, nullptr, 4);
}

extern "C" int sg_conv2d_bwd_data_v2(const float* dy, const float* w, const float* mask, float* dx, int B, int H, int W, int Cin,
                                     int Cout, int kh, int kw, int pad_same, int flags, void* stream) {
  if (!dy || !w || !dx || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemm2Args a{};
  a.a = (const u16*)dy; a.w = (const u16*)w; a.out = dx; a.mask = mask;
  a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm_bf16v2(a, (hipStream_t)stream, nullptr, 4);
}

// ==========================================================================================================
// Weight gradient, second generation:  dW_t[c][n] += sum_m P[pix(m) + tap_t][c] * Q[m][n]
//   P = the layer input (bf16 NHWC [B,H,W,Cp]), Q = the output gradient (bf16 NHWC [B,H,W,Cq], already multiplied by the
//   per-sample factors of the shared sweep where those apply: sg_cvt_bf16's rowscale), stride-1 convolutions only.
// The reduction index (pixels) is the SLOW index of both operands in memory, the MFMA wants it fastest in each lane's
// fragment.  Round 1 transposed in registers (8x4 blocks through VGPRs and ds_write).  Here the tiles go global -> LDS by
// DMA exactly as they lie in memory (64 pixels x CT / NT channels) and the transposition is done by the LDS read itself:
// ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block, i.e. four consecutive pixels
// of one channel -- two of them make the 8-pixel fragment of v_mfma_f32_32x32x16_bf16 for BOTH operands.
// LDS images (guide T10; the DMA applies the swizzle on the source side):
//   256-channel operand tile: two 128-channel sub-tiles with 256-byte rows, chunk ch of row r in slot
//                             ch ^ (((r & 3) << 2) | ((r >> 2) & 3))                                    (image (b))
//   64-channel operand tile : 8-row x 32-channel subtiles of 512 bytes,
//                             off = 1024 (r >> 3) + 512 (ch >> 2) + 64 (r & 7) + 16 ((ch & 3) ^ ((r >> 2) & 3))   (image (a))
// Workgroup = one (tap, CT-channel c-tile, NT-channel n-tile) x one chunk of pixels, 8 waves:
//   CT 256, NT 256: 2 x 4 waves, wave tile 128 x 64 (12 transposed reads per 8 MFMAs)            -- the large layers
//   CT  64, NT 256: 2 x 4 waves, wave tile  32 x 64                                              -- 64 -> 512 (3x3, 1x1)
//   CT  64, NT  64: 2 x 2 waves x 2 k-groups (each group takes two of a tile's four k-steps)    -- 64 -> 64
// Same two-stage pipeline and phantom-tile loop as sg_igemm_bf16v2_kernel.  Partial sums of the pixel chunks (and of the
// k-groups) meet in dW through float atomics (one 128-byte row segment per half-wave).
struct SgWgrad2Args {
  const u16* p;        // bf16 [Bn, H, W, Cp]
  const u16* q;        // bf16 [Bn, H, W, Cq]
  float* dw;           // per tap a [Cp x Cq] row-major fp32 matrix at dw + taps[t].w_off
  int Bn, H, W, Cp, Cq;
  int ntaps, flags;    // SG_RELU_IN applies to P
  int mchunk;          // pixels per workgroup (multiple of 64)
  int c_tiles, n_tiles, nchunks;
  int p_sy, p_sx;      // STRIDED kernels only: P lives on the (p_sy H) x (p_sx W) grid and pixel (y, x) of the base grid reads
                       // P at (p_sy y + dy, p_sx x + dx) -- the gradient operand of a transposed convolution
  SgTap taps[SG_MAX_TAPS];
};

template <int CT, int NT, bool STRIDED>
__global__ __launch_bounds__(512, 2) void sg_wgrad_bf16v2_kernel(const SgWgrad2Args p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr bool PA = CT == 64, QA = NT == 64;        // image (a) operands
  constexpr int PT = 64 * CT * 2, QT = 64 * NT * 2;   // tile bytes
  constexpr int NP = CT / 64, NQ = NT / 64;           // DMA instructions per thread: 4 (256 channels) or 1 (64)
  constexpr int KG = (CT == 64 && NT == 64) ? 2 : 1;  // k-groups
  constexpr int WN = NT == 256 ? 4 : 2, WM = 8 / KG / WN;
  constexpr int TM = CT / WM / 32, TN = NT / WN / 32;
  constexpr int SPT = 4 / KG;                         // k-steps per tile and wave
  constexpr int NF = 2 * (TM + TN);                   // transposed reads per k-step
  static_assert((TM == 1 || TM == 4) && (TN == 1 || TN == 2) && WM * WN * KG == 8, "wave layout");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave / (WM * WN), wm = (wave / WN) % WM, wn = wave % WN;
  const long M = (long)p.Bn * p.H * p.W;
  // block -> (pixel chunk, combo): the combos of one chunk are adjacent (they re-read the same pixels from L2 / MALL)
  const int combos = p.ntaps * p.c_tiles * p.n_tiles;
  const int chunk = blockIdx.x / combos;
  int combo = blockIdx.x - chunk * combos;
  const int tap = combo / (p.c_tiles * p.n_tiles);
  combo -= tap * p.c_tiles * p.n_tiles;
  const int c0 = (combo / p.n_tiles) * CT, n0 = (combo % p.n_tiles) * NT;
  const long m_begin = (long)chunk * p.mchunk;
  const long m_end = m_begin + p.mchunk < M ? m_begin + p.mchunk : M;
  const int KT = (int)((m_end - m_begin + 63) / 64);
  const int tdy = p.taps[tap].dy, tdx = p.taps[tap].dx;

  // ---- DMA lane roles.
  //  256-channel tile: instruction ii = 8 i + wave (i = 0..3) moves sub-tile (ii & 1), rows 4 (ii >> 1) .. + 3: lane -> row
  //                    (l >> 4) of the four, slot (l & 15); it fetches chunk slot ^ g(row) of its pixel's 128-channel half
  //  64-channel tile : the one instruction of wave w moves rows 8 w .. + 7: lane -> subtile (l >> 5), row ((l & 31) >> 2),
  //                    slot (l & 3); it fetches chunk (subtile << 2) | (slot ^ ((row >> 2) & 3))
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(sg2_zero_page) + 16 * (lane & 15);
  const int HW = p.H * p.W;
  constexpr bool SAMEROLE = (PA == QA) && (NP == NQ);   // P and Q rows of a lane coincide: one pixel cursor serves both
  int ry[NP], rx[NP];          // (y, x) cursor of the pixel this lane loads for P instruction i in the CURRENT load tile
  int ppix[STRIDED ? NP : 1];  // STRIDED: index of the P pixel (tap included) on P's own grid
  const int Hs = p.p_sy * p.H, Ws = p.p_sx * p.W;
  int rmp[NP], rmq[SAMEROLE ? 1 : NQ];       // pixel indices of its P / Q rows, relative to m_begin
  unsigned cswp[NP], cswq[NQ]; // byte offset of its (swizzled) chunk inside the pixel's channel vector (before c0 / n0)
  auto role = [&](bool img_a, int i, int& row, unsigned& csw) {
    if (img_a) {
      row = 8 * wave + ((lane & 31) >> 2);
      csw = 16u * (unsigned)(((lane >> 5) << 2) | ((lane & 3) ^ ((row >> 2) & 3)));
    } else {
      const int ii = 8 * i + wave;
      row = 4 * (ii >> 1) + (lane >> 4);
      const int g = ((row & 3) << 2) | ((row >> 2) & 3);
      csw = 2u * (unsigned)((ii & 1) * 128) + 16u * (unsigned)((lane & 15) ^ g);
    }
  };
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int row;
    role(PA, i, row, cswp[i]);
    const long m = m_begin + row;
    rmp[i] = row;
    const long mm = m < M ? m : 0;
    const int rem = (int)(mm % HW);
    ry[i] = rem / p.W;
    rx[i] = rem - ry[i] * p.W;
    if constexpr (STRIDED) ppix[i] = ((int)(mm / HW) * Hs + p.p_sy * ry[i] + tdy) * Ws + p.p_sx * rx[i] + tdx;
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    int row;
    role(QA, i, row, cswq[i]);
    if constexpr (!SAMEROLE) rmq[i] = row;
  }
  const int m_len = (int)(m_end - m_begin);
  const int adv_y = (64 % HW) / p.W, adv_x = (64 % HW) % p.W;       // cursor advance of 64 pixels (within one image plane)
  const int Hh = p.H;
  const unsigned long long p_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(p.p) + 2ull * (unsigned)c0 +
                                      (STRIDED ? 0ull : 2ull * (unsigned long long)m_begin * p.Cp);
  const unsigned long long q_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(p.q) + 2ull * (unsigned)n0 + 2ull * (unsigned long long)m_begin * p.Cq;
  const int pp_adv = p.p_sy * Ws * adv_y + p.p_sx * adv_x, pp_wrap = Ws * (p.p_sy - 1);      // STRIDED: advance of ppix per tile / per x wrap
  const unsigned long long z_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(zero);
  const long tap_shift = (long)tdy * p.W + tdx;                      // pixel shift of the tap (may be negative)
  // LDS: P stage 0 | P stage 1 | Q stage 0 | Q stage 1
  // part i (0..3) of the load tile at the cursors into stage `st`, then advance those cursors by one tile (64 pixels)
  auto issue_part = [&](int st, int i) {
    if (i < NP) {
      const bool live = rmp[i] < m_len;
      const int sy = (STRIDED ? p.p_sy * ry[i] : ry[i]) + tdy, sx = (STRIDED ? p.p_sx * rx[i] : rx[i]) + tdx;
      const bool okp = live && sy >= 0 && sy < (STRIDED ? Hs : Hh) && sx >= 0 && sx < (STRIDED ? Ws : p.W);
      const unsigned long long pa = p_base64 + (unsigned long long)((STRIDED ? (long)ppix[STRIDED ? i : 0] : (long)rmp[i] + tap_shift) * p.Cp * 2) + cswp[i];
      const unsigned plo = okp ? (unsigned)pa : (unsigned)z_base64, phi = okp ? (unsigned)(pa >> 32) : (unsigned)(z_base64 >> 32);
      const unsigned char* src_p = reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)phi << 32) | plo));
      const int ii = 8 * i + wave;
      unsigned char* dst_p = smem + st * PT + (PA ? wave * 1024 : (ii & 1) * 16384 + (ii >> 1) * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_p, (__attribute__((address_space(3))) void*)dst_p, 16, 0, 0);
      if constexpr (!SAMEROLE) rmp[i] += 64;
      rx[i] += adv_x;
      const int wx = rx[i] >= p.W ? 1 : 0;
      rx[i] -= wx * p.W;
      if constexpr (STRIDED) ppix[i] += pp_adv + wx * pp_wrap;
      ry[i] += adv_y + wx;
      ry[i] -= ry[i] >= Hh ? Hh : 0;
    }
    if (i < NQ) {
      const int rq = SAMEROLE ? rmp[i < NP ? i : 0] : rmq[SAMEROLE ? 0 : i];
      const bool live = rq < m_len;
      const unsigned long long qa = q_base64 + (unsigned long long)((long)rq * p.Cq * 2) + cswq[i];
      const unsigned qlo = live ? (unsigned)qa : (unsigned)z_base64, qhi = live ? (unsigned)(qa >> 32) : (unsigned)(z_base64 >> 32);
      const unsigned char* src_q = reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)qhi << 32) | qlo));
      const int ii = 8 * i + wave;
      unsigned char* dst_q = smem + 2 * PT + st * QT + (QA ? wave * 1024 : (ii & 1) * 16384 + (ii >> 1) * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_q, (__attribute__((address_space(3))) void*)dst_q, 16, 0, 0);
      if constexpr (SAMEROLE) rmp[i < NP ? i : 0] += 64;
      else rmq[SAMEROLE ? 0 : i] += 64;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- transposed fragment reads: lane = 16 grp4 + 4 q + pp; h = l >> 5 (pixel half of the k-step), grp = (l >> 4) & 1
  //      (16-channel half of a 32-channel group); read u (0,1) covers pixels 16 ks + 8 h + 4 u + q.  The k-step (and, for
  //      image (a), the 32-channel group) is an immediate of the read.
  typedef int v2i __attribute__((ext_vector_type(2)));
  typedef int v4i __attribute__((ext_vector_type(4)));
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int q4 = (lane >> 2) & 3, pp = lane & 3, grp = (lane >> 4) & 1, h = lane >> 5;
  unsigned pa_addr[TM][2], qb_addr[TN][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 8 * h + 4 * u + q4;                       // + 16 ks
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (PA) {
        const int ch = 4 * wm + 2 * grp + (pp >> 1);          // TM == 1: the wave's 32 channels = subtile wm
        pa_addr[i][u] = lds0 + (unsigned)(1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3)) + 8 * (pp & 1));
      } else {
        const int g = ((row & 3) << 2) | ((row >> 2) & 3);
        const int ch = 4 * i + 2 * grp + (pp >> 1);
        pa_addr[i][u] = lds0 + (unsigned)(wm * 16384 + 256 * row + 16 * (ch ^ g) + 8 * (pp & 1));
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (QA) {
        const int ch = 4 * wn + 2 * grp + (pp >> 1);          // TN == 1
        qb_addr[j][u] = lds0 + (unsigned)(2 * PT + 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3)) + 8 * (pp & 1));
      } else {
        const int g = ((row & 3) << 2) | ((row >> 2) & 3);
        const int ch = 8 * (wn & 1) + 4 * j + 2 * grp + (pp >> 1);
        qb_addr[j][u] = lds0 + (unsigned)(2 * PT + (wn >> 1) * 16384 + 256 * row + 16 * (ch ^ g) + 8 * (pp & 1));
      }
    }
  }
  // byte offset of k-step ks inside a tile: 16 pixel rows = 4096 B (256-byte rows) or 2048 B (image (a): two 8-row groups)
  constexpr int PKS = PA ? 2048 : 4096, QKS = QA ? 2048 : 4096;
  const bool relu_in = (p.flags & SG_RELU_IN) != 0;
  const short rfloor = relu_in ? (short)0 : (short)0x8000;
  const s16x8 rfloor8 = {rfloor, rfloor, rfloor, rfloor, rfloor, rfloor, rfloor, rfloor};
  v2i af[2][4][2], bfr[2][2][2];          // [slot][group][u]
#define SGW_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // ks = the step's index inside the tile for THIS wave's k-group: (step) for KG == 1, 2 kg + step for KG == 2 -- the
  // k-group offset is folded into the address registers below so that `ks` stays a literal
#define SGW_READ_FRAGS(st, ks, slot)                                                             \
  do {                                                                                           \
    SGW_TR(af[slot][0][0], pa_addr[0][0], (st) * PT + (ks) * PKS);                               \
    SGW_TR(af[slot][0][1], pa_addr[0][1], (st) * PT + (ks) * PKS);                               \
    if constexpr (TM > 1) {                                                                      \
      SGW_TR(af[slot][1][0], pa_addr[TM > 1 ? 1 : 0][0], (st) * PT + (ks) * PKS);                \
      SGW_TR(af[slot][1][1], pa_addr[TM > 1 ? 1 : 0][1], (st) * PT + (ks) * PKS);                \
      SGW_TR(af[slot][2][0], pa_addr[TM > 2 ? 2 : 0][0], (st) * PT + (ks) * PKS);                \
      SGW_TR(af[slot][2][1], pa_addr[TM > 2 ? 2 : 0][1], (st) * PT + (ks) * PKS);                \
      SGW_TR(af[slot][3][0], pa_addr[TM > 3 ? 3 : 0][0], (st) * PT + (ks) * PKS);                \
      SGW_TR(af[slot][3][1], pa_addr[TM > 3 ? 3 : 0][1], (st) * PT + (ks) * PKS);                \
    }                                                                                            \
    SGW_TR(bfr[slot][0][0], qb_addr[0][0], (st) * QT + (ks) * QKS);                              \
    SGW_TR(bfr[slot][0][1], qb_addr[0][1], (st) * QT + (ks) * QKS);                              \
    if constexpr (TN > 1) {                                                                      \
      SGW_TR(bfr[slot][1][0], qb_addr[TN > 1 ? 1 : 0][0], (st) * QT + (ks) * QKS);               \
      SGW_TR(bfr[slot][1][1], qb_addr[TN > 1 ? 1 : 0][1], (st) * QT + (ks) * QKS);               \
    }                                                                                            \
  } while (0)
  if constexpr (KG == 2) {                // fold the k-group's first step (2 kg) into the per-lane addresses
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pa_addr[0][u] += (unsigned)(2 * kg * PKS);
      qb_addr[0][u] += (unsigned)(2 * kg * QKS);
    }
  }
  auto mma = [&](int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const v4i a4 = {af[slot][i][0][0], af[slot][i][0][1], af[slot][i][1][0], af[slot][i][1][1]};
      const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a4), rfloor8));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const v4i b4 = {bfr[slot][j][0][0], bfr[slot][j][0][1], bfr[slot][j][1][0], bfr[slot][j][1][1]};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b4), acc[i][j], 0, 0, 0);
      }
    }
  };
  // the MFMAs of a step with the four DMA parts of the next load tile (stage `st`) between the MFMA groups: the whole tile is
  // issued one tile ahead of its use (see SG2_K_TILE)
  auto mma_dma = [&](int slot, int st) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const v4i a4 = {af[slot][i][0][0], af[slot][i][0][1], af[slot][i][1][0], af[slot][i][1][1]};
      const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a4), rfloor8));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const v4i b4 = {bfr[slot][j][0][0], bfr[slot][j][0][1], bfr[slot][j][1][0], bfr[slot][j][1][1]};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b4), acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = i * 4 / TM; q < (i + 1) * 4 / TM; ++q) issue_part(st, q);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // wait until only the NF transposed reads just issued are still in flight
#define SGW_WAIT_FRAGS()                                                                \
  do {                                                                                  \
    if constexpr (NF == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");         \
    else if constexpr (NF == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");      \
    else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                             \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)

  {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(1, i);
    SGW_READ_FRAGS(0, 0, 0);
  }
  // four k-steps per tile (one k-group)
#define SGW_K_TILE4(st, sn)                                        \
  do {                                                             \
    SGW_READ_FRAGS(st, 1, 1);                                      \
    SGW_WAIT_FRAGS();                                              \
    mma(0);                                                        \
    __builtin_amdgcn_sched_barrier(0);                             \
    SGW_READ_FRAGS(st, 2, 0);                                      \
    SGW_WAIT_FRAGS();                                              \
    mma(1);                                                        \
    __builtin_amdgcn_sched_barrier(0);                             \
    SGW_READ_FRAGS(st, 3, 1);                                      \
    SGW_WAIT_FRAGS();                                              \
    mma(0);                                                        \
    __builtin_amdgcn_sched_barrier(0);                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                                  \
    SGW_READ_FRAGS(sn, 0, 0);                                      \
    __builtin_amdgcn_sched_barrier(0);                             \
    mma_dma(1, st);                                                \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
  // two k-steps per tile and wave (two k-groups)
#define SGW_K_TILE2(st, sn)                                        \
  do {                                                             \
    SGW_READ_FRAGS(st, 1, 1);                                      \
    SGW_WAIT_FRAGS();                                              \
    mma(0);                                                        \
    __builtin_amdgcn_sched_barrier(0);                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                                  \
    SGW_READ_FRAGS(sn, 0, 0);                                      \
    __builtin_amdgcn_sched_barrier(0);                             \
    mma_dma(1, st);                                                \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
  for (int kt = 0; kt < KT; kt += 2) {
    if constexpr (SPT == 4) {
      SGW_K_TILE4(0, 1);
      SGW_K_TILE4(1, 0);
    } else {
      SGW_K_TILE2(0, 1);
      SGW_K_TILE2(1, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: accumulator row = c, lane = n; dW[c][n] += (atomics: nchunks x KG adders per address)
  float* dwt = p.dw + p.taps[tap].w_off;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        atomicAdd(dwt + (size_t)c * p.Cq + n, acc[i][j][r]);
      }
    }
  }
}

template <int CT, int NT, bool STRIDED = false>
static int sg2_launch_wgrad(SgWgrad2Args a, long M, hipStream_t s) {
  a.c_tiles = a.Cp / CT;
  a.n_tiles = a.Cq / NT;
  const int combos = a.ntaps * a.c_tiles * a.n_tiles;
  // Pixel chunks per combo.  One workgroup per CU at a time, so a launch of W workgroups takes ceil(W / 256) rounds of
  // (k-tiles per chunk) x t_tile + t_epi, where t_epi is the CT x NT x 4 bytes of float atomics every workgroup ends with
  // (measured for 256 x 256: WRITE_SIZE 550 MB per launch at 2304 workgroups, the chip adds ~1.3 TB/s -> ~50 us per round
  // against ~0.9 us per k-tile).  Few long workgroups win: the count that minimises the modelled time, >= 4 k-tiles each.
  static const int wg_env = getenv("SG_WGRAD2_CHUNKS") ? atoi(getenv("SG_WGRAD2_CHUNKS")) : 0;
  const double t_tile = 0.9 * (CT * NT) / 65536.0 + 0.15, t_epi = 50.0 * (CT * NT) / 65536.0 + 2.0;
  const long tiles_all = (M + 63) / 64;
  const long max_chunks = tiles_all / 4 > 0 ? tiles_all / 4 : 1;
  long nchunks = 1;
  double best = 1e30;
  for (long cc = 1; cc <= max_chunks && cc * combos <= 8192; ++cc) {
    const long Wg = combos * cc;
    const double t = (double)((Wg + 255) / 256) * ((double)((tiles_all + cc - 1) / cc) * t_tile + t_epi);
    if (t < best) { best = t; nchunks = cc; }
  }
  if (wg_env > 0) nchunks = wg_env < max_chunks ? wg_env : max_chunks;
  if (sg_deterministic()) nchunks = 1;        // one adder per dW address (the 64 x 64 config's two k-groups: see the entry points)
  long mchunk = (M + nchunks - 1) / nchunks;
  mchunk = (mchunk + 63) / 64 * 64;
  nchunks = (M + mchunk - 1) / mchunk;
  a.mchunk = (int)mchunk;
  a.nchunks = (int)nchunks;
  constexpr int LDS_BYTES = 2 * 64 * CT * 2 + 2 * 64 * NT * 2;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sg_wgrad_bf16v2_kernel<CT, NT, STRIDED>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return SG_ERR_UNSUPPORTED;
    }
    attr_done = true;
  }
  SG_KERNEL((sg_wgrad_bf16v2_kernel<CT, NT, STRIDED>), dim3((unsigned)(combos * nchunks)), dim3(512), LDS_BYTES, s, a);
  return sg_launch_status();
}

// x16 bf16 [B,H,W,Cin], dy16 bf16 [B,H,W,Cout] (per-sample factors already applied), dw fp32 [kh,kw,Cin,Cout] +=.
// SAME stride-1 convolutions (or 1x1) with (Cin, Cout) % (256, 256), (64, 256) or (64, 64) == 0, else SG_ERR_UNSUPPORTED
// (caller: sg_conv2d_bwd_weight).
extern "C" int sg_conv2d_bwd_weight_bf16v2(const void* x16, const void* dy16, float* dw, int B, int H, int W, int Cin, int Cout,
                                           int kh, int kw, int pad_same, int flags, void* stream) {
  if (!x16 || !dy16 || !dw || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  if (!pad_same && (kh != 1 || kw != 1)) return SG_ERR_UNSUPPORTED;
  if ((Cin % 64) || (Cout % 64)) return SG_ERR_UNSUPPORTED;
  const long M = (long)B * H * W;
  if (M <= 0) return SG_OK;
  if (2L * M * Cin >= (1L << 40) || 2L * M * Cout >= (1L << 40)) return SG_ERR_UNSUPPORTED;
  SgWgrad2Args a{};
  a.p = (const u16*)x16; a.q = (const u16*)dy16; a.dw = dw;
  a.Bn = B; a.H = H; a.W = W; a.Cp = Cin; a.Cq = Cout; a.ntaps = kh * kw; a.flags = flags; a.p_sy = 1; a.p_sx = 1;
  const int ph = kh / 2, pw = kw / 2;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  hipStream_t s = (hipStream_t)stream;
  if (Cin % 256 == 0 && Cout % 256 == 0) return sg2_launch_wgrad<256, 256>(a, M, s);
  if (Cout % 256 == 0) return sg2_launch_wgrad<64, 256>(a, M, s);
  if (Cin == 64 && Cout == 64 && !sg_deterministic()) return sg2_launch_wgrad<64, 64>(a, M, s);   // (two k-groups add into one address)
  return SG_ERR_UNSUPPORTED;
}

// Weight gradient of a transposed convolution (resnet_ops.py:57,69): dw [kh,kw,Cout,Cin] += sum over the INPUT pixels (b,i,j)
// of dy[b, sh i + ky - pad, sw j + kx - pad, co] * x[b,i,j,ci] -- the stride-1 kernel with the roles swapped (P = dy on its
// own, finer grid, read with a stride; Q = x) and the taps of sg_conv2d_transpose_bwd_weight.
// x16 bf16 [B,H,W,Cin], dy16 bf16 [B,sh H,sw W,Cout]; (Cout, Cin) % (256, 256), (64, 256) or (64, 64) == 0 and H W >= 64, else
// SG_ERR_UNSUPPORTED.
extern "C" int sg_conv2d_transpose_bwd_weight_bf16v2(const void* x16, const void* dy16, float* dw, int B, int H, int W, int Cin, int Cout,
                                                     int kh, int kw, int sh, int sw, void* stream) {
  if (!x16 || !dy16 || !dw || kh * kw > SG_MAX_TAPS || sh < 1 || sw < 1) return SG_ERR_ARG;
  if ((Cin % 64) || (Cout % 64) || H * W < 64) return SG_ERR_UNSUPPORTED;
  const long M = (long)B * H * W;
  if (M <= 0) return SG_OK;
  if ((long)B * sh * H * sw * W >= (1L << 31) || 2L * M * Cin >= (1L << 40)) return SG_ERR_UNSUPPORTED;
  SgWgrad2Args a{};
  a.p = (const u16*)dy16; a.q = (const u16*)x16; a.dw = dw;
  a.Bn = B; a.H = H; a.W = W; a.Cp = Cout; a.Cq = Cin; a.ntaps = kh * kw; a.flags = 0; a.p_sy = sh; a.p_sx = sw;
  const int pbh = sh == 1 ? kh / 2 : 0, pbw = sw == 1 ? kw / 2 : 0;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - pbh, kx - pbw, (ky * kw + kx) * Cin * Cout};
  hipStream_t s = (hipStream_t)stream;
  if (Cout % 256 == 0 && Cin % 256 == 0) return sg2_launch_wgrad<256, 256, true>(a, M, s);
  if (Cin % 256 == 0) return sg2_launch_wgrad<64, 256, true>(a, M, s);
  if (sg_deterministic()) return SG_ERR_UNSUPPORTED;
  return sg2_launch_wgrad<64, 64, true>(a, M, s);
}

// ==========================================================================================================
// fp8 (OCP e4m3) operands for the forward / data-grad convolutions of the D-shaped trunks (BASELINE config c5, first
// slice): per-tensor scaling, fp32 accumulation.   operand8 = e4m3(clamp(value * 448 / amax, +-448)),  amax = max|tensor|
// (device scalar: no host sync), result = sums * amax_a * amax_w / 448^2.  Weight gradients stay bf16 (gradient tensors
// need per-tensor scales of their own AND e5m2 range; not in this slice).
__global__ __launch_bounds__(256) void k_amax(const float* __restrict__ x, long n4, unsigned* amax_bits) {
  float m = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[e];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  m = sg_wave_max(m);
  if ((threadIdx.x & 63) == 0) sg_atomic_max_nonneg(reinterpret_cast<float*>(amax_bits), m);
}

// amax[0] = max(amax[0], max_i |x_i|); the caller zeroes amax first.  n % 4 == 0.
extern "C" int sg_amax_f32(const float* x, long n, float* amax, void* stream) {
  if (!x || !amax || n < 0 || (n & 3)) return SG_ERR_ARG;
  if (n == 0) return SG_OK;
  SG_KERNEL(k_amax, dim3(sg_grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, n / 4, reinterpret_cast<unsigned*>(amax));
  return sg_launch_status();
}

__device__ __forceinline__ float sg8_scale(const float* amax) {
  const float a = amax[0];
  return a > 0.f ? 448.f / a : 1.f;
}

__device__ __forceinline__ unsigned sg8_pack4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -448.f), 448.f), fminf(fmaxf(b, -448.f), 448.f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -448.f), 448.f), fminf(fmaxf(d, -448.f), 448.f), w, true);
  return (unsigned)w;
}

__global__ __launch_bounds__(256) void k_cvt_fp8(const float* __restrict__ x, uint2* __restrict__ out, long n8, int relu, const float* amax) {
  const float s = sg8_scale(amax);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (long)gridDim.x * blockDim.x) {
    float4 v0 = reinterpret_cast<const float4*>(x)[2 * e], v1 = reinterpret_cast<const float4*>(x)[2 * e + 1];
    if (relu) {
      v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
      v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
    }
    out[e] = make_uint2(sg8_pack4(v0.x * s, v0.y * s, v0.z * s, v0.w * s), sg8_pack4(v1.x * s, v1.y * s, v1.z * s, v1.w * s));
  }
}

// x fp32 [n] -> out fp8 e4m3 [n] = e4m3(relu?(x) * 448 / amax[0]); n % 8 == 0
extern "C" int sg_cvt_fp8(const float* x, void* out, long n, int relu, const float* amax, void* stream) {
  if (!x || !out || !amax || n < 0 || (n & 7)) return SG_ERR_ARG;
  if (n == 0) return SG_OK;
  SG_KERNEL(k_cvt_fp8, dim3(sg_grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (uint2*)out, n / 8, relu, amax);
  return sg_launch_status();
}

// the same conversion from a bf16 tensor (an operand-only conv result): out = e4m3(relu?(float(x16)) * 448 / amax[0]); n % 8 == 0
__global__ __launch_bounds__(256) void k_cvt_fp8_bf16(const bf16x8* __restrict__ x, uint2* __restrict__ out, long n8, int relu, const float* amax) {
  const float s = sg8_scale(amax);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (long)gridDim.x * blockDim.x) {
    const bf16x8 h = x[e];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = (float)h[j];
      if (relu) v[j] = fmaxf(v[j], 0.f);
      v[j] *= s;
    }
    out[e] = make_uint2(sg8_pack4(v[0], v[1], v[2], v[3]), sg8_pack4(v[4], v[5], v[6], v[7]));
  }
}

extern "C" int sg_cvt_fp8_bf16(const void* x16, void* out, long n, int relu, const float* amax, void* stream) {
  if (!x16 || !out || !amax || n < 0 || (n & 7)) return SG_ERR_ARG;
  if (n == 0) return SG_OK;
  SG_KERNEL(k_cvt_fp8_bf16, dim3(sg_grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x16, (uint2*)out, n / 8, relu, amax);
  return sg_launch_status();
}

// filter packing: fp32 [taps][K][N] (transpose = 1) or [taps][N][K] (transpose = 0) -> fp8 [taps][N][K], scaled by 448 / amax
__global__ __launch_bounds__(256) void k_pack_filter_fp8(const float* w, unsigned char* out, const float* amax, int K, int N, int transpose) {
  __shared__ float tile[32][33];
  const float s = sg8_scale(amax);
  const int t = blockIdx.z;
  const float* wt = w + (size_t)t * K * N;
  unsigned char* ot = out + (size_t)t * K * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    if (transpose) {
      const int k = k0 + r, n = n0 + tx;
      tile[r][tx] = (k < K && n < N) ? wt[(size_t)k * N + n] : 0.f;       // tile[k][n]
    } else {
      const int n = n0 + r, k = k0 + tx;
      tile[tx][r] = (k < K && n < N) ? wt[(size_t)n * K + k] : 0.f;       // tile[k][n]
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + tx;
    if (n < N && k < K) ot[(size_t)n * K + k] = (unsigned char)(sg8_pack4(tile[tx][r] * s, 0.f, 0.f, 0.f) & 0xffu);
  }
}

extern "C" int sg_pack_filter_fp8(const float* w, void* out, const float* amax, int taps, int K, int N, int transpose, void* stream) {
  if (!w || !out || !amax || taps < 1 || K < 1 || N < 1) return SG_ERR_ARG;
  SG_KERNEL(k_pack_filter_fp8, dim3(sg_cdiv(N, 32), sg_cdiv(K, 32), taps), dim3(256), 0, (hipStream_t)stream, w, (unsigned char*)out,
                     amax, K, N, transpose);
  return sg_launch_status();
}

// Contracts of sg_conv2d_fwd_bf16v2 / sg_conv2d_bwd_data_bf16v2 with fp8 operands and their amax scalars.
// SG_ERR_UNSUPPORTED unless reduction channels % 128 == 0 and output channels % 256 == 0 (caller: the bf16 entry points).
extern "C" int sg_conv2d_fwd_fp8(const void* x8, const float* amax_x, const void* wp8, const float* amax_w, const float* bias,
                                 const float* bias2, float* y, void* y16, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                 int pad_same, int flags, float* amax_y, void* stream) {
  if (!x8 || !wp8 || !amax_x || !amax_w || (!y && !y16) || (!y && (flags & SG_ACCUM)) || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  if (flags & SG_RELU_IN) return SG_ERR_UNSUPPORTED;           // fold the ReLU into sg_cvt_fp8
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemm2Args a{};
  a.a = (const u16*)x8; a.w = (const u16*)wp8; a.out = y; a.out16 = (u16*)y16; a.bias = bias; a.bias2 = bias2;
  a.amax_a = amax_x; a.amax_w = amax_w; a.amax_out = amax_y;
  a.Bn = B; a.Ha = H; a.Wa = W; a.Ca = Cin; a.Hg = Ho; a.Wg = Wo; a.a_sy = 1; a.a_sx = 1;
  a.Ho = Ho; a.Wo = Wo; a.N = Cout; a.o_sy = 1; a.o_sx = 1; a.o_oy = 0; a.o_ox = 0;
  a.ntaps = kh * kw; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  long done = 0;
  const int rc = sg_launch_igemm_bf16v2(a, (hipStream_t)stream, &done, 1);
  return rc != SG_OK ? rc : finish_twin(a, done, (hipStream_t)stream);
}

extern "C" int sg_conv2d_bwd_data_fp8(const void* dy8, const float* amax_dy, const void* wp8, const float* amax_w, const float* mask,
                                      const void* mask16, float* dx, void* dx16, int B, int H, int W, int Cin, int Cout, int kh,
                                      int kw, int pad_same, int flags, float* amax_dx, const float* amax_rowscale, void* stream) {
  if (!dy8 || !wp8 || !amax_dy || !amax_w || !dx || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemm2Args a{};
  a.a = (const u16*)dy8; a.w = (const u16*)wp8; a.out = dx; a.out16 = (u16*)dx16; a.mask = mask; a.mask16 = (const u16*)mask16;
  a.amax_a = amax_dy; a.amax_w = amax_w; a.amax_out = amax_dx; a.amax_rowscale = amax_rowscale;
  a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * Cin * Cout};
  long done = 0;
  const int rc = sg_launch_igemm_bf16v2(a, (hipStream_t)stream, &done, 1);
  return rc != SG_OK ? rc : finish_twin(a, done, (hipStream_t)stream);
}
