// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel serves Conv2D forward, Conv2D data-grad, Conv2DTranspose forward (one launch per
// output parity class) and Conv2DTranspose data-grad: see the tap-list formulation in sg_conv.h.
//   A (im2col rows)  : gathered from NHWC global memory as float4 along channels, transposed into
//                      LDS as As[k][m] (row stride BM+1 -> conflict-free b32 writes and reads)
//   B (filter slice) : Bs[k][n]; either a straight float4 copy ([K,N] row-major weights) or the
//                      same transposing loader as A when the weights are stored [N][K]
//   MFMA             : 32x32x2 f32, lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]
//   pipeline         : double-buffered LDS (BK = 16 -> 33 KB), next k-tile's global loads / LDS stores
//                      interleaved under the MFMAs, one barrier per k-tile; 3-4 workgroups per CU
//                      (measured: staggering co-resident workgroups changes nothing; MFMA pipe 80 % busy)
//   epilogue         : bias(+bias2), ReLU-backward mask, accumulate, ReLU, straight from the
//                      accumulators (lanes 0-31 of a register write one 128-byte row segment)
#include "sg_conv.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SG_BK 32

__device__ __forceinline__ int sg_xcd_remap(int orig, int nwg) {
  // bijective XCD-aware remap: blocks b and b+8 share an XCD (speed only, never correctness)
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

#define SG_IDENT_OUT 32   // internal flag: output pixel == base-grid pixel (idx = m*N + n, no div/mod)
#define SG_PREZEROED 64   // internal flag: the caller zeroed the whole output: partial tiles may be added without a memset

template <int BM, int BN, int WM, int WN, bool B_NK, int BK, int OCC>
__global__ __launch_bounds__(WM* WN * 64, OCC) void sg_igemm_kernel(const SgIgemmArgs p) {
  constexpr int NT = WM * WN * 64;
  constexpr int KQ = BK / 4;                // float4 chunks per operand row in a k-tile
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int LDA = BM + 1;
  constexpr int LDB = B_NK ? BN + 1 : BN;
  constexpr int A_P = BM * KQ / NT;         // float4 loads of A per thread per k-tile
  constexpr int BKN_RPP = NT / (BN / 4);    // B rows covered per pass ([K,N] loader)
  constexpr int BKN_P = BK / BKN_RPP;
  constexpr int BNK_P = BN * KQ / NT;       // [N,K] loader passes
  constexpr int B_P = B_NK ? BNK_P : BKN_P;
  constexpr int KS = BK / 2;                // MFMA k-steps per k-tile
  static_assert(A_P >= 1 && B_P >= 1, "tile/thread mismatch");

  __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int M = p.Bn * p.Hg * p.Wg;
  const int HW = p.Hg * p.Wg;
  const int n_tiles = (p.N + BN - 1) / BN;
  // Work decomposition: the first `full_tiles` workgroups own one output tile each (whole reduction, plain stores).
  // The tiles of the last, partly filled round of resident workgroups -- or all tiles when the grid cannot fill the
  // chip at all (small per-GPU batches) -- are cut `tail_split` ways along the reduction instead, so that the tail
  // of the launch still occupies every CU: such a workgroup reduces k-tiles [kt_begin, kt_begin + KT) and adds its
  // partial tile to the (pre-zeroed or accumulated-into) output with float atomics; every fused epilogue term used
  // with it (bias, 0/1 mask, accumulate) is linear in the partial sums.  Tail units are dispatched last.
  int wg, split, nsplit;
  if ((int)blockIdx.x < p.full_tiles) {
    wg = sg_xcd_remap(blockIdx.x, p.full_tiles);
    split = 0;
    nsplit = 1;
  } else {
    const int tail_tiles = p.n_tiles_total - p.full_tiles;
    const int u = sg_xcd_remap(blockIdx.x - p.full_tiles, tail_tiles * p.tail_split);
    nsplit = p.tail_split;
    split = u / tail_tiles;            // units of one k-range are neighbours: they share operand rows in L2
    wg = p.full_tiles + (u - split * tail_tiles);
  }
  const int m0 = (wg / n_tiles) * BM;
  const int n0 = (wg % n_tiles) * BN;

  const int kchunks = (p.Ca + BK - 1) / BK;
  const int KT_all = p.ntaps * kchunks;
  const int kt_begin = (int)(((long)KT_all * split) / nsplit);
  const int KT = (int)(((long)KT_all * (split + 1)) / nsplit) - kt_begin;
  const bool relu_in = (p.flags & SG_RELU_IN) != 0;

  // ---- per-thread operand rows: everything that does not depend on the k-tile is hoisted, so the
  //      loop body is one add + one bit test per load and can hide between the MFMAs.
  //      Operands are read with raw buffer loads (byte offsets in 32 bits, hardware range check): a lane that
  //      must contribute zero (padding tap, row / channel past the edge) gets an out-of-range offset and the load
  //      returns 0.0 without touching memory -- no select after the load, so the loaded registers are first
  //      consumed by the LDS store three k-steps later and the VMEM latency hides under the MFMAs. ----
  constexpr unsigned OOB = 0xFFFFFFE0u;
  const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  auto bload = [](decltype(rsrc_a) r, unsigned voff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return *reinterpret_cast<const float4*>(&v);
  };
  const int kc = tid % KQ;
  const int row0 = tid / KQ;
  unsigned a_base[A_P];       // byte offsets
  unsigned a_mask[A_P];
#pragma unroll
  for (int i = 0; i < A_P; ++i) {
    const int m = m0 + row0 + i * (NT / KQ);
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
  unsigned b_off[B_P];        // byte offsets
  bool b_ok[B_P];
#pragma unroll
  for (int i = 0; i < B_P; ++i) {
    if (B_NK) {
      const int n = n0 + row0 + i * (NT / KQ);
      b_ok[i] = n < p.N;
      b_off[i] = 4u * (unsigned)(n * p.ldw + 4 * kc);
    } else {
      const int k = tid / (BN / 4) + i * BKN_RPP;
      const int n = n0 + 4 * (tid % (BN / 4));
      b_ok[i] = n < p.N;
      b_off[i] = 4u * (unsigned)(k * p.ldw + n);
    }
  }

  float4 a_reg[A_P];
  float4 b_reg[B_P];
  // Reduction order.  The k-tiles of one output tile are visited CHANNEL-CHUNK-major with the taps innermost: the nine
  // taps of a 3x3 filter then read the same 64-byte channel slice of neighbouring pixels in consecutive k-tiles, so the
  // im2col re-reads hit L1 / L2 instead of re-streaming the activation rows once per tap (measured on the 1024->1024
  // layer at bs 128: FETCH_SIZE 4.27 GB per launch tap-major, 3.08 GB channel-major; same speed, MFMA-bound either way).
  // Tap constants (byte offsets) sit in a small LDS table: the tap changes every k-tile, a scalar-memory load per tile
  // would share the LDS wait counter with the fragment reads, and LDS reads stay in order with them.
  __shared__ int tap_tab[2 * SG_MAX_TAPS];
  if (tid < p.ntaps) {
    tap_tab[2 * tid] = 4 * (p.taps[tid].dy * p.Wa + p.taps[tid].dx) * p.Ca;
    tap_tab[2 * tid + 1] = 4 * p.taps[tid].w_off;
  }
  __syncthreads();
  const int nt1 = p.ntaps > 0 ? p.ntaps : 1;
  int lt = kt_begin % nt1, lc0 = (kt_begin / nt1) * BK;          // (tap, channel offset) of the next k-tile to fetch
  int tap_off = 0, w_tap = 0;
  auto set_tap = [&]() {
    tap_off = tap_tab[2 * lt];
    w_tap = tap_tab[2 * lt + 1];
  };
  set_tap();
  int wk = 4 * (B_NK ? lc0 : lc0 * p.ldw);              // byte offset of the k-tile's first weight row ([K,N]) / column ([N,K])
  const int wk_step = 4 * (B_NK ? BK : BK * p.ldw);
  const float relu_floor = relu_in ? 0.f : -__builtin_inff();   // ReLU on the operand without a branch in the loop
  auto relu = [&](float v) {                                      // exactly one VALU op (no canonicalisation pass in front)
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(relu_floor));
    return r;
  };

  auto load_a = [&]() {
    const unsigned toff = (unsigned)(tap_off + 4 * lc0);
    const bool cok = lc0 + 4 * kc < p.Ca;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const bool ok = ((a_mask[i] >> lt) & 1u) && cok;
      a_reg[i] = bload(rsrc_a, ok ? a_base[i] + toff : OOB);
    }
  };
  auto load_b = [&]() {
    const unsigned woff = (unsigned)(w_tap + wk);
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      bool ok;
      if (B_NK) {
        ok = b_ok[i] && lc0 + 4 * kc < p.Ca;
      } else {
        const int k = tid / (BN / 4) + i * BKN_RPP;
        ok = b_ok[i] && lc0 + k < p.Ca;
      }
      b_reg[i] = bload(rsrc_w, ok ? b_off[i] + woff : OOB);
    }
    ++lt;                            // advance the fetch cursor: next tap, or the next channel chunk at tap 0 (scalar selects)
    const bool wrap = lt >= p.ntaps;
    lt = wrap ? 0 : lt;
    lc0 += wrap ? BK : 0;
    wk += wrap ? wk_step : 0;
    set_tap();
  };
  auto store_a = [&](int buf) {
    float* as = As + buf * BK * LDA;
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
      const int r = row0 + i * (NT / KQ);
      const float4 v = a_reg[i];
      as[(4 * kc + 0) * LDA + r] = relu(v.x);
      as[(4 * kc + 1) * LDA + r] = relu(v.y);
      as[(4 * kc + 2) * LDA + r] = relu(v.z);
      as[(4 * kc + 3) * LDA + r] = relu(v.w);
    }
  };
  auto store_b = [&](int buf) {
    float* bs = Bs + buf * BK * LDB;
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
      if (B_NK) {
        const int r = row0 + i * (NT / KQ);
        const float4 v = b_reg[i];
        bs[(4 * kc + 0) * LDB + r] = v.x;
        bs[(4 * kc + 1) * LDB + r] = v.y;
        bs[(4 * kc + 2) * LDB + r] = v.z;
        bs[(4 * kc + 3) * LDB + r] = v.w;
      } else {
        const int k = tid / (BN / 4) + i * BKN_RPP;
        const int nn = 4 * (tid % (BN / 4));
        *reinterpret_cast<float4*>(bs + k * LDB + nn) = b_reg[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_col = wm * (BM / WM) + (lane & 31);
  const int b_col = wn * (BN / WN) + (lane & 31);
  const int khalf = lane >> 5;

  if (KT > 0) {
    load_a();
    load_b();
    store_a(0);
    store_b(0);
  }
  __syncthreads();

  // Software pipeline over k-steps (one k-step = TM*TN MFMAs = 2 k): the LDS fragment reads of step s+1 are issued
  // before the MFMAs of step s -- across k-tile boundaries too: the workgroup barrier sits one step before the end
  // of a tile, so the last step already reads the first fragments of the next LDS buffer -- and sched_barriers keep
  // the compiler from sinking them back next to their use.  The next tile's global loads go out in steps 0/1, its
  // LDS stores in steps KS-5/KS-4, all in MFMA shadows (each 32x32x2 f32 MFMA leaves most of its 64 cycles of
  // vector issue free).  `more` is a compile-time tag: the last tile is peeled.
  static_assert(KS % 2 == 0 && KS >= 6, "fragment double buffer parity");
  float af[2][TM], bf[2][TN];
  {
    const float* as = As + khalf * LDA + a_col;
    const float* bs = Bs + khalf * LDB + b_col;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = bs[j * 32];
  }
  auto k_tile = [&](int buf, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    const float* as = As + buf * BK * LDA + khalf * LDA + a_col;
    const float* bs = Bs + buf * BK * LDB + khalf * LDB + b_col;
    const float* as_n = As + (buf ^ 1) * BK * LDA + khalf * LDA + a_col;
    const float* bs_n = Bs + (buf ^ 1) * BK * LDB + khalf * LDB + b_col;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < KS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nxt][i] = as[(kk + 1) * 2 * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nxt][j] = bs[(kk + 1) * 2 * LDB + j * 32];
      } else if constexpr (more) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nxt][i] = as_n[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nxt][j] = bs_n[j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
      if constexpr (more) {
        if (kk == 0) load_a();
        if (kk == 1) load_b();
        if (kk == KS - 5) store_a(buf ^ 1);
        if (kk == KS - 4) store_b(buf ^ 1);
        if (kk == KS - 2) __syncthreads();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int kt = 0; kt + 1 < KT; ++kt) k_tile(kt & 1, std::true_type{});
  if (KT > 0) k_tile((KT - 1) & 1, std::false_type{});

  // ---- epilogue ----
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

__global__ __launch_bounds__(256) void k_relu_inplace(float* x, long n4) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<float4*>(x)[e];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    reinterpret_cast<float4*>(x)[e] = v;
  }
}

static int g_split_override = -1;   // tuning/debug: -1 = heuristic, 1 = never split, n > 1 = force n splits
extern "C" void sg_debug_set_splitk(int n) { g_split_override = n; }
extern "C" void sg_debug_set_splitk_v2(int n);          // conv_bf16v2.hip: the second-generation kernels' own override
static int g_deterministic = 0;
bool sg_deterministic() { return g_deterministic != 0; }
extern "C" int sg_set_deterministic(int on) {
  const int prev = g_deterministic;
  g_deterministic = on ? 1 : 0;
  g_split_override = on ? 1 : -1;
  sg_debug_set_splitk_v2(on ? 1 : -1);
  return prev;
}

template <int BM, int BN, int WM, int WN, int BK = SG_BK, int OCC = 2>
static int launch_cfg(const SgIgemmArgs& a_in, bool b_nk, hipStream_t s) {
  SgIgemmArgs a = a_in;
  if (a.o_sy == 1 && a.o_sx == 1 && a.o_oy == 0 && a.o_ox == 0 && a.Ho == a.Hg && a.Wo == a.Wg) a.flags |= SG_IDENT_OUT;
  const long M = (long)a.Bn * a.Hg * a.Wg;
  const int n_tiles = sg_cdiv(a.N, BN);
  const int tiles = sg_cdiv(M, BM) * n_tiles;
  if (tiles <= 0) return SG_OK;
  static const int split_env0 = getenv("SG_IGEMM_SPLITK") ? atoi(getenv("SG_IGEMM_SPLITK")) : 0;
  const int split_env = g_split_override >= 0 ? g_split_override : split_env0;   // 1 = never split, n > 1 = cut every tile n ways
  const int KT_all = a.ntaps * sg_cdiv(a.Ca, BK);
  // partial sums need a linear epilogue and an output that is zero (or holds the accumulate operand) beforehand;
  // strided output placement (transposed convolution classes) is only split when it accumulates
  // A fused output ReLU is not linear in the partial sums.  Where the grid cannot fill the chip (recognizer convs at
  // small per-GPU batches: a few dozen tiles) the launch is split anyway and the ReLU runs as a separate in-place
  // sweep over the (small) result afterwards.
  bool relu_after = false;
  if ((a.flags & SG_RELU_OUT) && (a.flags & SG_IDENT_OUT) && !(a.flags & SG_ACCUM) && !(a.N & 3) && tiles < 2 * 256 && KT_all >= 32 && split_env != 1) {
    a.flags &= ~SG_RELU_OUT;
    relu_after = true;
  }
  const bool can_split = !(a.flags & SG_RELU_OUT) && (a.flags & (SG_ACCUM | SG_IDENT_OUT | SG_PREZEROED));
  // Balance model: workgroups are handed to the 256 CUs round-robin and share a CU's matrix pipes, so a launch of T
  // equal tiles takes ceil(T / 256) tile-times.  The tiles beyond the last multiple of 256 (all of them when T < 256)
  // are cut `sp` ways along the reduction, sp chosen to minimise ceil(tail * sp / 256) / sp, with a small charge per
  // cut for the atomic epilogue and the extra prologues and a charge when a CU would hold fewer than 3 workgroups.
  constexpr int CUS = 256;
  int full = tiles, nsplit = 1;
  if (can_split && split_env != 1 && KT_all >= 16) {
    const int rem = tiles % CUS;
    if (split_env > 1) {
      full = 0;
      nsplit = split_env < KT_all ? split_env : 1;
    } else if (rem > 0) {
      // (with fewer than 3 whole tiles per CU every tile is cut: a CU then still holds 3 workgroups at a time)
      const int full_c = tiles < 3 * CUS ? 0 : (tiles - rem) / n_tiles * n_tiles;   // the tail starts on an m-tile boundary: its rows are contiguous
      const int tail = tiles - full_c;
      auto cost = [&](int sp) {
        const int units = tail * sp;
        const int per_cu = (units + CUS - 1) / CUS;
        const int resident = full_c > 0 ? OCC : (per_cu < OCC ? per_cu : OCC);
        const double eff = resident >= 3 ? 1.0 : (resident == 2 ? 0.9 : 0.7);
        return (double)per_cu / sp * (1.0 + 0.01 * (sp - 1)) / eff;
      };
      int best = 1;
      double best_cost = cost(1);
      const int sp_max = KT_all / 8 < 16 ? KT_all / 8 : 16;
      for (int sp = 2; sp <= sp_max; ++sp) {
        const double c = cost(sp);
        if (c < 0.97 * best_cost) {
          best = sp;
          best_cost = c;
        }
      }
      if (best > 1) {
        full = full_c;
        nsplit = best;
      }
    } else {
      // Whole rounds of 256, k = T / 256 tiles per CU, OCC of them resident at a time: when k is not a multiple of OCC the
      // last k mod OCC tiles of every CU run under-occupied (10 tiles at 3 per CU: 3-3-3-1, the lone workgroup at ~70 %
      // of the matrix rate).  Cutting those last tiles OCC ways keeps OCC workgroups resident to the end.
      static const int occ_tail_env = getenv("SG_IGEMM_OCCTAIL") ? atoi(getenv("SG_IGEMM_OCCTAIL")) : 1;
      const int k = tiles / CUS, r = k % OCC;
      if (occ_tail_env && r != 0 && k >= OCC && KT_all >= 16 * OCC) {
        const int full_c = (tiles - r * CUS) / n_tiles * n_tiles;
        full = full_c;
        nsplit = OCC;
      }
    }
  }
  if (nsplit > 1 && !(a.flags & (SG_ACCUM | SG_PREZEROED))) {   // zero the rows the partial tiles add into (IDENT_OUT: one contiguous range)
    const size_t row0 = (size_t)(full / n_tiles) * BM;
    if (hipMemsetAsync(a.out + row0 * a.N, 0, sizeof(float) * ((size_t)M - row0) * a.N, s) != hipSuccess) return SG_ERR_LAUNCH;
  }
  a.full_tiles = full;
  a.tail_split = nsplit;
  a.n_tiles_total = tiles;
  const dim3 g3(full + (tiles - full) * nsplit);
  if (b_nk)
    SG_KERNEL((sg_igemm_kernel<BM, BN, WM, WN, true, BK, OCC>), g3, dim3(WM * WN * 64), 0, s, a);
  else
    SG_KERNEL((sg_igemm_kernel<BM, BN, WM, WN, false, BK, OCC>), g3, dim3(WM * WN * 64), 0, s, a);
  if (relu_after) {
    const long n4 = M * a.N / 4;         // N % 4 == 0 on this path ([K,N] filters) or the tensor is padded to float4 by NHWC C % 4
    SG_KERNEL(k_relu_inplace, dim3(sg_grid_for(n4, 256)), dim3(256), 0, s, a.out, n4);
  }
  return sg_launch_status();
}

int sg_launch_igemm(const SgIgemmArgs& a_in, bool b_nk, hipStream_t s) {
  SgIgemmArgs a = a_in;
  if ((a.Ca & 3) || (a.ldw & 3) || (!b_nk && (a.N & 3))) return SG_ERR_ARG;
  if (a.ntaps < 0 || a.ntaps > SG_MAX_TAPS) return SG_ERR_ARG;
  // operands are addressed with 32-bit BYTE offsets (buffer loads), the output with 32-bit element indices
  const long a_elems = (long)a.Bn * a.Ha * a.Wa * a.Ca;
  long w_elems = 0;
  for (int t = 0; t < a.ntaps; ++t) w_elems = a.taps[t].w_off > w_elems ? a.taps[t].w_off : w_elems;
  w_elems += (long)(b_nk ? a.N : a.Ca) * a.ldw;
  if (a_elems >= (1L << 30) - 8 || w_elems >= (1L << 30) - 8 || (long)a.Bn * a.Ho * a.Wo * a.N >= (1L << 31)) return SG_ERR_ARG;
  a.a_bytes = (unsigned)(4 * a_elems);
  a.w_bytes = (unsigned)(4 * w_elems);
  static const int tile_env = getenv("SG_IGEMM_TILE") ? atoi(getenv("SG_IGEMM_TILE")) : 0;   // tuning knob
  // (BK = 32 k-tiles at 2 workgroups per CU were measured earlier and dropped: 16 at 3-4 per CU is faster everywhere)
  if (a.N > 64) {
    // 128x128 tiles; 128x64 only where the reduction cannot be split (non-linear epilogue, strided placement) and the
    // wider tile would leave the 256 CUs badly balanced (few tiles per CU with a large fractional remainder)
    const long M = (long)a.Bn * a.Hg * a.Wg;
    const long t128 = (long)sg_cdiv(M, 128) * sg_cdiv(a.N, 128), t64 = (long)sg_cdiv(M, 128) * sg_cdiv(a.N, 64);
    const double e128 = (double)t128 / (((t128 + 255) / 256) * 256.0);
    const double e64 = 0.93 * (double)t64 / (((t64 + 255) / 256) * 256.0);
    const bool ident = a.o_sy == 1 && a.o_sx == 1 && a.o_oy == 0 && a.o_ox == 0 && a.Ho == a.Hg && a.Wo == a.Wg;
    const bool can_split = !(a.flags & SG_RELU_OUT) && ((a.flags & (SG_ACCUM | SG_PREZEROED)) || ident);
    const bool narrow = tile_env == 64 || (tile_env == 0 && !can_split && t128 < 2048 && e64 > e128);
    if (narrow) return launch_cfg<128, 64, 2, 2, 16, 4>(a, b_nk, s);
    static const int occ_env = getenv("SG_IGEMM_OCC") ? atoi(getenv("SG_IGEMM_OCC")) : 3;
    if (occ_env == 4) return launch_cfg<128, 128, 2, 2, 16, 4>(a, b_nk, s);
    return launch_cfg<128, 128, 2, 2, 16, 3>(a, b_nk, s);
  }
  if (a.N > 32) {
    // (measured and dropped: 256x64 tiles for N <= 64, 2-wave workgroups with 128x64 wave tiles, s_setprio around the
    //  MFMA clusters, staggered workgroup starts -- none beat this configuration)
    return launch_cfg<128, 64, 2, 2, 16, 4>(a, b_nk, s);
  }
  return launch_cfg<128, 32, 4, 1>(a, b_nk, s);
}

// ------------------------------------------------------------------------------------------
// thin convolutions (Cin == 1 or Cout == 1): HBM-bound direct kernels, no matrix cores.
// Round 2 (expand, thin weight-grad): the one-channel rows a row of output needs are staged in LDS, so nothing in the inner loop is
// a dependent, bounds-checked global load (round 1: 1.9 TB/s of output for the expansion, now 4.7-5.4).
// ------------------------------------------------------------------------------------------
struct SgThinWin { int dy0, nrows, dx0, span; };      // tap window: rows dy0 .. dy0 + nrows - 1, columns dx0 .. dx0 + span - Wg

static SgThinWin sg_thin_window(const SgThinArgs& a) {
  int dy0 = a.taps[0].dy, dy1 = dy0, dx0 = a.taps[0].dx, dx1 = dx0;
  for (int t = 1; t < a.ntaps; ++t) {
    dy0 = a.taps[t].dy < dy0 ? a.taps[t].dy : dy0; dy1 = a.taps[t].dy > dy1 ? a.taps[t].dy : dy1;
    dx0 = a.taps[t].dx < dx0 ? a.taps[t].dx : dx0; dx1 = a.taps[t].dx > dx1 ? a.taps[t].dx : dx1;
  }
  return SgThinWin{dy0, dy1 - dy0 + 1, dx0, a.Wg + dx1 - dx0};
}

// expand: out[m, c] = sum_t a1[pix(m)+tap_t] * w_t[c] + bias[c]      (a has ONE channel)
// One output row (b, y) per workgroup iteration: the nrows input rows it needs are staged in LDS (zero outside the image, ReLU
// applied), then thread (channel group, pixel lane) walks the row: per output float4 a handful of LDS broadcasts and FMAs and
// one 16-byte store -- the kernel is a write stream.
__global__ __launch_bounds__(256) void sg_thin_expand_kernel(const SgThinArgs p, const SgThinWin win) {
  extern __shared__ float rows[];       // [nrows][span]
  const int cq = p.C >> 2, lanes = 256 / cq;          // cq divides 256
  const int cl = threadIdx.x % cq, pl = threadIdx.x / cq, c = 4 * cl;
  float4 wv[SG_MAX_TAPS];
  int toff[SG_MAX_TAPS];
#pragma unroll
  for (int t = 0; t < SG_MAX_TAPS; ++t) {
    wv[t] = t < p.ntaps ? *reinterpret_cast<const float4*>(p.w + p.taps[t].w_off + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    toff[t] = t < p.ntaps ? (p.taps[t].dy - win.dy0) * win.span + p.taps[t].dx - win.dx0 : 0;
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + c);
  const bool relu_in = p.flags & SG_RELU_IN, accum = p.flags & SG_ACCUM, relu_out = p.flags & SG_RELU_OUT;
  const int total_rows = p.Bn * p.Hg, stage_n = win.nrows * win.span;
  for (int row = blockIdx.x; row < total_rows; row += gridDim.x) {
    const int b = row / p.Hg, yg = row - b * p.Hg;
    __syncthreads();
    for (int e = threadIdx.x; e < stage_n; e += 256) {
      const int r = e / win.span, xx = e - r * win.span;
      const int iy = yg + win.dy0 + r, ix = xx + win.dx0;
      float v = 0.f;
      if (iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) v = p.a[((size_t)b * p.Ha + iy) * p.Wa + ix];
      rows[e] = relu_in ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
    for (int x = pl; x < p.Wg; x += lanes) {
      float4 o = bv;
#pragma unroll
      for (int t = 0; t < SG_MAX_TAPS; ++t) {
        if (t < p.ntaps) {
          const float a = rows[toff[t] + x];
          o.x += a * wv[t].x; o.y += a * wv[t].y; o.z += a * wv[t].z; o.w += a * wv[t].w;
        }
      }
      const size_t m = (size_t)row * p.Wg + x;
      float* op = p.out + m * p.C + c;
      if (p.mask) {
        const float4 mk = *reinterpret_cast<const float4*>(p.mask + m * p.C + c);
        if (mk.x <= 0.f) o.x = 0.f;
        if (mk.y <= 0.f) o.y = 0.f;
        if (mk.z <= 0.f) o.z = 0.f;
        if (mk.w <= 0.f) o.w = 0.f;
      }
      if (accum) {
        const float4 pv = *reinterpret_cast<const float4*>(op);
        o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w;
      }
      if (relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *reinterpret_cast<float4*>(op) = o;
    }
  }
}

// (Round 2 measured two LDS-staged scatter forms that read the wide tensor once -- 16 lanes per pixel with a butterfly sum per
//  tap, and lane = pixel over a transposed LDS tile -- at 0.54 and 0.87 TB/s against this kernel's 1.2: the barriers and the
//  short per-row phases cost more than the 9x re-reads, which come from L2.  Kept as it was.)
// contract: out[m] = sum_t sum_c aC[pix(m)+tap_t, c] * w_t[c] + bias[0]   (out has ONE channel)
// 16 lanes per pixel, float4 of channels per lane, 4-step shuffle reduction.
__global__ __launch_bounds__(256) void sg_thin_contract_kernel(const SgThinArgs p) {
  const long M = (long)p.Bn * p.Hg * p.Wg;
  const int HW = p.Hg * p.Wg;
  const int sub = threadIdx.x & 15;
  const long pix0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long pstride = ((long)gridDim.x * blockDim.x) >> 4;
  const bool relu_in = p.flags & SG_RELU_IN, accum = p.flags & SG_ACCUM, tanh_out = p.flags & SG_TANH_OUT;
  const float bias = p.bias ? p.bias[0] : 0.f;
  const long Mpad = (M + 3) / 4 * 4;  // keep all 64 lanes in the shuffles
  // pixel cursor advanced by pstride pixels per iteration (decoded once; a padding lane past M keeps computing on a
  // clamped pixel 0 as before)
  int cb = (int)(pix0 / HW), cy_, cx_;
  {
    const int rem = (int)(pix0 - (long)cb * HW);
    cy_ = rem / p.Wg;
    cx_ = rem - cy_ * p.Wg;
  }
  const int adv_b = (int)(pstride / HW), adv_r = (int)(pstride - (long)adv_b * HW), adv_y = adv_r / p.Wg, adv_x = adv_r - adv_y * p.Wg;
  for (long m = pix0; m < Mpad; m += pstride) {
    const bool live = m < M;
    const int b = live ? cb : 0, yg = live ? cy_ : 0, xg = live ? cx_ : 0;
    cx_ += adv_x;
    const int wx = cx_ >= p.Wg ? 1 : 0;
    cx_ -= wx * p.Wg;
    cy_ += adv_y + wx;
    const int wy = cy_ >= p.Hg ? 1 : 0;
    cy_ -= wy * p.Hg;
    cb += adv_b + wy;
    float s = 0.f;
    for (int t = 0; t < p.ntaps; ++t) {
      const int iy = yg + p.taps[t].dy, ix = xg + p.taps[t].dx;
      if (!(iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa)) continue;
      const float* ap = p.a + (((size_t)b * p.Ha + iy) * p.Wa + ix) * p.C;
      const float* wp = p.w + p.taps[t].w_off;
      for (int c = 4 * sub; c < p.C; c += 64) {
        float4 a = *reinterpret_cast<const float4*>(ap + c);
        const float4 w = *reinterpret_cast<const float4*>(wp + c);
        if (relu_in) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        s += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (live && sub == 0) {
      float v = s + bias;
      if (p.mask && p.mask[m] <= 0.f) v = 0.f;
      if (accum) v += p.out[m];
      if (tanh_out) v = tanhf(v);
      p.out[m] = v;
    }
  }
}

static int launch_thin(const SgThinArgs& a, bool expand, hipStream_t s) {
  const long M = (long)a.Bn * a.Hg * a.Wg;
  if (M <= 0) return SG_OK;
  if (a.C & 3) return SG_ERR_ARG;
  const SgThinWin win = sg_thin_window(a);
  if (expand) {
    if (256 % (a.C >> 2)) return SG_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * (size_t)win.nrows * win.span;
    if (lds > 60 * 1024) return SG_ERR_UNSUPPORTED;
    const long rows = (long)a.Bn * a.Hg;
    SG_KERNEL(sg_thin_expand_kernel, dim3((unsigned)(rows < 4096 ? rows : 4096)), dim3(256), lds, s, a, win);
  } else {
    SG_KERNEL(sg_thin_contract_kernel, dim3(sg_grid_for(M * 16, 256)), dim3(256), 0, s, a);
  }
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
// C-ABI entry points (include/scrabble_hip.h)
// ------------------------------------------------------------------------------------------
static inline int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

extern "C" int sg_conv2d_fwd(const float* x, const float* w, const float* bias, const float* bias2,
                             float* y, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                             int pad_same, int flags, void* stream) {
  if (!x || !w || !y || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1 || Cout == 1) {
    SgThinArgs a{};
    a.a = x; a.w = w; a.out = y; a.bias = bias; a.mask = nullptr;
    a.Bn = B; a.Ha = H; a.Wa = W; a.Hg = Ho; a.Wg = Wo; a.flags = flags; a.ntaps = kh * kw;
    if (bias2) return SG_ERR_UNSUPPORTED;
    a.C = (Cin == 1) ? Cout : Cin;
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * a.C};
    return launch_thin(a, Cin == 1, s);
  }
  SgIgemmArgs a{};
  a.a = x; a.w = w; a.out = y; a.bias = bias; a.bias2 = bias2; a.mask = nullptr;
  a.Bn = B; a.Ha = H; a.Wa = W; a.Ca = Cin; a.Hg = Ho; a.Wg = Wo; a.a_sy = 1; a.a_sx = 1;
  a.Ho = Ho; a.Wo = Wo; a.N = Cout; a.o_sy = 1; a.o_sx = 1; a.o_oy = 0; a.o_ox = 0;
  a.ntaps = kh * kw; a.ldw = Cout; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx)
      a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm(a, false, s);
}

// dx[b,i,j,ci] = sum_{ky,kx,co} dy[b, i+ph-ky, j+pw-kx, co] * w[ky,kx,ci,co]; optional ReLU mask / accumulate
extern "C" int sg_conv2d_bwd_data(const float* dy, const float* w, const float* mask, float* dx,
                                  int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                  int pad_same, int flags, void* stream) {
  if (!dy || !w || !dx || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1 || Cout == 1) {
    SgThinArgs a{};
    a.a = dy; a.w = w; a.out = dx; a.bias = nullptr; a.mask = mask;
    a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Hg = H; a.Wg = W; a.flags = flags; a.ntaps = kh * kw;
    a.C = (Cin == 1) ? Cout : Cin;
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * a.C};
    return launch_thin(a, /*expand=*/Cout == 1, s);
  }
  SgIgemmArgs a{};
  a.a = dy; a.w = w; a.out = dx; a.mask = mask;
  a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.ldw = Cout; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx)
      a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm(a, /*b_nk=*/true, s);
}

// fp32 filter transpose per tap: w [taps][K][N] -> out [taps][N][K]
__global__ __launch_bounds__(256) void k_transpose_filter(const float* w, float* out, int K, int N) {
  __shared__ float tile[32][33];
  const float* wt = w + (size_t)blockIdx.z * K * N;
  float* ot = out + (size_t)blockIdx.z * K * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
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
    if (n < N && k < K) ot[(size_t)n * K + k] = tile[tx][r];
  }
}

extern "C" int sg_transpose_filter(const float* w, float* out, int taps, int K, int N, void* stream) {
  if (!w || !out || taps < 1 || K < 1 || N < 1) return SG_ERR_ARG;
  SG_KERNEL(k_transpose_filter, dim3(sg_cdiv(N, 32), sg_cdiv(K, 32), taps), dim3(256), 0, (hipStream_t)stream, w, out, K, N);
  return sg_launch_status();
}

// Data-grad with a pre-transposed filter copy wt [kh,kw,Cout,Cin] (sg_transpose_filter(w, taps, K = Cin, N = Cout)):
// the reduction index (Cout) is then the row index of each tap matrix, so the launch takes the straight [K,N] filter
// loader of the forward pass instead of the transposing one (~5 % faster); same contract as sg_conv2d_bwd_data.
extern "C" int sg_conv2d_bwd_data_wt(const float* dy, const float* wt, const float* mask, float* dx, int B, int H, int W,
                                     int Cin, int Cout, int kh, int kw, int pad_same, int flags, void* stream) {
  if (!dy || !wt || !dx || kh * kw > SG_MAX_TAPS || Cin == 1 || Cout == 1) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  SgIgemmArgs a{};
  a.a = dy; a.w = wt; a.out = dx; a.mask = mask;
  a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.ldw = Cin; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm(a, /*b_nk=*/false, (hipStream_t)stream);
}

// Conv2DTranspose(padding='same'): x [B,H,W,Cin] -> y [B,sh*H,sw*W,Cout], w [kh,kw,Cout,Cin].
// One launch per output parity class; a class without taps still writes its bias.
extern "C" int sg_conv2d_transpose_fwd(const float* x, const float* w, const float* bias,
                                       const float* bias2, float* y, int B, int H, int W, int Cin,
                                       int Cout, int kh, int kw, int sh, int sw, int flags,
                                       void* stream) {
  if (!x || !w || !y || kh * kw > SG_MAX_TAPS || sh < 1 || sw < 1) return SG_ERR_ARG;
  const int pbh = (sh == 1) ? kh / 2 : 0, pbw = (sw == 1) ? kw / 2 : 0;
  // Small per-GPU batches leave each parity-class launch with a few dozen tiles.  The classes write disjoint pixels of
  // y, so y is zeroed ONCE here and every class may then cut its tiles along the reduction (atomic partial tiles).
  const long M1 = (long)B * H * W;
  if (!(flags & SG_ACCUM) && sh * sw > 1 && sg_cdiv(M1, 128) * sg_cdiv(Cout, 128) < 256) {
    if (hipMemsetAsync(y, 0, sizeof(float) * (size_t)M1 * sh * sw * Cout, (hipStream_t)stream) != hipSuccess) return SG_ERR_LAUNCH;
    flags |= SG_PREZEROED;
  }
  for (int py = 0; py < sh; ++py)
    for (int px = 0; px < sw; ++px) {
      SgIgemmArgs a{};
      a.a = x; a.w = w; a.out = y; a.bias = bias; a.bias2 = bias2;
      a.Bn = B; a.Ha = H; a.Wa = W; a.Ca = Cin; a.Hg = H; a.Wg = W; a.a_sy = 1; a.a_sx = 1;
      a.Ho = sh * H; a.Wo = sw * W; a.N = Cout; a.o_sy = sh; a.o_sx = sw; a.o_oy = py; a.o_ox = px;
      a.ldw = Cin; a.flags = flags; a.ntaps = 0;
      for (int ky = 0; ky < kh; ++ky) {
        if ((py + pbh - ky) % sh) continue;
        for (int kx = 0; kx < kw; ++kx) {
          if ((px + pbw - kx) % sw) continue;
          a.taps[a.ntaps++] =
              SgTap{floordiv(py + pbh - ky, sh), floordiv(px + pbw - kx, sw), (ky * kw + kx) * Cin * Cout};
        }
      }
      const int rc = sg_launch_igemm(a, /*b_nk=*/true, (hipStream_t)stream);
      if (rc != SG_OK) return rc;
    }
  return SG_OK;
}

// dx[b,i,j,ci] = sum_{ky,kx,co} dy[b, sh*i+ky-pbh, sw*j+kx-pbw, co] * w[ky,kx,co,ci]
extern "C" int sg_conv2d_transpose_bwd_data(const float* dy, const float* w, const float* mask,
                                            float* dx, int B, int H, int W, int Cin, int Cout,
                                            int kh, int kw, int sh, int sw, int flags, void* stream) {
  if (!dy || !w || !dx || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int pbh = (sh == 1) ? kh / 2 : 0, pbw = (sw == 1) ? kw / 2 : 0;
  SgIgemmArgs a{};
  a.a = dy; a.w = w; a.out = dx; a.mask = mask;
  a.Bn = B; a.Ha = sh * H; a.Wa = sw * W; a.Ca = Cout; a.Hg = H; a.Wg = W; a.a_sy = sh; a.a_sx = sw;
  a.Ho = H; a.Wo = W; a.N = Cin; a.o_sy = 1; a.o_sx = 1;
  a.ntaps = kh * kw; a.ldw = Cin; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx)
      a.taps[ky * kw + kx] = SgTap{ky - pbh, kx - pbw, (ky * kw + kx) * Cin * Cout};
  return sg_launch_igemm(a, /*b_nk=*/false, (hipStream_t)stream);
}
