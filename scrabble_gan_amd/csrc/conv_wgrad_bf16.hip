// bf16 variant of the weight-gradient contraction (config c3): dW_t[c][n] += sum_m s_b(m) * P_t[m][c] * Q[m][n] on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 atomics into dW.
//
// The reduction index is the pixel m, which is the SLOW index of both NHWC operands, while the MFMA wants 8
// consecutive reduction elements per lane.  So both operand tiles are transposed on their way into LDS: a thread
// loads an 8-pixel x 4-channel block (eight 16-byte rows), applies the per-sample factor, rounds to bf16, applies
// ReLU and writes four 16-byte column pieces Xs[c][m .. m+7] (row stride 40 elements: conflict-free fragment reads).
// Stride-1 convolutions only (both operands on the base grid: the IDENT form of sg_wgrad_kernel); the launcher
// falls back to the fp32 kernel for everything else.  Pipeline as in sg_igemm_bf16_kernel: global loads one k-tile
// (32 pixels) ahead in registers, LDS double buffer, fragment reads one k-step ahead.
#include "sg_conv.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// QSCALE: 0 = none, 1 = per-sample factor looked up per 4-pixel group, 2 = one factor per k-tile (Hg*Wg % 32 == 0: the 32
// pixels of a k-tile lie in one sample)
template <int QSCALE>
__global__ __launch_bounds__(256, 3) void sg_wgrad_bf16_kernel(const SgWgradArgs p) {
  constexpr int BC = 128, BN = 128, WM = 2, WN = 2, BK = 32, LDK = 40;
  constexpr int TM = BC / WM / 32, TN = BN / WN / 32;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BC * LDK + 2 * BN * LDK];
  unsigned short* Ps = smem;
  unsigned short* Qs = smem + 2 * BC * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int c_tiles = (p.Cp + BC - 1) / BC, n_tiles = (p.Cq + BN - 1) / BN;
  const int combos = p.ntaps * c_tiles * n_tiles;
  const int combo = blockIdx.x % combos;
  const int chunk = blockIdx.x / combos;
  const int t = combo / (c_tiles * n_tiles);
  const int c0 = ((combo / n_tiles) % c_tiles) * BC;
  const int n0 = (combo % n_tiles) * BN;
  const int M = p.Bn * p.Hg * p.Wg;
  const int HW = p.Hg * p.Wg;
  const int m_begin = chunk * p.mchunk;
  const int m_end = min(M, m_begin + p.mchunk);
  const int KT = (m_end - m_begin + BK - 1) / BK;
  const int dy = p.taps[t].dy, dx = p.taps[t].dx;
  const bool relu_in = (p.flags & SG_RELU_IN) != 0;

  constexpr unsigned OOB = 0xFFFFFFE0u;
  const auto rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.p), 0, (int)p.p_bytes, 0x00020000);
  const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.q), 0, (int)p.q_bytes, 0x00020000);
  const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(QSCALE ? p.qscale : p.q), 0, QSCALE ? 4 * p.Bn : 0, 0x00020000);
  auto bload = [](decltype(rsrc_p) r, unsigned voff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return *reinterpret_cast<const float4*>(&v);
  };

  // Operand roles are split over the waves: waves 0-1 stage P, waves 2-3 stage Q (wave-uniform, so descriptors and
  // constants are scalar selects).  A thread owns an 8-pixel x 4-channel block of its operand's k-tile: eight 16-byte
  // row loads, then FOUR 16-byte LDS writes Xs[c][m .. m+7] (the LDS pipe, not the matrix cores or VALU, bounds this
  // kernel: 8-byte column pieces cost twice the write slots).  The 8 pixels are two 4-pixel groups; Wg % 4 == 0
  // (checked by the launcher) keeps each group inside one image row of one sample, so a group needs one (b, y, x) cursor.
  const bool isP = __builtin_amdgcn_readfirstlane(tid) < 128;
  const int u = tid & 127;
  const int mq = u & 3, cq = u >> 2;                 // 4 pixel groups of 8 x 32 channel quads
  const int oc = (isP ? c0 : n0) + 4 * cq;           // first channel of the block
  const int oC = isP ? p.Cp : p.Cq;
  const bool cok = oc < oC;
  const int o_dy = isP ? dy : 0, o_dx = isP ? dx : 0;  // Q sits on the base grid: its validity test is always true
  const auto rsrc_o = isP ? rsrc_p : rsrc_q;
  unsigned short* const Xs = isP ? Ps : Qs;
  int gb[2], gy[2], gx[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int m = m_begin + 8 * mq + 4 * g;
    gb[g] = m / HW;
    const int rem = m - gb[g] * HW;
    gy[g] = rem / p.Wg;
    gx[g] = rem - gy[g] * p.Wg;
  }
  const int adv_b = BK / HW, adv_r = BK - adv_b * HW, adv_y = adv_r / p.Wg, adv_x = adv_r - adv_y * p.Wg;
  auto advance = [&](int& bb, int& yy, int& xx) {
    xx += adv_x;
    const int cx = xx >= p.Wg ? 1 : 0;
    xx -= cx * p.Wg;
    yy += adv_y + cx;
    const int cy = yy >= p.Hg ? 1 : 0;
    yy -= cy * p.Hg;
    bb += adv_b + cy;
  };
  unsigned o_lin = 4u * (unsigned)((m_begin + 8 * mq + o_dy * p.Wp + o_dx) * oC + oc);
  const unsigned o_row = 4u * oC, o_step = 4u * BK * oC;
  int m_next = m_begin;
  unsigned vzero = 0;
  asm volatile("" : "+v"(vzero));
  int qs_b = m_begin / HW, qs_rem = m_begin - (m_begin / HW) * HW;
  float q_sc[2] = {1.f, 1.f};

  float4 o_reg[8];
  auto load_tile = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const bool rowok = cok & ((unsigned)(gy[g] + o_dy) < (unsigned)p.Hp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool live = m_next + 8 * mq + 4 * g + r < m_end;
        const bool ok = live & rowok & ((unsigned)(gx[g] + r + o_dx) < (unsigned)p.Wp);
        o_reg[4 * g + r] = bload(rsrc_o, ok ? o_lin + (4 * g + r) * o_row : OOB);
      }
      if (QSCALE == 1)       // the group's sample
        q_sc[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            rsrc_s, m_next + 8 * mq + 4 * g < m_end ? (int)(4u * (unsigned)gb[g]) : (int)OOB, 0, 0));
      advance(gb[g], gy[g], gx[g]);
    }
    o_lin += o_step;
    if (QSCALE == 2) {
      q_sc[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, (int)(vzero + 4u * (unsigned)qs_b), 0, 0));
      q_sc[1] = q_sc[0];
      qs_rem += BK;
      if (qs_rem >= HW) { qs_rem -= HW; ++qs_b; }
    }
    m_next += BK;
  };
  const bool do_bias = p.dbias != nullptr && t == 0 && c0 == 0 && !isP;     // the Q-staging waves of one (tap, c-tile) column sum dy
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  // ReLU on the ROUNDED operand, two elements per instruction: a bf16 is negative iff it is negative as an int16, so
  // max(x, 0) is a packed signed 16-bit max (floor 0x8000 = most negative int16 switches it off; Q never takes it)
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const short rfloor = (relu_in && isP) ? (short)0 : (short)0x8000;
  const s16x8 rfloor8 = {rfloor, rfloor, rfloor, rfloor, rfloor, rfloor, rfloor, rfloor};
  auto store_tile = [&](int buf) {
    unsigned short* xs = Xs + buf * BC * LDK + (4 * cq) * LDK + 8 * mq;
    float4 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      v[r] = o_reg[r];
      if (QSCALE) {                      // packed fp32 multiplies; the P waves multiply by 1
        const float scr = isP ? 1.f : q_sc[r >> 2];
        f32x2 lo = {v[r].x, v[r].y}, hi = {v[r].z, v[r].w};
        const f32x2 sc = {scr, scr};
        lo *= sc; hi *= sc;
        v[r] = make_float4(lo.x, lo.y, hi.x, hi.y);
      }
    }
    if (do_bias) {                       // wave-uniform: fp32 column sums of Q (bias gradient)
#pragma unroll
      for (int r = 0; r < 8; ++r) { bsum.x += v[r].x; bsum.y += v[r].y; bsum.z += v[r].z; bsum.w += v[r].w; }
    }
    auto col = [&](float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
      bf16x8 h;
      h[0] = (__bf16)a0; h[1] = (__bf16)a1; h[2] = (__bf16)a2; h[3] = (__bf16)a3;
      h[4] = (__bf16)a4; h[5] = (__bf16)a5; h[6] = (__bf16)a6; h[7] = (__bf16)a7;
      return __builtin_elementwise_max(__builtin_bit_cast(s16x8, h), rfloor8);
    };
    *reinterpret_cast<s16x8*>(xs + 0 * LDK) = col(v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x);
    *reinterpret_cast<s16x8*>(xs + 1 * LDK) = col(v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y);
    *reinterpret_cast<s16x8*>(xs + 2 * LDK) = col(v[0].z, v[1].z, v[2].z, v[3].z, v[4].z, v[5].z, v[6].z, v[7].z);
    *reinterpret_cast<s16x8*>(xs + 3 * LDK) = col(v[0].w, v[1].w, v[2].w, v[3].w, v[4].w, v[5].w, v[6].w, v[7].w);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int khalf = lane >> 5;
  const int a_row = wm * (BC / WM) + (lane & 31);
  const int b_row = wn * (BN / WN) + (lane & 31);

  if (KT > 0) {
    load_tile();
    store_tile(0);
    if (KT > 1) load_tile();
  }
  __syncthreads();

  bf16x8 af[2][TM], bf[2][TN];
  auto read_frags = [&](int buf, int step, int slot) {
    const unsigned short* as = Ps + buf * BC * LDK + a_row * LDK + 16 * step + 8 * khalf;
    const unsigned short* bs = Qs + buf * BN * LDK + b_row * LDK + 16 * step + 8 * khalf;
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

  if (p.dbias != nullptr && t == 0 && c0 == 0) {      // (block-uniform) reduce the Q waves' column sums over the 4 pixel lanes
    float4* red = reinterpret_cast<float4*>(smem);
    __syncthreads();
    red[tid] = bsum;
    __syncthreads();
    if (!isP && mq == 0) {
      float4 s4 = red[tid];
      for (int k = 1; k < 4; ++k) {
        const float4 o = red[tid + k];
        s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
      }
      if (oc < p.Cq) {
        float* d = p.dbias + oc;
        atomicAdd(d + 0, s4.x); atomicAdd(d + 1, s4.y); atomicAdd(d + 2, s4.z); atomicAdd(d + 3, s4.w);
      }
    }
  }

  float* dwt = p.dw + p.taps[t].w_off;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
    if (n >= p.Cq) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wm * (BC / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (c < p.Cp) atomicAdd(dwt + (size_t)c * p.Cq + n, acc[i][j][r]);
      }
  }
}

// Launches the bf16 kernel when the shape qualifies; SG_ERR_UNSUPPORTED otherwise (caller falls back to fp32).
int sg_launch_wgrad_bf16(const SgWgradArgs& a_in, hipStream_t s) {
  SgWgradArgs a = a_in;
  const bool ident = a.p_sy == 1 && a.p_sx == 1 && a.q_sy == 1 && a.q_sx == 1 && a.Hp == a.Hg && a.Wp == a.Wg && a.Hq == a.Hg && a.Wq == a.Wg;
  const long HW = (long)a.Hg * a.Wg;
  if (!ident || a.Cp < 64 || a.Cq < 64 || (a.Cp & 3) || (a.Cq & 3) || (a.Wg & 3) || a.ntaps < 1 || a.ntaps > SG_MAX_TAPS)
    return SG_ERR_UNSUPPORTED;
  const long M = (long)a.Bn * HW;
  const long p_elems = (long)a.Bn * a.Hp * a.Wp * a.Cp, q_elems = (long)a.Bn * a.Hq * a.Wq * a.Cq;
  if (p_elems >= (1L << 30) - 8 || q_elems >= (1L << 30) - 8) return SG_ERR_ARG;
  a.p_bytes = (unsigned)(4 * p_elems);
  a.q_bytes = (unsigned)(4 * q_elems);
  const int combos = a.ntaps * sg_cdiv(a.Cp, 128) * sg_cdiv(a.Cq, 128);
  const long max_chunks = (M + 511) / 512;               // at least 16 k-tiles per workgroup
  long nchunks = 1;
  double best = 1e30;
  for (long c = (1536 + combos - 1) / combos; c <= (3072 + combos - 1) / combos; ++c) {
    const long cc = c < 1 ? 1 : (c > max_chunks ? max_chunks : c);
    const long W = combos * cc;
    const double loss = (double)((W + 255) / 256) * 256.0 / (double)W * (1.0 + 0.004 * cc);
    if (loss < best) { best = loss; nchunks = cc; }
  }
  if (sg_deterministic()) nchunks = 1;
  long mchunk = (M + nchunks - 1) / nchunks;
  mchunk = (mchunk + 31) / 32 * 32;
  nchunks = (M + mchunk - 1) / mchunk;
  a.mchunk = (int)mchunk;
  const dim3 grid((unsigned)(combos * nchunks)), block(256);
  if (a.qscale && HW % 32 == 0) SG_KERNEL((sg_wgrad_bf16_kernel<2>), grid, block, 0, s, a);
  else if (a.qscale) SG_KERNEL((sg_wgrad_bf16_kernel<1>), grid, block, 0, s, a);
  else SG_KERNEL((sg_wgrad_bf16_kernel<0>), grid, block, 0, s, a);
  return sg_launch_status();
}
