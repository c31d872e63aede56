// bf16 variant of the weight-gradient contraction (config c3): dW_t[c][n] += sum_m s_b(m) * P_t[m][c] * Q[m][n] on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 atomics into dW.
//
// The reduction index is the pixel m, which is the SLOW index of both NHWC operands, while the MFMA wants 8
// consecutive reduction elements per lane.  So both operand tiles are transposed on their way into LDS: a thread
// loads a 4-pixel x 4-channel block (four 16-byte rows), applies ReLU / the per-sample factor, rounds to bf16 and
// writes four 8-byte column pieces Ps[c][m .. m+3] (row stride 40 elements: conflict-free 16-byte fragment reads).
// Stride-1 convolutions only (both operands on the base grid: the IDENT form of sg_wgrad_kernel); the launcher
// falls back to the fp32 kernel for everything else.  Pipeline as in sg_igemm_bf16_kernel: global loads one k-tile
// (32 pixels) ahead in registers, LDS double buffer, fragment reads one k-step ahead.
#include "sg_conv.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// QSCALE: 0 = none, 1 = per-sample factor looked up per pixel row, 2 = one factor per k-tile (Hg*Wg % 32 == 0: the 32
// pixels of a k-tile lie in one sample)
template <int QSCALE>
__global__ __launch_bounds__(256, 3) void sg_wgrad_bf16_kernel(const SgWgradArgs p) {
  constexpr int BC = 128, BN = 128, WM = 2, WN = 2, BK = 32, LDK = 40, NT = 256;
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

  // thread -> 4x4 block: pixels 4*mq .. +3 of the k-tile, channels 4*cq .. +3 of the tile.  mq is the fast index, so
  // the 8-byte LDS writes of a wave (8 mq x 8 cq) spread over all banks; a global load instruction touches
  // 8 pixel rows x 128 contiguous bytes.
  const int mq = tid & 7, cq = tid >> 3;          // 8 x 32
  const int pc = c0 + 4 * cq, qn = n0 + 4 * cq;
  const bool p_cok = pc < p.Cp, q_cok = qn < p.Cq;
  // per-row pixel cursors of the 4 rows of this thread's block (tap validity of P follows (y, x))
  int pb[4], py[4], px[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m_begin + 4 * mq + r;
    pb[r] = m / HW;
    const int rem = m - pb[r] * HW;
    py[r] = rem / p.Wg;
    px[r] = rem - py[r] * p.Wg;
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
  unsigned p_lin = 4u * (unsigned)((m_begin + 4 * mq + dy * p.Wp + dx) * p.Cp + pc);
  unsigned q_lin = 4u * (unsigned)((m_begin + 4 * mq) * p.Cq + qn);
  const unsigned p_row = 4u * p.Cp, q_row = 4u * p.Cq;
  const unsigned p_step = 4u * BK * p.Cp, q_step = 4u * BK * p.Cq;
  int m_next = m_begin;
  unsigned vzero = 0;
  asm volatile("" : "+v"(vzero));
  int qs_b = m_begin / HW, qs_rem = m_begin - (m_begin / HW) * HW;
  float q_sc = 1.f, q_sc4[4] = {1.f, 1.f, 1.f, 1.f};

  float4 p_reg[4], q_reg[4];
  const float relu_floor = relu_in ? 0.f : -__builtin_inff();
  auto relu = [&](float v) {
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(relu_floor));
    return r;
  };
  auto load_tile = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m_next + 4 * mq + r;
      const int iy = py[r] + dy, ix = px[r] + dx;
      const bool live = m < m_end;
      const bool okp = live & p_cok & ((unsigned)iy < (unsigned)p.Hp) & ((unsigned)ix < (unsigned)p.Wp);
      p_reg[r] = bload(rsrc_p, okp ? p_lin + r * p_row : OOB);
      q_reg[r] = bload(rsrc_q, (live & q_cok) ? q_lin + r * q_row : OOB);
      if (QSCALE == 1) q_sc4[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, live ? (int)(4u * (unsigned)pb[r]) : (int)OOB, 0, 0));
      advance(pb[r], py[r], px[r]);
    }
    p_lin += p_step;
    q_lin += q_step;
    if (QSCALE == 2) {
      q_sc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, (int)(vzero + 4u * (unsigned)qs_b), 0, 0));
      qs_rem += BK;
      if (qs_rem >= HW) { qs_rem -= HW; ++qs_b; }
    }
    m_next += BK;
  };
  const bool do_bias = p.dbias != nullptr && t == 0 && c0 == 0;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto store_tile = [&](int buf) {
    unsigned short* ps = Ps + buf * BC * LDK + (4 * cq) * LDK + 4 * mq;
    unsigned short* qs = Qs + buf * BN * LDK + (4 * cq) * LDK + 4 * mq;
    float4 pv[4], qv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[r] = make_float4(relu(p_reg[r].x), relu(p_reg[r].y), relu(p_reg[r].z), relu(p_reg[r].w));
      qv[r] = q_reg[r];
      if (QSCALE) { const float sc = QSCALE == 2 ? q_sc : q_sc4[r]; qv[r].x *= sc; qv[r].y *= sc; qv[r].z *= sc; qv[r].w *= sc; }
      bsum.x += qv[r].x; bsum.y += qv[r].y; bsum.z += qv[r].z; bsum.w += qv[r].w;      // fp32 column sums of Q (bias gradient)
    }
    auto col = [](float a, float b, float c, float d) {
      bf16x4 h;
      h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
      return h;
    };
    *reinterpret_cast<bf16x4*>(ps + 0 * LDK) = col(pv[0].x, pv[1].x, pv[2].x, pv[3].x);
    *reinterpret_cast<bf16x4*>(ps + 1 * LDK) = col(pv[0].y, pv[1].y, pv[2].y, pv[3].y);
    *reinterpret_cast<bf16x4*>(ps + 2 * LDK) = col(pv[0].z, pv[1].z, pv[2].z, pv[3].z);
    *reinterpret_cast<bf16x4*>(ps + 3 * LDK) = col(pv[0].w, pv[1].w, pv[2].w, pv[3].w);
    *reinterpret_cast<bf16x4*>(qs + 0 * LDK) = col(qv[0].x, qv[1].x, qv[2].x, qv[3].x);
    *reinterpret_cast<bf16x4*>(qs + 1 * LDK) = col(qv[0].y, qv[1].y, qv[2].y, qv[3].y);
    *reinterpret_cast<bf16x4*>(qs + 2 * LDK) = col(qv[0].z, qv[1].z, qv[2].z, qv[3].z);
    *reinterpret_cast<bf16x4*>(qs + 3 * LDK) = col(qv[0].w, qv[1].w, qv[2].w, qv[3].w);
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

  if (do_bias) {      // (block-uniform) reduce the per-thread column sums over the 8 pixel lanes, one atomic per column
    float4* red = reinterpret_cast<float4*>(smem);
    __syncthreads();
    red[tid] = bsum;
    __syncthreads();
    if (mq == 0) {
      float4 s4 = red[tid];
      for (int k = 1; k < 8; ++k) {
        const float4 o = red[tid + k];
        s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
      }
      if (qn < p.Cq) {
        float* d = p.dbias + qn;
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
  if (!ident || a.Cp < 64 || a.Cq < 64 || (a.Cp & 3) || (a.Cq & 3) || a.ntaps < 1 || a.ntaps > SG_MAX_TAPS)
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
  long mchunk = (M + nchunks - 1) / nchunks;
  mchunk = (mchunk + 31) / 32 * 32;
  nchunks = (M + mchunk - 1) / mchunk;
  a.mchunk = (int)mchunk;
  const dim3 grid((unsigned)(combos * nchunks)), block(256);
  if (a.qscale && HW % 32 == 0) hipLaunchKernelGGL((sg_wgrad_bf16_kernel<2>), grid, block, 0, s, a);
  else if (a.qscale) hipLaunchKernelGGL((sg_wgrad_bf16_kernel<1>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((sg_wgrad_bf16_kernel<0>), grid, block, 0, s, a);
  return sg_launch_status();
}
