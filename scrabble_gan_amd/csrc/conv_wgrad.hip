// Weight-gradient contraction on the fp32 matrix cores: dW_t[c][n] += sum_m P_t[m][c] * Q[m][n].
//
// The reduction runs over base-grid pixels m, so both operands are already "k-major" in NHWC
// memory: a k-tile of 32 pixels x 128 channels is a straight float4 copy into Ps[k][c] / Qs[k][n]
// (no transpose, conflict-free b32 fragment reads).  One workgroup owns (tap, c-tile, n-tile,
// pixel-chunk); chunks of the same pixels are adjacent in the grid so that the 9 taps x c/n tiles
// that re-read one activation slab run together and hit L2/Infinity Cache.  Partial sums are
// added to dW with float atomics (one 128-byte row segment per half-wave: full atomic rate).
#include "sg_conv.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// IDENT: both operands live on the base grid with unit sampling stride (every stride-1 Conv2D weight-grad): their
// addresses are linear in the pixel index, so the k-loop needs one add per load; only the tap validity of P still
// follows a (y, x) cursor.  The general form (strided sampling: transposed convolutions) keeps full cursors.
// QSCALE: 0 = none; 1 = per-sample factor looked up per lane; 2 = the BK pixels of a k-tile always lie in ONE sample
// (Hg*Wg % BK == 0: every layer of the fixed-shape step), so the factor is one wave-uniform load per k-tile
template <int BC, int BN, int WM, int WN, int BK, int OCC, bool IDENT, int QSCALE>
__global__ __launch_bounds__(WM* WN * 64, OCC) void sg_wgrad_kernel(const SgWgradArgs p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BC / WM / 32, TN = BN / WN / 32;
  constexpr int P_RPP = NT / (BC / 4), P_P = BK / P_RPP;
  constexpr int Q_RPP = NT / (BN / 4), Q_P = BK / Q_RPP;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * BC + 2 * BK * BN];
  float* Ps = smem;
  float* Qs = smem + 2 * BK * BC;

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
  const bool grouped = p.p_plane > 0;                          // (uniform) conv_winograd.hip: tap t = independent product t
  const int dy = grouped ? 0 : p.taps[t < SG_MAX_TAPS ? t : 0].dy, dx = grouped ? 0 : p.taps[t < SG_MAX_TAPS ? t : 0].dx;
  const float* const p_base = p.p + (grouped ? (size_t)t * (size_t)p.p_plane : (size_t)0);
  const float* const q_base = p.q + (grouped ? (size_t)t * (size_t)p.q_plane : (size_t)0);
  const bool relu_in = (p.flags & SG_RELU_IN) != 0;

  float4 p_reg[P_P], q_reg[Q_P];
  float q_sc[Q_P];
  const int pc = c0 + 4 * (tid % (BC / 4));
  const int qn = n0 + 4 * (tid % (BN / 4));
  const bool p_cok = pc < p.Cp, q_cok = qn < p.Cq;
  constexpr int KS = BK / 2;

  // Operands come in through raw buffer loads (32-bit byte offsets, hardware range check): a lane that must
  // contribute zero (padding tap, pixel past the chunk, channel past the edge) gets an out-of-range offset and
  // reads 0.0 -- no select after the load, so the registers are first touched by the LDS store three k-steps later.
  constexpr unsigned OOB = 0xFFFFFFE0u;
  const auto rsrc_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_base), 0, (int)p.p_bytes, 0x00020000);
  const auto rsrc_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q_base), 0, (int)p.q_bytes, 0x00020000);
  auto bload = [](decltype(rsrc_p) r, unsigned voff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return *reinterpret_cast<const float4*>(&v);
  };

  // Per-slot pixel cursors (b, yg, xg), decoded once and advanced by BK pixels per k-tile with
  // compare/subtract wraps: no integer division inside the loop.
  int pb[P_P], py[P_P], px[P_P], qb[Q_P], qy[Q_P], qx[Q_P];
  auto decode = [&](int m, int& bb, int& yy, int& xx) {
    bb = m / HW;
    const int rem = m - bb * HW;
    yy = rem / p.Wg;
    xx = rem - yy * p.Wg;
  };
#pragma unroll
  for (int i = 0; i < P_P; ++i) decode(m_begin + tid / (BC / 4) + i * P_RPP, pb[i], py[i], px[i]);
#pragma unroll
  for (int i = 0; i < Q_P; ++i) decode(m_begin + tid / (BN / 4) + i * Q_RPP, qb[i], qy[i], qx[i]);
  int m_next = m_begin;          // first pixel of the next k-tile to fetch
  // advancing a cursor by BK pixels without branches: BK = adv_b*HW + adv_y*Wg + adv_x (uniform, computed once),
  // then at most one wrap per coordinate
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
  const float relu_floor = relu_in ? 0.f : -__builtin_inff();
  auto relu = [&](float v) {                                      // exactly one VALU op
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(relu_floor));
    return r;
  };
  // IDENT: both addresses are linear in the pixel index -> per-slot byte offsets advanced by one add per k-tile
  unsigned p_lin[P_P], q_lin[Q_P];
#pragma unroll
  for (int i = 0; i < P_P; ++i)
    p_lin[i] = 4u * (unsigned)((m_begin + tid / (BC / 4) + i * P_RPP + dy * p.Wp + dx) * p.Cp + pc);
#pragma unroll
  for (int i = 0; i < Q_P; ++i) q_lin[i] = 4u * (unsigned)((m_begin + tid / (BN / 4) + i * Q_RPP) * p.Cq + qn);
  const unsigned p_step = 4u * BK * p.Cp, q_step = 4u * BK * p.Cq;

  auto load_p = [&]() {
#pragma unroll
    for (int i = 0; i < P_P; ++i) {
      const int m = m_next + tid / (BC / 4) + i * P_RPP;
      const int iy = (IDENT ? py[i] : py[i] * p.p_sy) + dy, ix = (IDENT ? px[i] : px[i] * p.p_sx) + dx;
      // (bitwise &: one straight-line compare chain, no short-circuit branches in the loop)
      const bool ok = (m < m_end) & p_cok & ((unsigned)iy < (unsigned)p.Hp) & ((unsigned)ix < (unsigned)p.Wp);
      const unsigned off = IDENT ? p_lin[i] : 4u * (unsigned)(((pb[i] * p.Hp + iy) * p.Wp + ix) * p.Cp + pc);
      p_reg[i] = bload(rsrc_p, ok ? off : OOB);
      p_lin[i] += p_step;
      advance(pb[i], py[i], px[i]);
    }
  };
  const bool do_bias = p.dbias != nullptr && t == 0 && c0 == 0;     // one (tap, c-tile) column of workgroups sums dy
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(QSCALE ? p.qscale : q_base), 0, QSCALE ? 4 * p.Bn : 0, 0x00020000);
  unsigned vzero = 0;
  asm volatile("" : "+v"(vzero));                                 // opaque per-lane zero: keeps the factor load on the vector path
  int qs_b = m_begin / HW, qs_rem = m_begin - (m_begin / HW) * HW; // (QSCALE == 2) sample of the next k-tile
  auto load_q = [&]() {
#pragma unroll
    for (int i = 0; i < Q_P; ++i) {
      const int m = m_next + tid / (BN / 4) + i * Q_RPP;
      const bool ok = (m < m_end) & q_cok;
      const unsigned off = IDENT ? q_lin[i] : 4u * (unsigned)(((qb[i] * p.Hq + qy[i] * p.q_sy) * p.Wq + qx[i] * p.q_sx) * p.Cq + qn);
      q_reg[i] = bload(rsrc_q, ok ? off : OOB);
      q_lin[i] += q_step;
      if (QSCALE == 1) q_sc[i] = p.qscale[ok ? qb[i] : 0];        // the per-sample factor is applied at the LDS store
      if (!IDENT || QSCALE == 1) advance(qb[i], qy[i], qx[i]);
    }
    if (QSCALE == 2) {
      // a vector load (vmcnt, like the operands) of a uniform address: a scalar load would share the LDS wait counter
      q_sc[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, (int)(vzero + 4u * (unsigned)qs_b), 0, 0));
      qs_rem += BK;
      if (qs_rem >= HW) { qs_rem -= HW; ++qs_b; }
    }
    m_next += BK;
  };
  auto store_p = [&](int buf) {
    float* ps = Ps + buf * BK * BC;
#pragma unroll
    for (int i = 0; i < P_P; ++i) {
      float4 v = p_reg[i];
      v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w);
      *reinterpret_cast<float4*>(ps + (tid / (BC / 4) + i * P_RPP) * BC + 4 * (tid % (BC / 4))) = v;
    }
  };
  auto store_q = [&](int buf) {
    float* qs = Qs + buf * BK * BN;
#pragma unroll
    for (int i = 0; i < Q_P; ++i) {
      float4 v = q_reg[i];
      if (QSCALE) { const float sc = q_sc[QSCALE == 2 ? 0 : i]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
      bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;    // column sums of Q (used by the do_bias workgroups)
      *reinterpret_cast<float4*>(qs + (tid / (BN / 4) + i * Q_RPP) * BN + 4 * (tid % (BN / 4))) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int khalf = lane >> 5;
  const int a_col = wm * (BC / WM) + (lane & 31);
  const int b_col = wn * (BN / WN) + (lane & 31);

  if (KT > 0) {
    load_p();
    load_q();
    store_p(0);
    store_q(0);
  }
  __syncthreads();
  // Same software pipeline as sg_igemm_kernel: fragment reads of k-step s+1 issued before the MFMAs of step s, across
  // k-tile boundaries (workgroup barrier one step before the end of a tile), pinned with sched_barriers; next tile's
  // buffer loads in steps 0/1, its LDS stores in steps KS-5/KS-4.
  static_assert(KS % 2 == 0 && KS >= 6, "fragment double buffer parity");
  float af[2][TM], bf[2][TN];
  {
    const float* ps = Ps + khalf * BC + a_col;
    const float* qs = Qs + khalf * BN + b_col;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = ps[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = qs[j * 32];
  }
  auto k_tile = [&](int buf, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;       // the last tile (nothing left to prefetch) is peeled
    const float* ps = Ps + buf * BK * BC + khalf * BC + a_col;
    const float* qs = Qs + buf * BK * BN + khalf * BN + b_col;
    const float* ps_n = Ps + (buf ^ 1) * BK * BC + khalf * BC + a_col;
    const float* qs_n = Qs + (buf ^ 1) * BK * BN + khalf * BN + b_col;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < KS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nxt][i] = ps[(kk + 1) * 2 * BC + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nxt][j] = qs[(kk + 1) * 2 * BN + j * 32];
      } else if constexpr (more) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nxt][i] = ps_n[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nxt][j] = qs_n[j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
      if constexpr (more) {
        if (kk == 0) load_p();
        if (kk == 1) load_q();
        if (kk == KS - 5) store_p(buf ^ 1);
        if (kk == KS - 4) store_q(buf ^ 1);
        if (kk == KS - 2) __syncthreads();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int kt = 0; kt + 1 < KT; ++kt) k_tile(kt & 1, std::true_type{});
  if (KT > 0) k_tile((KT - 1) & 1, std::false_type{});

  if (do_bias) {      // (block-uniform) reduce the per-thread column sums over the Q_RPP row lanes, one atomic per column
    float4* red = reinterpret_cast<float4*>(smem);
    __syncthreads();                 // every wave is done with the operand tiles
    red[tid] = bsum;
    __syncthreads();
    if (tid < BN / 4) {
      float4 s4 = red[tid];
      for (int k = 1; k < Q_RPP; ++k) {
        const float4 o = red[tid + k * (BN / 4)];
        s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
      }
      if (qn < p.Cq) {
        float* d = p.dbias + qn;
        atomicAdd(d + 0, s4.x); atomicAdd(d + 1, s4.y); atomicAdd(d + 2, s4.z); atomicAdd(d + 3, s4.w);
      }
    }
  }

  float* dwt = p.dw + (grouped ? (size_t)t * (size_t)p.Cp * (size_t)p.Cq : (size_t)p.taps[t < SG_MAX_TAPS ? t : 0].w_off);
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

static inline bool wgrad_ident(const SgWgradArgs& a) {
  return a.p_sy == 1 && a.p_sx == 1 && a.q_sy == 1 && a.q_sx == 1 && a.Hp == a.Hg && a.Wp == a.Wg && a.Hq == a.Hg && a.Wq == a.Wg;
}

template <int BC, int BN, int WM, int WN, int BK = 32, int OCC = 2>
static int launch_wgrad_cfg(SgWgradArgs a, hipStream_t s) {
  const long M = (long)a.Bn * a.Hg * a.Wg;
  const int combos = a.ntaps * sg_cdiv(a.Cp, BC) * sg_cdiv(a.Cq, BN);
  if (combos <= 0 || M <= 0) return SG_OK;
  // ~1500-3000 workgroups (6-12 per CU), the count picked so that combos * nchunks fills whole rounds of the 256 CUs
  // (a launch of W equal workgroups takes ceil(W / 256) workgroup-times); at least 8 k-tiles per workgroup
  const long max_chunks = (M + 255) / 256;
  long nchunks = 1;
  double best = 1e30;
  for (long c = (1536 + combos - 1) / combos; c <= (3072 + combos - 1) / combos; ++c) {
    const long cc = c < 1 ? 1 : (c > max_chunks ? max_chunks : c);
    const long W = combos * cc;
    const long per_cu = (W + 255) / 256;
    const double occ_pen = (per_cu % OCC) ? 1.01 : 1.0;     // a last round with fewer than OCC resident workgroups per CU
    const double loss = (double)per_cu * 256.0 / (double)W * (1.0 + 0.002 * cc) * occ_pen;   // + epilogue cost per chunk
    if (loss < best) {
      best = loss;
      nchunks = cc;
    }
  }
  if (sg_deterministic()) nchunks = 1;        // one adder per dW (and bias-gradient) address: fixed summation order
  long mchunk = (M + nchunks - 1) / nchunks;
  mchunk = (mchunk + 31) / 32 * 32;
  nchunks = (M + mchunk - 1) / mchunk;
  a.mchunk = (int)mchunk;
  const dim3 grid((unsigned)(combos * nchunks)), block(WM * WN * 64);
  const bool ident = wgrad_ident(a), qs = a.qscale != nullptr;
  const bool qs_uniform = qs && ident && ((long)a.Hg * a.Wg) % BK == 0 && a.mchunk % BK == 0;
  if (ident && qs_uniform) SG_KERNEL((sg_wgrad_kernel<BC, BN, WM, WN, BK, OCC, true, 2>), grid, block, 0, s, a);
  else if (ident && qs) SG_KERNEL((sg_wgrad_kernel<BC, BN, WM, WN, BK, OCC, true, 1>), grid, block, 0, s, a);
  else if (ident) SG_KERNEL((sg_wgrad_kernel<BC, BN, WM, WN, BK, OCC, true, 0>), grid, block, 0, s, a);
  else if (qs) SG_KERNEL((sg_wgrad_kernel<BC, BN, WM, WN, BK, OCC, false, 1>), grid, block, 0, s, a);
  else SG_KERNEL((sg_wgrad_kernel<BC, BN, WM, WN, BK, OCC, false, 0>), grid, block, 0, s, a);
  return sg_launch_status();
}

int sg_launch_wgrad(const SgWgradArgs& a_in, hipStream_t s) {
  SgWgradArgs a = a_in;
  if ((a.Cp & 3) || (a.Cq & 3) || a.ntaps < 1 || a.ntaps > (a.p_plane > 0 ? SG_MAX_GROUPS : SG_MAX_TAPS)) return SG_ERR_ARG;
  if (a.p_plane > 0 && (a.qscale || a.dbias || a.q_plane <= 0)) return SG_ERR_ARG;
  const long p_elems = (long)a.Bn * a.Hp * a.Wp * a.Cp, q_elems = (long)a.Bn * a.Hq * a.Wq * a.Cq;
  if (p_elems >= (1L << 30) - 8 || q_elems >= (1L << 30) - 8) return SG_ERR_ARG;   // 32-bit byte offsets (buffer loads)
  a.p_bytes = (unsigned)(4 * p_elems);
  a.q_bytes = (unsigned)(4 * q_elems);
  const bool c_small = a.Cp <= 64, n_small = a.Cq <= 64;
  // (BK = 32 k-tiles at 2 workgroups per CU were measured earlier and dropped: 16 at 3-4 per CU is faster everywhere)
  static const int w8 = getenv("SG_WGRAD_W8") ? atoi(getenv("SG_WGRAD_W8")) : 0;      // (round 4 experiment) eight waves of 32 x 64 per 128 x 128 tile, two workgroups per CU
  if (!c_small && !n_small && w8 && (w8 == 2 || a.p_plane > 0)) return launch_wgrad_cfg<128, 128, 4, 2, 16, 2>(a, s);
  if (!c_small && !n_small) return launch_wgrad_cfg<128, 128, 2, 2, 16, 3>(a, s);
  if (!c_small) return launch_wgrad_cfg<128, 64, 2, 2, 16, 4>(a, s);
  if (!n_small) return launch_wgrad_cfg<64, 128, 2, 2, 16, 4>(a, s);
  return launch_wgrad_cfg<64, 64, 2, 2, 16, 4>(a, s);
}

// ------------------------------------------------------------------------------------------
// thin weight-grad: dW_t[c] += sum_m p1[pix(m)+tap_t] * Q[m, c]   (p1 has ONE channel)
// A workgroup owns a run of output rows.  Per row the one-channel rows it needs are staged in LDS (zero outside the image, ReLU
// applied), then thread (channel group, pixel lane) streams the row of Q with 16-byte loads: per float4 of Q a few LDS
// broadcasts and FMAs, no dependent global load, no bounds check (round 1: 1.2 TB/s of Q).  Ends with one LDS tree and one
// float atomic per (tap, channel) and workgroup.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_thin_wgrad_kernel(const SgThinArgs p, int rows_per_block, int dy0, int nrows, int dx0, int span) {
  extern __shared__ float rows[];      // [nrows][span]
  __shared__ float4 red[256];
  const int cqn = p.C >> 2;            // float4 groups per pixel (divides 256)
  const int lanes = 256 / cqn;         // pixels in flight per block
  const int cq = threadIdx.x % cqn, pl = threadIdx.x / cqn;
  const int total_rows = p.Bn * p.Hg;
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = row_begin + rows_per_block < total_rows ? row_begin + rows_per_block : total_rows;
  const bool relu_in = p.flags & SG_RELU_IN;
  const bool relu_q = p.flags & 16;    // internal: ReLU on the C-channel operand (Cout == 1 weight-grad)
  float4 acc[SG_MAX_TAPS];
  int toff[SG_MAX_TAPS];
#pragma unroll
  for (int t = 0; t < SG_MAX_TAPS; ++t) {
    acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    toff[t] = t < p.ntaps ? (p.taps[t].dy - dy0) * span + p.taps[t].dx - dx0 : 0;
  }
  float4 qsum = make_float4(0.f, 0.f, 0.f, 0.f);      // column sums of the C-channel operand (bias gradient when it is dy)
  constexpr int RB = 8;                // output rows per staging round (inside one image): few barriers, long load streams
  for (int row0 = row_begin; row0 < row_end;) {
    const int b = row0 / p.Hg, yg0 = row0 - b * p.Hg;
    int nb = row_end - row0 < RB ? row_end - row0 : RB;
    nb = p.Hg - yg0 < nb ? p.Hg - yg0 : nb;
    const int stage_n = (nb + nrows - 1) * span;
    __syncthreads();
    for (int e = threadIdx.x; e < stage_n; e += 256) {
      const int r = e / span, xx = e - r * span;
      const int iy = yg0 + dy0 + r, ix = xx + dx0;
      float v = 0.f;
      if (iy >= 0 && iy < p.Ha && ix >= 0 && ix < p.Wa) v = p.a[((size_t)b * p.Ha + iy) * p.Wa + ix];
      rows[e] = relu_in ? fmaxf(v, 0.f) : v;
    }
    __syncthreads();
    const float sc = p.qscale ? p.qscale[b] : 1.f;
    const float* qblk = p.w + (size_t)row0 * p.Wg * p.C + 4 * cq;
    const int npix = nb * p.Wg;
    int yl = 0, x = pl;                  // (yl, x) of pixel idx, advanced without divisions (lanes <= 64 <= Wg is not required: while)
    while (x >= p.Wg) { x -= p.Wg; ++yl; }
#pragma unroll 4
    for (int idx = pl; idx < npix; idx += lanes) {
      float4 q = *reinterpret_cast<const float4*>(qblk + (size_t)idx * p.C);
      q.x *= sc; q.y *= sc; q.z *= sc; q.w *= sc;
      if (relu_q) { q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f); }
      qsum.x += q.x; qsum.y += q.y; qsum.z += q.z; qsum.w += q.w;
      const int base = yl * span + x;
#pragma unroll
      for (int t = 0; t < SG_MAX_TAPS; ++t) {
        if (t < p.ntaps) {
          const float a = rows[toff[t] + base];
          acc[t].x += a * q.x; acc[t].y += a * q.y; acc[t].z += a * q.z; acc[t].w += a * q.w;
        }
      }
      x += lanes;
      while (x >= p.Wg) { x -= p.Wg; ++yl; }
    }
    row0 += nb;
  }
#pragma unroll
  for (int t = 0; t < SG_MAX_TAPS; ++t) {
    if (t >= p.ntaps) continue;   // block-uniform
    __syncthreads();
    red[threadIdx.x] = acc[t];
    __syncthreads();
    for (int s = lanes >> 1; s > 0; s >>= 1) {
      if (pl < s) {
        const float4 o = red[threadIdx.x + s * cqn];
        float4 v = red[threadIdx.x];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        red[threadIdx.x] = v;
      }
      __syncthreads();
    }
    if (pl == 0) {
      const float4 v = red[threadIdx.x];
      float* d = p.out + p.taps[t].w_off + 4 * cq;
      atomicAdd(d + 0, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
    }
  }

  if (p.bias) {        // block-uniform: dbias[c] += column sums
    __syncthreads();
    red[threadIdx.x] = qsum;
    __syncthreads();
    for (int s = lanes >> 1; s > 0; s >>= 1) {
      if (pl < s) {
        const float4 o = red[threadIdx.x + s * cqn];
        float4 v = red[threadIdx.x];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        red[threadIdx.x] = v;
      }
      __syncthreads();
    }
    if (pl == 0) {
      const float4 v = red[threadIdx.x];
      float* d = const_cast<float*>(p.bias) + 4 * cq;
      atomicAdd(d + 0, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
    }
  }
}

// p.a = one-channel operand [Bn,Ha,Wa], p.w = C-channel operand on the base grid [Bn,Hg,Wg,C],
// p.out = dW (per tap a [C] vector at taps[t].w_off)
static int launch_thin_wgrad(const SgThinArgs& a, hipStream_t s) {
  const long M = (long)a.Bn * a.Hg * a.Wg;
  if (M <= 0) return SG_OK;
  if ((a.C & 3) || 256 % (a.C >> 2)) return SG_ERR_UNSUPPORTED;
  int dy0 = a.taps[0].dy, dy1 = dy0, dx0 = a.taps[0].dx, dx1 = dx0;
  for (int t = 1; t < a.ntaps; ++t) {
    dy0 = a.taps[t].dy < dy0 ? a.taps[t].dy : dy0; dy1 = a.taps[t].dy > dy1 ? a.taps[t].dy : dy1;
    dx0 = a.taps[t].dx < dx0 ? a.taps[t].dx : dx0; dx1 = a.taps[t].dx > dx1 ? a.taps[t].dx : dx1;
  }
  const int nrows = dy1 - dy0 + 1, span = a.Wg + dx1 - dx0;
  const size_t lds = sizeof(float) * (size_t)(8 + nrows - 1) * span;      // RB = 8 output rows per staging round
  if (lds > 56 * 1024) return SG_ERR_UNSUPPORTED;
  // 256..512 workgroups of whole rows: each ends in one float atomic per (tap, channel), and same-address atomics serialise
  // (~90 ns each), so the adder count per address bounds this kernel on small inputs while large inputs want the parallelism
  const long total_rows = (long)a.Bn * a.Hg;
  long nb = total_rows < 512 ? total_rows : 512;
  if (nb < 1 || sg_deterministic()) nb = 1;      // (deterministic mode: one workgroup, one adder per address)
  const int rpb = (int)((total_rows + nb - 1) / nb);
  const int grid = (int)((total_rows + rpb - 1) / rpb);
  SG_KERNEL(sg_thin_wgrad_kernel, dim3(grid), dim3(256), lds, s, a, rpb, dy0, nrows, dx0, span);
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
// C-ABI entry points
// ------------------------------------------------------------------------------------------
extern "C" int sg_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, const float* sample_scale, int B, int H, int W,
                                    int Cin, int Cout, int kh, int kw, int pad_same, int flags,
                                    void* stream) {
  if (!x || !dy || !dw || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  const int ph = pad_same ? kh / 2 : 0, pw = pad_same ? kw / 2 : 0;
  const int Ho = pad_same ? H : H - kh + 1, Wo = pad_same ? W : W - kw + 1;
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 1) {  // dW[t, co] = sum_m relu?(x)[pix+tap] * dy[m, co]
    SgThinArgs a{};
    a.a = x; a.w = dy; a.out = dw; a.bias = dbias; a.qscale = sample_scale; a.Bn = B; a.Ha = H; a.Wa = W; a.Hg = Ho; a.Wg = Wo; a.C = Cout;
    a.ntaps = kh * kw; a.flags = flags;
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cout};
    return launch_thin_wgrad(a, s);
  }
  if (Cout == 1) {  // dW[t, ci] = sum_m' x[m', ci] * dy[m' - tap]
    if (dbias || sample_scale) return SG_ERR_UNSUPPORTED;   // the C-channel operand is x here: use sg_bias_grad on dy
    SgThinArgs a{};
    a.a = dy; a.w = x; a.out = dw; a.Bn = B; a.Ha = Ho; a.Wa = Wo; a.Hg = H; a.Wg = W; a.C = Cin;
    a.ntaps = kh * kw; a.flags = (flags & SG_RELU_IN) ? 16 : 0;
    for (int ky = 0; ky < kh; ++ky)
      for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ph - ky, pw - kx, (ky * kw + kx) * Cin};
    return launch_thin_wgrad(a, s);
  }
  SgWgradArgs a{};
  a.p = x; a.q = dy; a.dw = dw; a.dbias = dbias; a.qscale = sample_scale;
  a.Bn = B; a.Hp = H; a.Wp = W; a.Cp = Cin; a.p_sy = 1; a.p_sx = 1;
  a.Hq = Ho; a.Wq = Wo; a.Cq = Cout; a.q_sy = 1; a.q_sx = 1; a.Hg = Ho; a.Wg = Wo;
  a.ntaps = kh * kw; a.flags = flags;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx)
      a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  if (flags & SG_MMA_BF16) {            // bf16 matrix-core operands where the shape qualifies, fp32 otherwise
    a.flags &= ~SG_MMA_BF16;
    const int rc = sg_launch_wgrad_bf16(a, s);
    if (rc != SG_ERR_UNSUPPORTED) return rc;
  }
  return sg_launch_wgrad(a, s);
}

// dW[ky,kx,co,ci] += sum_{b,i,j} dy[b, sh*i+ky-pbh, sw*j+kx-pbw, co] * x[b,i,j,ci]
extern "C" int sg_conv2d_transpose_bwd_weight(const float* x, const float* dy, float* dw, int B,
                                              int H, int W, int Cin, int Cout, int kh, int kw,
                                              int sh, int sw, int flags, void* stream) {
  if (!x || !dy || !dw || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  if (flags & SG_RELU_IN) return SG_ERR_UNSUPPORTED;
  const int pbh = (sh == 1) ? kh / 2 : 0, pbw = (sw == 1) ? kw / 2 : 0;
  SgWgradArgs a{};
  a.p = dy; a.q = x; a.dw = dw;
  a.Bn = B; a.Hp = sh * H; a.Wp = sw * W; a.Cp = Cout; a.p_sy = sh; a.p_sx = sw;
  a.Hq = H; a.Wq = W; a.Cq = Cin; a.q_sy = 1; a.q_sx = 1; a.Hg = H; a.Wg = W;
  a.ntaps = kh * kw; a.flags = 0;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx)
      a.taps[ky * kw + kx] = SgTap{ky - pbh, kx - pbw, (ky * kw + kx) * Cin * Cout};
  return sg_launch_wgrad(a, (hipStream_t)stream);
}
