// fp8 weight gradient (BASELINE config c5):  dW_t[c][n] += s * sum_m P8[pix(m) + tap_t][c] * Q8[m][n]
//   P8 = the layer input as OCP e4m3 (the SAME fp8 copy the forward launch of the layer read: ReLU already folded in by
//        sg_cvt_fp8), Q8 = the output gradient as OCP e5m2 (wider exponent: gradients span more binades than activations),
//        already multiplied by the per-sample factors of the shared backward sweep; both with per-tensor scales
//        448 / amax_p and 57344 / amax_q, so s = amax_p * amax_q / (448 * 57344), applied to the fp32 sums.
//   v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = 0 (A = e4m3), blgp = 1 (B = e5m2), unit block scales, fp32 accumulation:
//   2x the bf16 MFMA rate per clock, half the operand bytes.
//
// Same structure as sg_wgrad_bf16v2_kernel<256, 256> (conv_bf16v2.hip): the reduction index (pixels) is the slow index of
// both operands in memory, so the tiles go global -> LDS by DMA exactly as they lie ([128 pixels][256 channels] bytes: the
// byte geometry of the bf16 kernel's [64 pixels][2 x 128 channels]) and the LDS READ transposes: ds_read_b64_tr_b8 hands
// lane i of a 16-lane group byte i of eight consecutive rows, i.e. eight consecutive pixels of one channel; four of them
// make the 32-byte (32-pixel) half of a 64-pixel MFMA k-step for a lane.  Both operands use the same pixel -> (half,
// read, byte) map, which is all the MFMA needs (the order of the reduction index is free when the operands agree).
// LDS image of a tile: row r (pixel) at 256 r; 16-byte chunk ch of the row in slot ch ^ (2 (r & 7)): a 32-lane half of a
// transposed read takes 8 rows x 2 adjacent chunks -- with this XOR the 16 slots are distinct (even / odd), 64 banks once.
// One fragment set (48 registers beside 128 accumulators), the next step's reads issued right behind the current MFMAs;
// two 64 KB stages; the DMAs of tile t+2 issued behind the barrier of tile t (a whole tile of lead).
// Workgroup = one (tap, 256-channel c-tile, 256-channel n-tile) x one chunk of pixels; partial sums meet in dW through float
// atomics; deterministic mode (sg_set_deterministic) runs ONE chunk per (tap, tile): one adder per address.
#include "sg_conv.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(256))) unsigned int sg8w_zero_page[64];      // 256 bytes of zeros (never written)

struct SgWgrad8Args {
  const unsigned char* p;    // e4m3 [Bn, H, W, Cp]
  const unsigned char* q;    // e5m2 [Bn, H, W, Cq]
  float* dw;                 // per tap a [Cp x Cq] row-major fp32 matrix at dw + taps[t].w_off
  const float* amax_p;       // device scalars behind the per-tensor scales
  const float* amax_q;
  int Bn, H, W, Cp, Cq;
  int ntaps;
  int mchunk;                // pixels per workgroup (multiple of 128)
  int c_tiles, n_tiles, nchunks;
  SgTap taps[SG_MAX_TAPS];
};

constexpr int W8_PIX = 128;                 // pixels per tile
constexpr int W8_TILE = W8_PIX * 256;       // bytes of an operand tile (32 KB)

__global__ __launch_bounds__(512, 2) void sg_wgrad_fp8_kernel(const SgWgrad8Args p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int WN = 4, WM = 2, TM = 4, TN = 2;       // 2 x 4 waves, wave tile 128 (c) x 64 (n)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const long M = (long)p.Bn * p.H * p.W;
  const int combos = p.ntaps * p.c_tiles * p.n_tiles;
  const int chunk = blockIdx.x / combos;
  int combo = blockIdx.x - chunk * combos;
  const int tap = combo / (p.c_tiles * p.n_tiles);
  combo -= tap * p.c_tiles * p.n_tiles;
  const int c0 = (combo / p.n_tiles) * 256, n0 = (combo % p.n_tiles) * 256;
  const long m_begin = (long)chunk * p.mchunk;
  const long m_end = m_begin + p.mchunk < M ? m_begin + p.mchunk : M;
  const int KT = (int)((m_end - m_begin + W8_PIX - 1) / W8_PIX);
  const int tdy = p.taps[tap].dy, tdx = p.taps[tap].dx;

  // ---- DMA lane roles: instruction ii = 8 i + wave (i = 0..3) moves rows 4 ii .. 4 ii + 3 of the 128-row tile: lane -> row
  //      (l >> 4) of the four, slot (l & 15); it fetches chunk slot ^ (2 (row & 7)) of its pixel's 256 channels
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(sg8w_zero_page) + 16 * (lane & 15);
  const int HW = p.H * p.W;
  int ry[4], rx[4], rm[4];
  unsigned csw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * (8 * i + wave) + (lane >> 4);
    csw[i] = 16u * (unsigned)((lane & 15) ^ (2 * (row & 7)));
    rm[i] = row;
    const long m = m_begin + row;
    const long mm = m < M ? m : 0;
    const int rem = (int)(mm % HW);
    ry[i] = rem / p.W;
    rx[i] = rem - ry[i] * p.W;
  }
  const int m_len = (int)(m_end - m_begin);
  const int Hh = p.H, Ww = p.W;
  const int adv_y = (W8_PIX % HW) / p.W, adv_x = (W8_PIX % HW) % p.W;       // cursor advance of one tile (within one image plane)
  const unsigned long long p_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(p.p) + (unsigned)c0 + (unsigned long long)m_begin * p.Cp;
  const unsigned long long q_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(p.q) + (unsigned)n0 + (unsigned long long)m_begin * p.Cq;
  const unsigned long long z_base64 = (unsigned long long)reinterpret_cast<uintptr_t>(zero);
  const long tap_shift = (long)tdy * p.W + tdx;
  // LDS: P stage 0 | P stage 1 | Q stage 0 | Q stage 1.  Part i of the load tile at the cursors into stage `st`, then advance
  auto issue_part = [&](int st, int i) {
    // (bitwise, not short-circuit: a branch in the k-loop splits it into several basic blocks, and the MFMAs -- whose
    //  results are not needed before the epilogue -- get sunk behind all the fragment reads: 299 spills, seen in the ISA)
    const bool live = rm[i] < m_len;
    const int sy = ry[i] + tdy, sx = rx[i] + tdx;
    const bool okp = live & ((unsigned)sy < (unsigned)Hh) & ((unsigned)sx < (unsigned)Ww);
    const unsigned long long pa = p_base64 + (unsigned long long)(((long)rm[i] + tap_shift) * p.Cp) + csw[i];
    const unsigned plo = okp ? (unsigned)pa : (unsigned)z_base64, phi = okp ? (unsigned)(pa >> 32) : (unsigned)(z_base64 >> 32);
    const unsigned char* src_p = reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)phi << 32) | plo));
    const unsigned long long qa = q_base64 + (unsigned long long)((long)rm[i] * p.Cq) + csw[i];
    const unsigned qlo = live ? (unsigned)qa : (unsigned)z_base64, qhi = live ? (unsigned)(qa >> 32) : (unsigned)(z_base64 >> 32);
    const unsigned char* src_q = reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)qhi << 32) | qlo));
    const int ii = 8 * i + wave;
    unsigned char* dst_p = smem + st * W8_TILE + ii * 1024;
    unsigned char* dst_q = smem + 2 * W8_TILE + st * W8_TILE + ii * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_p, (__attribute__((address_space(3))) void*)dst_p, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_q, (__attribute__((address_space(3))) void*)dst_q, 16, 0, 0);
    rm[i] += W8_PIX;
    rx[i] += adv_x;
    const int wx = rx[i] >= Ww ? 1 : 0;
    rx[i] -= wx * Ww;
    ry[i] += adv_y + wx;
    ry[i] -= ry[i] >= Hh ? Hh : 0;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- transposed fragment reads.  ds_read_b64_tr_b8, per 16-lane group: lane 2 q + e (q = 0..7, e = 0, 1) supplies the
  //      address of bytes 8 e .. 8 e + 7 of the block's row q; lane i receives byte i of rows 0..7 (row q in its byte q).
  //      Lane l: group (l >> 4): h = l >> 5 (which 32 pixels of the 64-pixel k-step), grp = (l >> 4) & 1 (which 16 of the
  //      operand's 32 channels); read u (0..3) covers pixels 64 ks + 32 h + 8 u + q.
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int q8 = (lane & 15) >> 1, e8 = lane & 1, grp = (lane >> 4) & 1, h = lane >> 5;
  const int rowl = 32 * h + q8;                                  // + 8 u + 64 ks: (row & 7) == q8 for every read
  unsigned pa_addr[TM], qb_addr[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ch = 8 * wm + 2 * i + grp;                         // 16-byte chunk: the wave's 128 channels = chunks 8 wm .. 8 wm + 7
    pa_addr[i] = lds0 + (unsigned)(256 * rowl + 16 * (ch ^ (2 * q8)) + 8 * e8);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ch = 4 * wn + 2 * j + grp;                         // the wave's 64 channels = chunks 4 wn .. 4 wn + 3
    qb_addr[j] = lds0 + (unsigned)(2 * W8_TILE + 256 * rowl + 16 * (ch ^ (2 * q8)) + 8 * e8);
  }
  v2i af[TM][4], bfr[TN][4];          // [group][u]
#define SG8W_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // rows of read u: + 8 u -> + 2048 u bytes; k-step ks: + 64 rows = 16384 bytes; stage: + 32768 bytes
#define SG8W_READ_FRAGS(st, ks)                                                                 \
  do {                                                                                          \
    {                                                                                           \
      SG8W_TR(af[0][0], pa_addr[0], (st) * W8_TILE + (ks) * 16384 + 0 * 2048);                  \
      SG8W_TR(af[0][1], pa_addr[0], (st) * W8_TILE + (ks) * 16384 + 1 * 2048);                  \
      SG8W_TR(af[0][2], pa_addr[0], (st) * W8_TILE + (ks) * 16384 + 2 * 2048);                  \
      SG8W_TR(af[0][3], pa_addr[0], (st) * W8_TILE + (ks) * 16384 + 3 * 2048);                  \
      SG8W_TR(af[1][0], pa_addr[1], (st) * W8_TILE + (ks) * 16384 + 0 * 2048);                  \
      SG8W_TR(af[1][1], pa_addr[1], (st) * W8_TILE + (ks) * 16384 + 1 * 2048);                  \
      SG8W_TR(af[1][2], pa_addr[1], (st) * W8_TILE + (ks) * 16384 + 2 * 2048);                  \
      SG8W_TR(af[1][3], pa_addr[1], (st) * W8_TILE + (ks) * 16384 + 3 * 2048);                  \
      SG8W_TR(af[2][0], pa_addr[2], (st) * W8_TILE + (ks) * 16384 + 0 * 2048);                  \
      SG8W_TR(af[2][1], pa_addr[2], (st) * W8_TILE + (ks) * 16384 + 1 * 2048);                  \
      SG8W_TR(af[2][2], pa_addr[2], (st) * W8_TILE + (ks) * 16384 + 2 * 2048);                  \
      SG8W_TR(af[2][3], pa_addr[2], (st) * W8_TILE + (ks) * 16384 + 3 * 2048);                  \
      SG8W_TR(af[3][0], pa_addr[3], (st) * W8_TILE + (ks) * 16384 + 0 * 2048);                  \
      SG8W_TR(af[3][1], pa_addr[3], (st) * W8_TILE + (ks) * 16384 + 1 * 2048);                  \
      SG8W_TR(af[3][2], pa_addr[3], (st) * W8_TILE + (ks) * 16384 + 2 * 2048);                  \
      SG8W_TR(af[3][3], pa_addr[3], (st) * W8_TILE + (ks) * 16384 + 3 * 2048);                  \
      SG8W_TR(bfr[0][0], qb_addr[0], (st) * W8_TILE + (ks) * 16384 + 0 * 2048);                 \
      SG8W_TR(bfr[0][1], qb_addr[0], (st) * W8_TILE + (ks) * 16384 + 1 * 2048);                 \
      SG8W_TR(bfr[0][2], qb_addr[0], (st) * W8_TILE + (ks) * 16384 + 2 * 2048);                 \
      SG8W_TR(bfr[0][3], qb_addr[0], (st) * W8_TILE + (ks) * 16384 + 3 * 2048);                 \
      SG8W_TR(bfr[1][0], qb_addr[1], (st) * W8_TILE + (ks) * 16384 + 0 * 2048);                 \
      SG8W_TR(bfr[1][1], qb_addr[1], (st) * W8_TILE + (ks) * 16384 + 1 * 2048);                 \
      SG8W_TR(bfr[1][2], qb_addr[1], (st) * W8_TILE + (ks) * 16384 + 2 * 2048);                 \
      SG8W_TR(bfr[1][3], qb_addr[1], (st) * W8_TILE + (ks) * 16384 + 3 * 2048);                 \
    }                                                                                           \
  } while (0)
  auto mma8 = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const v8i a = __builtin_shufflevector(__builtin_shufflevector(af[i][0], af[i][1], 0, 1, 2, 3),
                                            __builtin_shufflevector(af[i][2], af[i][3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const v8i b = __builtin_shufflevector(__builtin_shufflevector(bfr[j][0], bfr[j][1], 0, 1, 2, 3),
                                              __builtin_shufflevector(bfr[j][2], bfr[j][3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
        // cbsz = 0: A = e4m3; blgp = 1: B = e5m2; E8M0 scale 127 = 2^0 on both sides
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i][j], 0, 1, 0, 127, 0, 127);
      }
    }
  };
#define SG8W_WAIT_ALL() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define SG8W_K_TILE(st, sn)                                                            \
  do {                                                                                 \
    SG8W_WAIT_ALL();                                                                   \
    mma8();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    SG8W_READ_FRAGS(st, 1);                                                            \
    SG8W_WAIT_ALL();                                                                   \
    mma8();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
    __builtin_amdgcn_s_barrier();                                                      \
    SG8W_READ_FRAGS(sn, 0);                                                            \
    issue_part(st, 0);                                                                 \
    issue_part(st, 1);                                                                 \
    issue_part(st, 2);                                                                 \
    issue_part(st, 3);                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(1, i);
    SG8W_READ_FRAGS(0, 0);
  }
  for (int kt = 0; kt < KT; kt += 2) {          // (rows past m_end and whole phantom tiles come from the zero page)
    SG8W_K_TILE(0, 1);
    SG8W_K_TILE(1, 0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: accumulator row = c, lane = n; dW[c][n] += scale * acc (atomics: nchunks adders per address)
  const float ap = p.amax_p[0], aq = p.amax_q[0];
  const float scale = (ap > 0.f ? ap * (1.f / 448.f) : 1.f) * (aq > 0.f ? aq * (1.f / 57344.f) : 1.f);
  float* dwt = p.dw + p.taps[tap].w_off;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        atomicAdd(dwt + (size_t)c * p.Cq + n, acc[i][j][r] * scale);
      }
    }
  }
}

// x8 e4m3 [B,H,W,Cin] (ReLU folded in where the layer applies one), dy8 e5m2 [B,H,W,Cout] (per-sample factors folded in), with
// their amax device scalars; dw fp32 [kh,kw,Cin,Cout] +=.  SAME stride-1 convolutions (or 1x1) with Cin % 256 == 0 and
// Cout % 256 == 0, else SG_ERR_UNSUPPORTED (caller: sg_conv2d_bwd_weight_bf16v2).
extern "C" int sg_conv2d_bwd_weight_fp8(const void* x8, const float* amax_x, const void* dy8, const float* amax_dy, float* dw, int B, int H,
                                        int W, int Cin, int Cout, int kh, int kw, int pad_same, void* stream) {
  if (!x8 || !dy8 || !amax_x || !amax_dy || !dw || kh * kw > SG_MAX_TAPS) return SG_ERR_ARG;
  if (!pad_same && (kh != 1 || kw != 1)) return SG_ERR_UNSUPPORTED;
  if ((Cin % 256) || (Cout % 256)) return SG_ERR_UNSUPPORTED;
  const long M = (long)B * H * W;
  if (M <= 0) return SG_OK;
  if (M * Cin >= (1L << 40) || M * Cout >= (1L << 40)) return SG_ERR_UNSUPPORTED;
  SgWgrad8Args a{};
  a.p = (const unsigned char*)x8; a.q = (const unsigned char*)dy8; a.dw = dw; a.amax_p = amax_x; a.amax_q = amax_dy;
  a.Bn = B; a.H = H; a.W = W; a.Cp = Cin; a.Cq = Cout; a.ntaps = kh * kw;
  const int ph = kh / 2, pw = kw / 2;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) a.taps[ky * kw + kx] = SgTap{ky - ph, kx - pw, (ky * kw + kx) * Cin * Cout};
  a.c_tiles = Cin / 256;
  a.n_tiles = Cout / 256;
  const int combos = a.ntaps * a.c_tiles * a.n_tiles;
  // pixel chunks per combo: the cost model of sg2_launch_wgrad (conv_bf16v2.hip) with 128-pixel tiles at the fp8 rate:
  // few long workgroups -- every workgroup ends with 256 KB of float atomics
  static const int wg_env = getenv("SG_WGRAD8_CHUNKS") ? atoi(getenv("SG_WGRAD8_CHUNKS")) : 0;
  const double t_tile = 1.05, t_epi = 52.0;
  const long tiles_all = (M + W8_PIX - 1) / W8_PIX;
  const long max_chunks = tiles_all / 2 > 0 ? tiles_all / 2 : 1;
  long nchunks = 1;
  double best = 1e30;
  for (long cc = 1; cc <= max_chunks && cc * combos <= 8192; ++cc) {
    const long Wg = combos * cc;
    const double t = (double)((Wg + 255) / 256) * ((double)((tiles_all + cc - 1) / cc) * t_tile + t_epi);
    if (t < best) { best = t; nchunks = cc; }
  }
  if (wg_env > 0) nchunks = wg_env < max_chunks ? wg_env : max_chunks;
  if (sg_deterministic()) nchunks = 1;
  long mchunk = (M + nchunks - 1) / nchunks;
  mchunk = (mchunk + W8_PIX - 1) / W8_PIX * W8_PIX;
  nchunks = (M + mchunk - 1) / mchunk;
  a.mchunk = (int)mchunk;
  a.nchunks = (int)nchunks;
  constexpr int LDS_BYTES = 4 * W8_TILE;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sg_wgrad_fp8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return SG_ERR_UNSUPPORTED;
    }
    attr_done = true;
  }
  SG_KERNEL(sg_wgrad_fp8_kernel, dim3((unsigned)(combos * nchunks)), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  return sg_launch_status();
}

// ------------------------------------------------------------------------------------------
// Gradient operand of the fp8 weight-grad: ONE sweep over dy [M, C] (rows = pixels) yields
//   out_e5m2  = e5m2(clamp(rowscale[sample(m)] * dy * 57344 / amax_scaled))     (the weight-grad operand)
//   out_e4m3  = e4m3(clamp(dy * 448 / amax_plain))   (nullable: the data-grad launch's operand of the same gradient)
//   dbias[c] += sum_m rowscale * dy                   (nullable: fp32 column sums of the scaled fp32 values = the bias gradient)
// amax[0] = max |dy|, amax[1] = max |rowscale * dy| come from sg_amax2_f32 (one read sweep) or from the producing kernel.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned sg8w_pack4_e5m2(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(a, -57344.f), 57344.f), fminf(fmaxf(b, -57344.f), 57344.f), w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(c, -57344.f), 57344.f), fminf(fmaxf(d, -57344.f), 57344.f), w, true);
  return (unsigned)w;
}

__device__ __forceinline__ unsigned sg8w_pack4_e4m3(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -448.f), 448.f), fminf(fmaxf(b, -448.f), 448.f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -448.f), 448.f), fminf(fmaxf(d, -448.f), 448.f), w, true);
  return (unsigned)w;
}

__global__ __launch_bounds__(256) void k_amax2(const float* __restrict__ x, long n4, const float* __restrict__ rowscale, long rowlen,
                                               unsigned* amax_bits) {
  float m0 = 0.f, m1 = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[e];
    const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    m0 = fmaxf(m0, a);
    m1 = fmaxf(m1, rowscale ? a * fabsf(rowscale[(4 * e) / rowlen]) : a);
  }
  m0 = sg_wave_max(m0);
  m1 = sg_wave_max(m1);
  if ((threadIdx.x & 63) == 0) {
    sg_atomic_max_nonneg(reinterpret_cast<float*>(amax_bits), m0);
    sg_atomic_max_nonneg(reinterpret_cast<float*>(amax_bits) + 1, m1);
  }
}

// amax2[0] = max(amax2[0], max |x|), amax2[1] = max(amax2[1], max |rowscale[i / rowlen] * x_i|); the caller zeroes amax2.
// n % 4 == 0, rowlen % 4 == 0.
extern "C" int sg_amax2_f32(const float* x, long n, const float* rowscale, long rowlen, float* amax2, void* stream) {
  if (!x || !amax2 || n < 0 || (n & 3) || (rowscale && (rowlen <= 0 || (rowlen & 3)))) return SG_ERR_ARG;
  if (n == 0) return SG_OK;
  SG_KERNEL(k_amax2, dim3(sg_grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, n / 4, rowscale, rowlen,
                     reinterpret_cast<unsigned*>(amax2));
  return sg_launch_status();
}

__global__ __launch_bounds__(256) void k_cvt_fp8_grad(const float* __restrict__ x, uint2* __restrict__ out_e5m2, uint2* __restrict__ out_e4m3, long M,
                                                      int C, const float* __restrict__ rowscale, long rows_per_sample,
                                                      const float* __restrict__ amax2, float* __restrict__ dbias, int rows_per_block) {
  __shared__ float red[256 * 8];
  const float a_plain = amax2[0], a_scaled = amax2[1];
  const float s4 = a_plain > 0.f ? 448.f / a_plain : 1.f;
  const float s5 = a_scaled > 0.f ? 57344.f / a_scaled : 1.f;
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
        if (out_e4m3)
          out_e4m3[(m * C + 8 * cg) >> 3] = make_uint2(sg8w_pack4_e4m3(v0.x * s4, v0.y * s4, v0.z * s4, v0.w * s4),
                                                       sg8w_pack4_e4m3(v1.x * s4, v1.y * s4, v1.z * s4, v1.w * s4));
        if (rowscale) {
          const float f = rowscale[m / rows_per_sample];
          v0.x *= f; v0.y *= f; v0.z *= f; v0.w *= f; v1.x *= f; v1.y *= f; v1.z *= f; v1.w *= f;
        }
        out_e5m2[(m * C + 8 * cg) >> 3] = make_uint2(sg8w_pack4_e5m2(v0.x * s5, v0.y * s5, v0.z * s5, v0.w * s5),
                                                     sg8w_pack4_e5m2(v1.x * s5, v1.y * s5, v1.z * s5, v1.w * s5));
        s[0] += v0.x; s[1] += v0.y; s[2] += v0.z; s[3] += v0.w; s[4] += v1.x; s[5] += v1.y; s[6] += v1.z; s[7] += v1.w;
      }
    }
    if (!dbias) continue;
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

// x fp32 [M, C] -> out_e5m2 [M, C] (scaled by rowscale[m / rows_per_sample] when given), out_e4m3 (nullable) [M, C] = the unscaled
// e4m3 copy, dbias (nullable) [C] += column sums of the scaled fp32 values; amax2 = {max |x|, max |rowscale x|}; C % 8 == 0.
extern "C" int sg_cvt_fp8_grad(const float* x, void* out_e5m2, void* out_e4m3, long M, int C, const float* rowscale, long rows_per_sample,
                               const float* amax2, float* dbias, void* stream) {
  if (!x || !out_e5m2 || !amax2 || M < 0 || C <= 0 || (C & 7) || (rowscale && rows_per_sample <= 0)) return SG_ERR_ARG;
  if (M == 0) return SG_OK;
  long r = (M + 1023) / 1024;                 // ~1024 workgroups: enough to fill the chip, few enough atomics per column
  if (sg_deterministic() && dbias) r = M;     // one workgroup, one adder per column (slow: a reproducibility mode)
  const int rpb = (int)(r < 32 ? 32 : r);
  SG_KERNEL(k_cvt_fp8_grad, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream, x, (uint2*)out_e5m2,
                     (uint2*)out_e4m3, M, C, rowscale, rows_per_sample, amax2, dbias, rpb);
  return sg_launch_status();
}
