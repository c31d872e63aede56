// Argument block and launcher of the DMA-fed implicit-GEMM kernel (conv_bf16v2.hip), shared with the Winograd-domain
// path (conv_winograd.hip), which runs its sixteen per-frequency GEMMs as ONE grouped launch of that kernel.
#pragma once
#include "sg_conv.h"

typedef unsigned short u16;

#define SG2_IDENT_OUT 32

struct SgIgemm2Args {
  const u16* a;        // bf16 activation operand, NHWC [Bn, Ha, Wa, Ca], Ca % 64 == 0
  const u16* w;        // packed bf16 filter [tap][N][Ca] (sg_pack_filter_bf16), N % 64 == 0
  float* out;          // fp32 NHWC [Bn, Ho, Wo, N]
  u16* out16;          // nullable: bf16 copy of the result (after bias / mask / ReLU), same layout
  const float* bias;
  const float* bias2;
  const float* mask;   // nullable, fp32, same shape as out: result := 0 where mask <= 0
  const u16* mask16;   // nullable alternative to `mask`: the same tensor as bf16 (sign and zero are what matter)
  const float* amax_a; // fp8 operands only: device scalars max|activation| and max|filter| behind the per-tensor scales
  const float* amax_w; //   (operand = fp8(value * 448 / amax)); the epilogue multiplies the sums by amax_a * amax_w / 448^2
  float* amax_out;     // nullable (config c5): amax_out[0] = max(., max |result|), amax_out[1] = max(., max |amax_rowscale[b] * result|)
  const float* amax_rowscale;   //   -- the per-tensor scales of the NEXT fp8 launch that reads the result, taken while it is written
  int Bn, Ha, Wa, Ca;
  int Hg, Wg, a_sy, a_sx;
  int Ho, Wo, N, o_sy, o_sx, o_oy, o_ox;
  int ntaps, flags;
  int full_tiles, tail_split, n_tiles_total;
  // grouped launch (conv_winograd.hip): rows [g * group_rows, (g + 1) * group_rows) of the base grid form group g, whose activation
  // operand starts a_group_bytes and whose filter starts w_group_bytes behind group g - 1's (group_rows % 128 == 0, 128-row tiles: a tile never
  // straddles two groups; offsets inside a group stay 32-bit).  group_rows = 0: one group.
  int group_rows;
  long a_group_bytes, w_group_bytes;
  SgTap taps[SG_MAX_TAPS];   // w_off in elements of the packed filter
};

// -> SG_OK, and *twin_rows_done = number of leading output rows (pixels) whose bf16 copy the kernel wrote itself; es = bytes per
// operand element (4 fp32, 2 bf16, 1 fp8)
int sg_launch_igemm_bf16v2(const SgIgemm2Args& a_in, hipStream_t s, long* twin_rows_done, int es);
