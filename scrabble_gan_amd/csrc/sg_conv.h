// Internal argument blocks of the tap-list convolution kernels (conv_igemm.hip, conv_wgrad.hip).
//
// Every convolution-shaped op of the train step (Conv2D fwd / data-grad, Conv2DTranspose fwd /
// data-grad, and all their weight-grads) is expressed over a "base grid" of M = Bn*Hg*Wg pixels
// and a list of taps.  For tap t the activation operand is sampled at
//     (a_sy*yg + taps[t].dy, a_sx*xg + taps[t].dx)      (zero outside [0,Ha)x[0,Wa))
// and contracted with the per-tap weight matrix that starts at w + taps[t].w_off.  The result for
// base pixel (b,yg,xg) is written to output pixel (o_sy*yg + o_oy, o_sx*xg + o_ox).
#pragma once
#include "sg_common.h"

#define SG_MAX_TAPS 9

struct SgTap {
  int dy, dx, w_off;
};

struct SgIgemmArgs {
  const float* a;      // activation operand, NHWC [Bn, Ha, Wa, Ca]
  const float* w;      // weights; per tap a [Ca x N] matrix, see b_nk
  float* out;          // NHWC [Bn, Ho, Wo, N]
  const float* bias;   // [N] or null
  const float* bias2;  // [N] or null
  const float* mask;   // same shape as out, or null: result := 0 where mask <= 0 (ReLU backward)
  int Bn, Ha, Wa, Ca;
  int Hg, Wg, a_sy, a_sx;
  int Ho, Wo, N, o_sy, o_sx, o_oy, o_ox;
  int ntaps, ldw, flags;
  unsigned a_bytes, w_bytes;   // extents of a and w for the buffer-load range check (filled in by sg_launch_igemm)
  int full_tiles, tail_split, n_tiles_total;   // work decomposition (filled in by the launcher): see sg_igemm_kernel
  SgTap taps[SG_MAX_TAPS];
};

// weight-grad: dW_t[c][n] += sum_m P_t[m][c] * Q[m][n]
struct SgWgradArgs {
  const float* p;      // tap-shifted operand, NHWC [Bn, Hp, Wp, Cp]
  const float* q;      // base-grid operand,   NHWC [Bn, Hq, Wq, Cq] sampled at (q_sy*yg, q_sx*xg)
  float* dw;           // per tap a [Cp x Cq] row-major matrix at dw + taps[t].w_off
  const float* qscale; // nullable: per-sample factor [Bn] applied to the Q rows (weights the samples' contributions to dW / dbias)
  float* dbias;        // nullable: dbias[n] += sum_m Q[m][n] (bias gradient when Q is dy), done by the tap-0 / c-tile-0 workgroups
  int Bn, Hp, Wp, Cp, p_sy, p_sx;
  int Hq, Wq, Cq, q_sy, q_sx;
  int Hg, Wg;
  int ntaps, flags;    // SG_RELU_IN applies to P
  int mchunk;          // base pixels per block (multiple of 32)
  unsigned p_bytes, q_bytes;   // extents of p and q for the buffer-load range check (filled in by sg_launch_wgrad)
  SgTap taps[SG_MAX_TAPS];
  // grouped form (conv_winograd.hip; fp32 kernel only): p_plane > 0 makes "tap" t an independent product -- P = p + t * p_plane,
  // Q = q + t * q_plane (elements), no pixel shift, dW_t at dw + t * Cp * Cq; the tap table is not read and ntaps may reach SG_MAX_GROUPS
  long p_plane, q_plane;
};
#define SG_MAX_GROUPS 36

struct SgThinArgs {
  const float* a;
  const float* w;      // per tap a [C] vector at w + taps[t].w_off
  float* out;
  const float* bias;
  const float* mask;
  const float* qscale; // thin weight-grad only: per-sample factor on the C-channel operand
  int Bn, Ha, Wa, Hg, Wg, C;
  int ntaps, flags;
  SgTap taps[SG_MAX_TAPS];
};

int sg_launch_igemm(const SgIgemmArgs& a, bool b_nk, hipStream_t s);
int sg_launch_wgrad(const SgWgradArgs& a, hipStream_t s);
int sg_launch_wgrad_bf16(const SgWgradArgs& a, hipStream_t s);   // SG_ERR_UNSUPPORTED when the shape does not qualify
