// Recognizer tail: per-frame softmax -> log(p + 1e-7) -> ctc_loss's own log-softmax -> CTC
// forward/backward (alpha-beta) -> gradient back to the pre-softmax logits.
// Replaces Dense(softmax) + K.ctc_batch_cost of /root/reference/src/bigacgan/net_architecture.py:55-64
// (tf.compat.v1.nn.ctc_loss defaults: blank = last class, ctc_merge_repeated=True).
//
// One wavefront per sample: lanes are classes (C <= 64) in the per-frame phases and extended-label
// states (S = 2L+1 <= 64) in the alpha/beta recursions, where neighbours come from __shfl instead
// of LDS.  lp, p, alpha, beta live in dynamic LDS ((2*T*C + 2*T*S) floats).
#include "sg_common.h"

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

__global__ __launch_bounds__(64) void k_softmax_ctc(const float* logits, const int* labels, int label_stride, float* loss,
                                                    float* dlogits, int T, int C, int T_in, int L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = 2 * L + 1;
  float* lp = sm;                 // [T][C]
  float* pr = lp + T * C;         // [T][C]
  float* al = pr + T * C;         // [T][S]
  float* be = al + T * S;         // [T][S]
  __shared__ int labs[64];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int blank = C - 1;
  const float* lg = logits + (size_t)b * T * C;

  // ---- phase A: probabilities and log-probabilities per frame ----
  for (int t = 0; t < T; ++t) {
    const float a = lane < C ? lg[t * C + lane] : -INFINITY;
    const float mx = sg_wave_max(a);
    const float e = lane < C ? expf(a - mx) : 0.f;
    const float p = e / sg_wave_sum(e);
    const float yv = lane < C ? logf(p + 1e-7f) : -INFINITY;
    const float my = sg_wave_max(yv);
    const float ey = lane < C ? expf(yv - my) : 0.f;
    const float ls = my + logf(sg_wave_sum(ey));
    if (lane < C) {
      pr[t * C + lane] = p;
      lp[t * C + lane] = yv - ls;
    }
  }
  int lab = blank;
  bool bad = false;
  if (lane < S && (lane & 1)) {
    lab = labels[(size_t)b * label_stride + (lane >> 1)];
    bad = lab < 0 || lab >= blank;          // K.ctc_batch_cost rejects labels outside [0, C-2]; never train on a clamped target
    lab = bad ? 0 : lab;
  }
  if (__ballot(bad) != 0ull) {             // wave-uniform: poison this sample loudly (cost +inf, NaN gradient) instead of clamping
    if (lane == 0) loss[b] = INFINITY;
    if (dlogits)
      for (int e = lane; e < T * C; e += 64) dlogits[(size_t)b * T * C + e] = __builtin_nanf("");
    return;
  }
  labs[lane] = lab;
  __syncthreads();
  const int lab_m2 = lane >= 2 ? labs[lane - 2] : blank;
  const int lab_p2 = lane + 2 < 64 ? labs[lane + 2] : blank;
  const bool skip_in = lane >= 2 && lane < S && lab != blank && lab != lab_m2;        // s-2 -> s allowed
  const bool skip_out = lane + 2 < S && lab_p2 != blank && lab_p2 != lab;             // s -> s+2 allowed

  // ---- phase B: alpha ----
  float a_cur = (lane < 2 && lane < S) ? lp[lab] : -INFINITY;
  if (lane < S) al[lane] = a_cur;
  for (int t = 1; t < T_in; ++t) {
    float a1 = __shfl_up(a_cur, 1, 64), a2 = __shfl_up(a_cur, 2, 64);
    if (lane < 1) a1 = -INFINITY;
    if (!skip_in) a2 = -INFINITY;
    float v = lse3(a_cur, a1, a2);
    v = (lane < S) ? v + lp[t * C + lab] : -INFINITY;
    a_cur = v;
    if (lane < S) al[t * S + lane] = v;
  }
  const float aT1 = __shfl(a_cur, S - 1, 64), aT2 = __shfl(a_cur, S - 2, 64);
  const float nll = -lse2(aT1, aT2);
  if (lane == 0) loss[b] = nll;
  if (!dlogits) return;

  // ---- phase C: beta ----
  float b_cur = (lane < S && lane >= S - 2) ? lp[(T_in - 1) * C + lab] : -INFINITY;
  if (lane < S) be[(T_in - 1) * S + lane] = b_cur;
  for (int t = T_in - 2; t >= 0; --t) {
    float b1 = __shfl_down(b_cur, 1, 64), b2 = __shfl_down(b_cur, 2, 64);
    if (lane + 1 >= S) b1 = -INFINITY;
    if (!skip_out) b2 = -INFINITY;
    float v = lse3(b_cur, b1, b2);
    v = (lane < S) ? v + lp[t * C + lab] : -INFINITY;
    b_cur = v;
    if (lane < S) be[t * S + lane] = v;
  }
  __syncthreads();

  // ---- phase D: gradient w.r.t. the logits ----
  float* dl = dlogits + (size_t)b * T * C;
  for (int t = 0; t < T; ++t) {
    if (t >= T_in) {
      if (lane < C) dl[t * C + lane] = 0.f;
      continue;
    }
    float acc = -INFINITY;
    for (int s = 0; s < S; ++s)
      if (labs[s] == lane) acc = lse2(acc, al[t * S + s] + be[t * S + s]);
    const float lpv = lane < C ? lp[t * C + lane] : 0.f;
    const float pv = lane < C ? pr[t * C + lane] : 0.f;
    float g_lp = (lane < C && acc != -INFINITY) ? -expf(acc - lpv + nll) : 0.f;
    const float sg = sg_wave_sum(g_lp);
    const float g_y = lane < C ? g_lp - expf(lpv) * sg : 0.f;           // through ctc_loss's log-softmax
    const float g_p = lane < C ? g_y / (pv + 1e-7f) : 0.f;              // through log(p + 1e-7)
    const float dot = sg_wave_sum(pv * g_p);
    if (lane < C) dl[t * C + lane] = pv * (g_p - dot);                   // through the softmax
  }
}

// out[b, :] = s[b] * x[b, :]
__global__ __launch_bounds__(256) void k_rowscale(const float* x, const float* s, float* out, long rows, int rowlen) {
  const long total = rows * rowlen;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    out[e] = x[e] * s[e / rowlen];
}

// logits [B,T,C]; labels int32 [B,label_stride]; loss [B]; dlogits [B,T,C] or null (= d loss_b / d logits_b)
extern "C" int sg_softmax_ctc(const float* logits, const int* labels, int label_stride, float* loss, float* dlogits, int B, int T,
                              int C, int input_length, int label_length, void* stream) {
  if (!logits || !labels || !loss || C < 2 || C > 64 || label_length < 1 || 2 * label_length + 1 > 64 || input_length < 1 ||
      input_length > T || label_stride < label_length)
    return SG_ERR_ARG;
  const int S = 2 * label_length + 1;
  const size_t lds = (size_t)(2 * T * C + 2 * T * S) * sizeof(float);
  if (lds > 150 * 1024) return SG_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_softmax_ctc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return SG_ERR_UNSUPPORTED;
  }
  SG_KERNEL(k_softmax_ctc, dim3(B), dim3(64), lds, (hipStream_t)stream, logits, labels, label_stride, loss, dlogits, T, C,
                     input_length, label_length);
  return sg_launch_status();
}

extern "C" int sg_rowscale(const float* x, const float* s, float* out, long rows, int rowlen, void* stream) {
  if (!x || !s || !out) return SG_ERR_ARG;
  SG_KERNEL(k_rowscale, dim3(sg_grid_for(rows * rowlen, 256)), dim3(256), 0, (hipStream_t)stream, x, s, out, rows, rowlen);
  return sg_launch_status();
}
