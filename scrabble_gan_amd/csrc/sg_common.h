// Shared device/host helpers for the gfx950 kernels of the ScrabbleGAN train step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SG_OK 0
#define SG_ERR_ARG (-1)
#define SG_ERR_LAUNCH (-2)
#define SG_ERR_UNSUPPORTED (-3)

#define SG_WAVE 64

// flags shared by the conv entry points (include/scrabble_hip.h)
#define SG_RELU_IN 1    // apply max(x,0) to the activation operand while loading it
#define SG_ACCUM 2      // out += result instead of out = result
#define SG_RELU_OUT 4   // apply max(.,0) to the result
#define SG_TANH_OUT 8   // apply tanh to the result (thin Cout=1 path only)
#define SG_UPS2_IN 64   // Winograd F(4x4) data-grad / weight-grad: the gradient operand is given at HALF resolution, dy = 0.25 * upsample2x2(dy_half)
#define SG_POOL2_OUT 16 // Winograd F(4x4) forward only: the result is avg_pool2x2(conv + bias), written / accumulated as [B, H/2, W/2, N]
#define SG_MMA_BF16 256 // weight-grad: round the matrix-core operands to bf16 (fp32 accumulation); config c3

#include <stdio.h>
#include <stdlib.h>
// Kernel launch + status.  hipGetLastError() returns the last error of ANY earlier runtime call of the thread -- also one the
// host framework left behind (an event / stream query that was "not ready", a failed probe) -- so the slate is wiped right
// before the launch, and the launch's own status is folded into a per-thread flag right after it: an entry point that queues
// several launches (sg_spectral_norm: 3 per iteration + 1) and asks once at its end still sees a failure of ANY of them, not
// only of the last (sg_selftest_launch_status + tests/test_ops_gpu.py::test_launch_status_keeps_an_earlier_failure).
inline thread_local int sg_tls_launch_rc = SG_OK;
static inline void sg_note_launch(hipError_t e, const char* what) {
  if (e == hipSuccess) return;
  fprintf(stderr, "[libscrabble_hip] kernel launch failed: %s: %s (%s)\n", what, hipGetErrorName(e), hipGetErrorString(e));   // (the C-ABI returns a code only)
  sg_tls_launch_rc = SG_ERR_LAUNCH;
}
#define SG_KERNEL(kernel, grid, block, lds, stream, ...)                                  \
  do {                                                                                    \
    (void)hipGetLastError();                                                              \
    kernel<<<(grid), (block), (lds), (stream)>>>(__VA_ARGS__);                            \
    sg_note_launch(hipGetLastError(), #kernel);                                           \
  } while (0)
// -> SG_OK, or SG_ERR_LAUNCH if any SG_KERNEL launch of this thread failed since the last call; resets the flag
static inline int sg_launch_status() {
  const int rc = sg_tls_launch_rc;
  sg_tls_launch_rc = SG_OK;
  return rc;
}

// sg_set_deterministic (conv_igemm.hip): when on, no convolution launch lets two workgroups add into the same output address
// -- forward / data-grad launches are not cut along the reduction, weight-grad launches run ONE pixel chunk per
// (tap, c-tile, n-tile), so every dW element has exactly one adder and its summation order is fixed.
bool sg_deterministic();

static inline int sg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// memory-bound kernels: cap the grid and grid-stride the rest.  4 096 workgroups (round 4; 2 048 before): beside a matrix-bound launch of
// the other network's stream the sweeps get their CU slots one draining workgroup at a time, and a deeper queue of small workgroups keeps
// more of them resident -- two-stream c2 step 138.6 -> 137.9 ms; caps of 1 024 / 512 / 256 cost 1 / 2.5 / 6 % (SG_HBM_GRID_CAP)
static inline int sg_grid_cap() {
  static const int cap = getenv("SG_HBM_GRID_CAP") ? atoi(getenv("SG_HBM_GRID_CAP")) : 4096;
  return cap < 64 ? 64 : cap;
}
static inline int sg_grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > sg_grid_cap()) g = sg_grid_cap();
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float sg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float sg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// amax[0] = max(amax[0], m) for a non-negative float m (non-negative floats order like their bit patterns).  The value only
// grows, so a wave first LOOKS (agent-scope load: no stale L1 line) and adds only when it would raise it: a launch's tens
// of thousands of waves otherwise queue on ONE address (30 720 workgroups x 8 waves cost the 64-filter layers 20 ms per
// step in fp8 mode before this check -- same-address atomics serialise at ~70 ns each).
__device__ __forceinline__ void sg_atomic_max_nonneg(float* amax, float m) {
  if (m <= 0.f) return;
  unsigned* p = reinterpret_cast<unsigned*>(amax);
  const unsigned bits = __float_as_uint(m);
  if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= bits) return;
  atomicMax(p, bits);
}

__device__ __forceinline__ double sg_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
