// Sustained matrix-core rate of this device, per MFMA shape, on RANDOM operands (tools/README.md): the measured ceiling the
// conv kernels' TFLOP/s are read against next to the spec peak.  Bare loops: operands in registers, independent
// accumulators, 2 waves per SIMD on every CU, ~2 s of back-to-back launches before the timed ones (the chip lowers its clock
// under matrix load; MI355X_MICROARCH.md 'DVFS give-back').  In-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/mfma_peak.hip && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_loop(const float* src, float* sink, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63;
  const float* s = src + (size_t)(blockIdx.x * 512 + threadIdx.x) * 16;
  float acc_out = 0.f;
  unsigned long long t0 = 0, r0 = 0;
  if (MODE == 0) {            // f32 32x32x2
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = s[i];
    for (int i = 0; i < 2; ++i) b[i] = s[4 + i];
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc_out += acc[i][r];
  } else if (MODE == 1) {     // f32 16x16x4
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = s[i];
    for (int i = 0; i < 4; ++i) b[i] = s[8 + i];
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc_out += acc[i][r];
  } else if (MODE == 2) {     // bf16 32x32x16
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)s[(i + e) & 15];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)s[(5 + i + e) & 15];
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc_out += acc[i][r];
  } else if (MODE == 3) {     // bf16 16x16x32
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)s[(i + e) & 15];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)s[(9 + i + e) & 15];
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc_out += acc[i][r];
  } else if (MODE == 4) {     // fp8 (e4m3) block-scaled 32x32x64, scales 1.0
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    v8i a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__float_as_int(s[(i + e) & 15]) & 0x77777777);
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__float_as_int(s[(5 + i + e) & 15]) & 0x77777777);
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i * 2 + j], 0, 0, 0, 127, 0, 127);
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc_out += acc[i][r];
  } else if (MODE == 5) {     // fp8 (e4m3) block-scaled 16x16x128
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    v8i a[8], b[4];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__float_as_int(s[(i + e) & 15]) & 0x77777777);
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__float_as_int(s[(9 + i + e) & 15]) & 0x77777777);
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 4 + j], 0, 0, 0, 127, 0, 127);
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc_out += acc[i][r];
  }
  if (MODE == 6 || MODE == 7) {   // f32 32x32x2 as the fp32 conv loop issues it: max(a, floor) in front of each MFMA pair
    // (6), plus (7) the operands re-read from LDS every step (6 ds_read_b64 per 16 MFMAs, one step ahead) and a barrier per 8 steps
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = s[i & 15];
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 af[2][4], bf[2][2];
    const float floor_f = s[15] > 2.f ? 0.f : -__builtin_inff();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 8u * (threadIdx.x & 63);
#define RD(dst, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define READ6(slot, o) do { RD(af[slot][0], (o)); RD(af[slot][1], (o) + 512); RD(af[slot][2], (o) + 1024); RD(af[slot][3], (o) + 1536); RD(bf[slot][0], (o) + 2048); RD(bf[slot][1], (o) + 2560); } while (0)
    for (int i = 0; i < 4; ++i) { af[0][i] = f32x2{s[i], s[i + 1]}; af[1][i] = f32x2{s[i + 2], s[i + 3]}; }
    for (int i = 0; i < 2; ++i) { bf[0][i] = f32x2{s[8 + i], s[9 + i]}; bf[1][i] = f32x2{s[10 + i], s[11 + i]}; }
    auto mma = [&](int slot) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = fmaxf(af[slot][i][e], floor_f);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[slot][j][e], acc[i * 2 + j], 0, 0, 0);
        }
    };
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 6) {
      for (int it = 0; it < iters / 2; ++it) { mma(0); mma(1); }
    } else {
      READ6(0, 0);
      for (int it = 0; it < iters / 16; ++it) {
#define STEP(slot_next, slot_cur, o) do { READ6(slot_next, o); asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); mma(slot_cur); __builtin_amdgcn_sched_barrier(0); } while (0)
        STEP(1, 0, 4096); STEP(0, 1, 8192); STEP(1, 0, 12288); STEP(0, 1, 16384); STEP(1, 0, 20480); STEP(0, 1, 24576); STEP(1, 0, 28672);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        READ6(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc_out += acc[i][r];
  }
  if (MODE == 8 || MODE == 11 || MODE == 12) {   // as 7 (11: no barrier, 12: no max()), the six reads of the next step spread between the MFMAs of the current one
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = s[i & 15];
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 af[2][4], bf[2][2];
    const float floor_f = s[15] > 2.f ? 0.f : -__builtin_inff();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 8u * (threadIdx.x & 63);
    for (int i = 0; i < 4; ++i) { af[0][i] = f32x2{s[i], s[i + 1]}; af[1][i] = f32x2{s[i + 2], s[i + 3]}; }
    for (int i = 0; i < 2; ++i) { bf[0][i] = f32x2{s[8 + i], s[9 + i]}; bf[1][i] = f32x2{s[10 + i], s[11 + i]}; }
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MM(slot, e, i) do { const float a_ = MODE == 12 ? af[slot][i][e] : fmaxf(af[slot][i][e], floor_f); \
    acc[(i) * 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bf[slot][0][e], acc[(i) * 2], 0, 0, 0); \
    acc[(i) * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bf[slot][1][e], acc[(i) * 2 + 1], 0, 0, 0); SB(); } while (0)
#define STEP8(sn, sc, o) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); \
    MM(sc, 0, 0); RD(af[sn][0], (o)); SB(); MM(sc, 0, 1); RD(af[sn][1], (o) + 512); SB(); MM(sc, 0, 2); RD(af[sn][2], (o) + 1024); SB(); \
    MM(sc, 0, 3); RD(af[sn][3], (o) + 1536); SB(); MM(sc, 1, 0); RD(bf[sn][0], (o) + 2048); SB(); MM(sc, 1, 1); RD(bf[sn][1], (o) + 2560); SB(); \
    MM(sc, 1, 2); MM(sc, 1, 3); } while (0)
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters / 16; ++it) {
      STEP8(1, 0, 4096); STEP8(0, 1, 8192); STEP8(1, 0, 12288); STEP8(0, 1, 16384); STEP8(1, 0, 20480); STEP8(0, 1, 24576); STEP8(1, 0, 28672);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (MODE != 11) __builtin_amdgcn_s_barrier();
      STEP8(0, 1, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc_out += acc[i][r];
  }
  if (MODE == 13) {   // as 7, the eight max() of a step batched in front of its 16 MFMAs (separate registers)
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = s[i & 15];
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 af[2][4], bf[2][2];
    const int ifloor = s[15] > 2.f ? 0 : (int)0x80000000;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 8u * (threadIdx.x & 63);
    for (int i = 0; i < 4; ++i) { af[0][i] = f32x2{s[i], s[i + 1]}; af[1][i] = f32x2{s[i + 2], s[i + 3]}; }
    for (int i = 0; i < 2; ++i) { bf[0][i] = f32x2{s[8 + i], s[9 + i]}; bf[1][i] = f32x2{s[10 + i], s[11 + i]}; }
    auto mma13 = [&](int slot) {
      float ar[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float x_ = af[slot][i][e]; ar[e][i] = __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, x_), ifloor)); }
      SB();
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[e][i], bf[slot][j][e], acc[i * 2 + j], 0, 0, 0);
    };
#define STEP13(slot_next, slot_cur, o) do { READ6(slot_next, o); asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); SB(); mma13(slot_cur); SB(); } while (0)
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    READ6(0, 0);
    for (int it = 0; it < iters / 16; ++it) {
      STEP13(1, 0, 4096); STEP13(0, 1, 8192); STEP13(1, 0, 12288); STEP13(0, 1, 16384); STEP13(1, 0, 20480); STEP13(0, 1, 24576); STEP13(1, 0, 28672);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      READ6(0, 0);
      SB();
      mma13(1);
      SB();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc_out += acc[i][r];
  }
  if (MODE == 9 || MODE == 10) {   // bf16 32x32x16 with six ds_read_b128 per 8 MFMAs: grouped in front (9) or spread (10); barrier per 4 steps
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = s[i & 15];
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i af[2][4], bf[2][2];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 16u * (threadIdx.x & 63);
    for (int sl = 0; sl < 2; ++sl) {
      for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) af[sl][i][e] = __float_as_int(s[(i + e + sl) & 15]) & 0x3fff3fff;
      for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) bf[sl][i][e] = __float_as_int(s[(7 + i + e + sl) & 15]) & 0x3fff3fff;
    }
#define RDQ(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define MB(slot, i) do { acc[(i) * 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[slot][i]), __builtin_bit_cast(bf16x8, bf[slot][0]), acc[(i) * 2], 0, 0, 0); \
    acc[(i) * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[slot][i]), __builtin_bit_cast(bf16x8, bf[slot][1]), acc[(i) * 2 + 1], 0, 0, 0); SB(); } while (0)
#define STEP9(sn, sc, o) do { RDQ(af[sn][0], (o)); RDQ(af[sn][1], (o) + 1024); RDQ(af[sn][2], (o) + 2048); RDQ(af[sn][3], (o) + 3072); RDQ(bf[sn][0], (o) + 4096); RDQ(bf[sn][1], (o) + 5120); \
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); SB(); MB(sc, 0); MB(sc, 1); MB(sc, 2); MB(sc, 3); } while (0)
#define STEP10(sn, sc, o) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); \
    MB(sc, 0); RDQ(af[sn][0], (o)); RDQ(af[sn][1], (o) + 1024); SB(); MB(sc, 1); RDQ(af[sn][2], (o) + 2048); RDQ(af[sn][3], (o) + 3072); SB(); \
    MB(sc, 2); RDQ(bf[sn][0], (o) + 4096); RDQ(bf[sn][1], (o) + 5120); SB(); MB(sc, 3); } while (0)
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 9) {
      RDQ(af[0][0], 0); RDQ(af[0][1], 1024); RDQ(af[0][2], 2048); RDQ(af[0][3], 3072); RDQ(bf[0][0], 4096); RDQ(bf[0][1], 5120);
      for (int it = 0; it < iters / 4; ++it) {
        STEP9(1, 0, 8192); STEP9(0, 1, 16384); STEP9(1, 0, 24576);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RDQ(af[0][0], 0); RDQ(af[0][1], 1024); RDQ(af[0][2], 2048); RDQ(af[0][3], 3072); RDQ(bf[0][0], 4096); RDQ(bf[0][1], 5120);
        SB(); MB(1, 0); MB(1, 1); MB(1, 2); MB(1, 3);
      }
    } else {
      for (int it = 0; it < iters / 4; ++it) {
        STEP10(1, 0, 8192); STEP10(0, 1, 16384); STEP10(1, 0, 24576);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        STEP10(0, 1, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc_out += acc[i][r];
  }
  if (MODE == 14) {   // bf16 16x16x32 with the conv loop's LDS traffic: 12 ds_read_b128 per 32 MFMAs (the same bytes per FLOP as
    // six reads per eight 32x32x16 MFMAs), single fragment set read in front of the step, barrier per 2 steps (= 64 k)
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = s[i & 15];
    __syncthreads();
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i af[8], bf[4];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + 16u * (threadIdx.x & 63);
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) af[i][e] = __float_as_int(s[(i + e) & 15]) & 0x3fff3fff;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) bf[i][e] = __float_as_int(s[(9 + i + e) & 15]) & 0x3fff3fff;
#define RD12(o) do { RDQ(af[0], (o)); RDQ(af[1], (o) + 1024); RDQ(af[2], (o) + 2048); RDQ(af[3], (o) + 3072); RDQ(af[4], (o) + 4096); RDQ(af[5], (o) + 5120); \
    RDQ(af[6], (o) + 6144); RDQ(af[7], (o) + 7168); RDQ(bf[0], (o) + 8192); RDQ(bf[1], (o) + 9216); RDQ(bf[2], (o) + 10240); RDQ(bf[3], (o) + 11264); } while (0)
    auto mma16 = [&]() {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i * 4 + j], 0, 0, 0);
    };
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters / 2; ++it) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); mma16(); SB(); RD12(12288);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB(); mma16(); SB();
      __builtin_amdgcn_s_barrier();
      RD12(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc_out += acc[i][r];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 512 + threadIdx.x] = acc_out;
  if (lane == 0 && (threadIdx.x >> 6) == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int MODE>
static void run(const char* name, double flop_per_mfma, int mfma_per_iter, const float* src, float* sink, unsigned long long* stamps) {
  const int blocks = 256, iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  double warm = 0;
  while (warm < 2000.0) {      // ~2 s of back-to-back launches before the timed ones
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(512), 0, 0, src, sink, stamps, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    warm += ms;
  }
  std::vector<float> t;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(512), 0, 0, src, sink, stamps, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  std::vector<unsigned long long> st(2 * blocks);
  (void)hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);   // GHz (realtime = 100 MHz)
  std::sort(clk.begin(), clk.end());
  const double flops = (double)blocks * 8 * iters * mfma_per_iter * flop_per_mfma;
  // cycles per MFMA per SIMD: wall time x in-kernel clock over the MFMAs of the two waves that share a SIMD
  const double cyc = t[2] * 1e-3 * clk[blocks / 2] * 1e9 / ((double)iters * mfma_per_iter * 2);
  printf("%-34s %8.3f ms  %8.1f TFLOP/s   in-kernel clock %.3f GHz (median of %d workgroups)   %.1f cycles / MFMA / SIMD\n", name, t[2],
         flops / (t[2] * 1e-3) * 1e-12, clk[blocks / 2], blocks, cyc);
}

int main() {
  const size_t n = (size_t)256 * 512 * 16;
  std::vector<float> h(n);
  srand(7);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *src, *sink;
  unsigned long long* stamps;
  (void)hipMalloc(&src, n * 4);
  (void)hipMalloc(&sink, 256 * 512 * 4);
  (void)hipMalloc(&stamps, 512 * 8);
  (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  printf("# bare MFMA loops on random operands, 256 workgroups x 8 waves (2 per SIMD), operands in registers\n");
  run<0>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 8, src, sink, stamps);
  run<1>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4, 32, src, sink, stamps);
  run<2>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 8, src, sink, stamps);
  run<3>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 32, src, sink, stamps);
  run<4>("v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)", 2.0 * 32 * 32 * 64, 8, src, sink, stamps);
  run<5>("v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)", 2.0 * 16 * 16 * 128, 32, src, sink, stamps);
  run<6>("f32 32x32x2 + max() per MFMA pair", 2.0 * 32 * 32 * 2, 16, src, sink, stamps);
  run<7>("f32 32x32x2 + max() + LDS reads + barrier", 2.0 * 32 * 32 * 2, 8, src, sink, stamps);
  run<8>("f32 ... reads spread between the MFMAs", 2.0 * 32 * 32 * 2, 8, src, sink, stamps);
  run<11>("f32 ... spread reads, no barrier", 2.0 * 32 * 32 * 2, 8, src, sink, stamps);
  run<12>("f32 ... spread reads, barrier, no max()", 2.0 * 32 * 32 * 2, 8, src, sink, stamps);
  run<13>("f32 ... in-front reads, 8 max() batched per step", 2.0 * 32 * 32 * 2, 8, src, sink, stamps);
  run<9>("bf16 32x32x16 + 6 ds_read_b128 in front", 2.0 * 32 * 32 * 16, 8, src, sink, stamps);
  run<10>("bf16 32x32x16 + 6 ds_read_b128 spread", 2.0 * 32 * 32 * 16, 8, src, sink, stamps);
  run<14>("bf16 16x16x32 + 12 ds_read_b128 per 32 MFMAs (one fragment set)", 2.0 * 16 * 16 * 32, 32, src, sink, stamps);
  return 0;
}
