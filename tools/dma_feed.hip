// L2 -> LDS feed rate of global_load_lds_dwordx4 (the operand path of the DMA-fed conv kernels), without any MFMA work:
// one 512-thread workgroup per CU streams 64 KB "tiles" (8 wave-instructions of 1 KB per wave) out of a per-workgroup slice
// that it re-reads, either one tile per barrier with the queue drained at the barrier (what the conv
// loop does: two 64 KB stages), or with DEPTH tiles kept in flight (what 32 KB k-tiles in four stages would allow).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_feed tools/dma_feed.hip && /tmp/dma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int TILE_KB, int DEPTH>
__global__ __launch_bounds__(512) void k_feed(const unsigned char* src, int iters, unsigned* sink, unsigned slice) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int PER_THREAD = TILE_KB * 1024 / 512 / 16;       // 16-byte loads per thread and tile
  constexpr int STAGES = DEPTH + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x * slice;       // slice: power of two >= the tile
  auto issue = [&](int t) {
    const unsigned char* tsrc = base + (size_t)(((unsigned)t * TILE_KB * 1024u) & (slice - 1));
    unsigned char* dst = smem + (t % STAGES) * TILE_KB * 1024;
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) {
      const int ii = q * 8 + wave;                                // wave-instruction index inside the tile
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tsrc + ii * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + ii * 1024), 16, 0, 0);
    }
  };
  for (int t = 0; t < DEPTH; ++t) issue(t);
  for (int t = 0; t < iters; ++t) {
    // tile t must have landed; DEPTH - 1 younger tiles may stay in flight
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) { if (PER_THREAD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else { if (PER_THREAD == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    issue(t + DEPTH);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = reinterpret_cast<unsigned*>(smem)[lane];
}

template <int TILE_KB, int DEPTH>
static void run(const char* name, const unsigned char* src, unsigned* sink, unsigned slice) {
  const int blocks = 256, iters = 4000;
  const size_t lds = (size_t)(DEPTH + 1) * TILE_KB * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_feed<TILE_KB, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_feed<TILE_KB, DEPTH>), dim3(blocks), dim3(512), lds, 0, src, iters, sink, slice);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double bytes = (double)blocks * (iters + DEPTH) * TILE_KB * 1024;
  printf("%-64s %8.3f ms  %6.2f TB/s  (%5.1f GB/s per CU, %4.0f KB of LDS)\n", name, ts[2], bytes / (ts[2] * 1e-3) * 1e-12,
         bytes / (ts[2] * 1e-3) * 1e-9 / 256, lds / 1024.0);
}

int main() {
  unsigned char* src;
  unsigned* sink;
  (void)hipMalloc(&src, (size_t)256 * (2u << 20) + (1u << 20));
  (void)hipMemset(src, 1, (size_t)256 * (2u << 20) + (1u << 20));
  (void)hipMalloc(&sink, 256 * 4);
  printf("# global_load_lds_dwordx4 feed, 256 workgroups x 512 threads, no compute; each workgroup re-reads its own slice:\n");
  printf("# 64 KB slices = 2 MB per XCD (L2 hits), 2 MB slices = 512 MB in all (beyond L2 and the Infinity Cache)\n");
  for (unsigned slice : {64u << 10, 2u << 20}) {
    printf("slice %u KB\n", slice >> 10);
    run<64, 1>("  64 KB tiles, queue drained at every barrier (the conv loop: two 64 KB stages)", src, sink, slice);
    run<32, 1>("  32 KB tiles, queue drained at every barrier", src, sink, slice);
    run<32, 2>("  32 KB tiles, 2 in flight", src, sink, slice);
    run<32, 3>("  32 KB tiles, 3 in flight (4 stages = 128 KB)", src, sink, slice);
  }
  return 0;
}
