// Probe of ds_read_b64_tr_b8 (gfx950): which LDS bytes does each lane receive?  (No ISA text in the image's guides.)
// Pass 1: LDS byte a holds the id of the lane whose address register covers it (lane * 8 + j -> lane), pass 2: holds j.
// Every lane supplies address 8 * lane.  Prints, per lane, the (source lane, source byte) of its 8 result bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k(unsigned char* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  const int lane = threadIdx.x;
  for (int j = 0; j < 8; ++j) lds[lane * 8 + j] = mode == 0 ? (unsigned char)lane : (unsigned char)j;
  for (int j = 512 + lane * 8; j < 512 + lane * 8 + 8; ++j) lds[j] = 0xee;
  __syncthreads();
  typedef int v2i __attribute__((ext_vector_type(2)));
  v2i r;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + 8u * lane;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  reinterpret_cast<v2i*>(out)[lane] = r;
}

int main() {
  unsigned char *d, h[2][512];
  hipMalloc(&d, 512);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h[mode], d, 512, hipMemcpyDeviceToHost);
  }
  for (int lane = 0; lane < 64; ++lane) {
    printf("lane %2d:", lane);
    for (int j = 0; j < 8; ++j) printf(" (%2d,%d)", h[0][lane * 8 + j], h[1][lane * 8 + j]);
    printf("\n");
  }
  return 0;
}
