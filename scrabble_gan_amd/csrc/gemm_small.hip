// Small dense GEMM for the Dense layers of the nets (z projection, CBN gamma/beta, logits heads,
// recognizer per-frame classifier) and their gradients.  These are <0.1 % of the step's FLOPs and
// have awkward shapes (N = 1, 53, 128; K = 32): a plain LDS-tiled VALU kernel, 32x32 tile per
// workgroup, 2x2 outputs per thread.  C = alpha * op(A) * op(B) + beta * C (+ bias[n]).
#include "sg_common.h"

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm_small(const float* A, const float* Bm, float* C, const float* bias, int M, int N,
                                                    int K, int lda, int ldb, int ldc, float alpha, float beta) {
  __shared__ float As[32][33], Bs[32][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int k0 = 0; k0 < K; k0 += 32) {
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int r = e >> 5, c = e & 31;
      // As[r][c] = op(A)[m0+r][k0+c]
      float v = 0.f;
      if (TA) {  // A stored [K][M]: coalesce along m
        const int kk = k0 + r, mm = m0 + c;
        if (kk < K && mm < M) v = A[(size_t)kk * lda + mm];
        As[c][r] = v;
      } else {
        const int mm = m0 + r, kk = k0 + c;
        if (mm < M && kk < K) v = A[(size_t)mm * lda + kk];
        As[r][c] = v;
      }
      float w = 0.f;
      if (TB) {  // B stored [N][K]: coalesce along k
        const int nn = n0 + r, kk = k0 + c;
        if (nn < N && kk < K) w = Bm[(size_t)nn * ldb + kk];
        Bs[c][r] = w;
      } else {
        const int kk = k0 + r, nn = n0 + c;
        if (kk < K && nn < N) w = Bm[(size_t)kk * ldb + nn];
        Bs[r][c] = w;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float a0 = As[ty][k], a1 = As[ty + 16][k];
      const float b0 = Bs[k][tx], b1 = Bs[k][tx + 16];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
      if (m < M && n < N) {
        float v = alpha * acc[i][j];
        if (bias) v += bias[n];
        if (beta != 0.f) v += beta * C[(size_t)m * ldc + n];
        C[(size_t)m * ldc + n] = v;
      }
    }
}

extern "C" int sg_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                       int ldc, int transA, int transB, float alpha, float beta, void* stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0) return SG_ERR_ARG;
  if (M == 0 || N == 0) return SG_OK;
  dim3 grid(sg_cdiv(N, 32), sg_cdiv(M, 32)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (transA && transB) SG_KERNEL((k_gemm_small<true, true>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta);
  else if (transA) SG_KERNEL((k_gemm_small<true, false>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta);
  else if (transB) SG_KERNEL((k_gemm_small<false, true>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta);
  else SG_KERNEL((k_gemm_small<false, false>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta);
  return sg_launch_status();
}
