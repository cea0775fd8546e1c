// dense.hip -- nn.Linear on a batch of a few hundred rows (the classifier, /root/reference/models/resnet.py:242).
//
// C[M][N] (+ bias[N]) = A[M][Kd] * B[N][Kd]^T, both operands 16-bit with the reduction dimension contiguous, fp32
// accumulation.  The tiled implicit-GEMM kernel serves this shape with 128 x 128 tiles: M = 256 rows give 16 workgroups on
// 256 CUs (46.6 us forward + 27.4 us data gradient for a 1 GFLOP product, round 5).  Here one workgroup owns one
// 32 x 32 output tile - (N / 32) x (M / 32) = 256 ... 512 workgroups - and its four waves split the reduction: each wave
// feeds v_mfma_f32_32x32x16 straight from global memory (the MFMA A / B lane maps of cn_common.h are "8 consecutive
// reduction elements of one row per lane", i.e. a 16-byte load of a K-contiguous operand), the four partial tiles are
// summed through LDS in a fixed order and stored with bias / ReLU.  Forward: A = x [B][C], B = the KRSC filter [K][C];
// data gradient: A = dy [B][K], B = the CRSK filter copy [C][K].
#include "cn_api_internal.h"

namespace {

template <typename T> struct DenseMfma;
template <> struct DenseMfma<bf16_t> { static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) { return cn_mfma_32x32x16_bf16(a, b, c); } };
template <> struct DenseMfma<f16_t> { static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) { return cn_mfma_32x32x16_f16(a, b, c); } };

__device__ __forceinline__ s16x8 dense_ld(const char* base, long long row, int ld_elems, int k, bool ok) {
  if (!ok) { s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; return z; }
  const u32x4 v = cn_ld16(base + (row * (long long)ld_elems + k) * 2);
  return __builtin_bit_cast(s16x8, v);
}

#define DENSE_PITCH 33
template <typename T, bool OUTF32>
__global__ __launch_bounds__(256) void dense_smallm_kernel(const char* A, const char* B, char* C, const float* bias, int M, int N,
                                                          int Kd, int relu) {
  __shared__ float red[4][32 * DENSE_PITCH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row = blockIdx.y * 32 + (lane & 31), col = blockIdx.x * 32 + (lane & 31);
  const int koff = 8 * (lane >> 5);
  const int steps = (Kd + 15) / 16, per = (steps + 3) / 4;
  const int s0 = wave * per, s1 = s0 + per < steps ? s0 + per : steps;
  const bool rok = row < M, cok = col < N;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int s = s0;
  for (; s + 3 < s1; s += 4) {          // eight 16-byte loads in flight per lane
    s16x8 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = (s + u) * 16 + koff;
      a[u] = dense_ld(A, row, Kd, k, rok && k < Kd);
      b[u] = dense_ld(B, col, Kd, k, cok && k < Kd);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = DenseMfma<T>::run(a[u], b[u], acc);
  }
  for (; s < s1; ++s) {
    const int k = s * 16 + koff;
    acc = DenseMfma<T>::run(dense_ld(A, row, Kd, k, rok && k < Kd), dense_ld(B, col, Kd, k, cok && k < Kd), acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    red[wave][i * DENSE_PITCH + (lane & 31)] = acc[r];
  }
  __syncthreads();
  for (int idx = tid; idx < 1024; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    const int gr = blockIdx.y * 32 + i, gc = blockIdx.x * 32 + j;
    if (gr >= M || gc >= N) continue;
    float v = ((red[0][i * DENSE_PITCH + j] + red[1][i * DENSE_PITCH + j]) + red[2][i * DENSE_PITCH + j]) + red[3][i * DENSE_PITCH + j];
    if (bias != nullptr) v += bias[gc];
    if (relu) v = v > 0.f ? v : 0.f;
    if (OUTF32) ((float*)C)[(size_t)gr * N + gc] = v;
    else cn_store_elem<T>((T*)C + (size_t)gr * N + gc, v);
  }
}

}  // namespace

// M rows of a batch, N outputs, Kd reduction length (a multiple of 8; operands 16-byte aligned, row pitch = Kd).  Returns
// CN_OK after launching, or 1 when the shape is not one this kernel serves (the caller then uses the tiled kernel).
int cn_dense_smallm(const void* A, const void* B, void* C, const float* bias, int M, int N, int Kd, int dtype, int out_f32,
                    int relu, hipStream_t stream) {
  if (dtype != CN_BF16 && dtype != CN_F16) return 1;
  if (M < 1 || M > 1024 || N < 8 || Kd < 16 || Kd % 8 != 0) return 1;
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return 1;
  if (cn_get_option("dense_smallm", 1) == 0) return 1;
  const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
#define DSK(T, F) CN_LAUNCH((dense_smallm_kernel<T, F>), grid, dim3(256), stream, (const char*)A, (const char*)B, (char*)C, bias, M, N, Kd, relu)
  if (dtype == CN_BF16) { if (out_f32) DSK(bf16_t, true); else DSK(bf16_t, false); }
  else { if (out_f32) DSK(f16_t, true); else DSK(f16_t, false); }
#undef DSK
  cn_set_last_kernel("dense_smallm_kernel<%s, %s>", dtype == CN_BF16 ? "bf16_t" : "f16_t", out_f32 ? "true" : "false");
  return cn_check_launch("dense_smallm");
}
