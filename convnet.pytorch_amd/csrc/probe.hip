// probe.hip -- hardware lane-map probes.  The kernels in igemm.hip / wgrad.hip assume the gfx950
// MFMA operand / accumulator lane maps and the ds_read_b64_tr_b16 shuffle documented in
// cn_common.h; these probes run one instruction through exactly those wrappers so a GPU test
// (tests/test_gpu_probe.py) can pin the assumptions against a plain matrix product.
#include "cn_common.h"
#include "cn_api_internal.h"

// D[32][32] = A[32][16] * B[16][32] with one v_mfma_f32_32x32x16_bf16.
__global__ __launch_bounds__(64) void probe_mfma_bf16_kernel(const unsigned short* A, const unsigned short* B,
                                                            float* D) {
  const int l = threadIdx.x;
  s16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (l >> 5) + e;
    a[e] = (short)A[(l & 31) * 16 + k];
    b[e] = (short)B[k * 32 + (l & 31)];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = cn_mfma_32x32x16_bf16(a, b, c);
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[i * 32 + (l & 31)] = c[r];
  }
}

// the same with v_mfma_f32_32x32x16_f16 (operands as fp16 bit patterns)
__global__ __launch_bounds__(64) void probe_mfma_f16_kernel(const unsigned short* A, const unsigned short* B,
                                                           float* D) {
  const int l = threadIdx.x;
  s16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (l >> 5) + e;
    a[e] = (short)A[(l & 31) * 16 + k];
    b[e] = (short)B[k * 32 + (l & 31)];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = cn_mfma_32x32x16_f16(a, b, c);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// D[32][32] = A[32][2] * B[2][32] with one v_mfma_f32_32x32x2_f32.
__global__ __launch_bounds__(64) void probe_mfma_f32_kernel(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  const float a = A[(l & 31) * 2 + (l >> 5)];
  const float b = B[(l >> 5) * 32 + (l & 31)];
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = cn_mfma_32x32x2_f32(a, b, c);
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[i * 32 + (l & 31)] = c[r];
  }
}

// Every lane reads 8 bytes at LDS byte address 8*lane from an LDS image lds16[e] = src[e]
// (256 halfwords) through the transpose read; out[lane][0..3] receives what the lane got.
__global__ __launch_bounds__(64) void probe_tr16_kernel(const unsigned short* src, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds16[256];
  const int l = threadIdx.x;
  for (int e = l; e < 256; e += 64) lds16[e] = src[e];
  __syncthreads();
  s16x4 v = cn_lds_read_tr16_b64((const char*)lds16 + 8 * l);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

extern "C" int cn_probe_mfma_bf16(const unsigned short* A, const unsigned short* B, float* D, void* stream) {
  CN_LAUNCH(probe_mfma_bf16_kernel, dim3(1), dim3(64), (hipStream_t)stream, A, B, D);
  return cn_check_launch("probe_mfma_bf16");
}
extern "C" int cn_probe_mfma_f16(const unsigned short* A, const unsigned short* B, float* D, void* stream) {
  CN_LAUNCH(probe_mfma_f16_kernel, dim3(1), dim3(64), (hipStream_t)stream, A, B, D);
  return cn_check_launch("probe_mfma_f16");
}
extern "C" int cn_probe_mfma_f32(const float* A, const float* B, float* D, void* stream) {
  CN_LAUNCH(probe_mfma_f32_kernel, dim3(1), dim3(64), (hipStream_t)stream, A, B, D);
  return cn_check_launch("probe_mfma_f32");
}
extern "C" int cn_probe_tr16(const unsigned short* src, unsigned short* out, void* stream) {
  CN_LAUNCH(probe_tr16_kernel, dim3(1), dim3(64), (hipStream_t)stream, src, out);
  return cn_check_launch("probe_tr16");
}
