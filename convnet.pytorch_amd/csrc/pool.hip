// pool.hip -- MaxPool2d, global average pool, layout conversion and small elementwise helpers
// (NHWC, 16-byte channel chunks per lane, gfx950).
//
// Replaces nn.MaxPool2d(3, 2, 1) (/root/reference models/resnet.py:230), nn.AdaptiveAvgPool2d(1)
// (models/resnet.py:241), the `inputs.to(device, dtype)` boundary (trainer.py:116-117) and the
// autograd fan-in add of the residual blocks (models/resnet.py:115,162).
// Max-pool semantics follow ATen's CPU kernel the oracle runs: padding is -inf, the FIRST maximum
// in (kh, kw) scan order wins (strict >), and backward routes the gradient to that element only.
// The forward stores the winning tap (uint8) so backward is a gather: deterministic, no atomics.
#include "cn_common.h"
#include "cn_api_internal.h"

// Both max-pool kernels: blockIdx.y walks image rows (n, output row), blockIdx.x * 256 + tid walks the
// (pixel, 16-byte channel chunk) pairs of one row - all index math is 32-bit with one host-precomputed
// fast division (the first version spent three 64-bit divisions per thread and ran at 2.3 TB/s).
// AFF: the input is a pre-BatchNorm tensor and every tap is first mapped through relu(x*scale[c] + shift[c])
// - the stem's BatchNorm apply + ReLU + max-pool (models/resnet.py:228-230) in one pass, so the
// normalised 112x112 map is never written to or re-read from HBM.
template <typename T, bool AFF, int KS>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const char* x, char* y, unsigned char* idx, int N,
                                                         int H, int W, int C, int P, int Q, int k, int st,
                                                         int pad, FastDiv div_cpr, const float* scale,
                                                         const float* shift, char* xmax) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int cpr = C / CH;
  const int idr = blockIdx.x * 256 + threadIdx.x;
  if (idr >= Q * cpr) return;
  const int q = (int)cn_fastdiv((unsigned)idr, div_cpr);
  const int col = idr - q * cpr;
  const int w0 = q * st - pad;
  float sc[CH], sh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { sc[e] = AFF ? scale[col * CH + e] : 1.f; sh[e] = AFF ? shift[col * CH + e] : 0.f; }
  for (int row = blockIdx.y; row < N * P; row += gridDim.y) {
    const int n = row / P, pp = row - n * P;
    float best[CH], braw[CH];   // braw (AFF): the pre-BatchNorm value at the winning tap, for the backward sums
    int bi[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { best[e] = -INFINITY; bi[e] = 0; braw[e] = 0.f; }
    bool first = true;
    auto take = [&](const u32x4& raw, int t) {
      float f[CH], r0[CH];
      Chunk<T>::unpack(raw, f);
      if (AFF) {
#pragma unroll
        for (int e = 0; e < CH; ++e) r0[e] = f[e];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          const float v = fmaf(f[e], sc[e], sh[e]);
          f[e] = v > 0.f ? v : 0.f;
        }
        Chunk<T>::unpack(Chunk<T>::pack(f), f);   // compare what the unfused chain would have stored (rounded z)
      }
#pragma unroll
      for (int e = 0; e < CH; ++e)
        if (first || f[e] > best[e]) {
          best[e] = f[e];
          bi[e] = t;
          if (AFF) braw[e] = r0[e];
        }
      first = false;
    };
    if (KS > 0) {   // compile-time window: all KS*KS taps are requested before the first comparison
      constexpr int NT_ = KS > 0 ? KS * KS : 1;
      u32x4 raw[NT_];
      bool ok[NT_];
#pragma unroll
      for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int h = pp * st - pad + r, w = w0 + s;
          ok[r * KS + s] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
          const size_t o = ok[r * KS + s] ? (((size_t)(n * H + h) * W + w) * C + (size_t)col * CH) * EB : 0;
          raw[r * KS + s] = cn_ld16(x + o);
        }
#pragma unroll
      for (int t = 0; t < KS * KS; ++t)
        if (ok[t]) take(raw[t], t);
    } else {
      for (int r = 0; r < k; ++r) {
        const int h = pp * st - pad + r;
        if ((unsigned)h >= (unsigned)H) continue;
        const char* xr = x + ((size_t)(n * H + h) * W * C + (size_t)col * CH) * EB;
        for (int s = 0; s < k; ++s) {
          const int w = w0 + s;
          if ((unsigned)w >= (unsigned)W) continue;
          take(cn_ld16(xr + (size_t)w * C * EB), r * k + s);
        }
      }
    }
    const size_t o = ((size_t)row * Q + q) * C + (size_t)col * CH;
    cn_st16(y + o * EB, Chunk<T>::pack(best));
    if (AFF && xmax != nullptr) cn_st16(xmax + o * EB, Chunk<T>::pack(braw));   // exact: the values were stored as T
    if (CH == 8) {   // one 8-byte store of the chunk's winning taps
      unsigned long long pk = 0;
#pragma unroll
      for (int e = 0; e < CH; ++e) pk |= (unsigned long long)(unsigned char)bi[e] << (8 * e);
      *(unsigned long long*)(idx + o) = pk;
    } else {
      unsigned int pk = 0;
#pragma unroll
      for (int e = 0; e < CH; ++e) pk |= (unsigned int)(unsigned char)bi[e] << (8 * e);
      *(unsigned int*)(idx + o) = pk;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const char* dy, const unsigned char* idx, char* dx,
                                                         int N, int H, int W, int C, int P, int Q, int k,
                                                         int st, int pad, FastDiv div_cpr) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int cpr = C / CH;
  const int idr = blockIdx.x * 256 + threadIdx.x;
  if (idr >= W * cpr) return;
  const int w = (int)cn_fastdiv((unsigned)idr, div_cpr);
  const int col = idr - w * cpr;
  // windows q with q*st - pad <= w <= q*st - pad + k - 1
  int q_lo = w + pad - k + 1;
  q_lo = q_lo > 0 ? (q_lo + st - 1) / st : 0;
  int q_hi = (w + pad) / st;
  if (q_hi > Q - 1) q_hi = Q - 1;
  for (int row = blockIdx.y; row < N * H; row += gridDim.y) {
    const int n = row / H, h = row - n * H;
    float acc[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = 0.f;
    int p_lo = h + pad - k + 1;
    p_lo = p_lo > 0 ? (p_lo + st - 1) / st : 0;
    int p_hi = (h + pad) / st;
    if (p_hi > P - 1) p_hi = P - 1;
    for (int pp = p_lo; pp <= p_hi; ++pp)
      for (int q = q_lo; q <= q_hi; ++q) {
        const int t = (h - (pp * st - pad)) * k + (w - (q * st - pad));
        const size_t o = ((size_t)(n * P + pp) * Q + q) * C + (size_t)col * CH;
        float g[CH];
        Chunk<T>::unpack(cn_ld16(dy + o * EB), g);
        unsigned long long pk;
        if (CH == 8) pk = *(const unsigned long long*)(idx + o);
        else pk = *(const unsigned int*)(idx + o);
#pragma unroll
        for (int e = 0; e < CH; ++e)
          if ((int)((pk >> (8 * e)) & 0xffull) == t) acc[e] += g[e];
      }
    cn_st16_stream(dx + (((size_t)row * W + w) * C + (size_t)col * CH) * EB, Chunk<T>::pack(acc));
  }
}

// out[n][c] = mean over HW of x[n][hw][c]
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const char* x, char* y, int N, int HW, int C) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int cpr = C / CH;
  const int total = N * cpr;
  const float inv = 1.f / (float)HW;
  for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
    const int col = id % cpr, n = id / cpr;
    float acc[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = 0.f;
    for (int i = 0; i < HW; ++i) {
      float f[CH];
      Chunk<T>::unpack(cn_ld16(x + (((size_t)n * HW + i) * C + (size_t)col * CH) * EB), f);
#pragma unroll
      for (int e = 0; e < CH; ++e) acc[e] += f[e];
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] *= inv;
    cn_st16(y + ((size_t)n * C + (size_t)col * CH) * EB, Chunk<T>::pack(acc));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const char* dy, char* dx, int N, int HW, int C) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int cpr = C / CH;
  const long long total = (long long)N * HW * cpr;
  const float inv = 1.f / (float)HW;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int col = (int)(id % cpr);
    const long long pix = id / cpr;
    const int n = (int)(pix / HW);
    float g[CH];
    Chunk<T>::unpack(cn_ld16(dy + ((size_t)n * C + (size_t)col * CH) * EB), g);
#pragma unroll
    for (int e = 0; e < CH; ++e) g[e] *= inv;
    cn_st16(dx + ((size_t)pix * C + (size_t)col * CH) * EB, Chunk<T>::pack(g));
  }
}

// NCHW fp32 (host/loader layout, reference trainer.py:116-117) -> NHWC T with zero channel padding.
// Lanes run along pixels so each plane read is coalesced; one 16-byte store per (pixel, chunk).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, char* y, int N, int C, int HW,
                                                          int Cpad) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int cpr = Cpad / CH;
  const long long total = (long long)N * HW * cpr;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int hw = (int)(id % HW);
    const long long rest = id / HW;
    const int col = (int)(rest % cpr);
    const int n = (int)(rest / cpr);
    float f[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      const int c = col * CH + e;
      f[e] = c < C ? x[((size_t)n * C + c) * HW + hw] : 0.f;
    }
    cn_st16(y + (((size_t)n * HW + hw) * Cpad + (size_t)col * CH) * EB, Chunk<T>::pack(f));
  }
}

// fp32 NCHW with C <= 4 channels -> zero-padded "pixel pair" image for a stride-2 stem convolution
// (models/resnet.py:226: 7x7/2, pad 3, C = 3): y[n][hp][jp][8] (bf16) holds the two horizontally adjacent
// padded pixels wp = 2*jp, 2*jp + 1 with 4 channels each (source pixel (hp - pad_h, wp - pad_w), zero outside
// the image and for channels >= C).  One 16-byte chunk per pixel pair: the implicit GEMM then needs
// ceil(S/2) chunks per filter row instead of S, and no bounds tests (the padding is in the data).
__global__ __launch_bounds__(256) void nchw_to_pairs_kernel(const float* x, char* y, int N, int C, int H, int W,
                                                           int pad_h, int pad_w, int Hp, int Jp) {
  const long long total = (long long)N * Hp * Jp;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int jp = (int)(id % Jp);
    const long long rest = id / Jp;
    const int hp = (int)(rest % Hp);
    const int n = (int)(rest / Hp);
    const int h = hp - pad_h;
    float f[8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int w = 2 * jp + j - pad_w;
      const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        f[j * 4 + c] = (in && c < C) ? x[(((size_t)n * C + c) * H + h) * W + w] : 0.f;
    }
    cn_st16_stream(y + (size_t)id * 16, Chunk<bf16_t>::pack(f));
  }
}

// NHWC T -> NCHW fp32 (only used to hand feature maps back in the reference layout)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const char* x, float* y, int N, int C, int HW,
                                                          int Cpad) {
  const long long total = (long long)N * C * HW;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int hw = (int)(id % HW);
    const long long rest = id / HW;
    const int c = (int)(rest % C);
    const int n = (int)(rest / C);
    y[id] = cn_load_elem<T>((const T*)x + ((size_t)n * HW + hw) * Cpad + c);
  }
}

// a += b (gradient fan-in of a residual fork), a = relu(x), da = dz * (z > 0)
template <typename T, int OP>
__global__ __launch_bounds__(256) void eltwise_kernel(char* a, const char* b, const char* c, long long nchunks) {
  constexpr int CH = ElemTraits<T>::kChunk;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < nchunks; id += (long long)gridDim.x * 256) {
    float fa[CH], fb[CH];
    if (OP == 0) {  // a += b
      Chunk<T>::unpack(cn_ld16(a + id * 16), fa);
      Chunk<T>::unpack(cn_ld16(b + id * 16), fb);
#pragma unroll
      for (int e = 0; e < CH; ++e) fa[e] += fb[e];
    } else if (OP == 1) {  // a = relu(b)
      Chunk<T>::unpack(cn_ld16(b + id * 16), fb);
#pragma unroll
      for (int e = 0; e < CH; ++e) fa[e] = fb[e] > 0.f ? fb[e] : 0.f;
    } else if (OP == 2) {  // a = b * (c > 0)
      float fc[CH];
      Chunk<T>::unpack(cn_ld16(b + id * 16), fb);
      Chunk<T>::unpack(cn_ld16(c + id * 16), fc);
#pragma unroll
      for (int e = 0; e < CH; ++e) fa[e] = fc[e] > 0.f ? fb[e] : 0.f;
    } else if (OP == 4) {  // a = relu(b + c)   (residual junction of the unfused quantised blocks)
      float fc[CH];
      Chunk<T>::unpack(cn_ld16(b + id * 16), fb);
      Chunk<T>::unpack(cn_ld16(c + id * 16), fc);
#pragma unroll
      for (int e = 0; e < CH; ++e) { const float v = fb[e] + fc[e]; fa[e] = v > 0.f ? v : 0.f; }
    } else {  // a = b * c   (dropout: c is the pre-scaled keep mask)
      float fc[CH];
      Chunk<T>::unpack(cn_ld16(b + id * 16), fb);
      Chunk<T>::unpack(cn_ld16(c + id * 16), fc);
#pragma unroll
      for (int e = 0; e < CH; ++e) fa[e] = fb[e] * fc[e];
    }
    cn_st16(a + id * 16, Chunk<T>::pack(fa));
  }
}

// ------------------------------------------------------------------------------------------------
static unsigned pool_grid(long long total) {
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}
static int pool_check(const char* who, int C, int dtype) {
  if (!cn_dtype_ok(dtype)) { cn_set_error("%s: bad dtype %d", who, dtype); return CN_EINVAL; }
  const int CH = cn_dtype_chunk(dtype);
  if (C <= 0 || C % CH != 0) { cn_set_error("%s: C=%d must be a multiple of %d", who, C, CH); return CN_ESHAPE; }
  return CN_OK;
}
#define POOL_DISPATCH(kern, grid, stream, ...)                                          \
  do {                                                                                  \
    if (dtype == CN_BF16) CN_LAUNCH(kern<bf16_t>, grid, dim3(256), stream, __VA_ARGS__); \
    else if (dtype == CN_F16) CN_LAUNCH(kern<f16_t>, grid, dim3(256), stream, __VA_ARGS__); \
    else CN_LAUNCH(kern<float>, grid, dim3(256), stream, __VA_ARGS__);                   \
  } while (0)

extern "C" int cn_maxpool_fwd(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int k,
                              int stride, int pad, int dtype, void* stream) {
  int rc = pool_check("maxpool_fwd", C, dtype);
  if (rc) return rc;
  if (k * k > 255 || pad * 2 > k) { cn_set_error("maxpool_fwd: unsupported window"); return CN_ESHAPE; }
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const int CH = cn_dtype_chunk(dtype);
  const long long rows_f = (long long)N * P;
  dim3 grid((unsigned)((Q * (C / CH) + 255) / 256), (unsigned)(rows_f < 65535 ? rows_f : 65535));
  const FastDiv div_cpr = cn_make_fastdiv((unsigned)(C / CH));
  void* xmax_ = nullptr;
#define MP_GO(T, AFF, KS, SC, SH)                                                                               \
  CN_LAUNCH((maxpool_fwd_kernel<T, AFF, KS>), grid, dim3(256), (hipStream_t)stream, (const char*)x, (char*)y, idx, N, H, \
            W, C, P, Q, k, stride, pad, div_cpr, SC, SH, (char*)xmax_)
  const float* none = nullptr;
  if (dtype == CN_BF16) { if (k == 3) MP_GO(bf16_t, false, 3, none, none); else MP_GO(bf16_t, false, 0, none, none); }
  else if (dtype == CN_F16) { if (k == 3) MP_GO(f16_t, false, 3, none, none); else MP_GO(f16_t, false, 0, none, none); }
  else { if (k == 3) MP_GO(float, false, 3, none, none); else MP_GO(float, false, 0, none, none); }
  return cn_check_launch("maxpool_fwd");
}

// y = maxpool(relu(x*scale + shift)): x is the pre-BatchNorm tensor, scale/shift the per-channel
// coefficients cn_bn_fwd_train* wrote (stats_out + 2C / + 3C).  The pooled values are rounded to the
// compute dtype exactly like the unfused chain rounds z before pooling it.
static int maxpool_fwd_bnrelu_impl(const void* x, const float* scale, const float* shift, void* y,
                                   unsigned char* idx, void* xmax_, int N, int H, int W, int C, int k, int stride,
                                   int pad, int dtype, void* stream) {
  int rc = pool_check("maxpool_fwd_bnrelu", C, dtype);
  if (rc) return rc;
  if (k * k > 255 || pad * 2 > k || scale == nullptr || shift == nullptr) {
    cn_set_error("maxpool_fwd_bnrelu: unsupported window or missing coefficients");
    return CN_ESHAPE;
  }
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const int CH = cn_dtype_chunk(dtype);
  const long long rows_f = (long long)N * P;
  dim3 grid((unsigned)((Q * (C / CH) + 255) / 256), (unsigned)(rows_f < 65535 ? rows_f : 65535));
  const FastDiv div_cpr = cn_make_fastdiv((unsigned)(C / CH));
  if (dtype == CN_BF16) { if (k == 3) MP_GO(bf16_t, true, 3, scale, shift); else MP_GO(bf16_t, true, 0, scale, shift); }
  else if (dtype == CN_F16) { if (k == 3) MP_GO(f16_t, true, 3, scale, shift); else MP_GO(f16_t, true, 0, scale, shift); }
  else { if (k == 3) MP_GO(float, true, 3, scale, shift); else MP_GO(float, true, 0, scale, shift); }
#undef MP_GO
  return cn_check_launch("maxpool_fwd_bnrelu");
}

extern "C" int cn_maxpool_fwd_bnrelu(const void* x, const float* scale, const float* shift, void* y,
                                     unsigned char* idx, int N, int H, int W, int C, int k, int stride, int pad,
                                     int dtype, void* stream) {
  return maxpool_fwd_bnrelu_impl(x, scale, shift, y, idx, nullptr, N, H, W, C, k, stride, pad, dtype, stream);
}

// The same pass, additionally storing the pre-BatchNorm value of every winning tap (xmax, shaped like y): with it the
// BatchNorm-backward sums of the fused stem are taken over the pooled map (cn_bn_bwd_maxpool_xmax) instead of the
// 4x larger input map with the pool's gather per pixel.
extern "C" int cn_maxpool_fwd_bnrelu_xmax(const void* x, const float* scale, const float* shift, void* y,
                                          unsigned char* idx, void* xmax, int N, int H, int W, int C, int k,
                                          int stride, int pad, int dtype, void* stream) {
  if (xmax == nullptr) { cn_set_error("maxpool_fwd_bnrelu_xmax: no xmax buffer"); return CN_EINVAL; }
  return maxpool_fwd_bnrelu_impl(x, scale, shift, y, idx, xmax, N, H, W, C, k, stride, pad, dtype, stream);
}

extern "C" int cn_maxpool_bwd(const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C,
                              int k, int stride, int pad, int dtype, void* stream) {
  int rc = pool_check("maxpool_bwd", C, dtype);
  if (rc) return rc;
  const int P = (H + 2 * pad - k) / stride + 1, Q = (W + 2 * pad - k) / stride + 1;
  const int CH = cn_dtype_chunk(dtype);
  const long long rows_b = (long long)N * H;
  dim3 grid((unsigned)((W * (C / CH) + 255) / 256), (unsigned)(rows_b < 65535 ? rows_b : 65535));
  const FastDiv div_cpr = cn_make_fastdiv((unsigned)(C / CH));
  POOL_DISPATCH(maxpool_bwd_kernel, grid, (hipStream_t)stream, (const char*)dy, idx, (char*)dx, N, H, W, C, P, Q, k,
                stride, pad, div_cpr);
  return cn_check_launch("maxpool_bwd");
}

extern "C" int cn_avgpool_fwd(const void* x, void* y, int N, int HW, int C, int dtype, void* stream) {
  int rc = pool_check("avgpool_fwd", C, dtype);
  if (rc) return rc;
  const int CH = cn_dtype_chunk(dtype);
  dim3 grid(pool_grid((long long)N * (C / CH)));
  POOL_DISPATCH(avgpool_fwd_kernel, grid, (hipStream_t)stream, (const char*)x, (char*)y, N, HW, C);
  return cn_check_launch("avgpool_fwd");
}

extern "C" int cn_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype, void* stream) {
  int rc = pool_check("avgpool_bwd", C, dtype);
  if (rc) return rc;
  const int CH = cn_dtype_chunk(dtype);
  dim3 grid(pool_grid((long long)N * HW * (C / CH)));
  POOL_DISPATCH(avgpool_bwd_kernel, grid, (hipStream_t)stream, (const char*)dy, (char*)dx, N, HW, C);
  return cn_check_launch("avgpool_bwd");
}

// ToTensor + Normalize of the image pipeline on the device (preprocess.py:23-25: transforms.ToTensor(), Normalize(mean,
// std)): the loader ships the augmented crops as uint8 HWC (a quarter of the fp32 NCHW bytes over PCIe, and 0.6 ms of host
// arithmetic per image less), this pass writes the fp32 NCHW batch the step expects.  The value of a pixel depends on its
// byte and channel only: lut[c][u] = (float(u) / 255 - mean[c]) / std[c], computed by the CALLER with the reference's own
// operations (one division, one subtraction, one division in fp32) - the result is bit-identical by construction.
__global__ __launch_bounds__(256) void u8_nhwc_to_nchw_lut_kernel(const unsigned char* x, float* y, long long npix, int HW,
                                                                 int C, const float* lut) {
  __shared__ float s_lut[4 * 256];
  for (int i = threadIdx.x; i < C * 256; i += 256) s_lut[i] = lut[i];
  __syncthreads();
  const long long stride = (long long)gridDim.x * 256;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride) {
    const long long n = p / HW;
    const int hw = (int)(p - n * HW);
    const unsigned char* px = x + p * C;
    for (int c = 0; c < C; ++c) y[(n * C + c) * (long long)HW + hw] = s_lut[c * 256 + px[c]];
  }
}
extern "C" int cn_u8_nhwc_to_nchw_lut(const unsigned char* x_nhwc, float* y_nchw, int N, int H, int W, int C,
                                      const float* lut, void* stream) {
  if (x_nhwc == nullptr || y_nchw == nullptr || lut == nullptr) { cn_set_error("u8_nhwc_to_nchw_lut: null operand"); return CN_EINVAL; }
  if (N <= 0 || H <= 0 || W <= 0 || C < 1 || C > 4) { cn_set_error("u8_nhwc_to_nchw_lut: bad shape (C = %d, 1..4)", C); return CN_ESHAPE; }
  const long long npix = (long long)N * H * W;
  long long nb = (npix + 255) / 256;
  if (nb > 8192) nb = 8192;
  CN_LAUNCH(u8_nhwc_to_nchw_lut_kernel, dim3((unsigned)nb), dim3(256), (hipStream_t)stream, x_nhwc, y_nchw, npix, H * W, C, lut);
  return cn_check_launch("u8_nhwc_to_nchw_lut");
}

extern "C" int cn_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype,
                               void* stream) {
  int rc = pool_check("nchw_to_nhwc", Cpad, dtype);
  if (rc) return rc;
  if (Cpad < C) { cn_set_error("nchw_to_nhwc: Cpad < C"); return CN_ESHAPE; }
  const int CH = cn_dtype_chunk(dtype);
  dim3 grid(pool_grid((long long)N * H * W * (Cpad / CH)));
  POOL_DISPATCH(nchw_to_nhwc_kernel, grid, (hipStream_t)stream, x, (char*)y, N, C, H * W, Cpad);
  return cn_check_launch("nchw_to_nhwc");
}

extern "C" int cn_nchw_to_pairs(const float* x, void* y, int N, int C, int H, int W, int pad_h, int pad_w,
                                void* stream) {
  if (C < 1 || C > 4 || N <= 0 || H <= 0 || W <= 0 || pad_h < 0 || pad_w < 0 || ((W + 2 * pad_w) & 1)) {
    cn_set_error("nchw_to_pairs: need 1 <= C <= 4 and an even padded width (C=%d, W=%d, pad_w=%d)", C, W, pad_w);
    return CN_ESHAPE;
  }
  const int Hp = H + 2 * pad_h, Jp = (W + 2 * pad_w) / 2;
  dim3 grid(pool_grid((long long)N * Hp * Jp));
  CN_LAUNCH(nchw_to_pairs_kernel, grid, dim3(256), (hipStream_t)stream, x, (char*)y, N, C, H, W, pad_h, pad_w, Hp, Jp);
  return cn_check_launch("nchw_to_pairs");
}

extern "C" int cn_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int Cpad, int dtype,
                               void* stream) {
  if (!cn_dtype_ok(dtype)) { cn_set_error("nhwc_to_nchw: bad dtype"); return CN_EINVAL; }
  dim3 grid(pool_grid((long long)N * C * H * W));
  POOL_DISPATCH(nhwc_to_nchw_kernel, grid, (hipStream_t)stream, (const char*)x, y, N, C, H * W, Cpad);
  return cn_check_launch("nhwc_to_nchw");
}

// op: 0  a += b;  1  a = relu(b);  2  a = b * (c > 0);  3  a = b * c.   n = element count (multiple of the chunk).
extern "C" int cn_eltwise(int op, void* a, const void* b, const void* c, long long n, int dtype, void* stream) {
  if (!cn_dtype_ok(dtype)) { cn_set_error("eltwise: bad dtype"); return CN_EINVAL; }
  const int CH = cn_dtype_chunk(dtype);
  if (n % CH != 0) { cn_set_error("eltwise: n=%lld not a multiple of %d", n, CH); return CN_ESHAPE; }
  if (n == 0) return CN_OK;
  const long long nch = n / CH;
  dim3 grid(pool_grid(nch));
  hipStream_t s = (hipStream_t)stream;
#define ELT(T, OP) CN_LAUNCH((eltwise_kernel<T, OP>), grid, dim3(256), s, (char*)a, (const char*)b, (const char*)c, nch)
  if (dtype == CN_BF16) {
    if (op == 0) ELT(bf16_t, 0); else if (op == 1) ELT(bf16_t, 1); else if (op == 2) ELT(bf16_t, 2);
    else if (op == 3) ELT(bf16_t, 3); else if (op == 4) ELT(bf16_t, 4);
    else { cn_set_error("eltwise: bad op"); return CN_EINVAL; }
  } else if (dtype == CN_F16) {
    if (op == 0) ELT(f16_t, 0); else if (op == 1) ELT(f16_t, 1); else if (op == 2) ELT(f16_t, 2);
    else if (op == 3) ELT(f16_t, 3); else if (op == 4) ELT(f16_t, 4);
    else { cn_set_error("eltwise: bad op"); return CN_EINVAL; }
  } else {
    if (op == 0) ELT(float, 0); else if (op == 1) ELT(float, 1); else if (op == 2) ELT(float, 2);
    else if (op == 3) ELT(float, 3); else if (op == 4) ELT(float, 4);
    else { cn_set_error("eltwise: bad op"); return CN_EINVAL; }
  }
#undef ELT
  return cn_check_launch("eltwise");
}
