// qconv_i8.hip -- true int8 MFMA forward convolution for BASELINE config 5 (ResNet {'quantize': True}), gfx950.
//
// The reference (models/modules/quantize.py:185-220) only *simulates* 8-bit arithmetic: QConv2d snaps the
// activation to 256 levels (one zero point / scale per tensor) and each filter to 256 levels (zero point /
// scale per output channel), then runs a float convolution on the dequantised values.  Because both operands
// live on integer grids, that convolution is an integer GEMM plus a rank-1 style correction, evaluated here
// on the int8 matrix cores (v_mfma_i32_32x32x32_i8, exact int32 accumulation):
//
//   x = sx * a + zx',  w[k] = sw[k] * b + zw'[k]      a, b = level - 128 in [-128, 127]  (zx' = zx + 128 sx, ...)
//   y[p,k] = sum_{valid taps t, c} x * w
//          = sx sw[k] * ACC[p,k]  +  sx zw'[k] * A[p]  +  zx' sw[k] * B[cls(p)][k]  +  zx' zw'[k] * n_valid(p)
//   ACC = sum a*b (the int8 GEMM),  A[p] = sum of a over the valid window of pixel p (channel sums, then a
//   window sum),  B[cls][k] = sum of b over the taps that are valid for border class cls (zero padding
//   contributes x = 0, NOT the zero point, so border pixels see fewer taps: one table row per distinct
//   (valid rows, valid columns) pattern).
//
// The result equals the reference's float convolution up to fp32 rounding (it is the more exact of the two:
// nothing is rounded before the final combination).  Only the forward product has this form: the data
// gradient reduces over output channels, along which the per-channel filter scale varies, and the weight
// gradient uses the full-precision dy (quantize.py:115-121) -- both stay on the float MFMA kernels.
#include "cn_common.h"
#include "cn_api_internal.h"
#include <math.h>
#include <type_traits>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// Assumed gfx950 lane map of v_mfma_i32_32x32x32_i8 (wave64, lane l), pinned by cn_probe_mfma_i8:
//   A[i][k]: i = l & 31, k = 16 * (l >> 5) + e (e = 0..15, one byte each, 16 bytes per lane); B[k][j] likewise
//   with j = l & 31;  C/D as every other 32x32 MFMA: j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
#ifndef CN_EMULATE
__device__ __forceinline__ i32x16 cn_mfma_32x32x32_i8(u32x4 a, u32x4 b, i32x16 c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}
#else
static inline i32x16 cn_mfma_32x32x32_i8(u32x4 a, u32x4 b, i32x16 c) {
  struct P { u32x4 a, b; } mine{a, b};
  const void* const* all = cn_emul::wave_gather(&mine);
  const int l = cn_emul::lane(), j = l & 31;
  i32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    int acc = 0;
    for (int k = 0; k < 32; ++k) {
      const P* pa = (const P*)all[i + 32 * (k >> 4)];
      const P* pb = (const P*)all[j + 32 * (k >> 4)];
      const signed char va = ((const signed char*)&pa->a)[k & 15], vb = ((const signed char*)&pb->b)[k & 15];
      acc += (int)va * (int)vb;
    }
    d[r] += acc;
  }
  cn_emul::wave_release();
  return d;
}
#endif

#define QI_MAX_TAPS 64

struct QI8Params {
  const char* x;            // int8 NHWC levels - 128
  const char* w;            // int8 [Co][taps * Ci]
  char* y;                  // bf16 or fp32 NHWC
  const float* alpha;       // [Co]  sx * sw[k]
  const float* beta;        // [Co]  sx * zw'[k]
  const float* gamma;       // [ncls][Co]  zx' * (sw[k] * B[cls][k] + zw'[k] * n_valid(cls))
  const int* A;             // [M]  window sum of the activation levels of output pixel m
  const unsigned char* cls; // [M]  border class of output pixel m
  int N, Hi, Wi, Ci, P, Q, Co, stride_h, stride_w;
  int ntaps, cpt, nchunks, M, n_ntiles, n_mtiles, simple;
  unsigned int x_bytes, w_bytes, w_row;
  FastDiv div_pq, div_q, div_cpt;
  int tap_dhdw[QI_MAX_TAPS];
};

__device__ __forceinline__ int qi_slot(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// 256 threads, tile BN channels x BM pixels x 128 bytes (= 128 int8) of reduction per step; register-staged
// single LDS buffer (the short reductions of this path's 1x1 layers want occupancy, not depth), XOR-swizzled
// 16-byte slots, bounds-checked buffer loads (zero padding = a contribution of level 0 - handled by the
// correction terms: an out-of-image tap loads 0 bytes = a = 0, which is exactly "no contribution" to ACC).
template <int WC, int WP, int TI, int TJ, bool OUTF32>
__global__ __launch_bounds__(256) void qconv_i8_kernel(QI8Params p) {
  constexpr int BN = WC * TI * 32, BM = WP * TJ * 32, RS = 32, NPR = BM / RS, NWR = BN / RS;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int OEB = OUTF32 ? 4 : 2;
  constexpr int PITCH = BN * OEB + 16;
  constexpr int OUT_MAX = BM * PITCH;
  constexpr int MAIN = STAGE > OUT_MAX ? STAGE : OUT_MAX;
  __shared__ __attribute__((aligned(16))) char lds[MAIN + QI_MAX_TAPS * 8];
  int* s_taps = (int*)(lds + MAIN);   // per tap: {dhdw, x byte delta}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned int tile = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(tile % p.n_ntiles), mt = (int)(tile / p.n_ntiles);
  const int m0 = mt * BM, n0 = nt * BN;
  const int PQ = p.P * p.Q;
  if (tid < QI_MAX_TAPS) {
    const int t = tid < p.ntaps ? tid : 0;
    const int dhdw = p.tap_dhdw[t];
    const int dh = (int)(short)(dhdw & 0xffff), dw = dhdw >> 16;
    s_taps[2 * tid] = dhdw;
    s_taps[2 * tid + 1] = (dh * p.Wi + dw) * p.Ci;
  }
  const int cc = tid & 7, r0 = tid >> 3;
  const cn_buf_t xbuf = cn_make_buf(p.x, p.x_bytes);
  const cn_buf_t wbuf = cn_make_buf(p.w, p.w_bytes);
  int phin[NPR], pwin[NPR];
  unsigned int prow[NPR], wrow[NWR];
#pragma unroll
  for (int i = 0; i < NPR; ++i) {
    const int m = m0 + r0 + RS * i;
    const bool valid = m < p.M;
    const int mm = valid ? m : 0;
    const int n = (int)cn_fastdiv((unsigned)mm, p.div_pq);
    const int rem = mm - n * PQ;
    const int ph = (int)cn_fastdiv((unsigned)rem, p.div_q);
    const int pw = rem - ph * p.Q;
    phin[i] = valid ? ph * p.stride_h : -0x4000;
    pwin[i] = pw * p.stride_w;
    prow[i] = valid ? (unsigned int)(((n * p.Hi + ph * p.stride_h) * p.Wi + pw * p.stride_w) * p.Ci) : CN_OOB;
  }
#pragma unroll
  for (int i = 0; i < NWR; ++i) {
    const int co = n0 + r0 + RS * i;
    wrow[i] = co < p.Co ? (unsigned int)co * p.w_row : CN_OOB;
  }
  const int st0 = qi_slot(r0, cc);
  const int wc = wave % WC, wp = wave / WC;
  const int lrow = lane & 31, swz = (lrow >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = lrow * 128 + (((kk * 2 + (lane >> 5)) ^ swz) << 4);
  const int rd_w = wc * TI * 32 * 128;
  const int rd_p = BN * 128 + wp * TJ * 32 * 128;
  __syncthreads();

  i32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

  u32x4 preg[NPR], wreg[NWR];
  const int nkt = (p.nchunks + 7) >> 3;
  auto load_tile = [&](int kt) {
    const int kc = kt * 8 + cc;
    const bool kvalid = kc < p.nchunks;
    int tap = 0, cchunk = kc;
    if (p.ntaps > 1) {
      tap = kvalid ? (int)cn_fastdiv((unsigned)kc, p.div_cpt) : 0;
      cchunk = kc - tap * p.cpt;
    }
    const int dhdw = s_taps[2 * tap];
    const unsigned int kb = kvalid ? (unsigned int)(cchunk * 16) : CN_OOB;
    const unsigned int xofs = (unsigned int)s_taps[2 * tap + 1] + kb;
    const unsigned int wofs = (unsigned int)(tap * p.Ci) + kb;
    const int dh = (int)(short)(dhdw & 0xffff), dw = dhdw >> 16;
#pragma unroll
    for (int i = 0; i < NPR; ++i) {
      const bool ok = kb < CN_OOB && prow[i] < CN_OOB &&
                      (p.simple || ((unsigned)(phin[i] + dh) < (unsigned)p.Hi && (unsigned)(pwin[i] + dw) < (unsigned)p.Wi));
      preg[i] = cn_buf_ld16(xbuf, ok ? prow[i] + xofs : CN_OOB);
    }
#pragma unroll
    for (int i = 0; i < NWR; ++i) wreg[i] = cn_buf_ld16(wbuf, (wrow[i] | kb) >= CN_OOB ? CN_OOB : wrow[i] + wofs);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NWR; ++i) cn_st16(lds + st0 + i * RS * 128, wreg[i]);
#pragma unroll
    for (int i = 0; i < NPR; ++i) cn_st16(lds + st0 + BN * 128 + i * RS * 128, preg[i]);
  };
  if (nkt > 0) {
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4 af[TI], bfr[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) af[a] = cn_ld16(lds + rd_w + koff[kk] + a * 32 * 128);
#pragma unroll
        for (int b = 0; b < TJ; ++b) bfr[b] = cn_ld16(lds + rd_p + koff[kk] + b * 32 * 128);
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int b = 0; b < TJ; ++b) acc[a][b] = cn_mfma_32x32x32_i8(af[a], bfr[b], acc[a][b]);
      }
      __syncthreads();
      if (kt + 1 < nkt) {
        store_tile();
        __syncthreads();
      }
    }
  }
  // ---- epilogue: dequantise + corrections -> LDS out tile [BM pixels][BN channels] -> coalesced store
#pragma unroll
  for (int b = 0; b < TJ; ++b) {
    const int prow_l = (wp * TJ + b) * 32 + (lane & 31);
    const int m = m0 + prow_l;
    const float Am = m < p.M ? (float)p.A[m] : 0.f;
    const int cl = m < p.M ? (int)p.cls[m] : 0;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl0 = (wc * TI + a) * 32 + 8 * q + 4 * (lane >> 5);   // channel within the tile
        const int c = n0 + cl0;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < p.Co) {
          const f32x4 al = *(const f32x4*)(p.alpha + c), be = *(const f32x4*)(p.beta + c);
          const f32x4 ga = *(const f32x4*)(p.gamma + (size_t)cl * p.Co + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(al[e], (float)acc[a][b][q * 4 + e], fmaf(be[e], Am, ga[e]));
        }
        char* dst = lds + prow_l * PITCH + cl0 * OEB;
        if (OUTF32) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = v[e];
          *(f32x4*)dst = o;
        } else {
          u32x2 pk;
          pk[0] = cn_pack_bf16x2(v[0], v[1]);
          pk[1] = cn_pack_bf16x2(v[2], v[3]);
          *(u32x2*)dst = pk;
        }
      }
  }
  __syncthreads();
  constexpr int EPC = 16 / OEB, CPR = BN / EPC, NPASS = BM * CPR / 256;
  const int ecol = tid % CPR, erow0 = tid / CPR;
  const int c_first = n0 + ecol * EPC;
  if (c_first + EPC <= p.Co) {
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
      const int row = erow0 + k * (256 / CPR);
      const int m = m0 + row;
      if (m < p.M) cn_st16(p.y + ((size_t)m * p.Co + c_first) * OEB, cn_ld16(lds + row * PITCH + ecol * 16));
    }
  }
}

// ------------------------------------------------------------------------------------------------ pre-passes
// activation -> int8 levels - 128 (the same grid as cn_quantize: round_half_even(clamp((x - zp) / scale, 0, 255)))
template <typename T>
__global__ __launch_bounds__(256) void qi_levels_kernel(const T* x, signed char* q, long long n, const float* zero_point,
                                                       const float* range) {
  const float zp = zero_point[0];
  const float scale = (range[0] == 0.f ? 1.f : range[0]) / 255.f;
  const long long ngroups = n / 16;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long long)gridDim.x * 256) {
    union { signed char b[16]; u32x4 v; } out;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float t = (cn_load_elem<T>(x + g * 16 + e) + (-zp)) / scale;
      t = rintf(fminf(fmaxf(t, 0.f), 255.f));
      out.b[e] = (signed char)((int)t - 128);
    }
    cn_st16((char*)q + g * 16, out.v);
  }
}

// per input pixel: sum of its C levels (int32)
__global__ __launch_bounds__(256) void qi_chansum_kernel(const signed char* q, int* cs, long long npix, int C) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long long)gridDim.x * 256) {
    const signed char* row = q + pix * C;
    int s = 0;
    for (int c = 0; c < C; c += 16) {
      union { signed char b[16]; u32x4 v; } in;
      in.v = cn_ld16(row + c);
#pragma unroll
      for (int e = 0; e < 16; ++e) s += (int)in.b[e];
    }
    cs[pix] = s;
  }
}

// per output pixel: window sum of the channel sums over the taps inside the image, and its border class
// (row class of output row * ncolcls + column class of output column; classes enumerated by the host)
__global__ __launch_bounds__(256) void qi_window_kernel(const int* cs, int* A, unsigned char* cls, int N, int Hi, int Wi,
                                                       int P, int Q, int R, int S, int sh, int sw, int ph, int pw,
                                                       const unsigned char* rowcls, const unsigned char* colcls,
                                                       int ncolcls) {
  const long long total = (long long)N * P * Q;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < total; m += (long long)gridDim.x * 256) {
    const int q = (int)(m % Q);
    const long long t = m / Q;
    const int pr = (int)(t % P), n = (int)(t / P);
    int s = 0;
    for (int r = 0; r < R; ++r) {
      const int h = pr * sh - ph + r;
      if ((unsigned)h >= (unsigned)Hi) continue;
      for (int c = 0; c < S; ++c) {
        const int w = q * sw - pw + c;
        if ((unsigned)w >= (unsigned)Wi) continue;
        s += cs[((long long)n * Hi + h) * Wi + w];
      }
    }
    A[m] = s;
    cls[m] = (unsigned char)(rowcls[pr] * ncolcls + colcls[q]);
  }
}

// filter row k (fp32 master [taps][C]) -> int8 levels - 128, wsum[k][tap] = sum_c level - 128, wpar[k] = {sw, zw'}
__global__ __launch_bounds__(256) void qi_weight_kernel(const float* w, signed char* q, int* wsum, float* wpar, int taps,
                                                       int C) {
  __shared__ float red[8];
  __shared__ int isum[256];
  const int k = blockIdx.x, tid = threadIdx.x, J = taps * C;
  const float* row = w + (size_t)k * J;
  float mn = INFINITY, mx = -INFINITY;
  for (int e = tid; e < J; e += 256) { mn = fminf(mn, row[e]); mx = fmaxf(mx, row[e]); }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, cn_shfl_xor(mn, m)); mx = fmaxf(mx, cn_shfl_xor(mx, m)); }
  if ((tid & 63) == 0) { red[tid >> 6] = mn; red[4 + (tid >> 6)] = mx; }
  __syncthreads();
  mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  float range = mx - mn;
  if (range == 0.f) range = 1.f;
  const float scale = range / 255.f;
  for (int t = 0; t < taps; ++t) {
    int s = 0;
    for (int c = tid; c < C; c += 256) {
      float v = (row[t * C + c] + (-mn)) / scale;
      v = rintf(fminf(fmaxf(v, 0.f), 255.f));
      const int lv = (int)v - 128;
      q[(size_t)k * J + t * C + c] = (signed char)lv;
      s += lv;
    }
    isum[tid] = s;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int i = 0; i < 256; ++i) tot += isum[i];
      wsum[(size_t)k * taps + t] = tot;
    }
    __syncthreads();
  }
  if (tid == 0) {
    wpar[2 * k] = scale;
    wpar[2 * k + 1] = mn + 128.f * scale;
  }
}

// alpha / beta / gamma tables from the activation's device-side (zero point, range) and the filter tables.
// clsmask[cls][tap] = 1 when tap is inside the image for border class cls.
__global__ __launch_bounds__(256) void qi_tables_kernel(const float* zero_point, const float* range, const float* wpar,
                                                       const int* wsum, const unsigned char* clsmask, int ncls, int taps,
                                                       int C, int Co, float* alpha, float* beta, float* gamma) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= Co) return;
  const float sx = (range[0] == 0.f ? 1.f : range[0]) / 255.f;
  const float zx = zero_point[0] + 128.f * sx;
  const float sw = wpar[2 * k], zw = wpar[2 * k + 1];
  alpha[k] = sx * sw;
  beta[k] = sx * zw;
  for (int c = 0; c < ncls; ++c) {
    long long B = 0;
    int nv = 0;
    for (int t = 0; t < taps; ++t)
      if (clsmask[c * taps + t]) { B += wsum[(size_t)k * taps + t]; ++nv; }
    gamma[(size_t)c * Co + k] = zx * (sw * (float)B + zw * (float)(nv * C));
  }
}

static unsigned qi_grid(long long items) {
  long long nb = (items + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

// Activation (NHWC, dtype) -> int8 levels - 128 plus the per-output-pixel window sums / border classes of a
// convolution with the given geometry.  chansum: scratch int32 [N*H*W].  rowcls[P] / colcls[Q]: host-enumerated
// class ids of every output row / column (DEVICE arrays).
extern "C" int cn_i8_prepare_activation(const void* x, signed char* q, int* chansum, int* A, unsigned char* cls, int N,
                                        int H, int W, int C, int R, int S, int stride_h, int stride_w, int pad_h,
                                        int pad_w, int dtype, const float* zero_point, const float* range,
                                        const unsigned char* rowcls, const unsigned char* colcls, int ncolcls,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (x == nullptr || q == nullptr || chansum == nullptr || A == nullptr || cls == nullptr || zero_point == nullptr ||
      range == nullptr || rowcls == nullptr || colcls == nullptr) { cn_set_error("i8_prepare_activation: null operand"); return CN_EINVAL; }
  if (C % 16 != 0 || N <= 0 || H <= 0 || W <= 0) { cn_set_error("i8_prepare_activation: C=%d must be a multiple of 16", C); return CN_ESHAPE; }
  const int P = (H + 2 * pad_h - R) / stride_h + 1, Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0) { cn_set_error("i8_prepare_activation: empty output"); return CN_ESHAPE; }
  const long long npix = (long long)N * H * W, n = npix * C;
  if (dtype == CN_BF16)
    CN_LAUNCH(qi_levels_kernel<bf16_t>, dim3(qi_grid(n / 16)), dim3(256), stream, (const bf16_t*)x, q, n, zero_point, range);
  else if (dtype == CN_F32)
    CN_LAUNCH(qi_levels_kernel<float>, dim3(qi_grid(n / 16)), dim3(256), stream, (const float*)x, q, n, zero_point, range);
  else { cn_set_error("i8_prepare_activation: bad dtype"); return CN_EINVAL; }
  CN_LAUNCH(qi_chansum_kernel, dim3(qi_grid(npix)), dim3(256), stream, (const signed char*)q, chansum, npix, C);
  CN_LAUNCH(qi_window_kernel, dim3(qi_grid((long long)N * P * Q)), dim3(256), stream, (const int*)chansum, A, cls, N, H, W,
            P, Q, R, S, stride_h, stride_w, pad_h, pad_w, rowcls, colcls, ncolcls);
  return cn_check_launch("i8_prepare_activation");
}

// fp32 master filter [K][taps][C] -> int8 levels - 128 (same memory order), wsum[K][taps], wpar[K][2] = {scale, zero'}
extern "C" int cn_i8_prepare_weight(const float* w_master, signed char* q, int* wsum, float* wpar, int K, int taps, int C,
                                    void* stream) {
  if (w_master == nullptr || q == nullptr || wsum == nullptr || wpar == nullptr || K <= 0 || taps <= 0 || C <= 0) {
    cn_set_error("i8_prepare_weight: bad arguments");
    return CN_EINVAL;
  }
  CN_LAUNCH(qi_weight_kernel, dim3((unsigned)K), dim3(256), (hipStream_t)stream, w_master, q, wsum, wpar, taps, C);
  return cn_check_launch("i8_prepare_weight");
}

// y[N,P,Q,K] (dtype, or fp32 when out_f32) = the reference's QConv2d forward product, evaluated on the int8
// matrix cores.  tables: scratch of (2 + ncls) * K floats (alpha | beta | gamma).
extern "C" int cn_conv2d_fwd_i8(const signed char* xq, const signed char* wq, void* y, const int* A,
                                const unsigned char* cls, const float* zero_point, const float* range, const float* wpar,
                                const int* wsum, const unsigned char* clsmask, int ncls, float* tables, int N, int H, int W,
                                int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int out_dtype,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int P = (H + 2 * pad_h - R) / stride_h + 1, Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0 || N <= 0) { cn_set_error("conv2d_fwd_i8: empty output"); return CN_ESHAPE; }
  if (C % 16 != 0 || K % 8 != 0 || R * S > QI_MAX_TAPS) { cn_set_error("conv2d_fwd_i8: C=%d (x16), K=%d (x8), taps=%d (<=%d)", C, K, R * S, QI_MAX_TAPS); return CN_ESHAPE; }
  if (xq == nullptr || wq == nullptr || y == nullptr || A == nullptr || cls == nullptr || tables == nullptr ||
      wpar == nullptr || wsum == nullptr || clsmask == nullptr || ncls <= 0 || ncls > 255) { cn_set_error("conv2d_fwd_i8: bad operand"); return CN_EINVAL; }
  if (out_dtype != CN_BF16 && out_dtype != CN_F32) { cn_set_error("conv2d_fwd_i8: bad output dtype"); return CN_EINVAL; }
  const long long xb = (long long)N * H * W * C, wb = (long long)K * R * S * C;
  if (xb >= (1ll << 31) || wb >= (1ll << 31)) { cn_set_error("conv2d_fwd_i8: operand exceeds the 2 GiB buffer window"); return CN_ESHAPE; }
  float* alpha = tables;
  float* beta = tables + K;
  float* gamma = tables + 2 * K;
  CN_LAUNCH(qi_tables_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), stream, zero_point, range, wpar, wsum, clsmask,
            ncls, R * S, C, K, alpha, beta, gamma);
  QI8Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)xq; p.w = (const char*)wq; p.y = (char*)y;
  p.alpha = alpha; p.beta = beta; p.gamma = gamma; p.A = A; p.cls = cls;
  p.N = N; p.Hi = H; p.Wi = W; p.Ci = C; p.P = P; p.Q = Q; p.Co = K; p.stride_h = stride_h; p.stride_w = stride_w;
  p.ntaps = R * S; p.cpt = C / 16; p.nchunks = p.ntaps * p.cpt; p.M = N * P * Q;
  p.x_bytes = (unsigned int)xb; p.w_bytes = (unsigned int)wb; p.w_row = (unsigned int)(R * S * C);
  p.div_pq = cn_make_fastdiv((unsigned)(P * Q)); p.div_q = cn_make_fastdiv((unsigned)Q); p.div_cpt = cn_make_fastdiv((unsigned)p.cpt);
  p.simple = (pad_h == 0 && pad_w == 0) ? 1 : 0;
  for (int r = 0; r < R; ++r)
    for (int s = 0; s < S; ++s) p.tap_dhdw[r * S + s] = ((r - pad_h) & 0xffff) | ((s - pad_w) << 16);
  const bool narrow = K <= 64;
  const int BM = 128, BN = narrow ? 64 : 128;
  p.n_ntiles = (K + BN - 1) / BN;
  p.n_mtiles = (p.M + BM - 1) / BM;
  dim3 grid((unsigned)(p.n_ntiles * p.n_mtiles));
  const bool f32 = out_dtype == CN_F32;
  cn_set_last_kernel("qconv_i8_kernel<%s, %s>", narrow ? "1, 4, 2, 1" : "2, 2, 2, 2", f32 ? "true" : "false");
  if (narrow) {
    if (f32) CN_LAUNCH((qconv_i8_kernel<1, 4, 2, 1, true>), grid, dim3(256), stream, p);
    else CN_LAUNCH((qconv_i8_kernel<1, 4, 2, 1, false>), grid, dim3(256), stream, p);
  } else {
    if (f32) CN_LAUNCH((qconv_i8_kernel<2, 2, 2, 2, true>), grid, dim3(256), stream, p);
    else CN_LAUNCH((qconv_i8_kernel<2, 2, 2, 2, false>), grid, dim3(256), stream, p);
  }
  return cn_check_launch("conv2d_fwd_i8");
}

// ---- hardware lane-map probe (tests only): D[32][32] = A[32][32(k)] * B[32(k)][32], int8 operands
__global__ void qi_probe_kernel(const signed char* A, const signed char* B, int* D) {
  const int l = threadIdx.x;
  union { signed char b[16]; u32x4 v; } a, b;
  for (int e = 0; e < 16; ++e) {
    const int k = 16 * (l >> 5) + e;
    a.b[e] = A[(l & 31) * 32 + k];
    b.b[e] = B[k * 32 + (l & 31)];
  }
  i32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = cn_mfma_32x32x32_i8(a.v, b.v, c);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
extern "C" int cn_probe_mfma_i8(const signed char* A, const signed char* B, int* D, void* stream) {
  CN_LAUNCH(qi_probe_kernel, dim3(1), dim3(64), (hipStream_t)stream, A, B, D);
  return cn_check_launch("probe_mfma_i8");
}
