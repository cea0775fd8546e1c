// conv3x3.hip -- 3x3 / stride-1 / pad-1 convolution with 64 input and 64 output channels (the first stage's conv2 of a
// bottleneck ResNet, /root/reference models/resnet.py:126-132; forward and data gradient) as a halo kernel (round 3).
//
// Through the tiled implicit-GEMM kernel these layers re-gather every input pixel nine times from L2 into LDS (one K
// tile per tap, a workgroup barrier each) and run at a quarter of their roof (108 us for 206 MB of compulsory traffic and
// 59 GFLOP).  Here a persistent workgroup of four waves walks bands of 4 output rows of one image (two workgroups per
// CU: one stages its next band while the other multiplies):
//   * the band's 6 input rows (+ a zero column on either side) sit in LDS ONCE, 144 bytes per pixel (128 of data + 16:
//     sixteen consecutive pixels fall on sixteen different 16-byte bank groups);
//   * the MFMA pixel fragments of all nine taps are per-lane ds_read_b128 straight out of that halo (a tap is an offset);
//   * the 64 x 576 filter lives in registers: a wave owns 32 output channels (36 A-fragments; 2 channel tiles x 2 pixel
//     groups per workgroup); the pixel fragments of tap t + 1 are requested before the MFMAs of tap t (hipcc had
//     serialised every MFMA behind its own ds_read: 92 us with eight waves and register-prefetched bands).  (A form with
//     the whole filter in every wave - 288 registers, every pixel fragment feeding two MFMAs - measured slower, 98 us.)
//   * outputs leave through a wave-private transposition patch as 16-byte stores; the forward form keeps the BatchNorm
//     statistics of the stored values in registers (one partial row per workgroup).
// The data gradient is the same kernel on the gradient with the filter in CRSK order and the taps mirrored.
// Operand orientation and k order are igemm_kernel's (tap-major, channel chunks ascending): the same output bits.
#include "cn_common.h"
#include "cn_api_internal.h"
#include <type_traits>

struct C3Params {
  const char* x;    // [N][H][W][64]
  const char* w;    // [64][9][64]: row = output channel of this product, then tap, then input channel
  char* y;          // [N][H][W][64]
  float* partial;   // optional [nwg][128]: sum | sum of squares of the stored outputs
  int N, H, W, nbands, nwork, flip;
  FastDiv div_w;
  // XF ("lazy a", forward only): x is the INPUT of the BatchNorm in front of this convolution; a = relu?(x * scale + shift)
  // is formed on the way into the halo (bn_apply_kernel's arithmetic; rows / columns outside the image stay zero) and
  // the band's own rows are written to a_out
  const float* xf;   // [scale | shift] (2 * 64)
  char* a_out;       // [N][H][W][64]
  int relu;
};

#define C3_ROWS 4
#define C3_PPOS 144     /* bytes per halo position */
#define C3_MAXW 56

template <typename T, bool XF = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(C3Params p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int HR = C3_ROWS + 2;
  constexpr int PP = 80;              // wave-private patch pitch: 32 channels * 2 bytes + 16
  __shared__ __attribute__((aligned(16))) char lds[HR * (C3_MAXW + 2) * C3_PPOS + 4 * 32 * PP];
  char* halo = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int h = lane >> 5;
  const int ct = wave & 1, pg = wave >> 1;   // 2 channel tiles x 2 pixel groups
  char* priv = lds + HR * (C3_MAXW + 2) * C3_PPOS + wave * (32 * PP);
  const int W = p.W, H = p.H, WP = W + 2;

  // filter fragments: this wave's 32 output channels, all 36 k-steps (tap = kk / 4)
  s16x8 wf[36];
#pragma unroll
  for (int kk = 0; kk < 36; ++kk) {
    const int co = ct * 32 + (lane & 31);
    wf[kk] = __builtin_bit_cast(s16x8, cn_ld16(p.w + ((size_t)co * 576 + (kk >> 2) * 64 + (2 * (kk & 3) + h) * 8) * 2));
  }
  // zero border columns (never rewritten)
  for (int id = tid; id < HR * 2 * 9; id += 256) {
    const int row = id / 18, rem = id - row * 18, side = rem / 9, c = rem - side * 9;
    cn_st16(halo + (row * WP + (side ? W + 1 : 0)) * C3_PPOS + c * 16, cn_zero16());
  }

  // XF: the thread's chunk (tid & 7) is fixed; its coefficients are re-read from an LDS table per batch (held in
  // registers across the MFMA phase they pushed the kernel over its 256-VGPR budget)
  __shared__ float s_xf[XF ? 128 : 1];
  if constexpr (XF) {
    if (tid < 128) s_xf[tid] = p.xf[tid];
  }
  const int nchunks = HR * W * 8;
  auto stage_halo = [&](int work) {   // global -> LDS, a batch of four 16-byte chunks per thread in flight
    const int band = work % p.nbands, n = work / p.nbands;
    const int oy0 = band * C3_ROWS;
    for (int id0 = tid; id0 < nchunks; id0 += 4 * 256) {
      u32x4 v[4];
      int dst[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = id0 + 256 * i;
        const int pos = id >> 3, c = id & 7;
        const int row = (int)cn_fastdiv((unsigned)(id < nchunks ? pos : 0), p.div_w);
        const int col = pos - row * W;
        const int iy = oy0 - 1 + row;
        const bool ok = id < nchunks && (unsigned)iy < (unsigned)H;
        v[i] = ok ? cn_ld16(p.x + ((((size_t)n * H + (size_t)iy) * W + (size_t)col) * 64 + (size_t)c * 8) * 2) : cn_zero16();
        dst[i] = id < nchunks ? (row * WP + col + 1) * C3_PPOS + c * 16 : -1;
        if constexpr (XF) {   // bit 30: an image row; bit 29: one of the band's own rows (written to a_out)
          if (ok) dst[i] |= (1 << 30) | ((row >= 1 && row <= C3_ROWS) ? (1 << 29) : 0);
        }
      }
      float xsc[XF ? 8 : 1], xsh[XF ? 8 : 1];
      if constexpr (XF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xsc[e] = s_xf[(tid & 7) * 8 + e]; xsh[e] = s_xf[64 + (tid & 7) * 8 + e]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (dst[i] < 0) continue;
        if constexpr (XF) {
          const int flags = dst[i];
          dst[i] &= (1 << 29) - 1;
          if (flags & (1 << 30)) {
            float f[8];
            Chunk<T>::unpack(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], xsc[e], xsh[e]);
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : 0.f;
            }
            v[i] = Chunk<T>::pack(f);
            if (flags & (1 << 29)) {
              const int id = id0 + 256 * i;
              const int pos = id >> 3, c = id & 7;
              const int row = (int)cn_fastdiv((unsigned)pos, p.div_w);
              const int col = pos - row * W;
              cn_st16(p.a_out + ((((size_t)n * H + (size_t)(oy0 - 1 + row)) * W + (size_t)col) * 64 + (size_t)c * 8) * 2, v[i]);
            }
          }
        }
        cn_st16(halo + dst[i], v[i]);
      }
    }
  };

  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
  const int npx = C3_ROWS * W;
  const int ntiles = (npx + 31) / 32;
  const int ech = lane & 3, erow = lane >> 2;

  for (int work = blockIdx.x; work < p.nwork; work += gridDim.x) {
    const int band = work % p.nbands, n = work / p.nbands;
    const int oy0 = band * C3_ROWS;
    __syncthreads();          // the previous band's halo has been consumed
    stage_halo(work);
    __syncthreads();
    for (int tile = pg; tile < ntiles; tile += 2) {
      const int px = tile * 32 + (lane & 31);
      const int pxc = px < npx ? px : npx - 1;
      const int oyl = (int)cn_fastdiv((unsigned)pxc, p.div_w);
      const int ox = pxc - oyl * W;
      // halo position of tap (0, 0) for this pixel: rows oyl.., columns ox.. (halo row 0 = image row oy0 - 1, column 0 = -1)
      const char* base = halo + (oyl * WP + ox) * C3_PPOS + h * 16;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      auto tap_off = [&](int tap) {
        const int tr = tap / 3, ts = tap - tr * 3;
        const int dr = p.flip ? 2 - tr : tr, ds = p.flip ? 2 - ts : ts;
        return (dr * WP + ds) * C3_PPOS;
      };
      s16x8 bq[2][4];    // the four fragments of a tap, double-buffered: tap t + 1 is in flight during the MFMAs of tap t
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[0][j] = __builtin_bit_cast(s16x8, cn_ld16(base + tap_off(0) + j * 32));
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cur = tap & 1, nxt = cur ^ 1;
        if (tap < 8) {
          const int o = tap_off(tap + 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) bq[nxt][j] = __builtin_bit_cast(s16x8, cn_ld16(base + o + j * 32));
        }
        cn_sched_fence();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (std::is_same<T, f16_t>::value) acc = cn_mfma_32x32x16_f16(wf[tap * 4 + j], bq[cur][j], acc);
          else acc = cn_mfma_32x32x16_bf16(wf[tap * 4 + j], bq[cur][j], acc);
        }
        cn_sched_fence();
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x2 pk;
        pk[0] = cn_pack2<T>(acc[q * 4], acc[q * 4 + 1]);
        pk[1] = cn_pack2<T>(acc[q * 4 + 2], acc[q * 4 + 3]);
        *(u32x2*)(priv + (lane & 31) * PP + (8 * q + 4 * h) * 2) = pk;
      }
      cn_wave_sync();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pl = k * 16 + erow;
        const int pxo = tile * 32 + pl;
        const u32x4 v = cn_ld16(priv + pl * PP + ech * 16);
        const int oyo = (int)cn_fastdiv((unsigned)(pxo < npx ? pxo : 0), p.div_w);
        const int oxo = pxo - oyo * W;
        if (pxo < npx && oy0 + oyo < H) {
          if (p.partial != nullptr) {
            float f[8];
            Chunk<T>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { ssum[e] += f[e]; ssq[e] = fmaf(f[e], f[e], ssq[e]); }
          }
          cn_st16(p.y + ((((size_t)n * H + (size_t)(oy0 + oyo)) * W + (size_t)oxo) * 64 + (size_t)(ct * 32 + ech * 8)) * 2, v);
        }
      }
      cn_wave_sync();
    }
  }
  if (p.partial != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int msk = 4; msk <= 32; msk <<= 1) {
        ssum[e] += cn_shfl_xor(ssum[e], msk);
        ssq[e] += cn_shfl_xor(ssq[e], msk);
      }
    __syncthreads();
    float* red = (float*)lds;   // [4 waves][32 channels][2]
    if (lane < 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 32 + lane * 8 + e) * 2] = ssum[e];
        red[(wave * 32 + lane * 8 + e) * 2 + 1] = ssq[e];
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int c = tid, cc = c & 31, ctile = c >> 5;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g) {    // the two pixel groups of this channel tile, fixed order
        a += red[((g * 2 + ctile) * 32 + cc) * 2];
        b += red[((g * 2 + ctile) * 32 + cc) * 2 + 1];
      }
      float* dst = p.partial + (size_t)blockIdx.x * 128;
      dst[c] = a;
      dst[64 + c] = b;
    }
  }
}

static int c3_wgs(int nwork) {
  int n = cn_get_option("conv3x3_wgs", 512);
  if (n < 1) n = 1;
  return n < nwork ? n : nwork;
}
extern "C" int cn_conv3x3_c64_ok(int H, int W, int C, int K, int dtype) {
  return (dtype == CN_BF16 || dtype == CN_F16) && C == 64 && K == 64 && W >= 1 && W <= C3_MAXW && H >= 1 ? 1 : 0;
}
extern "C" int cn_conv3x3_c64_rows(int N, int H) { return c3_wgs(N * ((H + C3_ROWS - 1) / C3_ROWS)); }

// y = conv3x3(x, w), stride 1, pad 1, 64 -> 64 channels, NHWC.  flip = 0: forward with w = KRSC filter [64][3][3][64];
// flip = 1: data gradient (x = dy, w = the CRSK filter: rows = input channels of the convolution).  partial (optional,
// forward): cn_conv3x3_c64_rows(N, H) rows of 128 floats [sum | sum of squares] of the stored outputs for
// cn_bn_fwd_train_partials.  Same output bits as cn_conv2d_fwd / cn_conv2d_dgrad.
static int c3_impl(const char* who, const void* x, const float* xf, int relu, void* a_out, const void* w, void* y, int N,
                   int H, int W, int dtype, int flip, float* partial, int partial_rows, void* stream) {
  if (x == nullptr || w == nullptr || y == nullptr) { cn_set_error("%s: null operand", who); return CN_EINVAL; }
  if (!cn_conv3x3_c64_ok(H, W, 64, 64, dtype) || N <= 0) { cn_set_error("%s: unsupported shape", who); return CN_ESHAPE; }
  C3Params p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.w = (const char*)w; p.y = (char*)y; p.partial = partial;
  p.N = N; p.H = H; p.W = W; p.flip = flip ? 1 : 0;
  p.nbands = (H + C3_ROWS - 1) / C3_ROWS;
  p.nwork = N * p.nbands;
  p.div_w = cn_make_fastdiv((unsigned)W);
  p.xf = xf; p.a_out = (char*)a_out; p.relu = relu;
  const int nwg = c3_wgs(p.nwork);
  if (partial != nullptr && partial_rows < nwg) { cn_set_error("%s: partial buffer of %d rows < %d", who, partial_rows, nwg); return CN_EWORKSPACE; }
  cn_set_last_kernel("conv3x3_c64_kernel<%s%s>%s", dtype == CN_F16 ? "f16_t" : "bf16_t",
                     xf != nullptr ? ", true" : "", flip ? " [dgrad]" : "");
  if (xf != nullptr) {
    if (dtype == CN_F16) CN_LAUNCH((conv3x3_c64_kernel<f16_t, true>), dim3((unsigned)nwg), dim3(256), (hipStream_t)stream, p);
    else CN_LAUNCH((conv3x3_c64_kernel<bf16_t, true>), dim3((unsigned)nwg), dim3(256), (hipStream_t)stream, p);
  } else {
    if (dtype == CN_F16) CN_LAUNCH((conv3x3_c64_kernel<f16_t>), dim3((unsigned)nwg), dim3(256), (hipStream_t)stream, p);
    else CN_LAUNCH((conv3x3_c64_kernel<bf16_t>), dim3((unsigned)nwg), dim3(256), (hipStream_t)stream, p);
  }
  return cn_check_launch("conv3x3_c64");
}
extern "C" int cn_conv3x3_c64(const void* x, const void* w, void* y, int N, int H, int W, int dtype, int flip,
                              float* partial, int partial_rows, void* stream) {
  return c3_impl("conv3x3_c64", x, nullptr, 0, nullptr, w, y, N, H, W, dtype, flip, partial, partial_rows, stream);
}
// "Lazy a" (forward): cn_conv3x3_c64 whose input is still the INPUT bn_y of the BatchNorm in front of the convolution
// (stats = [mean | invstd | scale | shift] of 64 channels): a = relu?(bn_y * scale + shift) is formed on the way into the
// halo and written to a_out [N][H][W][64] (cn_conv1x1_stream_fwd_lazya's contract for the 3x3 halo kernel; padding stays
// zero: it pads a, not bn_y).  Same a and y bits as the apply pass followed by cn_conv3x3_c64.
extern "C" int cn_conv3x3_c64_lazya(const void* bn_y, const float* stats, int relu, void* a_out, const void* w, void* y,
                                    int N, int H, int W, int dtype, float* partial, int partial_rows, void* stream) {
  if (stats == nullptr || a_out == nullptr) { cn_set_error("conv3x3_c64_lazya: null operand"); return CN_EINVAL; }
  return c3_impl("conv3x3_c64_lazya", bn_y, stats + 2 * 64, relu, a_out, w, y, N, H, W, dtype, 0, partial, partial_rows, stream);
}
