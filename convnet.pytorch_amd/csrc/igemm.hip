// igemm.hip -- implicit-GEMM "gather GEMM" on MFMA for gfx950: Conv2d forward, Conv2d data-gradient
// and Linear forward / input-gradient all run through this one kernel.
//
// Replaces (reference, /root/reference): nn.Conv2d forward + the dgrad half of its backward
// (models/resnet.py:75-78,126-132,178-179,226-227) and nn.Linear (models/resnet.py:242), which the
// reference delegates to ATen (MIOpen / oneDNN).
//
//   Out[n, oh, ow, co] = sum_{t < ntaps} sum_{ci < Ci}  In[n, g_h*a_h + dh[t], g_w*a_w + dw[t], ci]
//                                                      * Wt[co][woff[t] + ci]
//   (oh, ow) = (g_h*oh_mul + oh_off, g_w*ow_mul + ow_off),   (g_h, g_w) in [0,Hg) x [0,Wg)
//
// Layout: activations NHWC, filters [Co][taps*Ci] ("KRSC"; "CRSK" for dgrad), 16-byte channel
// chunks.  MFMA roles: A operand (i) = output channels (filter rows), B operand (j) = pixels, so a
// lane of the 32x32 accumulator owns one pixel and quads of consecutive channels -> the epilogue
// packs 8/16-byte LDS writes and the global store is fully coalesced 16 B per lane.
//
// Block = 256 threads (4 waves); tile = BN channels x BM pixels x 128 bytes of reduction per step;
// double-buffered LDS with register prefetch of the next K tile; XOR-swizzled 16-byte slots so the
// ds_read_b128 fragment reads are bank-conflict free.
#include "cn_common.h"
#include "cn_api_internal.h"

#define IG_MAX_TAPS 64

struct IgemmParams {
  const char* x;
  const char* w;
  char* y;
  const char* addend;   // optional tensor added to the output (same layout / dtype as y)
  const float* bias;
  int N, Hi, Wi, Ci;
  int Hg, Wg, a_h, a_w;
  int Ho, Wo, Co;
  int oh_mul, oh_off, ow_mul, ow_off;
  int ntaps, cpt, cpt_shift, nchunks;
  long long w_row;
  int out_f32, relu;
  int M, n_ntiles;
  FastDiv div_hw, div_w, div_cpt;
  int tap_dhdw[IG_MAX_TAPS];
  int tap_woff[IG_MAX_TAPS];
};

template <typename T>
__device__ __forceinline__ void ig_mma(const u32x4& a, const u32x4& b, f32x16& acc);
template <>
__device__ __forceinline__ void ig_mma<bf16_t>(const u32x4& a, const u32x4& b, f32x16& acc) {
  acc = cn_mfma_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), acc);
}
template <>
__device__ __forceinline__ void ig_mma<float>(const u32x4& a, const u32x4& b, f32x16& acc) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    union { unsigned int u; float f; } fa, fb;
    fa.u = a[e];
    fb.u = b[e];
    acc = cn_mfma_32x32x2_f32(fa.f, fb.f, acc);
  }
}

__device__ __forceinline__ int ig_slot(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// STAGES = 2: double-buffered LDS staging (one barrier per K tile, 2 workgroups/CU at 128x128);
// STAGES = 1: single staging buffer + register prefetch (two barriers per K tile, but half the LDS
//             so 4 workgroups/CU hide the global-load latency of short reductions, e.g. 1x1 convs).
// OUTF32 sizes the LDS output tile for fp32 results (fp32 compute or fp32 logits).
template <typename T, int WC, int WP, int TI, int TJ, int STAGES, bool OUTF32>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
  constexpr int BN = WC * TI * 32;  // output channels per block
  constexpr int BM = WP * TJ * 32;  // pixels per block
  static_assert(WC * WP == 4, "4 waves");
  constexpr int EB = ElemTraits<T>::kBytes;
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int NPR = BM / 32;  // pixel rows staged per thread
  constexpr int NWR = BN / 32;  // filter rows staged per thread
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int OUT_MAX = BM * (BN * (OUTF32 ? 4 : EB) + 16);
  constexpr int MAIN = (STAGES * STAGE > OUT_MAX) ? STAGES * STAGE : OUT_MAX;
  constexpr int LDS_BYTES = MAIN + IG_MAX_TAPS * 8 + BM * 4;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  int* s_taps = (int*)(lds + MAIN);
  int* s_outpix = (int*)(lds + MAIN + IG_MAX_TAPS * 8);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const unsigned int tile = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = tile % p.n_ntiles;
  const int mt = tile / p.n_ntiles;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int HgWg = p.Hg * p.Wg;

  if (tid < IG_MAX_TAPS) {
    int t = tid < p.ntaps ? tid : 0;
    s_taps[2 * tid] = p.ntaps > 0 ? p.tap_dhdw[t] : 0;
    s_taps[2 * tid + 1] = p.ntaps > 0 ? p.tap_woff[t] : 0;
  }
  if (tid < BM) {
    int m = m0 + tid;
    int pix = -1;
    if (m < p.M) {
      int n = (int)cn_fastdiv((unsigned)m, p.div_hw);
      int rem = m - n * HgWg;
      int hg = (int)cn_fastdiv((unsigned)rem, p.div_w);
      int wg = rem - hg * p.Wg;
      pix = (n * p.Ho + hg * p.oh_mul + p.oh_off) * p.Wo + wg * p.ow_mul + p.ow_off;
    }
    s_outpix[tid] = pix;
  }

  // per-thread staging coordinates (fixed for the whole reduction loop)
  const int cc = tid & 7;
  const int r0 = tid >> 3;
  int pbase[NPR], phin[NPR], pwin[NPR];
  bool pvalid[NPR];
#pragma unroll
  for (int i = 0; i < NPR; ++i) {
    int m = m0 + r0 + 32 * i;
    pvalid[i] = m < p.M;
    int mm = pvalid[i] ? m : 0;
    int n = (int)cn_fastdiv((unsigned)mm, p.div_hw);
    int rem = mm - n * HgWg;
    int hg = (int)cn_fastdiv((unsigned)rem, p.div_w);
    int wg = rem - hg * p.Wg;
    pbase[i] = n * p.Hi * p.Wi;
    phin[i] = hg * p.a_h;
    pwin[i] = wg * p.a_w;
  }
  __syncthreads();

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  u32x4 preg[NPR], wreg[NWR];
  const int nkt = (p.nchunks + 7) >> 3;

  auto load_tile = [&](int kt) {
    const int kc = kt * 8 + cc;
    const bool kvalid = kc < p.nchunks;
    int tap = 0, cchunk = kc;
    if (p.ntaps > 1) {
      tap = kvalid ? (int)cn_fastdiv((unsigned)kc, p.div_cpt) : 0;
      cchunk = kvalid ? kc - tap * p.cpt : 0;
    }
    const int dhdw = s_taps[2 * tap];
    const int woff = s_taps[2 * tap + 1];
    const int dh = (int)(short)(dhdw & 0xffff);
    const int dw = dhdw >> 16;
    const int coff = cchunk * CH;
#pragma unroll
    for (int i = 0; i < NPR; ++i) {
      int hi = phin[i] + dh, wi = pwin[i] + dw;
      bool ok = kvalid && pvalid[i] && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
      size_t off = ((size_t)(pbase[i] + hi * p.Wi + wi) * (size_t)p.Ci + (size_t)coff) * EB;
      preg[i] = ok ? cn_ld16(p.x + off) : cn_zero16();
    }
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      int co = n0 + r0 + 32 * i;
      bool ok = kvalid && co < p.Co;
      size_t off = ((size_t)co * (size_t)p.w_row + (size_t)(woff + coff)) * EB;
      wreg[i] = ok ? cn_ld16(p.w + off) : cn_zero16();
    }
  };
  auto store_tile = [&](int buf) {
    char* wt = lds + buf * STAGE;
    char* pt = wt + BN * 128;
#pragma unroll
    for (int i = 0; i < NWR; ++i) cn_st16(wt + ig_slot(r0 + 32 * i, cc), wreg[i]);
#pragma unroll
    for (int i = 0; i < NPR; ++i) cn_st16(pt + ig_slot(r0 + 32 * i, cc), preg[i]);
  };

  const int wc = wave % WC;
  const int wp = wave / WC;
  auto compute = [&](int buf) {
    const char* wt = lds + buf * STAGE;
    const char* pt = wt + BN * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int chunk = kk * 2 + (lane >> 5);
      u32x4 af[TI], bfr[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) af[a] = cn_ld16(wt + ig_slot((wc * TI + a) * 32 + (lane & 31), chunk));
#pragma unroll
      for (int b = 0; b < TJ; ++b) bfr[b] = cn_ld16(pt + ig_slot((wp * TJ + b) * 32 + (lane & 31), chunk));
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) ig_mma<T>(af[a], bfr[b], acc[a][b]);
    }
  };

  if (nkt > 0) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      if (STAGES == 2) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        compute(buf);
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
      } else {
        if (kt + 1 < nkt) load_tile(kt + 1);
        compute(0);
        __syncthreads();
        if (kt + 1 < nkt) {
          store_tile(0);
          __syncthreads();
        }
      }
    }
  }

  // ---- epilogue: accumulators -> LDS out tile [BM pixels][BN channels] -> coalesced global store
  const int OEB = OUTF32 ? 4 : EB;
  const int pitch = BN * OEB + 16;
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int prow = (wp * TJ + b) * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = (wc * TI + a) * 32 + 8 * q + 4 * (lane >> 5);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float f = acc[a][b][q * 4 + e];
          if (p.bias != nullptr && n0 + c + e < p.Co) f += p.bias[n0 + c + e];
          if (p.relu) f = f > 0.f ? f : 0.f;
          v[e] = f;
        }
        char* dst = lds + prow * pitch + c * OEB;
        if (OEB == 4) {
          cn_st16(dst, Chunk<float>::pack(v));
        } else {
          u32x2 pk;
          pk[0] = cn_pack_bf16x2(v[0], v[1]);
          pk[1] = cn_pack_bf16x2(v[2], v[3]);
          *(u32x2*)dst = pk;
        }
      }
    }
  __syncthreads();
  const int epc = 16 / OEB;            // elements per 16-byte chunk of the output
  const int cpr = BN / epc;            // chunks per tile row
  const bool vec_ok = ((p.Co * OEB) & 15) == 0;
  for (int id = tid; id < BM * cpr; id += 256) {
    const int row = id / cpr, col = id - row * cpr;
    const int pix = s_outpix[row];
    if (pix < 0) continue;
    const int c_first = n0 + col * epc;
    if (c_first >= p.Co) continue;
    const char* src = lds + row * pitch + col * 16;
    char* dst = p.y + ((size_t)pix * (size_t)p.Co + (size_t)c_first) * OEB;
    const size_t goff = ((size_t)pix * (size_t)p.Co + (size_t)c_first) * OEB;
    if (vec_ok && c_first + epc <= p.Co) {
      u32x4 v = cn_ld16(src);
      if (p.addend != nullptr) {   // e.g. the residual-branch gradient folded into dgrad
        const u32x4 a = cn_ld16(p.addend + goff);
        if (OEB == 4) {
          float fv[4], fa[4];
          Chunk<float>::unpack(v, fv);
          Chunk<float>::unpack(a, fa);
#pragma unroll
          for (int e = 0; e < 4; ++e) fv[e] += fa[e];
          v = Chunk<float>::pack(fv);
        } else {
          float fv[8], fa[8];
          Chunk<bf16_t>::unpack(v, fv);
          Chunk<bf16_t>::unpack(a, fa);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += fa[e];
          v = Chunk<bf16_t>::pack(fv);
        }
      }
      cn_st16(dst, v);
    } else {
      for (int e = 0; e < epc && c_first + e < p.Co; ++e) {
        if (OEB == 4) {
          float f = ((const float*)src)[e];
          if (p.addend != nullptr) f += ((const float*)(p.addend + goff))[e];
          ((float*)dst)[e] = f;
        } else {
          float f = cn_bf16_to_f32(((const unsigned short*)src)[e]);
          if (p.addend != nullptr) f += cn_bf16_to_f32(((const unsigned short*)(p.addend + goff))[e]);
          ((unsigned short*)dst)[e] = cn_f32_to_bf16(f);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
static int ig_log2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

template <typename T, bool OUTF32>
static int ig_launch(IgemmParams& p, hipStream_t stream) {
  const int nkt = (p.nchunks + 7) / 8;
  int stages = cn_get_option("igemm_stages", 0);
  if (stages != 1 && stages != 2) stages = nkt <= 24 ? 1 : 2;   // measured crossover (profiles/r01_conv_layers)
  const int BM = 128, BN = p.Co <= 64 ? 64 : 128;
  p.n_ntiles = (p.Co + BN - 1) / BN;
  const int n_mtiles = (p.M + BM - 1) / BM;
  dim3 grid((unsigned)(p.n_ntiles * n_mtiles));
  if (p.Co <= 64) {
    if (stages == 1) CN_LAUNCH((igemm_kernel<T, 1, 4, 2, 1, 1, OUTF32>), grid, dim3(256), stream, p);
    else CN_LAUNCH((igemm_kernel<T, 1, 4, 2, 1, 2, OUTF32>), grid, dim3(256), stream, p);
  } else {
    if (stages == 1) CN_LAUNCH((igemm_kernel<T, 2, 2, 2, 2, 1, OUTF32>), grid, dim3(256), stream, p);
    else CN_LAUNCH((igemm_kernel<T, 2, 2, 2, 2, 2, OUTF32>), grid, dim3(256), stream, p);
  }
  return cn_check_launch("igemm");
}

static int ig_dispatch(IgemmParams& p, int dtype, hipStream_t stream) {
  if (p.M <= 0 || p.Co <= 0) return CN_OK;
  if (dtype == CN_BF16) return p.out_f32 ? ig_launch<bf16_t, true>(p, stream) : ig_launch<bf16_t, false>(p, stream);
  if (dtype == CN_F32) { p.out_f32 = 1; return ig_launch<float, true>(p, stream); }
  cn_set_error("igemm: bad dtype %d", dtype);
  return CN_EINVAL;
}

static int ig_common(IgemmParams& p, int dtype, int Ci, int ntaps) {
  const int CH = dtype == CN_BF16 ? 8 : 4;
  if (Ci % CH != 0) {
    cn_set_error("igemm: reduction channels %d not a multiple of the 16-byte chunk (%d elems)", Ci, CH);
    return CN_ESHAPE;
  }
  if (ntaps > IG_MAX_TAPS) {
    cn_set_error("igemm: %d taps > %d", ntaps, IG_MAX_TAPS);
    return CN_ESHAPE;
  }
  p.cpt = Ci / CH;
  p.cpt_shift = ig_log2_exact(p.cpt);
  p.div_cpt = cn_make_fastdiv((unsigned)p.cpt);
  p.ntaps = ntaps;
  p.nchunks = ntaps * p.cpt;
  p.M = p.N * p.Hg * p.Wg;
  p.div_hw = cn_make_fastdiv((unsigned)(p.Hg * p.Wg));
  p.div_w = cn_make_fastdiv((unsigned)p.Wg);
  return CN_OK;
}

extern "C" int cn_conv2d_fwd(const void* x, const void* w_krsc, void* y, const float* bias, int N, int H,
                             int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                             int pad_w, int dtype, int out_f32, int relu, void* stream) {
  const int P = (H + 2 * pad_h - R) / stride_h + 1;
  const int Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0 || N <= 0) { cn_set_error("conv2d_fwd: empty output"); return CN_ESHAPE; }
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.w = (const char*)w_krsc; p.y = (char*)y; p.bias = bias;
  p.N = N; p.Hi = H; p.Wi = W; p.Ci = C;
  p.Hg = P; p.Wg = Q; p.a_h = stride_h; p.a_w = stride_w;
  p.Ho = P; p.Wo = Q; p.Co = K;
  p.oh_mul = 1; p.ow_mul = 1; p.oh_off = 0; p.ow_off = 0;
  p.w_row = (long long)R * S * C;
  p.out_f32 = out_f32; p.relu = relu;
  int rc = ig_common(p, dtype, C, R * S);
  if (rc) return rc;
  for (int r = 0; r < R; ++r)
    for (int s = 0; s < S; ++s) {
      int t = r * S + s;
      p.tap_dhdw[t] = ((r - pad_h) & 0xffff) | ((s - pad_w) << 16);
      p.tap_woff[t] = t * C;
    }
  return ig_dispatch(p, dtype, (hipStream_t)stream);
}

extern "C" int cn_conv2d_dgrad(const void* dy, const void* w_crsk, void* dx, const void* addend, int N, int H,
                               int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                               int pad_w, int dtype, int out_f32, void* stream) {
  const int P = (H + 2 * pad_h - R) / stride_h + 1;
  const int Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0 || N <= 0) { cn_set_error("conv2d_dgrad: empty output"); return CN_ESHAPE; }
  // One launch per output-parity class (ph, pw): dX[n, hg*st+ph, wg*st+pw, :] receives exactly the
  // taps r with (ph + pad - r) % st == 0, reading dY at row hg + (ph + pad - r)/st.
  for (int ph = 0; ph < stride_h; ++ph)
    for (int pw = 0; pw < stride_w; ++pw) {
      IgemmParams p;
      memset(&p, 0, sizeof(p));
      p.x = (const char*)dy; p.w = (const char*)w_crsk; p.y = (char*)dx; p.bias = nullptr;
      p.addend = (const char*)addend;
      p.N = N; p.Hi = P; p.Wi = Q; p.Ci = K;
      p.Hg = (H - ph + stride_h - 1) / stride_h;
      p.Wg = (W - pw + stride_w - 1) / stride_w;
      if (p.Hg <= 0 || p.Wg <= 0) continue;
      p.a_h = 1; p.a_w = 1;
      p.Ho = H; p.Wo = W; p.Co = C;
      p.oh_mul = stride_h; p.oh_off = ph; p.ow_mul = stride_w; p.ow_off = pw;
      p.w_row = (long long)R * S * K;
      p.out_f32 = out_f32; p.relu = 0;
      int nt = 0;
      int dhdw[IG_MAX_TAPS], woff[IG_MAX_TAPS];
      for (int r = 0; r < R; ++r) {
        int nh = ph + pad_h - r;
        if (((nh % stride_h) + stride_h) % stride_h != 0) continue;
        for (int s = 0; s < S; ++s) {
          int nw = pw + pad_w - s;
          if (((nw % stride_w) + stride_w) % stride_w != 0) continue;
          if (nt >= IG_MAX_TAPS) { cn_set_error("conv2d_dgrad: too many taps"); return CN_ESHAPE; }
          int dh = nh / stride_h, dw = nw / stride_w;  // exact
          dhdw[nt] = (dh & 0xffff) | (dw << 16);
          woff[nt] = (r * S + s) * K;
          ++nt;
        }
      }
      int rc = ig_common(p, dtype, K, nt);
      if (rc) return rc;
      for (int t = 0; t < nt; ++t) { p.tap_dhdw[t] = dhdw[t]; p.tap_woff[t] = woff[t]; }
      rc = ig_dispatch(p, dtype, (hipStream_t)stream);
      if (rc) return rc;
    }
  return CN_OK;
}
