// stem.hip -- the 7x7 / stride-2 stem convolution (/root/reference models/resnet.py:226) as a halo kernel (round 3).
//
// The stem runs on the pixel-pair image (nchw_to_pairs: [N][Hp][Jp][8], one 16-byte chunk = two horizontally adjacent
// padded pixels x 4 channels) as a 7 x 4-tap convolution with stride (2, 1) and 8 input channels.  Through the tiled
// implicit-GEMM kernel every 128-pixel tile re-gathers its 28 chunks per pixel from L2 into LDS (each input chunk about
// 12 times per launch, 1.4 GB of L2 -> LDS traffic) behind a barrier per 64-element K tile: 0.39 ms, 0.27 of its roof.
// Here a workgroup owns ROWS output rows of one image:
//   * the input rows they need (2*ROWS + 5 rows of Jp chunks, 38 KB for ROWS = 8) are copied to LDS ONCE;
//   * the MFMA pixel fragments are read straight out of that halo - the 16 bytes lane (pixel, h) needs for k-step kk are
//     the chunk at (2*oy + r, ox + s2) with (r, s2) = divmod(2*kk + h, 4): a per-lane ds_read_b128, consecutive pixels on
//     consecutive 16-byte slots (conflict-free), no re-staging, no barrier inside the reduction;
//   * the 64 x 224 filter lives in registers as A-fragments (2 x 14 of them per wave);
//   * outputs leave through a wave-private transposition patch as 16-byte stores of full 128-byte pixel rows; the
//     BatchNorm statistics of the stored values stay in registers: one partial row per workgroup.
// Same operand orientation and k order as igemm_kernel on the pair image (its zero padding chunks add exact zeros): the
// same output bits; the statistics partials are associated differently (fp32).
#include "cn_common.h"
#include "cn_api_internal.h"
#include <type_traits>

struct StemParams {
  const char* xp;   // [N][Hp][Jp][8]
  const char* wp;   // [64][7][4][8]
  char* y;          // [N][P][Q][64]
  float* partial;   // [N * nbands][2 * 64]: sum | sum of squares of the stored outputs
  int N, Hp, Jp, P, Q, nbands;
  FastDiv div_q;
};

#define STEM_ROWS 8
#define STEM_R 7
#define STEM_S2 4
#define STEM_NKK 14   /* 28 chunks of 8 = 14 k-steps of 16 */
#define STEM_MAX_JP 120   /* pair columns the static LDS halo is sized for (224-pixel images: 115) */

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemParams p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int HR = 2 * STEM_ROWS + STEM_R - 2;   // halo rows: 2*(ROWS-1) + 7
  constexpr int PP = 144;                          // wave-private patch pitch: 64 channels * 2 bytes + 16
  __shared__ __attribute__((aligned(16))) char lds[HR * STEM_MAX_JP * 16 + 4 * 32 * PP];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int h = lane >> 5;
  const int band = blockIdx.x % p.nbands, n = blockIdx.x / p.nbands;
  const int oy0 = band * STEM_ROWS;
  const int Jp = p.Jp;
  char* halo = lds;
  char* priv = lds + HR * Jp * 16 + wave * (32 * PP);

  // filter fragments: rows = output channels, all 14 k-steps
  s16x8 wf[2][STEM_NKK];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < STEM_NKK; ++kk) {
      const int co = t * 32 + (lane & 31);
      wf[t][kk] = __builtin_bit_cast(s16x8, cn_ld16(p.wp + ((size_t)co * (STEM_R * STEM_S2) + 2 * kk + h) * 16));
    }
  // halo: input rows 2*oy0 .. 2*oy0 + HR - 1 (rows past the image are zero)
  {
    const int total = HR * Jp;
    const char* src = p.xp + ((size_t)n * p.Hp + (size_t)(2 * oy0)) * Jp * 16;
    const int rows_in = p.Hp - 2 * oy0;   // rows that exist
    for (int id = tid; id < total; id += 256) {
      const int row = id / Jp;
      const u32x4 v = row < rows_in ? cn_ld16(src + (size_t)id * 16) : cn_zero16();
      cn_st16(halo + id * 16, v);
    }
  }
  __syncthreads();

  const int ech = lane & 7, erow = lane >> 3;
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
  const int npx = STEM_ROWS * p.Q;
  const int ntiles = (npx + 31) / 32;
  for (int tile = wave; tile < ntiles; tile += 4) {
    const int px = tile * 32 + (lane & 31);
    const int pxc = px < npx ? px : npx - 1;                 // clamped: lanes past the band compute a valid address
    const int oyl = (int)cn_fastdiv((unsigned)pxc, p.div_q);
    const int ox = pxc - oyl * p.Q;
    const char* base = halo + ((2 * oyl) * Jp + ox) * 16;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // the two fragments of filter row r + 1 are requested before the MFMAs of row r (hipcc serialises every MFMA behind
    // its own ds_read otherwise)
    s16x8 bq[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bq[0][j] = __builtin_bit_cast(s16x8, cn_ld16(base + (2 * j + h) * 16));
#pragma unroll
    for (int r = 0; r < STEM_R; ++r) {
      const int cur = r & 1, nxt = cur ^ 1;
      if (r + 1 < STEM_R) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bq[nxt][j] = __builtin_bit_cast(s16x8, cn_ld16(base + ((r + 1) * Jp + 2 * j + h) * 16));
      }
      cn_sched_fence();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if constexpr (std::is_same<T, f16_t>::value) acc[t] = cn_mfma_32x32x16_f16(wf[t][2 * r + j], bq[cur][j], acc[t]);
          else acc[t] = cn_mfma_32x32x16_bf16(wf[t][2 * r + j], bq[cur][j], acc[t]);
        }
      cn_sched_fence();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        u32x2 pk;
        pk[0] = cn_pack2<T>(acc[t][qq * 4], acc[t][qq * 4 + 1]);
        pk[1] = cn_pack2<T>(acc[t][qq * 4 + 2], acc[t][qq * 4 + 3]);
        *(u32x2*)(priv + (lane & 31) * PP + (t * 32 + 8 * qq + 4 * h) * 2) = pk;
      }
    cn_wave_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pl = k * 8 + erow;
      const int pxo = tile * 32 + pl;
      const u32x4 v = cn_ld16(priv + pl * PP + ech * 16);
      const int oyo = (int)cn_fastdiv((unsigned)(pxo < npx ? pxo : 0), p.div_q);
      const int oxo = pxo - oyo * p.Q;
      if (pxo < npx && oy0 + oyo < p.P) {
        float f[8];
        Chunk<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { ssum[e] += f[e]; ssq[e] = fmaf(f[e], f[e], ssq[e]); }
        cn_st16(p.y + ((((size_t)n * p.P + (size_t)(oy0 + oyo)) * p.Q + (size_t)oxo) * 64 + (size_t)ech * 8) * 2, v);
      }
    }
    cn_wave_sync();
  }
  // ---- statistics of this workgroup's outputs: one partial row [sum(64) | sum of squares(64)]
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int msk = 8; msk <= 32; msk <<= 1) {
      ssum[e] += cn_shfl_xor(ssum[e], msk);
      ssq[e] += cn_shfl_xor(ssq[e], msk);
    }
  __syncthreads();
  float* red = (float*)lds;   // [4 waves][64][2]
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(wave * 64 + lane * 8 + e) * 2] = ssum[e];
      red[(wave * 64 + lane * 8 + e) * 2 + 1] = ssq[e];
    }
  }
  __syncthreads();
  if (tid < 64) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += red[(w * 64 + tid) * 2]; b += red[(w * 64 + tid) * 2 + 1]; }
    float* dst = p.partial + (size_t)blockIdx.x * 128;
    dst[tid] = a;
    dst[64 + tid] = b;
  }
}

// Shapes the halo kernel is built for: 64 output channels, 7 x 4 taps on the pair image, stride (2, 1), 16-bit storage,
// and a pair row that fits the LDS budget.
extern "C" int cn_stem_fwd_ok(int K, int R, int S2, int Jp, int dtype) {
  return (dtype == CN_BF16 || dtype == CN_F16) && K == 64 && R == STEM_R && S2 == STEM_S2 && Jp >= STEM_S2 && Jp <= STEM_MAX_JP ? 1 : 0;
}
extern "C" int cn_stem_fwd_rows(int N, int P) { return N * ((P + STEM_ROWS - 1) / STEM_ROWS); }

// y[n][oy][ox][k] = sum_{r < 7, s2 < 4, e < 8} xp[n][2*oy + r][ox + s2][e] * wp[k][r][s2][e]   (P = (Hp - 7) / 2 + 1 rows,
// Q = Jp - 3 columns): cn_conv2d_fwd_bnstats on the pair image (R = 7, S = 4, stride (2, 1), no padding) as a halo
// kernel; partial: cn_stem_fwd_rows(N, P) rows of 128 floats [sum | sum of squares] for cn_bn_fwd_train_partials (or the
// fused stem pooling).  Same output bits as the tiled kernel.
extern "C" int cn_stem_fwd(const void* xp, const void* wp, void* y, int N, int Hp, int Jp, int dtype, float* partial,
                           int partial_rows, void* stream) {
  if (xp == nullptr || wp == nullptr || y == nullptr || partial == nullptr) { cn_set_error("stem_fwd: null operand"); return CN_EINVAL; }
  if (!cn_stem_fwd_ok(64, STEM_R, STEM_S2, Jp, dtype) || Hp < STEM_R || N <= 0) { cn_set_error("stem_fwd: unsupported shape"); return CN_ESHAPE; }
  StemParams p;
  memset(&p, 0, sizeof(p));
  p.xp = (const char*)xp; p.wp = (const char*)wp; p.y = (char*)y; p.partial = partial;
  p.N = N; p.Hp = Hp; p.Jp = Jp;
  p.P = (Hp - STEM_R) / 2 + 1;
  p.Q = Jp - STEM_S2 + 1;
  p.nbands = (p.P + STEM_ROWS - 1) / STEM_ROWS;
  p.div_q = cn_make_fastdiv((unsigned)p.Q);
  if (partial_rows < N * p.nbands) { cn_set_error("stem_fwd: partial buffer of %d rows < %d", partial_rows, N * p.nbands); return CN_EWORKSPACE; }
  cn_set_last_kernel("stem_fwd_kernel<%s>", dtype == CN_F16 ? "f16_t" : "bf16_t");
  dim3 grid((unsigned)(N * p.nbands));
  if (dtype == CN_F16) CN_LAUNCH((stem_fwd_kernel<f16_t>), grid, dim3(256), (hipStream_t)stream, p);
  else CN_LAUNCH((stem_fwd_kernel<bf16_t>), grid, dim3(256), (hipStream_t)stream, p);
  return cn_check_launch("stem_fwd");
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the stem on the pair image (round 3):
//     dwp[k][r][s2][e] = sum_{n, oy, ox} dy[n][oy][ox][k] * xp[n][2*oy + r][ox + s2][e]
// The tiled kernel (wgrad_kernel on the pair image) re-gathers 28 chunks per pixel from L2 like the forward pass did,
// and it is the LAST kernel of the backward pass: nothing overlaps it and the optimizer step waits for it.  Here a
// workgroup walks bands of 8 output rows: the band's input rows sit in an LDS halo (as in stem_fwd_kernel), dy comes in
// stages of 128 pixels; the MFMA reduction index is the pixel, so both operands are transpose reads
// (ds_read_b64_tr_b16): dy from its pixel-major tile, the activation STRAIGHT out of the halo - lane L of a 16-lane group
// points at the 4 elements (pixel ox0 + L/4, pair s2, element e0..e0+3) it contributes, no gathered tile is ever
// written.  Seven waves own the seven filter rows (32 columns = 4 pairs x 8 elements each) x 64 output channels; the
// eighth helps with the loads.  Partial sums per workgroup, fixed-order reduction by wgrad_reduce_kernel.
struct StemWgParams {
  const char* xp;   // [N][Hp][Jp][8]
  const char* dy;   // [N][P][Q][64]
  float* part;      // [nwg][64][224]
  int N, Hp, Jp, P, Q, nbands, nwork;
  FastDiv div_q;
};

int wg_launch_reduce(hipStream_t stream, const float* part, float* dw, int nsplit, int Co, int ntaps, int Ci, int Creal,
                     float beta, float scale);   // wgrad.hip

template <typename T>
__global__ __launch_bounds__(512, 4) void stem_wgrad_kernel(StemWgParams p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int HR = 2 * STEM_ROWS + STEM_R - 2;
  constexpr int PD = 64 * 2 + 64;      // dy tile pitch: four consecutive pixel rows on four bank quarters
  constexpr int BMS = 128;             // pixels per stage
  __shared__ __attribute__((aligned(16))) char lds[HR * STEM_MAX_JP * 16 + BMS * PD];
  char* halo = lds;
  char* dyT = lds + HR * STEM_MAX_JP * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int L = lane & 15, g1 = (lane >> 4) & 1, h = lane >> 5;
  const int Jp = p.Jp, Q = p.Q;
  const int npx = STEM_ROWS * Q;
  const int nst = (npx + BMS - 1) / BMS;

  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  for (int work = blockIdx.x; work < p.nwork; work += gridDim.x) {
    const int band = work % p.nbands, n = work / p.nbands;
    const int oy0 = band * STEM_ROWS;
    const int vrows = p.P - oy0 < STEM_ROWS ? p.P - oy0 : STEM_ROWS;
    const int vpx = vrows * Q;                       // pixels of this band that exist
    __syncthreads();                                 // the previous band's halo / dy tile have been consumed
    {
      const int total = HR * Jp;
      const char* src = p.xp + ((size_t)n * p.Hp + (size_t)(2 * oy0)) * Jp * 16;
      const int rows_in = p.Hp - 2 * oy0;
      for (int id = tid; id < total; id += 512) {
        const int row = id / Jp;
        cn_st16(halo + id * 16, row < rows_in ? cn_ld16(src + (size_t)id * 16) : cn_zero16());
      }
    }
    const char* dyb = p.dy + (((size_t)n * p.P + (size_t)oy0) * Q) * 128;
    u32x4 dreg[2];
    auto load_dy = [&](int st) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i;
        const int px = st * BMS + (id >> 3);
        dreg[i] = px < vpx ? cn_ld16(dyb + (size_t)px * 128 + (id & 7) * 16) : cn_zero16();
      }
    };
    load_dy(0);
    for (int st = 0; st < nst; ++st) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i;
        cn_st16(dyT + (id >> 3) * PD + (id & 7) * 16, dreg[i]);
      }
      __syncthreads();
      if (st + 1 < nst) load_dy(st + 1);
      if (wave < STEM_R) {
#pragma unroll
        for (int kk = 0; kk < BMS / 16; ++kk) {
          int pk = st * BMS + kk * 16;                 // 16 consecutive pixels of one output row (Q % 16 == 0)
          if (pk >= npx) pk = npx - 16;                // (a stage past the band: its dy rows are zero)
          const int oyl = (int)cn_fastdiv((unsigned)pk, p.div_q);
          const int ox0 = pk - oyl * Q;
          const int rbase = kk * 16 + h * 8 + (L >> 2);
          const int cbase = g1 * 16 + (L & 3) * 4;
          s16x8 af[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const char* q = dyT + rbase * PD + (a * 32 + cbase) * 2;
            const s16x4 lo = cn_lds_read_tr16_b64(q);
            const s16x4 hi = cn_lds_read_tr16_b64(q + 4 * PD);
            af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
          // activation: columns of this wave's filter row r = wave: pair s2 = g1*2 + ((L&3) >> 1), elements e0..e0+3
          const char* qx = halo + ((2 * oyl + wave) * Jp + ox0 + h * 8 + (L >> 2) + g1 * 2 + ((L & 3) >> 1)) * 16 +
                           ((L & 3) & 1) * 8;
          const s16x4 xl = cn_lds_read_tr16_b64(qx);
          const s16x4 xh = cn_lds_read_tr16_b64(qx + 4 * 16);
          const s16x8 bf = __builtin_shufflevector(xl, xh, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            if constexpr (std::is_same<T, f16_t>::value) acc[a] = cn_mfma_32x32x16_f16(af[a], bf, acc[a]);
            else acc[a] = cn_mfma_32x32x16_bf16(af[a], bf, acc[a]);
          }
        }
      }
      __syncthreads();
    }
  }
  if (wave < STEM_R) {
    float* out = p.part + (size_t)blockIdx.x * 64 * (STEM_R * STEM_S2 * 8);
    const int col = wave * 32 + (lane & 31);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(size_t)co * (STEM_R * STEM_S2 * 8) + col] = acc[a][r];
      }
  }
}

static int stem_wg_count(int nwork) {
  int n = cn_get_option("stem_wgrad_wgs", 256);
  if (n < 1) n = 1;
  return n < nwork ? n : nwork;
}
extern "C" int cn_stem_wgrad_ok(int K, int R, int S2, int Jp, int dtype) {
  return cn_stem_fwd_ok(K, R, S2, Jp, dtype) && (Jp - STEM_S2 + 1) % 16 == 0 ? 1 : 0;
}
extern "C" size_t cn_stem_wgrad_workspace(int N, int Hp) {
  const int P = (Hp - STEM_R) / 2 + 1;
  return (size_t)stem_wg_count(N * ((P + STEM_ROWS - 1) / STEM_ROWS)) * 64 * (STEM_R * STEM_S2 * 8) * sizeof(float);
}
// dwp [64][7][4][8] (fp32, the layout cn_wgrad_unpack_pairs takes) = beta*dwp + scale * the stem's weight gradient on the
// pair image: cn_conv2d_wgrad(xp, dy, ...) with R = 7, S = 4, stride (2, 1) as a halo kernel.  Differs from the tiled
// kernel by fp32 summation order only.
extern "C" int cn_stem_wgrad(const void* xp, const void* dy, float* dwp, int N, int Hp, int Jp, int dtype, float beta,
                             float scale, void* workspace, size_t ws_bytes, void* stream) {
  if (xp == nullptr || dy == nullptr || dwp == nullptr) { cn_set_error("stem_wgrad: null operand"); return CN_EINVAL; }
  if (!cn_stem_wgrad_ok(64, STEM_R, STEM_S2, Jp, dtype) || Hp < STEM_R || N <= 0) { cn_set_error("stem_wgrad: unsupported shape"); return CN_ESHAPE; }
  StemWgParams p;
  memset(&p, 0, sizeof(p));
  p.xp = (const char*)xp; p.dy = (const char*)dy; p.part = (float*)workspace;
  p.N = N; p.Hp = Hp; p.Jp = Jp;
  p.P = (Hp - STEM_R) / 2 + 1;
  p.Q = Jp - STEM_S2 + 1;
  p.nbands = (p.P + STEM_ROWS - 1) / STEM_ROWS;
  p.nwork = N * p.nbands;
  p.div_q = cn_make_fastdiv((unsigned)p.Q);
  const int nwg = stem_wg_count(p.nwork);
  if (workspace == nullptr || ws_bytes < cn_stem_wgrad_workspace(N, Hp)) { cn_set_error("stem_wgrad: workspace too small"); return CN_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const int phase = cn_get_option("wgrad_phase", 0);
  if (phase != 2) {
    cn_set_last_kernel("stem_wgrad_kernel<%s>", dtype == CN_F16 ? "f16_t" : "bf16_t");
    if (dtype == CN_F16) CN_LAUNCH((stem_wgrad_kernel<f16_t>), dim3((unsigned)nwg), dim3(512), st, p);
    else CN_LAUNCH((stem_wgrad_kernel<bf16_t>), dim3((unsigned)nwg), dim3(512), st, p);
    int rc = cn_check_launch("stem_wgrad");
    if (rc) return rc;
  }
  if (phase == 1) return CN_OK;
  return wg_launch_reduce(st, (const float*)workspace, dwp, nwg, 64, STEM_R * STEM_S2, 8, 8, beta, scale);
}
