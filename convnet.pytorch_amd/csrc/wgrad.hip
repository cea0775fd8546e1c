// wgrad.hip -- Conv2d / Linear weight gradient on MFMA for gfx950.
//
// Replaces the wgrad half of nn.Conv2d / nn.Linear backward that the reference reaches through
// loss.backward() (/root/reference trainer.py:162; layers built at models/resnet.py:75-78,126-132,
// 178-179,226-227,242).
//
//   dW[co][t][ci] = sum_m dY[m][co] * X[gather(m, t)][ci],   m = (n, ho, wo),
//   gather(m, t) = (n, ho*stride_h + dh[t], wo*stride_w + dw[t])   (zero outside the image)
//
// Both GEMM operands are stored with the *reduction* index (pixel m) as the slow dimension, so the
// MFMA fragments need a transpose: bf16 uses the gfx950 LDS transpose read (ds_read_b64_tr_b16),
// fp32 reads single dwords (32 consecutive channels per half-wave, conflict free).
// The reduction over up to 3.2 M pixels is split across workgroups; partial tiles go to an fp32
// workspace and a second kernel reduces them in a fixed order (deterministic, no atomics) while
// writing the KRSC gradient (optionally accumulating, for chunked batches).
#include "cn_common.h"
#include "cn_api_internal.h"
#include <type_traits>

#define WG_MAX_TAPS 64

struct WgradParams {
  const char* x;
  const char* dy;
  float* part;
  int N, Hi, Wi, Ci, Ho, Wo, Co;
  int stride_h, stride_w;
  int ntaps, cpt, ncols;
  int M, m_per_split, nsplit;
  int n_itiles, n_jtiles;
  unsigned int x_bytes, dy_bytes;
  int simple;   // 1x1, stride 1, no padding: the gather is the identity (row m of x)
  // "lazy dy" (register-staged kernel only): dy[m][k] = c1[k]*dy[m][k] + c2[k]*dy2[m][k] + c3[k] formed on load
  // (dy = masked gradient g, dy2 = BatchNorm input y, coef = [c1 | c2 | c3] of the Co channels), rounded to T
  const char* dy2;
  const float* coef;
  FastDiv div_hw, div_w, div_cpt;
  int tap_dhdw[WG_MAX_TAPS];
};

template <typename T, int BI, int BJ, int LAZY = 0>
// LAZY = 2: the lazy form with its 3 x CH per-thread coefficients read from an LDS table at each stage instead of held in
// 24 registers, and the kernel held to 168 VGPRs (three workgroups per CU like the plain kernel; LAZY = 1 compiles to
// 180 = two).  Same arithmetic, same bits.
__global__ __launch_bounds__(256, LAZY == 2 ? 3 : 1) void wgrad_kernel(WgradParams p) {
  constexpr int EB = ElemTraits<T>::kBytes;
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr bool kBf16 = (EB == 2);
  constexpr int BKP = kBf16 ? 64 : 32;        // pixels per stage
  constexpr int TI = BI / 64, TJ = BJ / 64;   // 32x32 MFMA tiles per wave (2x2 waves)
  constexpr int PI = BI * EB + 64;            // LDS row pitches (bytes)
  constexpr int PJ = BJ * EB + 64;
  constexpr int CI_ = BI / CH, CJ_ = BJ / CH;  // 16-byte chunks per tile row
  constexpr int RI = 256 / CI_, RJ = 256 / CJ_;  // rows covered per pass of the 256 threads
  constexpr int NI = BKP / RI, NJ = BKP / RJ;    // passes
  static_assert(NI >= 1 && NJ >= 1, "tile too wide for one pass");
  __shared__ __attribute__((aligned(16))) char lds[BKP * (PI + PJ)];
  char* tI = lds;
  char* tJ = lds + BKP * PI;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int wi = wave & 1, wj = wave >> 1;

  unsigned int tile = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int jt = tile % p.n_jtiles;
  tile /= p.n_jtiles;
  const int it = tile % p.n_itiles;
  const int split = tile / p.n_itiles;
  const int i0 = it * BI, j0 = jt * BJ;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split;
  if (m_end > p.M) m_end = p.M;
  const int HoWo = p.Ho * p.Wo;

  // fixed per-thread columns; all global addressing is 32-bit through bounds-checked buffer loads
  // (out-of-range / padded elements arrive as zeros: offset CN_OOB)
  const cn_buf_t xbuf = cn_make_buf(p.x, p.x_bytes);
  const cn_buf_t dybuf = cn_make_buf(p.dy, p.dy_bytes);
  const int colI = tid % CI_, rowI0 = tid / CI_;
  const int colJ = tid % CJ_, rowJ0 = tid / CJ_;
  const bool validI = i0 + colI * CH < p.Co;
  const int jc = j0 / CH + colJ;  // global column chunk in [tap][ci] space
  const bool validJ = jc < p.ntaps * p.cpt;
  int tap = 0, cchunk = jc;
  if (p.ntaps > 1) {
    tap = validJ ? (int)cn_fastdiv((unsigned)jc, p.div_cpt) : 0;
    cchunk = validJ ? jc - tap * p.cpt : 0;
  }
  const int dhdw = p.tap_dhdw[tap];
  const int dh = (int)(short)(dhdw & 0xffff), dw = dhdw >> 16;
  const unsigned int colI_b = validI ? (unsigned int)((i0 + colI * CH) * EB) : CN_OOB;
  const unsigned int colJ_b = validJ ? (unsigned int)(cchunk * 16) : CN_OOB;
  const unsigned int rowI_pitch = (unsigned int)(p.Co * EB), rowJ_pitch = (unsigned int)(p.Ci * EB);

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  u32x4 regI[NI], regJ[NJ];
  u32x4 regI2[LAZY ? NI : 1];            // lazy dy: the BatchNorm input chunks (LAZY instantiations only: the plain
  constexpr bool lazy = LAZY != 0;       //  kernel keeps its register budget - 3 workgroups per CU)
  __shared__ float s_coef[LAZY == 2 ? 3 * BI : 1];
  const cn_buf_t dy2buf = cn_make_buf(lazy ? p.dy2 : p.dy, p.dy_bytes);
  float lc1[CH], lc2[CH], lc3[CH];   // this thread's channel chunk is fixed: its coefficients live in registers
#pragma unroll
  for (int e = 0; e < CH; ++e) { lc1[e] = 1.f; lc2[e] = 0.f; lc3[e] = 0.f; }
  if (LAZY == 2) {
    for (int c = tid; c < 3 * BI; c += 256) {
      const int k = c / BI, cc = c - k * BI;
      s_coef[c] = (i0 + cc < p.Co) ? p.coef[k * p.Co + i0 + cc] : (k == 0 ? 1.f : 0.f);
    }
    __syncthreads();
  } else if (lazy && validI) {
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      const int c = i0 + colI * CH + e;
      lc1[e] = p.coef[c];
      lc2[e] = p.coef[p.Co + c];
      lc3[e] = p.coef[2 * p.Co + c];
    }
  }
  unsigned int okI = 0;
  auto load_stage = [&](int mb) {
    okI = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = mb + rowI0 + i * RI;
      const bool ok = m < m_end && colI_b < CN_OOB;
      const unsigned int oI = ok ? (unsigned int)m * rowI_pitch + colI_b : CN_OOB;
      okI |= (ok ? 1u : 0u) << i;
      regI[i] = cn_buf_ld16(dybuf, oI);
      if (lazy) regI2[i] = cn_buf_ld16(dy2buf, oI);
    }
    if (p.simple) {
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const int m = mb + rowJ0 + i * RJ;
        const bool ok = m < m_end && colJ_b < CN_OOB;
        const unsigned int oJ = ok ? (unsigned int)m * rowJ_pitch + colJ_b : CN_OOB;
        regJ[i] = cn_buf_ld16(xbuf, oJ);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const int m = mb + rowJ0 + i * RJ;
        bool ok = m < m_end && colJ_b < CN_OOB;
        const int mm = ok ? m : 0;
        const int n = (int)cn_fastdiv((unsigned)mm, p.div_hw);
        const int rem = mm - n * HoWo;
        const int ho = (int)cn_fastdiv((unsigned)rem, p.div_w);
        const int wo = rem - ho * p.Wo;
        const int hi = ho * p.stride_h + dh, wq = wo * p.stride_w + dw;
        ok = ok && (unsigned)hi < (unsigned)p.Hi && (unsigned)wq < (unsigned)p.Wi;
        const unsigned int oJ = ok ? (unsigned int)((n * p.Hi + hi) * p.Wi + wq) * rowJ_pitch + colJ_b : CN_OOB;
        regJ[i] = cn_buf_ld16(xbuf, oJ);
      }
    }
  };
  auto store_stage = [&]() {
    if (lazy) {   // dy = c1*g + c2*y + c3 in bn_bwd_apply_kernel's operation order; rows past the split stay zero
      if (LAZY == 2) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          lc1[e] = s_coef[colI * CH + e];
          lc2[e] = s_coef[BI + colI * CH + e];
          lc3[e] = s_coef[2 * BI + colI * CH + e];
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float g[CH], v[CH];
        Chunk<T>::unpack(regI[i], g);
        Chunk<T>::unpack(regI2[i], v);
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = fmaf(lc1[e], g[e], fmaf(lc2[e], v[e], lc3[e]));
        const u32x4 o = Chunk<T>::pack(g);
        regI[i] = ((okI >> i) & 1u) ? o : cn_zero16();
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) cn_st16(tI + (rowI0 + i * RI) * PI + colI * 16, regI[i]);
#pragma unroll
    for (int i = 0; i < NJ; ++i) cn_st16(tJ + (rowJ0 + i * RJ) * PJ + colJ * 16, regJ[i]);
  };
  auto compute = [&]() {
    if constexpr (kBf16) {
      const int L = lane & 15, g1 = (lane >> 4) & 1, h = lane >> 5;
#pragma unroll
      for (int kk = 0; kk < BKP / 16; ++kk) {
        const int rbase = kk * 16 + h * 8 + (L >> 2);
        const int cbase = g1 * 16 + (L & 3) * 4;
        s16x8 af[TI], bfr[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) {
          const char* q = tI + rbase * PI + ((wi * TI + a) * 32 + cbase) * 2;
          s16x4 lo = cn_lds_read_tr16_b64(q);
          s16x4 hi = cn_lds_read_tr16_b64(q + 4 * PI);
          af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int b = 0; b < TJ; ++b) {
          const char* q = tJ + rbase * PJ + ((wj * TJ + b) * 32 + cbase) * 2;
          s16x4 lo = cn_lds_read_tr16_b64(q);
          s16x4 hi = cn_lds_read_tr16_b64(q + 4 * PJ);
          bfr[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int b = 0; b < TJ; ++b) {
            if constexpr (std::is_same<T, f16_t>::value) acc[a][b] = cn_mfma_32x32x16_f16(af[a], bfr[b], acc[a][b]);
            else acc[a][b] = cn_mfma_32x32x16_bf16(af[a], bfr[b], acc[a][b]);
          }
      }
    } else {
#pragma unroll 4
      for (int kk = 0; kk < BKP / 2; ++kk) {
        const int row = kk * 2 + (lane >> 5);
        float af[TI], bfr[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a)
          af[a] = *(const float*)(tI + row * PI + ((wi * TI + a) * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int b = 0; b < TJ; ++b)
          bfr[b] = *(const float*)(tJ + row * PJ + ((wj * TJ + b) * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int b = 0; b < TJ; ++b) acc[a][b] = cn_mfma_32x32x2_f32(af[a], bfr[b], acc[a][b]);
      }
    }
  };

  if (m_begin < m_end) {
    load_stage(m_begin);
    for (int mb = m_begin; mb < m_end; mb += BKP) {
      store_stage();
      __syncthreads();
      if (mb + BKP < m_end) load_stage(mb + BKP);
      compute();
      __syncthreads();
    }
  }

  // partial tile -> workspace [split][Co][ncols]
  float* out = p.part + (size_t)split * (size_t)p.Co * (size_t)p.ncols;
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int col = j0 + (wj * TJ + b) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = i0 + (wi * TI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Co && col < p.ncols) out[(size_t)co * p.ncols + col] = acc[a][b][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 variant with LDS-DMA staging (`buffer_load ... lds`), double-buffered, one barrier per stage.
// The register-staged kernel above moves every operand byte through ds_write_b128 (13 LDS-path
// cycles per wave instruction): 8 of them per thread and stage = 416 cycles per workgroup-stage on
// top of 256 cycles of transpose reads, against 512 MFMA cycles - the LDS path, not the matrix
// core, was the limiter.  Here the tiles land in LDS without passing through VGPRs.
//
// LDS image of a tile: [64 pixel rows][row bytes], unpadded (a DMA wave instruction fills one
// contiguous KiB).  The transpose read touches 4 consecutive rows x 64 bytes per half-wave, so the
// 64-byte granule index is XOR-swizzled with the row (256-byte rows: ^ (row & 3); 128-byte rows:
// ^ ((row >> 1) & 1)) to keep it bank-conflict free; the swizzle is applied on the *source* side of
// the DMA (the lane that owns LDS slot s fetches global chunk s ^ swizzle).
template <int BI>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(WgradParams p) {
  typedef bf16_t T;
  constexpr int BJ = 128, BKP = 64;
  constexpr int TI = BI / 64, TJ = BJ / 64;
  constexpr int RBI = BI * 2, RBJ = BJ * 2;         // row bytes
  constexpr int SZI = BKP * RBI, SZJ = BKP * RBJ;   // tile bytes per stage
  constexpr int STAGE = SZI + SZJ;
  constexpr int NI = SZI / 1024 / 4, NJ = SZJ / 1024 / 4;   // DMA instructions per thread and stage
  constexpr int CPRI = RBI / 16, CPRJ = RBJ / 16;           // 16-byte chunks per row
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int wi = wave & 1, wj = wave >> 1;

  unsigned int tile = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int jt = tile % p.n_jtiles;
  tile /= p.n_jtiles;
  const int it = tile % p.n_itiles;
  const int split = tile / p.n_itiles;
  const int i0 = it * BI, j0 = jt * BJ;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split;
  if (m_end > p.M) m_end = p.M;
  const int HoWo = p.Ho * p.Wo;

  const cn_buf_t xbuf = cn_make_buf(p.x, p.x_bytes);
  const cn_buf_t dybuf = cn_make_buf(p.dy, p.dy_bytes);
  const unsigned int rowI_pitch = (unsigned int)(p.Co * 2), rowJ_pitch = (unsigned int)(p.Ci * 2);

  // per-thread DMA coordinates: instruction i of wave w fills KiB (i * 4 + w) of the tile
  const int rI = lane / CPRI, sI = lane % CPRI;   // row inside the KiB, slot inside the row
  const int rJ = lane / CPRJ, sJ = lane % CPRJ;
  auto swzI = [](int row) { return BI == 128 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); };
  // rows of one thread differ by multiples of 4 (BI = 128: 1024 / 256) or 8 (BI = 64) between KiB
  // blocks, so the swizzle and hence the fetched chunk column are fixed per thread
  const int cI = sI ^ swzI(rI);
  const int cJ = sJ ^ ((rJ & 3) << 2);
  const bool validI = i0 + cI * 8 < p.Co;
  const unsigned int colI_b = validI ? (unsigned int)((i0 + cI * 8) * 2) : CN_OOB;
  const int jc = j0 / 8 + cJ;
  const bool validJ = jc < p.ntaps * p.cpt;
  int tap = 0, cchunk = jc;
  if (p.ntaps > 1) {
    tap = validJ ? (int)cn_fastdiv((unsigned)jc, p.div_cpt) : 0;
    cchunk = validJ ? jc - tap * p.cpt : 0;
  }
  const int dhdw = p.tap_dhdw[tap];
  const int dh = (int)(short)(dhdw & 0xffff), dw = dhdw >> 16;
  const unsigned int colJ_b = validJ ? (unsigned int)(cchunk * 16) : CN_OOB;

  auto load_stage = [&](int mb, int buf) {
    char* baseI = lds + buf * STAGE;
    char* baseJ = baseI + SZI;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int blk = i * 4 + wave;
      const int m = mb + blk * (1024 / RBI) + rI;
      const bool ok = m < m_end && colI_b < CN_OOB;
      const unsigned int oI = ok ? (unsigned int)m * rowI_pitch + colI_b : CN_OOB;
      cn_buf_ld16_lds(dybuf, oI, baseI + blk * 1024);
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int blk = i * 4 + wave;
      const int m = mb + blk * (1024 / RBJ) + rJ;
      bool ok = m < m_end && colJ_b < CN_OOB;
      unsigned int off;
      if (p.simple) {
        off = (unsigned int)m * rowJ_pitch + colJ_b;
      } else {
        const int mm = ok ? m : 0;
        const int n = (int)cn_fastdiv((unsigned)mm, p.div_hw);
        const int rem = mm - n * HoWo;
        const int ho = (int)cn_fastdiv((unsigned)rem, p.div_w);
        const int wo = rem - ho * p.Wo;
        const int hi = ho * p.stride_h + dh, wq = wo * p.stride_w + dw;
        ok = ok && (unsigned)hi < (unsigned)p.Hi && (unsigned)wq < (unsigned)p.Wi;
        off = (unsigned int)((n * p.Hi + hi) * p.Wi + wq) * rowJ_pitch + colJ_b;
      }
      cn_buf_ld16_lds(xbuf, ok ? off : CN_OOB, baseJ + blk * 1024);
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment addressing (see cn_lds_read_tr16_b64): loop-invariant byte offsets inside a stage
  const int L = lane & 15, g1 = (lane >> 4) & 1, h = lane >> 5;
  const int rlo = h * 8 + (L >> 2);                       // + kk * 16 (+ 4 for the upper half)
  const int inner = g1 * 32 + (L & 3) * 8;                // byte offset inside the 64-byte granule
  int offI[TI], offJ[TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a) {
    const int g = wi * TI + a;
    const int gs = BI == 128 ? (g ^ (rlo & 3)) : (g ^ ((rlo >> 1) & 1));
    offI[a] = rlo * RBI + (gs << 6) + inner;
  }
#pragma unroll
  for (int b = 0; b < TJ; ++b) offJ[b] = rlo * RBJ + (((wj * TJ + b) ^ (rlo & 3)) << 6) + inner;
  // (rows rlo + 4 and rlo + 16*kk keep the same row & 3; for 128-byte rows (row >> 1) & 1 flips with +4
  //  only through bit 2, which the swizzle does not use, so one offset per tile serves all reads)

  auto compute = [&](int buf) {
    const char* tI = lds + buf * STAGE;
    const char* tJ = tI + SZI;
#pragma unroll
    for (int kk = 0; kk < BKP / 16; ++kk) {
      s16x8 af[TI], bfr[TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) {
        const char* q = tI + kk * 16 * RBI + offI[a];
        s16x4 lo = cn_lds_read_tr16_b64(q);
        s16x4 hi = cn_lds_read_tr16_b64(q + 4 * RBI);
        af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < TJ; ++b) {
        const char* q = tJ + kk * 16 * RBJ + offJ[b];
        s16x4 lo = cn_lds_read_tr16_b64(q);
        s16x4 hi = cn_lds_read_tr16_b64(q + 4 * RBJ);
        bfr[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = cn_mfma_32x32x16_bf16(af[a], bfr[b], acc[a][b]);
    }
  };

  if (m_begin < m_end) {
    load_stage(m_begin, 0);
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += BKP, buf ^= 1) {
      __syncthreads();   // stage `buf` has landed (hipcc drains vmcnt first); the other buffer is free
      if (mb + BKP < m_end) load_stage(mb + BKP, buf ^ 1);
      compute(buf);
    }
  }

  float* out = p.part + (size_t)split * (size_t)p.Co * (size_t)p.ncols;
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int col = j0 + (wj * TJ + b) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = i0 + (wi * TI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Co && col < p.ncols) out[(size_t)co * p.ncols + col] = acc[a][b][r];
      }
    }
}

// Fixed-order reduction of the split partials into the KRSC fp32 gradient (C_real <= Ci channels
// kept per tap: the stem's input is channel-padded).  beta = 1 accumulates (chunked batches).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dw, int nsplit, int Co,
                                                          int ntaps, int Ci, int Creal, float beta,
                                                          float scale) {
  const long long total = (long long)Co * ntaps * Creal;
  const long long stride_split = (long long)Co * ntaps * Ci;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * 256) {
    int c = (int)(idx % Creal);
    long long rest = idx / Creal;  // co*ntaps + t
    const float* src = part + rest * Ci + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // 16 loads in flight; summation order is fixed
    int k = 0;
    for (; k + 15 < nsplit; k += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(long long)(k + u) * stride_split];
      s0 += (v[0] + v[4]) + (v[8] + v[12]);
      s1 += (v[1] + v[5]) + (v[9] + v[13]);
      s2 += (v[2] + v[6]) + (v[10] + v[14]);
      s3 += (v[3] + v[7]) + (v[11] + v[15]);
    }
    if (k < nsplit) {   // < 16 slabs left: requested together (clamped index), added in slab order
      float v[15];
#pragma unroll
      for (int u = 0; u < 15; ++u) {
        const int kc = k + u < nsplit ? k + u : nsplit - 1;
        const float x = src[(long long)kc * stride_split];
        v[u] = k + u < nsplit ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 15; ++u) s0 += v[u];
    }
    float s = ((s0 + s1) + (s2 + s3)) * scale;
    dw[idx] = beta != 0.f ? beta * dw[idx] + s : s;
  }
}

// Second stage of every split weight gradient (the tile kernels, the band kernel, the junction pair, the stem).  One thread
// per output on purpose: spreading the split dimension over more lanes makes the launch itself 2x faster and the STEP
// 1 % slower - it puts 4-8x the workgroups beside the backward chain (profiles/r04_ab_wgrad_reduce_lanes_and_side_streams_rejected.txt).
int wg_launch_reduce(hipStream_t stream, const float* part, float* dw, int nsplit, int Co, int ntaps, int Ci,
                     int Creal, float beta, float scale) {
  const long long total = (long long)Co * ntaps * Creal;
  unsigned nb = (unsigned)((total + 255) / 256);
  if (nb > 1024) nb = 1024;   // (few, long workgroups beside the chain: 14.71k vs 14.64k img/s with 8192, 256 the same; profiles/README.md)
  CN_LAUNCH(wgrad_reduce_kernel, dim3(nb), dim3(256), stream, part, dw, nsplit, Co, ntaps, Ci, Creal, beta, scale);
  return cn_check_launch("wgrad_reduce");
}

// ------------------------------------------------------------------------------------------------
// Junction pair (round 3): data gradient AND weight gradient of a 1x1 / stride-1 convolution whose upstream gradient
// is a "lazy dy" (dy = c1*g + c2*y + c3, see WgradParams::dy2), in ONE pass over g and y.
//
// The two lazy kernels (cn_conv2d_dgrad_lazy on the backward chain, cn_conv2d_wgrad_lazy beside it) each read the two
// junction-sized tensors g and y - at the 56x56 layers of ResNet-50 4 x 411 MB per convolution, the largest reads of
// the step.  Here a workgroup owns a range of pixels and ALL channels: per stage of BM pixels it loads g, y (and the
// convolution's input x) once, forms the dy tile in LDS, and feeds it to both products:
//     dx[m][ci]  = sum_co dy[m][co] * W[co][ci]      (MFMA A = W rows (ci) from an LDS copy of the filter, B = dy rows)
//     dW[co][ci] += sum_m dy[m][co] * x[m][ci]       (both operands pixel-major: LDS transpose reads, as wgrad_kernel)
// (reads per pixel: 2*CO + CI elements instead of 4*CO + CI).  The dy tile is stored pixel-major at a pitch of
// CO*2 + 64 bytes (four consecutive rows on four bank quarters: conflict-free transpose reads) with the 16-byte chunk
// index XOR-ed by (row >> 2) & 3 inside each 64-byte granule, which leaves the transpose reads alone (their four rows
// share row >> 2) and spreads the 16 rows a ds_read_b128 of the data-gradient product touches over all 16 bank
// groups.  dx: same operand orientation and k order as igemm_kernel => the bits of cn_conv2d_dgrad_lazy.  dW partials
// per pixel range go to the fp32 workspace; wgrad_reduce_kernel sums them in a fixed order.
struct JbParams {
  const char* g;      // [M][CO] masked upstream gradient of the BatchNorm
  const char* y;      // [M][CO] BatchNorm input
  const float* coef;  // [3][CO]
  const char* w;      // [CI][CO] filter in data-gradient (CRSK) order
  const char* x;      // [M][CI] convolution input
  char* dx;           // [M][CI]
  float* part;        // [nsplit][CO][CI]
  int M, m_per_split, nsplit;
  unsigned int gy_bytes, x_bytes;
};

template <typename T, int CO, int CI, int BM>
__global__ __launch_bounds__(512) void jbwd_kernel(JbParams p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  static_assert(CO % 32 == 0 && CI % 32 == 0 && BM % 64 == 0, "tile shapes");
  constexpr int PD = CO * 2 + 64;   // dy tile pitch (bytes)
  constexpr int PX = CI * 2 + 64;   // x tile pitch
  constexpr int PW = CO * 2;        // filter rows (ci), chunk-swizzled
  constexpr int NCG = CO / 8;       // 16-byte chunks per g / y row
  constexpr int NCX = CI / 8;
  constexpr int RG = 512 / NCG, RX = 512 / NCX;   // rows per load pass
  constexpr int NG = BM / RG, NX = BM / RX;
  static_assert(NG >= 1 && NX >= 1 && BM % RG == 0 && BM % RX == 0, "load passes");
  constexpr int NPT = BM / 32, NCT = CI / 32, NOT = CO / 32;   // 32 x 32 output tiles: dx is NPT x NCT, dW is NOT x NCT
  static_assert(NPT * NCT == 8, "one data-gradient tile per wave");
  static_assert(NOT == 8, "one weight-gradient row of tiles per wave");
  __shared__ __attribute__((aligned(16))) char lds[BM * PD + BM * PX + CI * PW + 3 * CO * 4];
  char* dyT = lds;
  char* xT = lds + BM * PD;
  char* wT = lds + BM * PD + BM * PX;
  float* s_coef = (float*)(lds + BM * PD + BM * PX + CI * PW);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int split = blockIdx.x;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split;
  if (m_end > p.M) m_end = p.M;

  // filter copy (whole launch) and the lazy-dy coefficients
  for (int id = tid; id < CI * NCG; id += 512) {
    const int row = id / NCG, c = id - row * NCG;
    cn_st16(wT + row * PW + ((c ^ (row & (NCG - 1))) << 4), cn_ld16(p.w + ((size_t)row * CO + (size_t)c * 8) * 2));
  }
  for (int c = tid; c < 3 * CO; c += 512) s_coef[c] = p.coef[c];
  __syncthreads();

  const cn_buf_t gbuf = cn_make_buf(p.g, p.gy_bytes);
  const cn_buf_t ybuf = cn_make_buf(p.y, p.gy_bytes);
  const cn_buf_t xbuf = cn_make_buf(p.x, p.x_bytes);
  const int colG = tid % NCG, rowG0 = tid / NCG;
  const int colX = tid % NCX, rowX0 = tid / NCX;

  u32x4 regG[NG], regY[NG], regX[NX];
  unsigned int okG = 0;
  auto load_stage = [&](int mb) {
    okG = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int m = mb + rowG0 + i * RG;
      const bool ok = m < m_end;
      const unsigned int o = ok ? ((unsigned int)m * (unsigned int)CO + (unsigned int)colG * 8u) * 2u : CN_OOB;
      okG |= (ok ? 1u : 0u) << i;
      regG[i] = cn_buf_ld16(gbuf, o);
      regY[i] = cn_buf_ld16(ybuf, o);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int m = mb + rowX0 + i * RX;
      const unsigned int o = m < m_end ? ((unsigned int)m * (unsigned int)CI + (unsigned int)colX * 8u) * 2u : CN_OOB;
      regX[i] = cn_buf_ld16(xbuf, o);
    }
  };
  auto store_stage = [&]() {
    float c1[8], c2[8], c3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      c1[e] = s_coef[colG * 8 + e];
      c2[e] = s_coef[CO + colG * 8 + e];
      c3[e] = s_coef[2 * CO + colG * 8 + e];
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {   // dy = c1*g + c2*y + c3: bn_bwd_apply_kernel's operation order and rounding
      float gg[8], vv[8];
      Chunk<T>::unpack(regG[i], gg);
      Chunk<T>::unpack(regY[i], vv);
#pragma unroll
      for (int e = 0; e < 8; ++e) gg[e] = fmaf(c1[e], gg[e], fmaf(c2[e], vv[e], c3[e]));
      const u32x4 o = Chunk<T>::pack(gg);
      const int row = rowG0 + i * RG;
      const int cs = (colG & ~3) | ((colG & 3) ^ ((row >> 2) & 3));
      cn_st16(dyT + row * PD + (cs << 4), ((okG >> i) & 1u) ? o : cn_zero16());   // rows past the range stay zero
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) cn_st16(xT + (rowX0 + i * RX) * PX + colX * 16, regX[i]);
  };

  f32x16 accw[NCT];
#pragma unroll
  for (int b = 0; b < NCT; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[b][r] = 0.f;
  const int L = lane & 15, g1 = (lane >> 4) & 1, h = lane >> 5;
  const int pt = wave / NCT, ct = wave % NCT;   // this wave's data-gradient tile: pixels pt*32.., input channels ct*32..

  auto mma = [&](const s16x8& a, const s16x8& b, f32x16& c) {
    if constexpr (std::is_same<T, f16_t>::value) c = cn_mfma_32x32x16_f16(a, b, c);
    else c = cn_mfma_32x32x16_bf16(a, b, c);
  };
  auto compute = [&](int mb) {
    // ---- data gradient: D[i = ci][j = pixel], k = co ascending in steps of 16 (igemm_kernel's order)
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
    {
      const int wrow = ct * 32 + (lane & 31);
      const int prow = pt * 32 + (lane & 31);
      const char* wbase = wT + wrow * PW;
      const char* dbase = dyT + prow * PD;
      const int wsw = wrow & (NCG - 1), dsw = (prow >> 2) & 3;
#pragma unroll 4
      for (int kk = 0; kk < CO / 16; ++kk) {
        const int c = 2 * kk + h;
        const s16x8 a = __builtin_bit_cast(s16x8, cn_ld16(wbase + ((c ^ wsw) << 4)));
        const s16x8 b = __builtin_bit_cast(s16x8, cn_ld16(dbase + (((c & ~3) | ((c & 3) ^ dsw)) << 4)));
        mma(a, b, accd);
      }
      const int m = mb + prow;
      if (m < m_end) {
        char* dst = p.dx + ((size_t)m * CI + (size_t)(ct * 32 + 4 * h)) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // 4 consecutive input channels per lane and q
          u32x2 pk;
          pk[0] = cn_pack2<T>(accd[q * 4], accd[q * 4 + 1]);
          pk[1] = cn_pack2<T>(accd[q * 4 + 2], accd[q * 4 + 3]);
          *(u32x2*)(dst + q * 16) = pk;
        }
      }
    }
    // ---- weight gradient: D[i = co][j = ci], k = the BM pixels of this stage
#pragma unroll
    for (int kk = 0; kk < BM / 16; ++kk) {
      const int rbase = kk * 16 + h * 8 + (L >> 2);
      const int cbase = g1 * 16 + (L & 3) * 4;
      const int col = wave * 32 + cbase;            // this wave's 32 output channels
      const int chunk = col >> 3, within = (col & 7) * 2;
      const int sw = (rbase >> 2) & 3;               // (rbase + 4) >> 2 = that + 1
      const char* q0 = dyT + rbase * PD + ((((chunk & ~3) | ((chunk & 3) ^ sw)) << 4) + within);
      const char* q1 = dyT + (rbase + 4) * PD + ((((chunk & ~3) | ((chunk & 3) ^ ((sw + 1) & 3))) << 4) + within);
      const s16x4 lo = cn_lds_read_tr16_b64(q0);
      const s16x4 hi = cn_lds_read_tr16_b64(q1);
      const s16x8 af = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int b = 0; b < NCT; ++b) {
        const char* qx = xT + rbase * PX + (b * 32 + cbase) * 2;
        const s16x4 xl = cn_lds_read_tr16_b64(qx);
        const s16x4 xh = cn_lds_read_tr16_b64(qx + 4 * PX);
        const s16x8 bf = __builtin_shufflevector(xl, xh, 0, 1, 2, 3, 4, 5, 6, 7);
        mma(af, bf, accw[b]);
      }
    }
  };

  if (m_begin < m_end) {
    load_stage(m_begin);
    for (int mb = m_begin; mb < m_end; mb += BM) {
      store_stage();
      __syncthreads();
      if (mb + BM < m_end) load_stage(mb + BM);
      compute(mb);
      __syncthreads();
    }
  }
  float* out = p.part + (size_t)split * (size_t)CO * (size_t)CI;
#pragma unroll
  for (int b = 0; b < NCT; ++b) {
    const int col = b * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(size_t)co * CI + col] = accw[b][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 weight gradient with the activation staged ONCE per band of image rows (round 3).
//
// The tile kernels above treat every tap as its own block of GEMM columns: each (co, tap*ci) tile re-gathers its
// window of x and re-reads its dy rows, so a 3x3 layer moves 9 x (|x| + |dy|) through L2 -> LDS (0.9 - 1.8 GB per launch
// at B = 256) and runs at that rate (105 - 133 us whatever the layer: 430 - 560 TFLOP/s, 0.19 - 0.27 of its roof).
// Here a workgroup owns BI output channels x 32 input channels x ALL NINE taps (9 accumulators of 32 x 32 per wave):
//   * the pixel reduction walks "bands" of whole image rows (<= 112 pixels = 7 MFMA k-steps of 16);
//   * per band the 32-channel slice of x is written to LDS once, as a zero-padded 2-D image: buffer rows = the
//     band's image rows plus one row above / below (the zero row between two images where the band crosses an image
//     boundary), W + 2 positions of 64 bytes per row with zero border columns.  A tap is then a UNIFORM shift of
//     dh*(W+2) + dw positions: no bounds tests, no per-tap gather - one v_add per transpose read;
//   * 64-byte positions put four consecutive pixels on four distinct 64-byte bank quarters, so the
//     ds_read_b64_tr_b16 fragment reads are conflict free without a swizzle;
//   * dy (the band's pixels x BI channels, contiguous rows) arrives by LDS-DMA, double buffered.
// L2 -> LDS traffic per launch: (Ci/32) x |dy| + (Co/BI) x (1 + 2/rows per band) x |x|.
// BI = 128: four waves = four 32-channel blocks of co.  BI = 64 (64-channel layers): two co blocks x two waves
// that split the k-steps of a band (alternating parity per band), summed through LDS at the end.
// Partial sums go to the same [split][Co][9*Ci] workspace as the tile kernels (wgrad_reduce_kernel).
#define WG3_NPX 112
#define WG3_XBYTES 20480
struct Wg3Params {
  const char* x;
  const char* dy;
  float* part;
  int N, H, W, Ci, Co;
  int NR;               // image rows per band (NR * W <= WG3_NPX)
  int rows_per_split;   // global rows (n*H + h) per workgroup: a multiple of NR
  int total_rows;       // N * H
  int n_itiles, n_jtiles;
  unsigned int x_bytes, dy_bytes;
  FastDiv div_hp1, div_w, div_wp2, div_h;
};

template <typename T, int BI>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(Wg3Params p) {
  constexpr int NPX = WG3_NPX, KK = NPX / 16;
  constexpr int RBI = BI * 2;              // dy row bytes in LDS
  constexpr int DYSZ = NPX * RBI;          // one dy band tile
  constexpr int NBLK = DYSZ / 1024;        // KiB blocks of a dy tile (28 / 14)
  constexpr int NDI = (NBLK + 3) / 4;      // DMA instructions per wave and band
  constexpr int CPR = RBI / 16;            // 16-byte chunks per dy row
  constexpr int RPB = 1024 / RBI;          // dy rows per KiB block
  constexpr int NSLOT = WG3_XBYTES / 16 / 256;   // 16-byte x slots per thread and band (5)
  constexpr bool KSPLIT = (BI == 64);
  constexpr int NW = BI / 32;              // co blocks
  // (round 6: padding the allocation so that only ONE workgroup of this kernel fits a CU - half the side stream's footprint, twice
  // its duration - measured 15030 vs 15109 img/s over four interleaved rounds: not kept, profiles/r06_ab_whole_step.txt)
  __shared__ __attribute__((aligned(1024))) char lds[2 * DYSZ + WG3_XBYTES + NPX * 4];
  char* xb = lds + 2 * DYSZ;
  int* s_pos = (int*)(lds + 2 * DYSZ + WG3_XBYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int wi = KSPLIT ? (wave & 1) : wave;     // co block of this wave
  const int wk = KSPLIT ? (wave >> 1) : 0;       // k-step parity of this wave (BI = 64)

  unsigned int tile = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int jt = tile % p.n_jtiles;
  tile /= p.n_jtiles;
  const int it = tile % p.n_itiles;
  const int split = tile / p.n_itiles;
  const int i0 = it * BI, j0 = jt * 32;
  const int g_lo = split * p.rows_per_split;
  int g_hi = g_lo + p.rows_per_split;
  if (g_hi > p.total_rows) g_hi = p.total_rows;
  const int W = p.W, H = p.H, WP = W + 2;

  const cn_buf_t xbuf = cn_make_buf(p.x, p.x_bytes);
  const cn_buf_t dybuf = cn_make_buf(p.dy, p.dy_bytes);
  const unsigned int dy_pitch = (unsigned int)(p.Co * 2), x_pitch = (unsigned int)(p.Ci * 2);

  // dy DMA coordinates (as wgrad_dma_kernel: instruction i of wave w fills KiB block i*4 + w; XOR swizzle of the
  // 64-byte granule applied on the source side)
  const int rI = lane / CPR, sI = lane % CPR;
  const int cI = BI == 128 ? (sI ^ ((rI & 3) << 2)) : (sI ^ (((rI >> 1) & 1) << 2));
  const unsigned int colI_b = (unsigned int)((i0 + cI * 8) * 2);
  auto load_dy = [&](int gb, int buf) {   // the band that starts at global row gb
    char* base = lds + buf * DYSZ;
    int nrow = g_hi - gb;
    if (nrow > p.NR) nrow = p.NR;
    const int npx = nrow > 0 ? nrow * W : 0;
    const unsigned int m0 = (unsigned int)gb * (unsigned int)W;
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const int blk = i * 4 + wave;
      if (NBLK % 4 != 0 && blk >= NBLK) break;
      const int r = blk * RPB + rI;               // pixel index inside the band
      const unsigned int o = r < npx ? (m0 + (unsigned int)r) * dy_pitch + colI_b : CN_OOB;
      cn_buf_ld16_lds(dybuf, o, base + blk * 1024);
    }
  };

  // x band image: slot = (buffer row, position, 16-byte chunk); thread t owns slots t, t + 256, ...
  u32x4 xr[NSLOT];
  int nbr_cur = 0;     // buffer rows of the band held in xr
  auto band_rows = [&](int gb, int& prb0) {   // padded-row range [prb0, prb0 + nbr) of the band starting at gb
    int ge = gb + p.NR;
    if (ge > g_hi) ge = g_hi;
    // padded row of global row g = g + g / H + 1 (one zero row ahead of image 0, one after every image)
    prb0 = gb + (int)cn_fastdiv((unsigned)gb, p.div_h);                     // pr(gb) - 1
    const int pre = (ge - 1) + (int)cn_fastdiv((unsigned)(ge - 1), p.div_h) + 2;   // pr(ge - 1) + 1
    return pre - prb0 + 1;
  };
  auto load_x = [&](int gb) {
    int prb0;
    const int nbr = band_rows(gb, prb0);
    nbr_cur = nbr;
    const int nslots = nbr * WP * 4;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int s = tid + i * 256;
      const int q = s >> 2, chunk = s & 3;                    // position index, chunk
      const int br = (int)cn_fastdiv((unsigned)q, p.div_wp2);
      const int c = q - br * WP;
      const int t = prb0 + br - 1;                            // (padded row - 1) = n * (H + 1) + h
      const int n = t >= 0 ? (int)cn_fastdiv((unsigned)t, p.div_hp1) : 0;
      const int h = t - n * (H + 1);
      const bool ok = (s < nslots) & (t >= 0) & (h < H) & (n < p.N) & (c >= 1) & (c <= W);
      const unsigned int o = ok ? (unsigned int)((n * H + h) * W + (c - 1)) * x_pitch + (unsigned int)((j0 + chunk * 8) * 2)
                                : CN_OOB;
      xr[i] = cn_buf_ld16(xbuf, o);
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int s = tid + i * 256;
      if (s < nbr_cur * WP * 4) cn_st16(xb + s * 16, xr[i]);
    }
  };
  // position (in 64-byte units) of band pixel p inside the x band image; pixels past the band's end point at the
  // image's first interior position (every tap of it lies in rows 0..2, which every band writes): their dy rows are
  // zeros, and 0 x (written, finite data) stays 0 - never 0 x (uninitialised LDS bits)
  auto fill_pos = [&](int gb) {
    if (tid < NPX) {
      int nrow = g_hi - gb;
      if (nrow > p.NR) nrow = p.NR;
      int pos = WP + 1;
      if (tid < nrow * W) {
        const int r = (int)cn_fastdiv((unsigned)tid, p.div_w);
        const int w = tid - r * W;
        const int g = gb + r;
        const int prb0 = gb + (int)cn_fastdiv((unsigned)gb, p.div_h);
        const int br = g + (int)cn_fastdiv((unsigned)g, p.div_h) + 1 - prb0;
        pos = br * WP + w + 1;
      }
      s_pos[tid] = pos;
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // fragment addressing (cn_lds_read_tr16_b64): lane (L, g1, hh) supplies pixel rows kk*16 + hh*8 + (L>>2) (+4)
  const int L = lane & 15, g1 = (lane >> 4) & 1, hh = lane >> 5;
  const int rlo = hh * 8 + (L >> 2);
  const int inner = g1 * 32 + (L & 3) * 8;
  const int gsw = BI == 128 ? (wi ^ (rlo & 3)) : (wi ^ ((rlo >> 1) & 1));
  const int offA = rlo * RBI + (gsw << 6) + inner;      // + kk*16*RBI (+ 4*RBI): the swizzle bits do not change
  int tapsh[9];                                           // byte shift of tap (dh, dw) in the x band image
#pragma unroll
  for (int t = 0; t < 9; ++t) tapsh[t] = ((t / 3 - 1) * WP + (t % 3 - 1)) * 64;

  auto compute = [&](int buf, int parity) {
    const char* tA = lds + buf * DYSZ + offA;
    const char* tX = xb + inner;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (KSPLIT && ((kk + parity) & 1) != wk) continue;     // wave-uniform
      const s16x4 alo = cn_lds_read_tr16_b64(tA + kk * 16 * RBI);
      const s16x4 ahi = cn_lds_read_tr16_b64(tA + kk * 16 * RBI + 4 * RBI);
      const s16x8 af = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
      const int plo = s_pos[kk * 16 + rlo] * 64, phi = s_pos[kk * 16 + rlo + 4] * 64;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const s16x4 blo = cn_lds_read_tr16_b64(tX + plo + tapsh[t]);
        const s16x4 bhi = cn_lds_read_tr16_b64(tX + phi + tapsh[t]);
        const s16x8 bf = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (std::is_same<T, f16_t>::value) acc[t] = cn_mfma_32x32x16_f16(af, bf, acc[t]);
        else acc[t] = cn_mfma_32x32x16_bf16(af, bf, acc[t]);
      }
    }
  };

  if (g_lo < g_hi) {
    load_dy(g_lo, 0);
    load_x(g_lo);
    int buf = 0, parity = 0;
    for (int gb = g_lo; gb < g_hi; gb += p.NR, buf ^= 1, parity ^= 1) {
      __syncthreads();            // every wave is done with the previous band's x image and position table
      store_x();                  // x band image of THIS band (requested one band ago)
      fill_pos(gb);
      __syncthreads();            // image + table visible; dy tile `buf` has landed (requested one band ago)
      if (gb + p.NR < g_hi) {     // next band: dy by DMA into the other buffer, x into registers
        load_dy(gb + p.NR, buf ^ 1);
        load_x(gb + p.NR);
      }
      compute(buf, parity);
    }
  }

  if (KSPLIT) {   // sum the two k-parity waves of each co block through LDS (three accumulators at a time)
    float* red = (float*)lds;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      __syncthreads();
      if (wk == 1) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((wi * 3 + t) * 16 + r) * 64 + lane] = acc[c * 3 + t][r];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c * 3 + t][r] += red[((wi * 3 + t) * 16 + r) * 64 + lane];
      }
    }
    if (wk != 0) return;
  }
  // partial tile -> workspace [split][Co][9 * Ci]   (A rows = co, B columns = ci)
  float* out = p.part + (size_t)split * (size_t)p.Co * (size_t)(9 * p.Ci);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int col = t * p.Ci + j0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      out[(size_t)co * (size_t)(9 * p.Ci) + col] = acc[t][r];
    }
  }
  (void)NW;
}

// ------------------------------------------------------------------------------------------------
// (A 256 x 256 tile on eight waves - half the L2 -> LDS bytes per flop - was built in round 2: faster alone on the gathering
// layers, slower in the step, where a 512-thread workgroup with 128 KiB of LDS evicts the chain's kernels; removed in
// round 4: profiles/r02d_ab_wgrad_256sq_whole_step.txt.)
struct WgradPlan {
  int BI, BJ, BKP, n_itiles, n_jtiles, nsplit, m_per_split, ncols;
};

// Plan of the band kernel (wgrad3x3_kernel); ok = false: the shape is served by the tile kernels
struct Wg3Plan {
  bool ok;
  int BI, NR, rows_per_split, nsplit, n_itiles, n_jtiles;
};
static Wg3Plan wg3_plan(int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                        int dtype) {
  Wg3Plan pl;
  memset(&pl, 0, sizeof(pl));
  if (cn_get_option("wgrad_3x3", 1) == 0) return pl;
  if (!(R == 3 && S == 3 && stride_h == 1 && stride_w == 1 && pad_h == 1 && pad_w == 1)) return pl;
  if (dtype == CN_F32 || K % 64 != 0 || C % 32 != 0 || W > WG3_NPX || H < 1) return pl;
  int NR = WG3_NPX / W;
  if (NR > 16) NR = 16;
  // buffer rows of a band: NR + one above + one below + one zero row per image boundary inside the band
  while (NR >= 1 && (NR + 3 + (NR - 1) / H) * (W + 2) * 64 > WG3_XBYTES) --NR;
  if (NR < 1) return pl;
  pl.BI = K % 128 == 0 ? 128 : 64;
  pl.NR = NR;
  pl.n_itiles = K / pl.BI;
  pl.n_jtiles = C / 32;
  const int tiles = pl.n_itiles * pl.n_jtiles;
  const long long total_rows = (long long)N * H;
  const long long bands = (total_rows + NR - 1) / NR;
  int target = cn_get_option("wgrad_3x3_wgs", 256);      // whole-step A/B (profiles/r03_ab_whole_step_knobs.txt): fewer, longer workgroups intrude less on the main stream
  long long want = (target + tiles - 1) / tiles;          // splits wanted
  long long bps = (bands + want - 1) / want;              // bands per split
  if (bps < 2) bps = bands < 2 ? 1 : 2;
  pl.rows_per_split = (int)(bps * NR);
  pl.nsplit = (int)((total_rows + pl.rows_per_split - 1) / pl.rows_per_split);
  pl.ok = true;
  return pl;
}

static WgradPlan wg_plan(int M, int Co, int ntaps, int Ci, int dtype, bool simple) {
  WgradPlan pl;
  pl.ncols = ntaps * Ci;
  pl.BI = Co <= 64 ? 64 : 128;
  pl.BJ = 128;
  pl.BKP = dtype == CN_F32 ? 32 : 64;
  pl.n_itiles = (Co + pl.BI - 1) / pl.BI;
  pl.n_jtiles = (pl.ncols + pl.BJ - 1) / pl.BJ;
  int tiles = pl.n_itiles * pl.n_jtiles;
  int stages = (M + pl.BKP - 1) / pl.BKP;
  // workgroups per launch: ~2 per CU for the 128-wide tile (fewer, longer splits = less partial traffic),
  // ~4 per CU for the 64-wide one when it gathers (3x3 / 7x7: small LDS / register footprint; identity-gather
  // 64-wide layers measured better at 512: profiles/r01_conv_layers_b256_bf16.txt).
  // Tuning knob "wgrad_target_wgs" overrides both.
  int target = cn_get_option("wgrad_target_wgs", 0);
  if (target <= 0) target = (pl.BI == 64 && !simple) ? 1024 : 512;
  int want = (target + tiles - 1) / tiles;
  int max_split = (stages + 7) / 8;               // at least 8 stages per split
  if (max_split < 1) max_split = 1;
  int nsplit = want < max_split ? want : max_split;
  if (nsplit < 1) nsplit = 1;
  int sps = (stages + nsplit - 1) / nsplit;       // stages per split
  pl.m_per_split = sps * pl.BKP;
  pl.nsplit = (M + pl.m_per_split - 1) / pl.m_per_split;
  return pl;
}

extern "C" size_t cn_conv2d_wgrad_workspace(int N, int H, int W, int C, int K, int R, int S, int stride_h,
                                            int stride_w, int pad_h, int pad_w, int dtype) {
  const int P = (H + 2 * pad_h - R) / stride_h + 1;
  const int Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0 || N <= 0) return 0;
  const bool simple = R == 1 && S == 1 && stride_h == 1 && stride_w == 1 && pad_h == 0 && pad_w == 0;
  WgradPlan pl = wg_plan(N * P * Q, K, R * S, C, dtype, simple);
  size_t need = (size_t)pl.nsplit * (size_t)K * (size_t)pl.ncols * sizeof(float);
  const Wg3Plan p3 = wg3_plan(N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype);
  if (p3.ok) {   // (either kernel may serve the call, depending on the knobs at launch time)
    const size_t n3 = (size_t)p3.nsplit * (size_t)K * (size_t)(9 * C) * sizeof(float);
    if (n3 > need) need = n3;
  }
  return need;
}

template <typename T>
static void wg_launch(const WgradParams& p, const WgradPlan& pl, hipStream_t stream) {
  dim3 grid((unsigned)(pl.n_itiles * pl.n_jtiles * pl.nsplit));
  // bf16, identity gather (1x1 stride 1), more than 64 output channels: LDS-DMA staging (+5-10 % on those layers alone).
  // With a real gather (3x3, strided) the register-staged kernel at 3 workgroups/CU measured 20-30 % faster than the DMA
  // one at 2 (profiles/README.md); for <= 64 output channels (the 256 -> 64 convolutions on the 56x56 maps) the DMA kernel
  // is faster alone (5.6 vs 5.0 TB/s) and the STEP is 0.5 % faster without it - half the LDS beside the chain - so it was
  // removed in round 4 (14760 vs 14678 img/s, four interleaved rounds).  Knob "wgrad_variant": 0 = this heuristic,
  // 1 = register-staged everywhere (tests).
  const int wv = p.dy2 != nullptr ? 1 : cn_get_option("wgrad_variant", 0);   // lazy dy: register-staged only
  if (std::is_same<T, bf16_t>::value && wv == 0 && p.simple && pl.BI != 64) {   // (the LDS-DMA kernel is a bf16 instantiation)
    cn_set_last_kernel("wgrad_dma_kernel<128>");
    CN_LAUNCH((wgrad_dma_kernel<128>), grid, dim3(256), stream, p);
    return;
  }
  cn_set_last_kernel("wgrad_kernel<%s, %d, 128%s>",
                     std::is_same<T, float>::value ? "float" : (std::is_same<T, f16_t>::value ? "f16_t" : "bf16_t"), pl.BI == 64 ? 64 : 128,
                     p.dy2 != nullptr ? ", true" : "");
  if (p.dy2 != nullptr) {
    if (pl.BI == 64) CN_LAUNCH((wgrad_kernel<T, 64, 128, 1>), grid, dim3(256), stream, p);
    else CN_LAUNCH((wgrad_kernel<T, 128, 128, 2>), grid, dim3(256), stream, p);
    return;
  }
  if (pl.BI == 64) CN_LAUNCH((wgrad_kernel<T, 64, 128>), grid, dim3(256), stream, p);
  else CN_LAUNCH((wgrad_kernel<T, 128, 128>), grid, dim3(256), stream, p);
}

static int wg_conv2d_wgrad(const void* x, const void* dy, float* dw_krsc, int C_real, int N, int H, int W,
                           int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                           int dtype, float beta, float scale, void* workspace, size_t ws_bytes,
                           void* stream, const void* lazy_y, const float* lazy_coef);

extern "C" int cn_conv2d_wgrad(const void* x, const void* dy, float* dw_krsc, int C_real, int N, int H, int W,
                               int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                               int dtype, float beta, float scale, void* workspace, size_t ws_bytes,
                               void* stream) {
  return wg_conv2d_wgrad(x, dy, dw_krsc, C_real, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype, beta,
                         scale, workspace, ws_bytes, stream, nullptr, nullptr);
}

// "Lazy dy" weight gradient (see cn_conv2d_dgrad_lazy): dy = c1*g + c2*y + c3 is formed on the operand load of the
// register-staged kernel; same bits as cn_bn_bwd_partials(dy) + cn_conv2d_wgrad(dy) for the kernels it replaces.
extern "C" int cn_conv2d_wgrad_lazy(const void* x, const void* g, const void* bn_y, const float* coef, float* dw_krsc,
                                    int C_real, int N, int H, int W, int C, int K, int R, int S, int stride_h,
                                    int stride_w, int pad_h, int pad_w, int dtype, float beta, float scale,
                                    void* workspace, size_t ws_bytes, void* stream) {
  if (g == nullptr || bn_y == nullptr || coef == nullptr) { cn_set_error("conv2d_wgrad_lazy: needs g, y and the coefficients"); return CN_EINVAL; }
  return wg_conv2d_wgrad(x, g, dw_krsc, C_real, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype, beta, scale,
                         workspace, ws_bytes, stream, bn_y, coef);
}

static int wg_conv2d_wgrad(const void* x, const void* dy, float* dw_krsc, int C_real, int N, int H, int W,
                           int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                           int dtype, float beta, float scale, void* workspace, size_t ws_bytes,
                           void* stream, const void* lazy_y, const float* lazy_coef) {
  const int P = (H + 2 * pad_h - R) / stride_h + 1;
  const int Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0 || N <= 0) { cn_set_error("conv2d_wgrad: empty output"); return CN_ESHAPE; }
  const int CH = cn_dtype_chunk(dtype);
  if (!cn_dtype_ok(dtype)) { cn_set_error("conv2d_wgrad: bad dtype"); return CN_EINVAL; }
  if (C % CH != 0 || K % CH != 0) {
    cn_set_error("conv2d_wgrad: C=%d / K=%d must be multiples of the 16-byte chunk (%d)", C, K, CH);
    return CN_ESHAPE;
  }
  if (R * S > WG_MAX_TAPS) { cn_set_error("conv2d_wgrad: too many taps"); return CN_ESHAPE; }
  if (C_real <= 0 || C_real > C) { cn_set_error("conv2d_wgrad: bad C_real"); return CN_EINVAL; }
  const bool simple_gather = R == 1 && S == 1 && stride_h == 1 && stride_w == 1 && pad_h == 0 && pad_w == 0;
  Wg3Plan p3 = wg3_plan(N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype);
  if (lazy_y != nullptr) p3.ok = false;   // (lazy dy feeds 1x1 layers; the register-staged tile kernel forms it)
  if (p3.ok) {
    // 3x3 / stride 1 / pad 1: the band kernel (activation staged once per band of image rows, all nine taps per tile)
    const size_t need3 = (size_t)p3.nsplit * (size_t)K * (size_t)(9 * C) * sizeof(float);
    const long long xb3 = (long long)N * H * W * C * 2, dyb3 = (long long)N * H * W * K * 2;
    if (ws_bytes < need3 || workspace == nullptr) {
      cn_set_error("conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need3);
      return CN_EWORKSPACE;
    }
    if (xb3 >= (1ll << 31) || dyb3 >= (1ll << 31)) {
      cn_set_error("conv2d_wgrad: operand exceeds the 2 GiB buffer-descriptor window");
      return CN_ESHAPE;
    }
    const int phase3 = cn_get_option("wgrad_phase", 0);
    if (phase3 != 2) {
      Wg3Params q;
      memset(&q, 0, sizeof(q));
      q.x = (const char*)x; q.dy = (const char*)dy; q.part = (float*)workspace;
      q.N = N; q.H = H; q.W = W; q.Ci = C; q.Co = K;
      q.NR = p3.NR; q.rows_per_split = p3.rows_per_split; q.total_rows = N * H;
      q.n_itiles = p3.n_itiles; q.n_jtiles = p3.n_jtiles;
      q.x_bytes = (unsigned int)xb3; q.dy_bytes = (unsigned int)dyb3;
      q.div_hp1 = cn_make_fastdiv((unsigned)(H + 1));
      q.div_w = cn_make_fastdiv((unsigned)W);
      q.div_wp2 = cn_make_fastdiv((unsigned)(W + 2));
      q.div_h = cn_make_fastdiv((unsigned)H);
      dim3 grid((unsigned)(p3.n_itiles * p3.n_jtiles * p3.nsplit));
      cn_set_last_kernel("wgrad3x3_kernel<%s, %d>", dtype == CN_F16 ? "f16_t" : "bf16_t", p3.BI);
      if (dtype == CN_F16) {
        if (p3.BI == 128) CN_LAUNCH((wgrad3x3_kernel<f16_t, 128>), grid, dim3(256), (hipStream_t)stream, q);
        else CN_LAUNCH((wgrad3x3_kernel<f16_t, 64>), grid, dim3(256), (hipStream_t)stream, q);
      } else {
        if (p3.BI == 128) CN_LAUNCH((wgrad3x3_kernel<bf16_t, 128>), grid, dim3(256), (hipStream_t)stream, q);
        else CN_LAUNCH((wgrad3x3_kernel<bf16_t, 64>), grid, dim3(256), (hipStream_t)stream, q);
      }
      int rc3 = cn_check_launch("wgrad3x3");
      if (rc3) return rc3;
    }
    if (phase3 == 1) return CN_OK;
    return wg_launch_reduce((hipStream_t)stream, (const float*)workspace, dw_krsc, p3.nsplit, K, 9, C, C_real, beta, scale);
  }
  WgradPlan pl = wg_plan(N * P * Q, K, R * S, C, dtype, simple_gather);
  size_t need = (size_t)pl.nsplit * (size_t)K * (size_t)pl.ncols * sizeof(float);
  if (ws_bytes < need || workspace == nullptr) {
    cn_set_error("conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
    return CN_EWORKSPACE;
  }
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.dy = (const char*)dy; p.part = (float*)workspace;
  p.N = N; p.Hi = H; p.Wi = W; p.Ci = C; p.Ho = P; p.Wo = Q; p.Co = K;
  p.stride_h = stride_h; p.stride_w = stride_w;
  p.ntaps = R * S; p.cpt = C / CH; p.ncols = pl.ncols;
  p.div_cpt = cn_make_fastdiv((unsigned)p.cpt);
  const long long EBl = cn_dtype_bytes(dtype);
  const long long xb = (long long)N * H * W * C * EBl, dyb = (long long)N * P * Q * K * EBl;
  if (xb >= (1ll << 31) || dyb >= (1ll << 31)) {
    cn_set_error("conv2d_wgrad: operand exceeds the 2 GiB buffer-descriptor window");
    return CN_ESHAPE;
  }
  p.x_bytes = (unsigned int)xb; p.dy_bytes = (unsigned int)dyb;
  p.simple = (R == 1 && S == 1 && stride_h == 1 && stride_w == 1 && pad_h == 0 && pad_w == 0) ? 1 : 0;
  p.dy2 = (const char*)lazy_y; p.coef = lazy_coef;
  p.M = N * P * Q; p.m_per_split = pl.m_per_split; p.nsplit = pl.nsplit;
  p.n_itiles = pl.n_itiles; p.n_jtiles = pl.n_jtiles;
  p.div_hw = cn_make_fastdiv((unsigned)(P * Q));
  p.div_w = cn_make_fastdiv((unsigned)Q);
  for (int r = 0; r < R; ++r)
    for (int s = 0; s < S; ++s) p.tap_dhdw[r * S + s] = ((r - pad_h) & 0xffff) | ((s - pad_w) << 16);
  // measurement only (knob "wgrad_phase"): 1 = the partial-product launch alone, 2 = the reduction launch alone (on the
  // partials a phase-1 call left in the workspace), so that a profiler can time the two kernels of this call separately
  const int phase = cn_get_option("wgrad_phase", 0);
  if (phase != 2) {
    if (dtype == CN_BF16) wg_launch<bf16_t>(p, pl, (hipStream_t)stream);
    else if (dtype == CN_F16) wg_launch<f16_t>(p, pl, (hipStream_t)stream);
    else wg_launch<float>(p, pl, (hipStream_t)stream);
    int rc = cn_check_launch("wgrad");
    if (rc) return rc;
  }
  if (phase == 1) return CN_OK;
  return wg_launch_reduce((hipStream_t)stream, (const float*)workspace, dw_krsc, pl.nsplit, K, R * S, C, C_real, beta, scale);
}

// Junction pair entry point (see jbwd_kernel): shapes it is instantiated for.
extern "C" int cn_conv2d_bwd1x1_lazy_ok(int C, int K, int dtype) {
  return (dtype == CN_BF16 || dtype == CN_F16) && K == 256 && C == 64 ? 1 : 0;
}
static int jb_splits(long long M) {
  int ns = cn_get_option("jbwd_splits", 256);
  if (ns < 1) ns = 1;
  const long long stages = (M + 127) / 128;
  if (ns > stages) ns = (int)stages;
  return ns;
}
extern "C" size_t cn_conv2d_bwd1x1_lazy_workspace(int N, int H, int W, int C, int K) {
  return (size_t)jb_splits((long long)N * H * W) * (size_t)K * (size_t)C * sizeof(float);
}
// dx = conv1x1 data gradient and dw_krsc = beta*dw + scale * weight gradient of y = conv1x1(x, w) (x [N,H,W,C],
// K output channels) for the upstream gradient dy = c1*g + c2*bn_y + c3 (coef = [c1 | c2 | c3], 3*K floats), formed
// on load: cn_conv2d_dgrad_lazy + cn_conv2d_wgrad_lazy in one pass over g and bn_y.  dx has the bits of
// cn_conv2d_dgrad_lazy; dw differs from cn_conv2d_wgrad_lazy by fp32 summation order only (other pixel ranges).
extern "C" int cn_conv2d_bwd1x1_lazy(const void* x, const void* g, const void* bn_y, const float* coef,
                                     const void* w_crsk, void* dx, float* dw_krsc, int N, int H, int W, int C, int K,
                                     int dtype, float beta, float scale, void* workspace, size_t ws_bytes,
                                     void* stream) {
  if (x == nullptr || g == nullptr || bn_y == nullptr || coef == nullptr || w_crsk == nullptr || dx == nullptr ||
      dw_krsc == nullptr) { cn_set_error("conv2d_bwd1x1_lazy: null operand"); return CN_EINVAL; }
  if (!cn_conv2d_bwd1x1_lazy_ok(C, K, dtype)) { cn_set_error("conv2d_bwd1x1_lazy: C=%d K=%d dtype %d is not an instantiated shape", C, K, dtype); return CN_ESHAPE; }
  const long long M = (long long)N * H * W;
  if (M <= 0) { cn_set_error("conv2d_bwd1x1_lazy: empty"); return CN_ESHAPE; }
  const long long gyb = M * K * 2, xb = M * C * 2;
  if (gyb >= (1ll << 31) || xb >= (1ll << 31)) { cn_set_error("conv2d_bwd1x1_lazy: operand exceeds the 2 GiB buffer-descriptor window"); return CN_ESHAPE; }
  const int ns0 = jb_splits(M);
  long long mps = (M + ns0 - 1) / ns0;
  mps = (mps + 127) / 128 * 128;
  const int nsplit = (int)((M + mps - 1) / mps);
  if (workspace == nullptr || ws_bytes < (size_t)nsplit * K * C * sizeof(float)) { cn_set_error("conv2d_bwd1x1_lazy: workspace too small"); return CN_EWORKSPACE; }
  JbParams p;
  memset(&p, 0, sizeof(p));
  p.g = (const char*)g; p.y = (const char*)bn_y; p.coef = coef; p.w = (const char*)w_crsk; p.x = (const char*)x;
  p.dx = (char*)dx; p.part = (float*)workspace;
  p.M = (int)M; p.m_per_split = (int)mps; p.nsplit = nsplit;
  p.gy_bytes = (unsigned int)gyb; p.x_bytes = (unsigned int)xb;
  hipStream_t st = (hipStream_t)stream;
  const int phase = cn_get_option("wgrad_phase", 0);   // measurement only, as cn_conv2d_wgrad
  if (phase != 2) {
    cn_set_last_kernel("jbwd_kernel<%s, 256, 64, 128>", dtype == CN_F16 ? "f16_t" : "bf16_t");
    if (dtype == CN_F16) CN_LAUNCH((jbwd_kernel<f16_t, 256, 64, 128>), dim3((unsigned)nsplit), dim3(512), st, p);
    else CN_LAUNCH((jbwd_kernel<bf16_t, 256, 64, 128>), dim3((unsigned)nsplit), dim3(512), st, p);
    int rc = cn_check_launch("jbwd");
    if (rc) return rc;
  }
  if (phase == 1) return CN_OK;
  return wg_launch_reduce(st, (const float*)workspace, dw_krsc, nsplit, K, 1, C, C, beta, scale);
}
