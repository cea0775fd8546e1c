// junction.hip -- the data gradient of a residual block's first 1x1 convolution fused with everything that meets it at
// the block input, as ONE streaming kernel (round 3; replaces the igemm_kernel<..., EPI> launches of the large
// junctions).
//
// What happens at the block input in backward (/root/reference models/resnet.py:141-165 run in reverse by
// loss.backward(), trainer.py:162): the gradient of conv1's input, dx = dy1 * W, is added to the gradient arriving over
// the shortcut; the sum is masked by the ReLU of the junction that produced the block input; the result g is the
// upstream gradient of that junction's BatchNorm, whose backward needs sum(g) and sum(g * xhat) per channel.
// cn_conv2d_dgrad_bnbwd(_sa) does all of that in the epilogue of the tiled implicit-GEMM kernel.  That kernel is a GEMM
// kernel: one 128 x 128 output tile per workgroup, half a dozen workgroup barriers per tile, a fresh workgroup (launch,
// index tables) every ~100 KB - and this operation is not a GEMM but a STREAM: per output element 2 bytes of addend, 2
// of BatchNorm input, 2 of g and 1/8 of mask against 2*KD flops with KD = 64 ... 128, i.e. 6+ bytes per 128 - 256 flops.
// It ran at 3.5 TB/s alone where the streaming kernels of this library reach 5.
//
// Here a workgroup is persistent over a contiguous range of pixels and owns ALL CO output channels:
//   * the filter (CO x KD) lives in registers as MFMA A-fragments for the whole launch (a wave owns 64 channels);
//   * per stage of BM = 32 or 64 pixels the dy tile (BM x KD, a few KB) goes through a double-buffered LDS tile - one
//     workgroup barrier per stage -; every wave multiplies its 64 channels x 32 pixels;
//   * the accumulators are transposed to pixel rows through a WAVE-PRIVATE LDS patch (no workgroup barrier): a lane then
//     holds 16 bytes of 8 consecutive channels of one pixel, so the addend / BatchNorm-input loads and the g stores are
//     16 bytes per lane and 128 contiguous bytes per pixel row, and they are requested a stage AHEAD;
//   * the BatchNorm-backward sums stay in registers for the whole pixel range: one partial row per workgroup (a few
//     hundred rows per launch instead of one per 128 pixels: no row compression before the finalize).
// g has the bits of cn_conv2d_dgrad_bnbwd_sa (same operand orientation, k order and rounding sequence); the partial sums
// are associated differently (fp32).
#include "cn_common.h"
#include "cn_api_internal.h"
#include <type_traits>

struct JdParams {
  const char* dy;                // [M][KD] upstream gradient of conv1's output
  const char* w;                 // [CO][KD] filter in data-gradient order (CRSK of a 1x1 convolution)
  const char* addend;            // [M][CO], or [N][add_H][add_W][CO] (addend_sub = 2: even (h, w) pixels only)
  const char* bn_y;              // [M][CO] input of the BatchNorm whose output is conv1's input
  const unsigned char* bn_mask;  // [M][CO / 8] ReLU bits of that junction
  const float* bn_coef;          // [mean | invstd | scale | shift]
  char* g;                       // [M][CO]
  float* partial;                // [nsplit][2 * CO]
  int M, m_per_split, nsplit;
  int co_total;                  // channels of g (a workgroup computes CO of them, from blockIdx.y * CO on)
  int addend_sub, H, W, add_H, add_W;
  FastDiv div_hw, div_w;
  unsigned int dy_bytes, out_bytes, add_bytes, mask_bytes;
};

// The epilogue operands of stage s + 1 are requested before stage s is processed (72 registers).  A workgroup computes
// the CO channels from blockIdx.y * CO on of p.co_total.
template <typename T, int KD, int CO>
__global__ __launch_bounds__(512) void jdgrad_kernel(JdParams p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int NCW = CO / 64;          // wave columns of 64 channels
  static_assert(NCW == 4 || NCW == 8, "256 or 512 output channels");
  constexpr int PH = 8 / NCW;           // 32-pixel groups per stage
  constexpr int BM = 32 * PH;
  constexpr int NKK = KD / 16;
  constexpr int NCD = KD / 8;           // 16-byte chunks per dy row
  constexpr int DYB = BM * KD * 2;      // bytes of one dy tile
  constexpr int ND = BM * NCD / 512;    // dy chunks staged per thread
  static_assert(ND >= 1 && BM * NCD % 512 == 0, "dy tile staging");
  constexpr int PP = 144;               // pitch of the wave-private transposition patch (32 rows of 128 + 16 bytes)
  constexpr int PRIV = 32 * PP;
  __shared__ __attribute__((aligned(16))) char lds[2 * DYB + 8 * PRIV];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int cw = wave % NCW, ph = wave / NCW;
  const int h = lane >> 5;
  char* priv = lds + 2 * DYB + wave * PRIV;

  const int split = blockIdx.x;
  const int co0 = blockIdx.y * CO, COT = p.co_total;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split;
  if (m_end > p.M) m_end = p.M;

  // filter fragments: this wave's 2 x 32 channels (MFMA rows), all of k
  s16x8 wf[2][NKK];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int c = co0 + cw * 64 + t * 32 + (lane & 31);
      wf[t][kk] = __builtin_bit_cast(s16x8, cn_ld16(p.w + ((size_t)c * KD + 16 * kk + 8 * h) * 2));
    }

  const cn_buf_t dybuf = cn_make_buf(p.dy, p.dy_bytes);
  const cn_buf_t abuf = cn_make_buf(p.addend, p.add_bytes);
  const cn_buf_t ybuf = cn_make_buf(p.bn_y, p.out_bytes);
  // epilogue coordinates of this lane: pixel rows pass*8 + (lane >> 3) of the wave's 32, chunk (8 channels) lane & 7
  const int ech = lane & 7, erow = lane >> 3;
  const int cb = co0 + cw * 64 + ech * 8;     // first channel of the lane's chunk
  const int HW = p.H * p.W;

  struct Epi {
    u32x4 a[4], y[4];
    unsigned int bits[4];
  };
  u32x4 dreg[ND];
  auto load_dy = [&](int mb) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = tid + 512 * i;
      const int row = id / NCD, c = id - row * NCD;
      const int m = mb + row;
      dreg[i] = cn_buf_ld16(dybuf, m < m_end ? ((unsigned int)m * (unsigned int)KD + (unsigned int)c * 8u) * 2u : CN_OOB);
    }
  };
  auto load_epi = [&](int mb, Epi& e) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int m = mb + ph * 32 + k * 8 + erow;
      const bool ok = m < m_end;
      const unsigned int o = ok ? ((unsigned int)m * (unsigned int)COT + (unsigned int)cb) * 2u : CN_OOB;
      unsigned int oa = o;
      if (p.addend_sub == 2) {   // the addend exists at even (h, w) only, stored compactly
        const int mm = ok ? m : 0;
        const int n = (int)cn_fastdiv((unsigned)mm, p.div_hw);
        const int rem = mm - n * HW;
        const int ho = (int)cn_fastdiv((unsigned)rem, p.div_w);
        const int wo = rem - ho * p.W;
        const bool even = ((ho | wo) & 1) == 0;
        const int apx = (n * p.add_H + (ho >> 1)) * p.add_W + (wo >> 1);
        oa = (ok && even) ? ((unsigned int)apx * (unsigned int)COT + (unsigned int)cb) * 2u : CN_OOB;
      }
      e.a[k] = cn_buf_ld16(abuf, oa);
      e.y[k] = cn_buf_ld16(ybuf, o);
      e.bits[k] = ok ? (unsigned int)p.bn_mask[(size_t)m * (COT / 8) + (cb >> 3)] : 0u;
    }
  };
  auto store_dy = [&](int buf) {
    char* t = lds + buf * DYB;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = tid + 512 * i;
      const int row = id / NCD, c = id - row * NCD;
      const int cs = NCD == 8 ? (c ^ ((row >> 1) & 7)) : (c ^ (row & (NCD - 1)));
      cn_st16(t + row * (KD * 2) + (cs << 4), dreg[i]);
    }
  };

  float bs1[8], bs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }

  auto compute = [&](int mb, int buf, const Epi& ep) {
    const char* t = lds + buf * DYB;
    f32x16 acc[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    const int prow = ph * 32 + (lane & 31);
    const char* rowp = t + prow * (KD * 2);
    const int sw = NCD == 8 ? ((prow >> 1) & 7) : (prow & (NCD - 1));
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const s16x8 b = __builtin_bit_cast(s16x8, cn_ld16(rowp + (((2 * kk + h) ^ sw) << 4)));
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if constexpr (std::is_same<T, f16_t>::value) acc[x] = cn_mfma_32x32x16_f16(wf[x][kk], b, acc[x]);
        else acc[x] = cn_mfma_32x32x16_bf16(wf[x][kk], b, acc[x]);
      }
    }
    // accumulators -> stored precision -> wave-private patch [32 pixels][64 channels]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x2 pk;
        pk[0] = cn_pack2<T>(acc[x][q * 4], acc[x][q * 4 + 1]);
        pk[1] = cn_pack2<T>(acc[x][q * 4 + 2], acc[x][q * 4 + 3]);
        *(u32x2*)(priv + (lane & 31) * PP + (x * 32 + 8 * q + 4 * h) * 2) = pk;
      }
    cn_wave_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int m = mb + ph * 32 + k * 8 + erow;
      const u32x4 v0 = cn_ld16(priv + (k * 8 + erow) * PP + ech * 16);
      if (m < m_end) {
        float fv[8], fa[8], yv[8];
        Chunk<T>::unpack(v0, fv);
        Chunk<T>::unpack(ep.a[k], fa);
        Chunk<T>::unpack(ep.y[k], yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] += fa[e];
        const unsigned int bits = ep.bits[k];
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] = ((bits >> e) & 1u) ? fv[e] : 0.f;
        const u32x4 v = Chunk<T>::pack(fv);
        Chunk<T>::unpack(v, fv);     // statistics of the values as stored
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bs1[e] += fv[e];
          bs2[e] = fmaf(fv[e], yv[e], bs2[e]);
        }
        cn_st16(p.g + ((size_t)m * COT + (size_t)cb) * 2, v);
      }
    }
    cn_wave_sync();   // the patch is rewritten by the next stage
  };

  if (m_begin < m_end) {
    Epi cur, nxt;
    load_dy(m_begin);
    load_epi(m_begin, cur);
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += BM) {
      store_dy(buf);
      __syncthreads();    // (two dy tiles: the tile of stage s + 1 is written while stage s is still being read: one barrier per stage)
      const bool more = mb + BM < m_end;
      if (more) { load_dy(mb + BM); load_epi(mb + BM, nxt); }
      compute(mb, buf, cur);
      if (more) cur = nxt;
      buf ^= 1;
    }
  }
  // ---- BatchNorm-backward partial sums of this workgroup's pixel range: [sum g | sum g * xhat]
  float r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float mu = p.bn_coef[cb + e], is = p.bn_coef[COT + cb + e];
    r1[e] = bs1[e];
    r2[e] = is * (bs2[e] - mu * bs1[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int msk = 8; msk <= 32; msk <<= 1) {   // the eight row groups of the wave hold the same channels
      r1[e] += cn_shfl_xor(r1[e], msk);
      r2[e] += cn_shfl_xor(r2[e], msk);
    }
  __syncthreads();
  float* red = (float*)lds;   // [8 waves][64 channels][2]
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(wave * 64 + lane * 8 + e) * 2] = r1[e];
      red[(wave * 64 + lane * 8 + e) * 2 + 1] = r2[e];
    }
  }
  __syncthreads();
  if (tid < CO) {
    const int c = tid, wcol = c / 64, cc = c % 64;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < PH; ++g2) {   // fixed order
      a1 += red[((g2 * NCW + wcol) * 64 + cc) * 2];
      a2 += red[((g2 * NCW + wcol) * 64 + cc) * 2 + 1];
    }
    float* dst = p.partial + (size_t)split * 2 * COT;
    dst[co0 + c] = a1;
    dst[COT + co0 + c] = a2;
  }
}

// The 256-channel reductions (the 28x28 -> 14x14 and the 14x14 junctions: K = 256 gradient channels, C = 512 / 1024) with
// a wave owning 32 channels instead of 64: the filter fragments of a wave are 64 registers instead of 128, so the
// stage-ahead prefetch of the epilogue operands fits without spills (the 64-channel-wave form needs 128 filter registers and
// still spills 20 registers).  A workgroup owns a pixel range and a 256-channel slice (blockIdx.y) - eight waves of 32
// channels, all 32 pixels of a stage each -; the dy tile (16 KB per stage) is read once per slice (C / 256 times, from
// L2 after the first).  Epilogue accesses are 64 contiguous bytes per pixel row and wave.  Same g bits.
template <typename T, int KD>
__global__ __launch_bounds__(512) void jdgrad_w32_kernel(JdParams p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int CO = 256, BM = 32;
  constexpr int NKK = KD / 16;
  constexpr int NCD = KD / 8;
  constexpr int DYB = BM * KD * 2;
  constexpr int ND = BM * NCD / 512;
  static_assert(ND >= 1 && BM * NCD % 512 == 0 && 512 % NCD == 0, "dy tile staging");
  constexpr int PP = 80;                // patch pitch: 32 channels * 2 bytes + 16
  constexpr int PRIV = 32 * PP;
  __shared__ __attribute__((aligned(16))) char lds[2 * DYB + 8 * PRIV];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int h = lane >> 5;
  char* priv = lds + 2 * DYB + wave * PRIV;
  const int split = blockIdx.x;
  const int co0 = blockIdx.y * CO, COT = p.co_total;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split;
  if (m_end > p.M) m_end = p.M;

  s16x8 wf[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    const int c = co0 + wave * 32 + (lane & 31);
    wf[kk] = __builtin_bit_cast(s16x8, cn_ld16(p.w + ((size_t)c * KD + 16 * kk + 8 * h) * 2));
  }
  const cn_buf_t dybuf = cn_make_buf(p.dy, p.dy_bytes);
  const cn_buf_t abuf = cn_make_buf(p.addend, p.add_bytes);
  const cn_buf_t ybuf = cn_make_buf(p.bn_y, p.out_bytes);
  const int ech = lane & 3, erow = lane >> 2;      // epilogue: pixel rows k*16 + erow, chunk (8 channels) ech of the wave's 4
  const int cb = co0 + wave * 32 + ech * 8;
  const int HW = p.H * p.W;
  struct Epi {
    u32x4 a[2], y[2];
    unsigned int bits[2];
  };
  u32x4 dreg[ND];
  auto load_dy = [&](int mb) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = tid + 512 * i;
      const int row = id / NCD, c = id - row * NCD;
      const int m = mb + row;
      dreg[i] = cn_buf_ld16(dybuf, m < m_end ? ((unsigned int)m * (unsigned int)KD + (unsigned int)c * 8u) * 2u : CN_OOB);
    }
  };
  auto load_epi = [&](int mb, Epi& e) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m = mb + k * 16 + erow;
      const bool ok = m < m_end;
      const unsigned int o = ok ? ((unsigned int)m * (unsigned int)COT + (unsigned int)cb) * 2u : CN_OOB;
      unsigned int oa = o;
      if (p.addend_sub == 2) {
        const int mm = ok ? m : 0;
        const int n = (int)cn_fastdiv((unsigned)mm, p.div_hw);
        const int rem = mm - n * HW;
        const int ho = (int)cn_fastdiv((unsigned)rem, p.div_w);
        const int wo = rem - ho * p.W;
        const bool even = ((ho | wo) & 1) == 0;
        const int apx = (n * p.add_H + (ho >> 1)) * p.add_W + (wo >> 1);
        oa = (ok && even) ? ((unsigned int)apx * (unsigned int)COT + (unsigned int)cb) * 2u : CN_OOB;
      }
      e.a[k] = cn_buf_ld16(abuf, oa);
      e.y[k] = cn_buf_ld16(ybuf, o);
      e.bits[k] = ok ? (unsigned int)p.bn_mask[(size_t)m * (COT / 8) + (cb >> 3)] : 0u;
    }
  };
  auto store_dy = [&](int buf) {
    char* t = lds + buf * DYB;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = tid + 512 * i;
      const int row = id / NCD, c = id - row * NCD;
      cn_st16(t + row * (KD * 2) + ((c ^ (row & (NCD - 1))) << 4), dreg[i]);
    }
  };
  float bs1[8], bs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; }

  auto compute = [&](int mb, int buf, const Epi& ep) {
    const char* t = lds + buf * DYB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int prow = lane & 31;
    const char* rowp = t + prow * (KD * 2);
    const int sw = prow & (NCD - 1);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const s16x8 b = __builtin_bit_cast(s16x8, cn_ld16(rowp + (((2 * kk + h) ^ sw) << 4)));
      if constexpr (std::is_same<T, f16_t>::value) acc = cn_mfma_32x32x16_f16(wf[kk], b, acc);
      else acc = cn_mfma_32x32x16_bf16(wf[kk], b, acc);
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
      const int m = mb + pl;
      const u32x4 v0 = cn_ld16(priv + pl * PP + ech * 16);
      if (m < m_end) {
        float fv[8], fa[8], yv[8];
        Chunk<T>::unpack(v0, fv);
        Chunk<T>::unpack(ep.a[k], fa);
        Chunk<T>::unpack(ep.y[k], yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] += fa[e];
        const unsigned int bits = ep.bits[k];
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] = ((bits >> e) & 1u) ? fv[e] : 0.f;
        const u32x4 v = Chunk<T>::pack(fv);
        Chunk<T>::unpack(v, fv);     // statistics of the values as stored
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bs1[e] += fv[e];
          bs2[e] = fmaf(fv[e], yv[e], bs2[e]);
        }
        cn_st16(p.g + ((size_t)m * COT + (size_t)cb) * 2, v);
      }
    }
    cn_wave_sync();
  };

  if (m_begin < m_end) {
    Epi cur, nxt;
    load_dy(m_begin);
    load_epi(m_begin, cur);
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += BM) {
      store_dy(buf);
      __syncthreads();
      const bool more = mb + BM < m_end;
      if (more) { load_dy(mb + BM); load_epi(mb + BM, nxt); }
      compute(mb, buf, cur);
      if (more) cur = nxt;
      buf ^= 1;
    }
  }
  float r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float mu = p.bn_coef[cb + e], is = p.bn_coef[COT + cb + e];
    r1[e] = bs1[e];
    r2[e] = is * (bs2[e] - mu * bs1[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int msk = 4; msk <= 32; msk <<= 1) {   // the sixteen row groups of the wave hold the same channels
      r1[e] += cn_shfl_xor(r1[e], msk);
      r2[e] += cn_shfl_xor(r2[e], msk);
    }
  if (lane < 4) {     // a channel belongs to exactly one wave: no cross-wave sum
    float* dst = p.partial + (size_t)split * 2 * COT;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dst[cb + e] = r1[e];
      dst[COT + cb + e] = r2[e];
    }
  }
}

static int jd_splits(long long M, int BM) {
  int ns = cn_get_option("jdgrad_splits", 256);
  if (ns < 1) ns = 1;
  const long long stages = (M + BM - 1) / BM;
  if (ns > stages) ns = (int)stages;
  return ns;
}
static int jd_bm(int C) { return C == 256 ? 64 : 32; }   // (C >= 512: 512-channel slices of 32-pixel stages)
// pixel ranges: whole stages per workgroup; returns the number of ranges (= partial rows) and their length
static int jd_plan(long long M, int BM, long long* mps_out, int slices = 1) {
  int ns0 = jd_splits(M, BM);
  if (slices > 1) { ns0 = (ns0 + slices - 1) / slices; if (ns0 < 1) ns0 = 1; }   // (slices x ranges workgroups in all)
  long long mps = (M + ns0 - 1) / ns0;
  mps = (mps + BM - 1) / BM * BM;
  if (mps_out != nullptr) *mps_out = mps;
  return (int)((M + mps - 1) / mps);
}

// Shapes the streaming junction kernel is instantiated for: K channels of dy (conv1's outputs), C channels of g.
extern "C" int cn_conv2d_dgrad_junction_ok(int C, int K, int dtype) {
  if (dtype != CN_BF16 && dtype != CN_F16) return 0;
  // 256-channel reductions (the 28x28 -> 14x14 and 14x14 junctions): jdgrad_w32_kernel, a wave owns 32 channels.  (Measured
  // and not kept: the same reductions on 64-channel waves - 128 filter registers, 20 spilled, step neutral - and the
  // last stage's 512-channel reductions on 32-channel waves - 28 spilled, step +0.3 %: they stay on the tiled kernel;
  // profiles/r03_ab_second_session_whole_step.txt.)
  if ((C == 512 || C == 1024) && K == 256) return 1;
  return ((C == 256 && (K == 64 || K == 128)) || (C == 512 && K == 128)) ? 1 : 0;
}
extern "C" int cn_conv2d_dgrad_junction_rows(int N, int H, int W, int C) {   // (the K <= 128 forms)
  return jd_plan((long long)N * H * W, jd_bm(C), nullptr);
}
// partial rows cn_conv2d_dgrad_junction writes for K -> C channels (the 256-channel reductions plan per channel slice)
extern "C" int cn_conv2d_dgrad_junction_rows_k(int N, int H, int W, int C, int K) {
  return jd_plan((long long)N * H * W, jd_bm(C), nullptr, K >= 256 ? C / 256 : 1);
}

// cn_conv2d_dgrad_bnbwd_sa for a 1x1 / stride-1 / unpadded convolution with K -> C channels of an instantiated shape,
// ReLU bits given (bn_mask) and an addend (dense, or addend_sub = 2: the even pixels of a stride-2 projection's
// gradient), as one persistent streaming kernel (see the head of this file).  partial: cn_conv2d_dgrad_junction_rows
// rows of 2*C floats for cn_bn_bwd_partials.  g: the bits of cn_conv2d_dgrad_bnbwd_sa.
extern "C" int cn_conv2d_dgrad_junction(const void* dy, const void* w_crsk, void* g, const void* addend, int addend_sub,
                                        int N, int H, int W, int C, int K, int dtype, const void* bn_y,
                                        const unsigned char* bn_mask, const float* bn_coef, float* partial,
                                        int partial_rows, void* stream) {
  if (!cn_conv2d_dgrad_junction_ok(C, K, dtype)) { cn_set_error("conv2d_dgrad_junction: K=%d -> C=%d dtype %d is not an instantiated shape", K, C, dtype); return CN_ESHAPE; }
  if (dy == nullptr || w_crsk == nullptr || g == nullptr || addend == nullptr || bn_y == nullptr || bn_mask == nullptr ||
      bn_coef == nullptr || partial == nullptr) { cn_set_error("conv2d_dgrad_junction: null operand"); return CN_EINVAL; }
  if (addend_sub != 1 && addend_sub != 2) { cn_set_error("conv2d_dgrad_junction: addend subsampling %d (1 or 2)", addend_sub); return CN_EINVAL; }
  const long long M = (long long)N * H * W;
  if (M <= 0) { cn_set_error("conv2d_dgrad_junction: empty"); return CN_ESHAPE; }
  const long long ob = M * C * 2, db = M * K * 2;
  const int aH = (H + 1) / 2, aW = (W + 1) / 2;
  const long long ab = addend_sub == 2 ? (long long)N * aH * aW * C * 2 : ob;
  if (ob >= (1ll << 31) || db >= (1ll << 31)) { cn_set_error("conv2d_dgrad_junction: operand exceeds the 2 GiB buffer-descriptor window"); return CN_ESHAPE; }
  const int BM = jd_bm(C);
  const bool w32 = K >= 256;
  long long mps = 0;
  const int nsplit = jd_plan(M, BM, &mps, w32 ? C / 256 : 1);
  if (partial_rows < nsplit) { cn_set_error("conv2d_dgrad_junction: partial buffer of %d rows < %d", partial_rows, nsplit); return CN_EWORKSPACE; }
  JdParams p;
  memset(&p, 0, sizeof(p));
  p.dy = (const char*)dy; p.w = (const char*)w_crsk; p.addend = (const char*)addend; p.bn_y = (const char*)bn_y;
  p.bn_mask = bn_mask; p.bn_coef = bn_coef; p.g = (char*)g; p.partial = partial;
  p.M = (int)M; p.m_per_split = (int)mps; p.nsplit = nsplit;
  p.co_total = C;
  p.addend_sub = addend_sub; p.H = H; p.W = W; p.add_H = aH; p.add_W = aW;
  p.div_hw = cn_make_fastdiv((unsigned)(H * W)); p.div_w = cn_make_fastdiv((unsigned)W);
  p.dy_bytes = (unsigned int)db; p.out_bytes = (unsigned int)ob; p.add_bytes = (unsigned int)ab;
  hipStream_t st = (hipStream_t)stream;
  const char* tn = dtype == CN_F16 ? "f16_t" : "bf16_t";
  if (w32) {
    cn_set_last_kernel("jdgrad_w32_kernel<%s, %d> [%d channels]", tn, K, C);
    dim3 g32((unsigned)nsplit, (unsigned)(C / 256));
    if (dtype == CN_F16) CN_LAUNCH((jdgrad_w32_kernel<f16_t, 256>), g32, dim3(512), st, p);
    else CN_LAUNCH((jdgrad_w32_kernel<bf16_t, 256>), g32, dim3(512), st, p);
    return cn_check_launch("jdgrad_w32");
  }
  cn_set_last_kernel("jdgrad_kernel<%s, %d, %d>", tn, K, C);
  dim3 grid((unsigned)nsplit, (unsigned)(C > 512 ? C / 512 : 1));
#define JD_GO(KD, CO)                                                                           \
  do {                                                                                          \
    if (dtype == CN_F16) CN_LAUNCH((jdgrad_kernel<f16_t, KD, CO>), grid, dim3(512), st, p);      \
    else CN_LAUNCH((jdgrad_kernel<bf16_t, KD, CO>), grid, dim3(512), st, p);                     \
  } while (0)
  if (C == 256 && K == 64) JD_GO(64, 256);
  else if (C == 256 && K == 128) JD_GO(128, 256);
  else JD_GO(128, 512);
#undef JD_GO
  return cn_check_launch("jdgrad");
}

// ------------------------------------------------------------------------------------------------
// The forward counterpart for the block's LAST 1x1 convolution (conv3, the stride-1 projection: K -> 4K channels,
// /root/reference models/resnet.py:130-132,176-181): the same streaming structure - filter in registers, a pixel range
// and all CO channels per workgroup, wave-private transposition, 16-byte stores of full pixel rows - with the
// BatchNorm statistics of the stored values (cn_conv2d_fwd_bnstats' epilogue) kept in registers: one partial row
// [sum | sum of squares] per workgroup.  Output bits = cn_conv2d_fwd's.
struct JfParams {
  const char* x;      // [M][KD]
  const char* w;      // [CO][KD] (KRSC of a 1x1 convolution)
  char* y;            // [M][CO]
  float* partial;     // [nsplit][2 * co_total]
  int M, m_per_split, nsplit;
  int co_total;       // output channels of the convolution; a workgroup computes the CO of them from blockIdx.y * CO on
  unsigned int x_bytes;
  // XF ("lazy a"): x is the INPUT of the BatchNorm in front of this convolution; the kernel forms
  // a = relu?(x * scale + shift) on its way to the LDS tile (bn_apply_kernel's arithmetic) and writes it to a_out
  const float* xf;    // [scale | shift] (2 * KD)
  char* a_out;        // [M][KD]
  int relu;
};

template <typename T, int KD, int CO, bool XF = false>
__global__ __launch_bounds__(512) void jfwd_kernel(JfParams p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int NCW = CO / 64;
  static_assert(NCW == 4 || NCW == 8, "256 or 512 output channels");
  constexpr int PH = 8 / NCW;
  constexpr int BM = 32 * PH;
  constexpr int NKK = KD / 16;
  constexpr int NCD = KD / 8;
  constexpr int DYB = BM * KD * 2;
  constexpr int ND = BM * NCD / 512;
  static_assert(ND >= 1 && BM * NCD % 512 == 0, "x tile staging");
  constexpr int PP = 144;
  constexpr int PRIV = 32 * PP;
  __shared__ __attribute__((aligned(16))) char lds[2 * DYB + 8 * PRIV];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int cw = wave % NCW, ph = wave / NCW;
  const int h = lane >> 5;
  char* priv = lds + 2 * DYB + wave * PRIV;
  const int split = blockIdx.x;
  const int co0 = blockIdx.y * CO, COT = p.co_total;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split;
  if (m_end > p.M) m_end = p.M;

  s16x8 wf[2][NKK];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int c = co0 + cw * 64 + t * 32 + (lane & 31);
      wf[t][kk] = __builtin_bit_cast(s16x8, cn_ld16(p.w + ((size_t)c * KD + 16 * kk + 8 * h) * 2));
    }
  const cn_buf_t xbuf = cn_make_buf(p.x, p.x_bytes);
  const int ech = lane & 7, erow = lane >> 3;
  const int cb = co0 + cw * 64 + ech * 8;
  u32x4 dreg[ND];
  auto load_stage = [&](int mb) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = tid + 512 * i;
      const int row = id / NCD, c = id - row * NCD;
      const int m = mb + row;
      dreg[i] = cn_buf_ld16(xbuf, m < m_end ? ((unsigned int)m * (unsigned int)KD + (unsigned int)c * 8u) * 2u : CN_OOB);
    }
  };
  // XF: the thread's chunk column is fixed (512 % NCD == 0): its 2 x 8 coefficients live in registers
  float xsc[XF ? 8 : 1], xsh[XF ? 8 : 1];
  if constexpr (XF) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { xsc[e] = p.xf[(tid % NCD) * 8 + e]; xsh[e] = p.xf[KD + (tid % NCD) * 8 + e]; }
  }
  auto store_x = [&](int mb, int buf) {
    char* t = lds + buf * DYB;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = tid + 512 * i;
      const int row = id / NCD, c = id - row * NCD;
      const int cs = NCD == 8 ? (c ^ ((row >> 1) & 7)) : (c ^ (row & (NCD - 1)));
      if constexpr (XF) {
        float f[8];
        Chunk<T>::unpack(dreg[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], xsc[e], xsh[e]);
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : 0.f;
        }
        dreg[i] = Chunk<T>::pack(f);
        const int m = mb + row;
        if (m < m_end && blockIdx.y == 0) cn_st16(p.a_out + ((size_t)m * KD + (size_t)c * 8) * 2, dreg[i]);
      }
      cn_st16(t + row * (KD * 2) + (cs << 4), dreg[i]);
    }
  };
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }

  if (m_begin < m_end) {
    load_stage(m_begin);
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += BM) {
      store_x(mb, buf);
      __syncthreads();
      if (mb + BM < m_end) load_stage(mb + BM);
      const char* t = lds + buf * DYB;
      f32x16 acc[2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
      const int prow = ph * 32 + (lane & 31);
      const char* rowp = t + prow * (KD * 2);
      const int sw = NCD == 8 ? ((prow >> 1) & 7) : (prow & (NCD - 1));
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const s16x8 b = __builtin_bit_cast(s16x8, cn_ld16(rowp + (((2 * kk + h) ^ sw) << 4)));
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          if constexpr (std::is_same<T, f16_t>::value) acc[x] = cn_mfma_32x32x16_f16(wf[x][kk], b, acc[x]);
          else acc[x] = cn_mfma_32x32x16_bf16(wf[x][kk], b, acc[x]);
        }
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x2 pk;
          pk[0] = cn_pack2<T>(acc[x][q * 4], acc[x][q * 4 + 1]);
          pk[1] = cn_pack2<T>(acc[x][q * 4 + 2], acc[x][q * 4 + 3]);
          *(u32x2*)(priv + (lane & 31) * PP + (x * 32 + 8 * q + 4 * h) * 2) = pk;
        }
      cn_wave_sync();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = mb + ph * 32 + k * 8 + erow;
        const u32x4 v = cn_ld16(priv + (k * 8 + erow) * PP + ech * 16);
        if (m < m_end) {
          float f[8];
          Chunk<T>::unpack(v, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { ssum[e] += f[e]; ssq[e] = fmaf(f[e], f[e], ssq[e]); }
          cn_st16(p.y + ((size_t)m * COT + (size_t)cb) * 2, v);
        }
      }
      cn_wave_sync();
      buf ^= 1;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int msk = 8; msk <= 32; msk <<= 1) {
      ssum[e] += cn_shfl_xor(ssum[e], msk);
      ssq[e] += cn_shfl_xor(ssq[e], msk);
    }
  __syncthreads();
  float* red = (float*)lds;
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(wave * 64 + lane * 8 + e) * 2] = ssum[e];
      red[(wave * 64 + lane * 8 + e) * 2 + 1] = ssq[e];
    }
  }
  __syncthreads();
  if (tid < CO && p.partial != nullptr) {
    const int c = tid, wcol = c / 64, cc = c % 64;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < PH; ++g2) {
      a1 += red[((g2 * NCW + wcol) * 64 + cc) * 2];
      a2 += red[((g2 * NCW + wcol) * 64 + cc) * 2 + 1];
    }
    float* dst = p.partial + (size_t)split * 2 * COT;
    dst[co0 + c] = a1;
    dst[COT + co0 + c] = a2;
  }
}

extern "C" int cn_conv1x1_stream_fwd_ok(int C, int K, int dtype) {   // C input, K output channels
  if (dtype != CN_BF16 && dtype != CN_F16) return 0;
  return ((K == 256 && (C == 64 || C == 128)) || (K == 512 && C == 128) || (K == 1024 && C == 256)) ? 1 : 0;
}
extern "C" int cn_conv1x1_stream_fwd_rows(int N, int H, int W, int K) {
  return jd_plan((long long)N * H * W, jd_bm(K >= 512 ? 512 : K), nullptr);
}
// y = conv1x1(x, w) (stride 1, C -> K channels of an instantiated shape: 64 / 128 -> 256, 128 -> 512, 256 -> 1024 - the
// last in two 512-channel slices, grid.y) with the
// statistics partials of cn_conv2d_fwd_bnstats (cn_conv1x1_stream_fwd_rows rows of 2*K floats, one per workgroup;
// partial may be NULL) as a persistent streaming kernel.  Output bits = cn_conv2d_fwd's.
static int jfwd_impl(const char* who, const void* x, const float* xf, int relu, void* a_out, const void* w_krsc, void* y,
                     int N, int H, int W, int C, int K, int dtype, float* partial, int partial_rows, void* stream) {
  if (!cn_conv1x1_stream_fwd_ok(C, K, dtype)) { cn_set_error("%s: C=%d -> K=%d dtype %d is not an instantiated shape", who, C, K, dtype); return CN_ESHAPE; }
  if (x == nullptr || w_krsc == nullptr || y == nullptr) { cn_set_error("%s: null operand", who); return CN_EINVAL; }
  const long long M = (long long)N * H * W;
  if (M <= 0) { cn_set_error("%s: empty", who); return CN_ESHAPE; }
  if (M * K * 2 >= (1ll << 31)) { cn_set_error("%s: operand exceeds the 2 GiB buffer-descriptor window", who); return CN_ESHAPE; }
  long long mps = 0;
  const int nsplit = jd_plan(M, jd_bm(K >= 512 ? 512 : K), &mps);
  if (partial != nullptr && partial_rows < nsplit) { cn_set_error("%s: partial buffer of %d rows < %d", who, partial_rows, nsplit); return CN_EWORKSPACE; }
  JfParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.w = (const char*)w_krsc; p.y = (char*)y; p.partial = partial;
  p.M = (int)M; p.m_per_split = (int)mps; p.nsplit = nsplit;
  p.co_total = K;
  p.x_bytes = (unsigned int)(M * C * 2);
  p.xf = xf; p.a_out = (char*)a_out; p.relu = relu;
  hipStream_t st = (hipStream_t)stream;
  cn_set_last_kernel(xf != nullptr ? "jfwd_kernel<%s, %d, %d, true>" : "jfwd_kernel<%s, %d, %d>", dtype == CN_F16 ? "f16_t" : "bf16_t", C, K);
  dim3 grid((unsigned)nsplit, (unsigned)(K > 512 ? K / 512 : 1));
#define JF_GO(KD, CO)                                                                             \
  do {                                                                                            \
    if (xf != nullptr) {                                                                          \
      if (dtype == CN_F16) CN_LAUNCH((jfwd_kernel<f16_t, KD, CO, true>), grid, dim3(512), st, p);  \
      else CN_LAUNCH((jfwd_kernel<bf16_t, KD, CO, true>), grid, dim3(512), st, p);                 \
    } else {                                                                                      \
      if (dtype == CN_F16) CN_LAUNCH((jfwd_kernel<f16_t, KD, CO>), grid, dim3(512), st, p);        \
      else CN_LAUNCH((jfwd_kernel<bf16_t, KD, CO>), grid, dim3(512), st, p);                       \
    }                                                                                             \
  } while (0)
  if (K == 256 && C == 64) JF_GO(64, 256);
  else if (K == 256 && C == 128) JF_GO(128, 256);
  else if (K == 512) JF_GO(128, 512);
  else JF_GO(256, 512);
#undef JF_GO
  return cn_check_launch("jfwd");
}
extern "C" int cn_conv1x1_stream_fwd(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C, int K,
                                     int dtype, float* partial, int partial_rows, void* stream) {
  return jfwd_impl("conv1x1_stream_fwd", x, nullptr, 0, nullptr, w_krsc, y, N, H, W, C, K, dtype, partial, partial_rows, stream);
}
// "Lazy a": cn_conv1x1_stream_fwd whose input is still the INPUT bn_y of the BatchNorm in front of the convolution
// (statistics finalised: stats = [mean | invstd | scale | shift], cn_bn_fwd_train*'s layout): the kernel forms
// a = relu?(bn_y * scale + shift) on its way into the LDS tile - bn_apply_kernel's arithmetic and rounding - and writes
// it to a_out [M][C] (the convolution's saved input for the weight gradient), so the apply pass of an inner BatchNorm
// (one read of bn_y + one write of a, then the convolution's read of a) becomes one read of bn_y + one write of a.
// Same a and y bits as cn_bn_fwd_train's apply pass + cn_conv1x1_stream_fwd.
extern "C" int cn_conv1x1_stream_fwd_lazya(const void* bn_y, const float* stats, int relu, void* a_out, const void* w_krsc,
                                           void* y, int N, int H, int W, int C, int K, int dtype, float* partial,
                                           int partial_rows, void* stream) {
  if (stats == nullptr || a_out == nullptr) { cn_set_error("conv1x1_stream_fwd_lazya: null operand"); return CN_EINVAL; }
  return jfwd_impl("conv1x1_stream_fwd_lazya", bn_y, stats + 2 * C, relu, a_out, w_krsc, y, N, H, W, C, K, dtype, partial,
                   partial_rows, stream);
}

// ------------------------------------------------------------------------------------------------
// "Lazy dy" data gradient of a 1x1 convolution with a LONG reduction as a streaming kernel (conv3 of the second stage:
// 512 gradient channels -> 128 input channels): dx[m][c] = sum_k dy[m][k] * W[k][c] with dy = c1*g + c2*y + c3 formed
// on load (cn_conv2d_dgrad_lazy's contract).  The streams are g and y (2 x 1 KB per pixel); the product is small.  A
// workgroup owns a pixel range: per stage of 32 pixels every thread loads, transforms and stages its share of the dy tile
// (LDS, double-buffered, requested a stage ahead); waves 0-3 own one 32-channel output tile each with their 128 x 32
// filter slice in registers (32 A-fragments) and multiply; the accumulators leave through a wave-private patch as
// 16-byte stores.  Operand orientation, k order and rounding of igemm_kernel: the same bits.
struct JlParams {
  const char* g;      // [M][KD]
  const char* y;      // [M][KD]
  const float* coef;  // [3][KD]
  const char* w;      // [CO][KD] (CRSK of the 1x1 convolution: rows = its input channels)
  char* dx;           // [M][CO]
  int M, m_per_split, nsplit;
  unsigned int gy_bytes;
};

template <typename T, int KD, int CO>
__global__ __launch_bounds__(512) void jdlazy_kernel(JlParams p) {
  static_assert(sizeof(T) == 2 && (CO == 128 || CO == 256) && KD % 64 == 0, "16-bit storage, four or eight 32-channel output tiles");
  constexpr int NW = CO / 32;           // waves that own an output tile (registers are allocated for all eight anyway)
  constexpr int BM = 32;
  constexpr int NKK = KD / 16;
  constexpr int NCD = KD / 8;             // 16-byte chunks per row
  constexpr int TB = BM * KD * 2;         // dy tile bytes
  constexpr int ND = BM * NCD / 512;      // chunks per thread and operand
  static_assert(512 % NCD == 0 && BM * NCD % 512 == 0, "a thread keeps one chunk column");
  constexpr int RS = 512 / NCD;           // rows per staging pass
  constexpr int PP = 80;                  // patch pitch: 32 channels * 2 bytes + 16
  __shared__ __attribute__((aligned(16))) char lds[2 * TB + NW * 32 * PP];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int h = lane >> 5;
  const int split = blockIdx.x;
  const int m_begin = split * p.m_per_split;
  int m_end = m_begin + p.m_per_split;
  if (m_end > p.M) m_end = p.M;

  s16x8 wf[NKK];
  if (wave < NW) {
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int c = wave * 32 + (lane & 31);
      wf[kk] = __builtin_bit_cast(s16x8, cn_ld16(p.w + ((size_t)c * KD + 16 * kk + 8 * h) * 2));
    }
  }
  const cn_buf_t gbuf = cn_make_buf(p.g, p.gy_bytes);
  const cn_buf_t ybuf = cn_make_buf(p.y, p.gy_bytes);
  const int col = tid % NCD, row0 = tid / NCD;
  float c1[8], c2[8], c3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    c1[e] = p.coef[col * 8 + e];
    c2[e] = p.coef[KD + col * 8 + e];
    c3[e] = p.coef[2 * KD + col * 8 + e];
  }
  u32x4 rg[ND], ry[ND];
  unsigned int okm = 0;
  auto load_stage = [&](int mb) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int m = mb + row0 + i * RS;
      const bool ok = m < m_end;
      const unsigned int o = ok ? ((unsigned int)m * (unsigned int)KD + (unsigned int)col * 8u) * 2u : CN_OOB;
      okm |= (ok ? 1u : 0u) << i;
      rg[i] = cn_buf_ld16(gbuf, o);
      ry[i] = cn_buf_ld16(ybuf, o);
    }
  };
  auto store_stage = [&](int buf) {
    char* t = lds + buf * TB;
#pragma unroll
    for (int i = 0; i < ND; ++i) {   // dy = c1*g + c2*y + c3: bn_bwd_apply_kernel's operation order and rounding
      float gg[8], vv[8];
      Chunk<T>::unpack(rg[i], gg);
      Chunk<T>::unpack(ry[i], vv);
#pragma unroll
      for (int e = 0; e < 8; ++e) gg[e] = fmaf(c1[e], gg[e], fmaf(c2[e], vv[e], c3[e]));
      const u32x4 o = Chunk<T>::pack(gg);
      const int row = row0 + i * RS;
      cn_st16(t + row * (KD * 2) + ((col ^ (row & (NCD - 1))) << 4), ((okm >> i) & 1u) ? o : cn_zero16());
    }
  };
  char* priv = lds + 2 * TB + (wave & (NW - 1)) * (32 * PP);
  const int ech = lane & 3, erow = lane >> 2;

  if (m_begin < m_end) {
    load_stage(m_begin);
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += BM) {
      store_stage(buf);
      __syncthreads();
      if (mb + BM < m_end) load_stage(mb + BM);
      if (wave < NW) {
        const char* t = lds + buf * TB;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int prow = lane & 31;
        const char* rowp = t + prow * (KD * 2);
        const int sw = prow & (NCD - 1);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          const s16x8 b = __builtin_bit_cast(s16x8, cn_ld16(rowp + (((2 * kk + h) ^ sw) << 4)));
          if constexpr (std::is_same<T, f16_t>::value) acc = cn_mfma_32x32x16_f16(wf[kk], b, acc);
          else acc = cn_mfma_32x32x16_bf16(wf[kk], b, acc);
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
          const int m = mb + pl;
          const u32x4 v = cn_ld16(priv + pl * PP + ech * 16);
          if (m < m_end) cn_st16(p.dx + ((size_t)m * CO + (size_t)(wave * 32 + ech * 8)) * 2, v);
        }
        cn_wave_sync();
      }
      buf ^= 1;
    }
  }
}

extern "C" int cn_conv2d_dgrad_lazy_stream_ok(int C, int K, int dtype) {   // K gradient channels -> C input channels
  // 512 -> 128 (conv3 of the second stage) and 512 -> 256 (that stage's stride-2 projection on its coarse grid: once per
  // step, ~276 -> ~130 us on this kernel; the step does not move either way, round 3 and round 4 A/Bs)
  return (dtype == CN_BF16 || dtype == CN_F16) && K == 512 && (C == 128 || C == 256) ? 1 : 0;
}
// cn_conv2d_dgrad_lazy for a 1x1 / stride-1 convolution of an instantiated shape (512 -> 128 or 256 channels) as a
// persistent streaming kernel.  Same bits.
extern "C" int cn_conv2d_dgrad_lazy_stream(const void* g, const void* bn_y, const float* coef, const void* w_crsk, void* dx,
                                           int N, int H, int W, int C, int K, int dtype, void* stream) {
  if (!cn_conv2d_dgrad_lazy_stream_ok(C, K, dtype)) { cn_set_error("conv2d_dgrad_lazy_stream: K=%d -> C=%d dtype %d is not an instantiated shape", K, C, dtype); return CN_ESHAPE; }
  if (g == nullptr || bn_y == nullptr || coef == nullptr || w_crsk == nullptr || dx == nullptr) { cn_set_error("conv2d_dgrad_lazy_stream: null operand"); return CN_EINVAL; }
  const long long M = (long long)N * H * W;
  if (M <= 0) { cn_set_error("conv2d_dgrad_lazy_stream: empty"); return CN_ESHAPE; }
  if (M * K * 2 >= (1ll << 31)) { cn_set_error("conv2d_dgrad_lazy_stream: operand exceeds the 2 GiB buffer-descriptor window"); return CN_ESHAPE; }
  long long mps = 0;
  const int nsplit = jd_plan(M, 32, &mps);
  JlParams p;
  memset(&p, 0, sizeof(p));
  p.g = (const char*)g; p.y = (const char*)bn_y; p.coef = coef; p.w = (const char*)w_crsk; p.dx = (char*)dx;
  p.M = (int)M; p.m_per_split = (int)mps; p.nsplit = nsplit;
  p.gy_bytes = (unsigned int)(M * K * 2);
  cn_set_last_kernel("jdlazy_kernel<%s, %d, %d>", dtype == CN_F16 ? "f16_t" : "bf16_t", K, C);
  if (C == 128) {
    if (dtype == CN_F16) CN_LAUNCH((jdlazy_kernel<f16_t, 512, 128>), dim3((unsigned)nsplit), dim3(512), (hipStream_t)stream, p);
    else CN_LAUNCH((jdlazy_kernel<bf16_t, 512, 128>), dim3((unsigned)nsplit), dim3(512), (hipStream_t)stream, p);
  } else {   // (the stride-2 projection of the second stage on its coarse grid: 512 -> 256)
    if (dtype == CN_F16) CN_LAUNCH((jdlazy_kernel<f16_t, 512, 256>), dim3((unsigned)nsplit), dim3(512), (hipStream_t)stream, p);
    else CN_LAUNCH((jdlazy_kernel<bf16_t, 512, 256>), dim3((unsigned)nsplit), dim3(512), (hipStream_t)stream, p);
  }
  return cn_check_launch("jdlazy");
}
