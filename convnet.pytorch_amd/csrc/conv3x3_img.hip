// conv3x3_img.hip -- 3x3 / stride-1 / pad-1 convolution with C = K = 128 / 256 / 512 channels on the 28 x 28 / 14 x 14 /
// 7 x 7 maps of a bottleneck ResNet (conv2 of the second / third / fourth stage, /root/reference models/resnet.py:126-132;
// forward and data gradient) as an IMAGE-RESIDENT kernel (round 5).
//
// Through the tiled implicit-GEMM kernels these layers run at 0.26 - 0.34 of the dense MFMA peak: every layer of the
// family is 59.2 GFLOP = 231 MFLOP per CU, i.e. 24 us at the peak, and takes 70 - 95 us.  Three things cost the tiles:
// 196 (or 98) tiles for 256 CUs; nine re-gathers of every input pixel from L2 behind one workgroup barrier per K tile; a
// launch-wide ramp per tile.  Here ONE workgroup of four waves (one per SIMD, up to 512 registers each) owns a band of
// whole image rows - half a 28 x 28 image, a whole 14 x 14 or 7 x 7 image: 512 / 256 / 256 equal work items for 256 CUs:
//   * the band's input rows + one row above / below + a zero column on either side sit in LDS ONCE (120 / 128 / 81 KB),
//     16-byte channel chunks XOR-swizzled by the position index so that the 16 pixels of a ds_read_b128 lane group fall
//     on 16 different bank groups; a tap is an address offset;
//   * a wave owns a quarter of the output channels (32 / 64 / 128) and ALL pixels of the band: 13 / 14 / 8 independent
//     32 x 32 accumulators, one filter fragment feeds 13 / 7 / 2 MFMAs, one pixel fragment 1 / 2 / 4;
//   * the filter streams: per k-step (one tap, 16 reduction channels) a wave needs 1 / 2 / 4 KB of it - ITS OWN output
//     channels only - which it fetches itself by LDS-DMA into a wave-private ring (8 / 4 / 4 slots) from a copy of the
//     filter laid out in exactly that order (cn_weight_prep_tiled writes it: [tap][k block][32-channel tile][lane][8]).
//     No wave reads what another wave loads: THE MAIN LOOP HAS NO WORKGROUP BARRIER; a wave orders its ring by counted
//     vmcnt waits, the fragment reads of step s + 1 are issued between the MFMAs of step s;
//   * outputs leave through a wave-private transposition patch as 16-byte stores; the forward form keeps the BatchNorm
//     statistics of the stored values in registers (one partial row per workgroup).
// The data gradient is the same kernel on the gradient with the CRSK filter copy and mirrored taps.
// Operand orientation and k order are igemm_kernel's (tap-major, channel chunks ascending): the same output bits.
#include "cn_common.h"
#include "cn_api_internal.h"
#include <type_traits>

struct CimgParams {
  const char* x;       // [N][H][W][C]
  const char* wslab;   // [9][C/16][C/32][64 lanes][8]: lane l = (row l & 31 of the 32-channel tile, k half l >> 5)
  char* y;             // [N][H][W][C]
  float* partial;      // optional [nwg][2 * C]: sum | sum of squares of the stored outputs
  int N, H, nbands, nwork, flip;
  unsigned int w_bytes;
  int dbg;
};

template <int C, int W>
struct CimgGeom {
  static constexpr int RB = W == 28 ? 14 : W;              // image rows per band
  static constexpr int D = C == 128 ? 8 : 4;               // ring slots per wave
  static constexpr int WP = W + 2, HR = RB + 2, NPOS = HR * WP;
  static constexpr int PB = C * 2;                         // bytes per halo position
  static constexpr int HALO = NPOS * PB;
  static constexpr int NPX = RB * W, NPF = (NPX + 31) / 32;
  static constexpr int NCW = C / 128;                      // 32-channel output tiles per wave
  static constexpr int NKB = C / 16;                       // k blocks per tap
  static constexpr int NSTEP = 9 * NKB;
  static constexpr int SLOT = NCW * 1024;
  static constexpr int RING = D * SLOT;
  static constexpr int NCH = C / 8;                        // 16-byte chunks per position
  static constexpr int PP = NCW * 64 + 16;                 // patch pitch (bytes): a wave's channels of one pixel + 16
  static constexpr int LDS = HALO + 4 * RING;
  static_assert(32 * PP <= RING, "the transposition patch lives in the wave's ring");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <typename T, int C, int W>
__global__ __launch_bounds__(256, 1) void conv3x3_img_kernel(CimgParams p) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  using G = CimgGeom<C, W>;
  constexpr int RB = G::RB, D = G::D, WP = G::WP, HR = G::HR, NPOS = G::NPOS, PB = G::PB, NPX = G::NPX, NPF = G::NPF;
  constexpr int NCW = G::NCW, NKB = G::NKB, NSTEP = G::NSTEP, SLOT = G::SLOT, RING = G::RING, NCH = G::NCH, PP = G::PP;
  constexpr int NM = NCW * NPF;            // MFMAs per k-step and wave
  constexpr int NR = NCW + NPF;            // fragment reads per k-step and wave
  __shared__ __attribute__((aligned(1024))) char lds[G::LDS];
  char* halo = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);
  const int h = lane >> 5;
  char* ring = lds + G::HALO + wave * RING;
  const int H = p.H;
  const cn_buf_t wbuf = cn_make_buf(p.wslab, p.w_bytes);

  // pixel fragments: lane's pixel of fragment f (clamped past the band) -> halo position of tap (0, 0)
  int pos0[NPF];
#pragma unroll
  for (int f = 0; f < NPF; ++f) {
    int px = f * 32 + (lane & 31);
    if (px > NPX - 1) px = NPX - 1;
    const int oy = px / W, ox = px - oy * W;
    pos0[f] = oy * WP + ox;
  }
  // epilogue coordinates: 16-byte chunk `ech` of the wave's channels, pixel rows pass * RPP + erow of a fragment
  constexpr int CPR = NCW * 4, RPP = 64 / CPR;
  const int ech = lane % CPR, erow = lane / CPR;
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }

  // the wave's filter slice of k-step s: piece i (1 KB) -> ring slot `slot` (past the last step: zeros nobody reads)
  auto dma_piece = [&](int s, int slot, int i) {
    const unsigned int off = ((unsigned int)(s * 4 + wave) * (unsigned int)NCW + (unsigned int)i) * 1024u + (unsigned int)lane * 16u;
    cn_buf_ld16_lds(wbuf, s < NSTEP ? off : CN_OOB, ring + slot * SLOT + i * 1024);
  };

  for (int work = blockIdx.x; work < p.nwork; work += gridDim.x) {
    const int band = work % p.nbands, n = work / p.nbands;
    const int oy0 = band * RB;
    __syncthreads();                 // the previous band's halo has been consumed by every wave
    // the first D k-steps of the filter are on their way while the halo is staged
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
      for (int i = 0; i < NCW; ++i) dma_piece(s, s, i);
    // ---- halo: rows oy0 - 1 .. oy0 + RB, columns -1 .. W; chunk j of position pos at slot j ^ (pos & 15)
    if (!(p.dbg & 2)) {
      constexpr int NCHUNK = NPOS * NCH;
      for (int id0 = tid; id0 < NCHUNK; id0 += 8 * 256) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int id = id0 + 256 * i;
          const int pos = id / NCH, j = id - pos * NCH;
          const int r = pos / WP, c = pos - r * WP;
          const int iy = oy0 - 1 + r, ix = c - 1;
          const bool ok = id < NCHUNK && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
          v[i] = ok ? cn_ld16(p.x + ((((size_t)n * H + (size_t)iy) * W + (size_t)ix) * C + (size_t)j * 8) * 2) : cn_zero16();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int id = id0 + 256 * i;
          if (id < NCHUNK) {
            const int pos = id / NCH, j = id - pos * NCH;
            cn_st16(halo + pos * PB + ((j ^ (pos & 15)) << 4), v[i]);
          }
        }
      }
    }
    __syncthreads();

    f32x16 acc[NCW][NPF];
#pragma unroll
    for (int a = 0; a < NCW; ++a)
#pragma unroll
      for (int f = 0; f < NPF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][f][r] = 0.f;

    // fragments of one k-step: NCW filter fragments (ring) + NPF pixel fragments (halo).  The halo byte base and swizzle
    // key of every pixel fragment change with the tap only: kept in registers, refreshed every NKB steps
    int bpos[NPF], bsw[NPF];
    auto prep_tap = [&](int tap) {
      const int tr = tap / 3, ts = tap - tr * 3;
      const int dr = p.flip ? 2 - tr : tr, ds = p.flip ? 2 - ts : ts;
      const int toff = dr * WP + ds;
#pragma unroll
      for (int f = 0; f < NPF; ++f) {
        const int pos = pos0[f] + toff;
        bpos[f] = pos * PB;
        bsw[f] = pos & 15;
      }
    };
    auto frag_a = [&](int slot, int i) -> const char* { return ring + slot * SLOT + i * 1024 + lane * 16; };
    auto frag_b = [&](int kb2h, int f) -> const char* { return halo + bpos[f] + ((kb2h ^ bsw[f]) << 4); };
    auto mma = [&](const u32x4& a, const u32x4& b, f32x16& c) {
      if constexpr (std::is_same<T, f16_t>::value) c = cn_mfma_32x32x16_f16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c);
      else c = cn_mfma_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c);
    };
    // One k-step with ring slot U (compile time).  One wave per SIMD: nothing hides what is not issued between two MFMAs,
    // so everything else of the step sits behind its first MFMAs: `cur` holds the fragments of step s (their reads were
    // issued early in step s - 1 and have returned); slot U is refilled with step s + D (piece i behind MFMA i); the
    // pixel fragments of step s + 1 are requested from the first MFMA on, its filter fragments once slot U + 1 has
    // landed (requested D - 1 steps ago) - two reads per MFMA, all out by the middle of the step.
    auto kstep = [&](auto U_, int s, u32x4 (&cur)[NR], u32x4 (&nxt)[NR]) {
      constexpr int U = decltype(U_)::value, UN = (U + 1) % D;
      const int kbn = (s + 1) % NKB;
      int rb = 0, ra = 0;
#pragma unroll
      for (int a = 0; a < NCW; ++a)
#pragma unroll
        for (int f = 0; f < NPF; ++f) {
          const int m = a * NPF + f;
          mma(cur[a], cur[NCW + f], acc[a][f]);
          if (m == 0) {
#ifndef CN_EMULATE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of step s (slot U among them) has returned
#endif
            cn_sched_fence();
            if (kbn == 0) prep_tap((s + 1) / NKB);               // (wave-uniform; past the last tap nothing is used)
          }
          if (m < NCW && !(p.dbg & 8)) dma_piece(s + D, U, m);
          if (m == NCW - 1) {
            // outstanding: steps s + 1 .. s + D; the oldest (s + 1, slot UN) is what the filter reads below need
            if constexpr (NCW == 1) CN_WAIT_VMCNT(7);
            else if constexpr (NCW == 2) CN_WAIT_VMCNT(6);
            else CN_WAIT_VMCNT(12);
            cn_sched_fence();
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            if (rb < NPF) { if (!(p.dbg & 16)) nxt[NCW + rb] = cn_ld16(frag_b(kbn * 2 + h, rb)); ++rb; }
            else if (m >= NCW - 1 && ra < NCW) { if (!(p.dbg & 16)) nxt[ra] = cn_ld16(frag_a(UN, ra)); ++ra; }
          }
          cn_sched_fence();
        }
    };
    static_assert((D == 8 && NCW == 1) || (D == 4 && NCW == 2) || (D == 4 && NCW == 4), "vmcnt literals above: (D - 1) * NCW");
    static_assert(NSTEP % D == 0 && D % 2 == 0 && NKB % D == 0, "D k-steps per loop trip, fragment sets alternate");

    u32x4 fa[NR], fb[NR];
    CN_WAIT_VMCNT(0);                 // (the halo's own loads and the first D filter slices)
    prep_tap(0);
#pragma unroll
    for (int q = 0; q < NCW; ++q) fa[q] = cn_ld16(frag_a(0, q));
#pragma unroll
    for (int q = 0; q < NPF; ++q) fa[NCW + q] = cn_ld16(frag_b(h, q));
    for (int s0 = 0; s0 < ((p.dbg & 1) ? 0 : NSTEP); s0 += D) {
      kstep(std::integral_constant<int, 0>(), s0, fa, fb);
      kstep(std::integral_constant<int, 1>(), s0 + 1, fb, fa);
      kstep(std::integral_constant<int, 2>(), s0 + 2, fa, fb);
      kstep(std::integral_constant<int, 3>(), s0 + 3, fb, fa);
      if constexpr (D == 8) {
        kstep(std::integral_constant<int, 4>(), s0 + 4, fa, fb);
        kstep(std::integral_constant<int, 5>(), s0 + 5, fb, fa);
        kstep(std::integral_constant<int, 6>(), s0 + 6, fa, fb);
        kstep(std::integral_constant<int, 7>(), s0 + 7, fb, fa);
      }
    }
    CN_WAIT_VMCNT(0);                 // the zero slices fetched past the end have landed: the ring is free
#ifndef CN_EMULATE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    cn_sched_fence();

    // ---- epilogue: 32 pixels x the wave's NCW * 32 channels at a time through the wave-private patch
    char* priv = ring;
#pragma unroll
    for (int f = 0; f < ((p.dbg & 4) ? 0 : NPF); ++f) {
#pragma unroll
      for (int a = 0; a < NCW; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x2 pk;
          pk[0] = cn_pack2<T>(acc[a][f][q * 4], acc[a][f][q * 4 + 1]);
          pk[1] = cn_pack2<T>(acc[a][f][q * 4 + 2], acc[a][f][q * 4 + 3]);
          *(u32x2*)(priv + (lane & 31) * PP + (a * 32 + 8 * q + 4 * h) * 2) = pk;
        }
      cn_wave_sync();
#pragma unroll
      for (int k = 0; k < 32 / RPP; ++k) {
        const int pl = k * RPP + erow;
        const int px = f * 32 + pl;
        const u32x4 v = cn_ld16(priv + pl * PP + ech * 16);
        const int oy = (px < NPX ? px : 0) / W, ox = (px < NPX ? px : 0) - oy * W;
        if (px < NPX && oy0 + oy < H) {
          if (p.partial != nullptr) {
            float fv[8];
            Chunk<T>::unpack(v, fv);
#pragma unroll
            for (int e = 0; e < 8; ++e) { ssum[e] += fv[e]; ssq[e] = fmaf(fv[e], fv[e], ssq[e]); }
          }
          cn_st16(p.y + ((((size_t)n * H + (size_t)(oy0 + oy)) * W + (size_t)ox) * C + (size_t)(wave * NCW * 32 + ech * 8)) * 2, v);
        }
      }
      cn_wave_sync();
    }
  }

  if (p.partial != nullptr) {      // one row per workgroup: the lanes of a wave that share a chunk column, fixed order
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int msk = CPR; msk <= 32; msk <<= 1) {
        ssum[e] += cn_shfl_xor(ssum[e], msk);
        ssq[e] += cn_shfl_xor(ssq[e], msk);
      }
    if (erow == 0) {
      float* dst = p.partial + (size_t)blockIdx.x * 2 * C + wave * NCW * 32 + ech * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { dst[e] = ssum[e]; dst[C + e] = ssq[e]; }
    }
  }
}

static int cimg_width(int C) { return C == 128 ? 28 : (C == 256 ? 14 : (C == 512 ? 7 : 0)); }
extern "C" int cn_conv3x3_img_ok(int H, int W, int C, int K, int dtype) {
  return (dtype == CN_BF16 || dtype == CN_F16) && C == K && cimg_width(C) != 0 && W == cimg_width(C) && H >= 1 ? 1 : 0;
}
static int cimg_bands(int H, int C) { const int rb = C == 128 ? 14 : cimg_width(C); return (H + rb - 1) / rb; }
// workgroups of a launch = rows of the statistics partials (one per workgroup)
extern "C" int cn_conv3x3_img_rows(int N, int H, int C) {
  if (cimg_width(C) == 0 || N <= 0 || H <= 0) return 0;
  const int nwork = N * cimg_bands(H, C);
  int n = cn_get_option("conv3x3_img_wgs", 256);
  if (n < 1) n = 1;
  return n < nwork ? n : nwork;
}

// y = conv3x3(x, w), stride 1, pad 1, C -> C channels (128 on 28-wide, 256 on 14-wide, 512 on 7-wide maps), NHWC.
// w_slab: the filter in k-step order (cn_weight_prep_tiled's slab copies).  flip = 0: forward with the slab of the KRSC
// filter; flip = 1: data gradient (x = dy, the slab of the CRSK filter: rows = input channels of the convolution), taps
// mirrored.  partial (optional, forward): cn_conv3x3_img_rows(N, H, C) rows of 2 * C floats [sum | sum of squares] of
// the stored outputs for cn_bn_fwd_train_partials.  Same output bits as cn_conv2d_fwd / cn_conv2d_dgrad.
extern "C" int cn_conv3x3_img(const void* x, const void* w_slab, void* y, int N, int H, int W, int C, int dtype, int flip,
                              float* partial, int partial_rows, void* stream) {
  if (x == nullptr || w_slab == nullptr || y == nullptr) { cn_set_error("conv3x3_img: null operand"); return CN_EINVAL; }
  if (!cn_conv3x3_img_ok(H, W, C, C, dtype) || N <= 0) { cn_set_error("conv3x3_img: unsupported shape (H=%d W=%d C=%d)", H, W, C); return CN_ESHAPE; }
  CimgParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.wslab = (const char*)w_slab; p.y = (char*)y; p.partial = partial;
  p.N = N; p.H = H; p.flip = flip ? 1 : 0;
  p.nbands = cimg_bands(H, C);
  p.nwork = N * p.nbands;
  p.w_bytes = (unsigned int)(9u * (unsigned int)C * (unsigned int)C * 2u);
  p.dbg = cn_get_option("dbg_img", 0);
  const int nwg = cn_conv3x3_img_rows(N, H, C);
  if (partial != nullptr && partial_rows < nwg) { cn_set_error("conv3x3_img: partial buffer of %d rows < %d", partial_rows, nwg); return CN_EWORKSPACE; }
  cn_set_last_kernel("conv3x3_img_kernel<%s, %d, %d>%s", dtype == CN_F16 ? "f16_t" : "bf16_t", C, W, flip ? " [dgrad]" : "");
  const dim3 grid((unsigned)nwg), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CN_F16) {
    if (C == 128) CN_LAUNCH((conv3x3_img_kernel<f16_t, 128, 28>), grid, block, st, p);
    else if (C == 256) CN_LAUNCH((conv3x3_img_kernel<f16_t, 256, 14>), grid, block, st, p);
    else CN_LAUNCH((conv3x3_img_kernel<f16_t, 512, 7>), grid, block, st, p);
  } else {
    if (C == 128) CN_LAUNCH((conv3x3_img_kernel<bf16_t, 128, 28>), grid, block, st, p);
    else if (C == 256) CN_LAUNCH((conv3x3_img_kernel<bf16_t, 256, 14>), grid, block, st, p);
    else CN_LAUNCH((conv3x3_img_kernel<bf16_t, 512, 7>), grid, block, st, p);
  }
  return cn_check_launch("conv3x3_img");
}
