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
#include <type_traits>

#define IG_MAX_TAPS 64

struct IgemmParams {
  const char* x;
  const char* w;
  char* y;
  const char* addend;   // optional tensor added to the output (same layout / dtype as y)
  int addend_sub, add_H, add_W;   // addend_sub = 2: the addend holds the even (h, w) pixels only, [N][add_H][add_W][Co] (the rest is zero)
  const float* bias;
  float* stats;         // optional [n_mtiles][2*Co]: per pixel-tile sum / sum of squares of the stored outputs
  const float* stats_pivot;   // optional [Co]: the sums are taken of (output - pivot[c]) (centred statistics)
  // optional fused BatchNorm-backward reduction (dgrad): the output is the gradient w.r.t. z = act(BN(bn_y));
  // the epilogue applies the ReLU mask, stores g = dz*mask and emits per-tile sum(g), sum(g*xhat)
  const char* bn_y;              // BN input, same layout / dtype as the output
  const unsigned char* bn_mask;  // optional bit mask (one byte per 16-byte chunk) written by bn_apply
  const float* bn_coef;          // [mean | invstd | scale | shift], 4*Co floats
  float* bn_partial;             // [rows][2*Co]
  int bn_row0, bn_relu;
  int stats_rows;                // rows of `stats` (one per 128 pixels)
  int N, Hi, Wi, Ci;
  int Hg, Wg, a_h, a_w;
  int Ho, Wo, Co;
  int oh_mul, oh_off, ow_mul, ow_off;
  int ntaps, cpt, nchunks;
  long long w_row;
  int out_f32, relu;
  int M, n_ntiles, n_mtiles;
  unsigned int x_bytes, w_bytes;   // extents of the gather source / filter tensors (buffer descriptors)
  int simple;                      // 1: no tap of a valid row ever leaves the image (skip bounds tests)
  const float* xf;                 // optional per-channel fp32 tables of an operand transform applied on load (XF; modes below)
  int xf_relu;                     // (mode 3: the junction has a ReLU)
  // XF mode 2 ("lazy dy", round 3): the gathered operand is the BatchNorm-backward result c1[c]*x + c2[c]*x2 + c3[c]
  // (x = masked upstream gradient g, x2 = BatchNorm input y, xf = [c1 | c2 | c3] of the Ci channels), rounded to T
  // exactly as bn_bwd_apply_kernel stores it - so that apply pass (read g, read y, write dy) never runs
  const char* x2;
  int xf_mode;
  // XF mode 3 ("lazy z", round 3; kernel instantiations XF = 3): the gathered operand is a residual JUNCTION's output
  //   z = relu?( x*scale[c] + shift[c] + r ),  r = x2            (xf2 == NULL: x2 is the materialised shortcut tensor)
  //                                            r = round_T(x2*rscale[c] + rshift[c])   (xf2 = [rscale | rshift]: x2 is
  //                                                the projection shortcut's BatchNorm input, cf. cn_bn_apply_dual)
  // formed between the global load and the LDS store with bn_apply_kernel's operation order and rounding, and ALSO
  // stored to xz (+ the ReLU bits to xz_mask): the junction's apply pass never runs, z is written once by the 1x1
  // convolution that consumes it and never re-read by it.  1x1 / stride 1 / one channel tile only (every element of x
  // is visited exactly once).
  const float* xf2;
  char* xz;
  unsigned char* xz_mask;
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
__device__ __forceinline__ void ig_mma<f16_t>(const u32x4& a, const u32x4& b, f32x16& acc) {
  acc = cn_mfma_32x32x16_f16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), acc);
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

// STAGES = 1 (register-staged): single staging buffer + register prefetch (two barriers per K tile, half the LDS, so
//             more workgroups per CU hide the global-load latency of short reductions, e.g. 1x1 convs);
// STAGES = 2 (GLDS): LDS-DMA double buffer (one barrier per K tile).
// (Measured and removed in round 4: a register-staged double buffer, 4-deep DMA rings with counted vmcnt on 4 / 8 waves,
//  a 256 x 128 tile, 64-pixel tiles for the epilogue launches, pixel-tiles-fastest order, non-temporal operand loads:
//  none faster on any layer, profiles/README.md.)
// OUTF32 sizes the LDS output tile for fp32 results (fp32 compute or fp32 logits).
//
// The reduction loop is kept lean in VALU work (a first version spent 23 VALU instructions per MFMA
// on 64-bit im2col address math, see profiles/r01_pmc_*): operands come in through bounds-checked
// buffer loads (zero fill = offset CN_OOB), every row's byte offset is computed once, a tap only
// adds a per-tap byte delta read from an LDS table, and all LDS addresses are loop invariant.
// GLDS: operands are DMA'd straight into the (source-permuted, hence still XOR-swizzled) LDS tiles
//       with `buffer_load ... lds`: no prefetch registers, no ds_write pass (the register-staged
//       variant spends ~416 LDS cycles per K tile on ds_write_b128 against 512 MFMA cycles).
// XF: the pixel operand is transformed between the global load and the LDS store (register-staged variants only), with
//     the arithmetic and rounding of the streaming pass it replaces: XF = 1 "lazy dy" (IgemmParams::xf_mode 2), XF = 3
//     "lazy z" (mode 3).  (A plain BatchNorm apply folded in the same way was built in round 2, measured slower than
//     apply + convolution on the tiled kernels and removed in round 4: profiles/r02e_bn_apply_folded_into_conv_operand_load.txt.)
#define IG_XF_MAX 512
#define IG_XF_TAB_BYTES (IG_XF_MAX * 16)   /* up to four per-channel fp32 tables */
// ILV (round 3, LDS-DMA double buffer only): the DMA instructions of K tile kt+1 are issued BETWEEN the MFMAs of tile kt
//     (one after each of the first NPR+NWR MFMAs, pinned by scheduling fences) instead of in a block ahead of them.  In
//     the block form every wave of the workgroup issues its 8 DMA instructions (~100 cycles of issue each while the
//     texture path is busy) right after the barrier, i.e. with the matrix pipe of all four SIMDs idle; interleaved,
//     each one is issued while the MFMA before it executes.  Same instructions, same results (bit-identical).
template <typename T, int WC, int WP, int TI, int TJ, int STAGES, bool OUTF32, bool GLDS, bool FRAGDB, bool EPI, int XF = 0, bool ILV = false>
// (the eight-wave lazy-z tile is held to 128 VGPRs - four waves per SIMD, two workgroups per CU -: at 130 it ran one
//  workgroup per CU and streamed at 3.5 instead of 5 TB/s)
__global__ __launch_bounds__(WC * WP * 64, (XF != 0 && WC * WP == 8) ? 4 : 1) void igemm_kernel(IgemmParams p) {
  static_assert(!XF || (!GLDS && !EPI), "operand transform needs the register-staged path");
  static_assert(!ILV || (GLDS && STAGES == 2), "interleaved DMA issue: LDS-DMA double buffer");
  static_assert(GLDS ? STAGES == 2 : STAGES == 1, "LDS-DMA double buffer, or register-staged single buffer");
  constexpr int BN = WC * TI * 32;  // output channels per block
  constexpr int BM = WP * TJ * 32;  // pixels per block
  constexpr int NT = WC * WP * 64;  // threads (4 or 8 waves)
  static_assert(NT == 256 || NT == 512, "4 or 8 waves");
  constexpr int RS = NT / 8;        // tile rows covered by one staging pass (8 chunks per row)
  constexpr int EB = ElemTraits<T>::kBytes;
  constexpr int NPR = BM / RS;  // pixel rows staged per thread
  constexpr int NWR = BN / RS;  // filter rows staged per thread
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int OUT_MAX = BM * (BN * (OUTF32 ? 4 : EB) + 16);
  constexpr int MAIN = (STAGES * STAGE > OUT_MAX) ? STAGES * STAGE : OUT_MAX;
  constexpr int LDS_BYTES = MAIN + IG_MAX_TAPS * 16 + BM * 4 + (XF ? IG_XF_TAB_BYTES : 0) + (EPI ? BM * 4 : 0);
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  int* s_taps = (int*)(lds + MAIN);                       // per tap: {dhdw, woff bytes, x delta bytes, 0}
  int* s_outpix = (int*)(lds + MAIN + IG_MAX_TAPS * 16);
  float* s_xf = (float*)(lds + MAIN + IG_MAX_TAPS * 16 + BM * 4);   // XF: [scale | shift] of the Ci input channels
  int* s_addpix = (int*)(lds + MAIN + IG_MAX_TAPS * 16 + BM * 4);   // EPI (never together with XF): pixel index into a subsampled addend

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = cn_uniform(tid >> 6);   // wave-uniform by construction: lets LDS-DMA bases live in SGPRs (no waterfall)
  const unsigned int tile = cn_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = (int)(tile % p.n_ntiles);   // channel tiles fastest: neighbours share the activation tile
  const int mt = (int)(tile / p.n_ntiles);
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int HgWg = p.Hg * p.Wg;

  if (tid < IG_MAX_TAPS) {
    const int t = tid < p.ntaps ? tid : 0;
    int dhdw = 0, woffb = 0, delta = 0;
    if (p.ntaps > 0) {
      dhdw = p.tap_dhdw[t];
      woffb = p.tap_woff[t] * EB;
      const int dh = (int)(short)(dhdw & 0xffff), dw = dhdw >> 16;
      delta = (dh * p.Wi + dw) * p.Ci * EB;
    }
    s_taps[4 * tid] = dhdw;
    s_taps[4 * tid + 1] = woffb;
    s_taps[4 * tid + 2] = delta;
    s_taps[4 * tid + 3] = 0;
  }
  if (tid < BM) {
    int m = m0 + tid;
    int pix = -1, apix = -1;
    if (m < p.M) {
      int n = (int)cn_fastdiv((unsigned)m, p.div_hw);
      int rem = m - n * HgWg;
      int hg = (int)cn_fastdiv((unsigned)rem, p.div_w);
      int wg = rem - hg * p.Wg;
      const int ho = hg * p.oh_mul + p.oh_off, wo = wg * p.ow_mul + p.ow_off;
      pix = (n * p.Ho + ho) * p.Wo + wo;
      if (EPI && p.addend_sub == 2 && ((ho | wo) & 1) == 0) apix = (n * p.add_H + (ho >> 1)) * p.add_W + (wo >> 1);
    }
    s_outpix[tid] = pix;
    if (EPI) s_addpix[tid] = apix;
  }
  if (XF) {
    const int ntab = (XF == 3 ? 2 : 3) * p.Ci;   // lazy z: [scale | shift]; lazy dy: [c1 | c2 | c3]
    for (int c = tid; c < ntab; c += NT) s_xf[c] = p.xf[c];
    if (XF == 3 && p.xf2 != nullptr)
      for (int c = tid; c < 2 * p.Ci; c += NT) s_xf[2 * p.Ci + c] = p.xf2[c];
  }

  // per-thread staging coordinates (fixed for the whole reduction loop)
  const int cc = tid & 7;
  const int r0 = tid >> 3;
  // LDS slot cc of row r0 holds chunk cc ^ swizzle(row): with DMA the lane must FETCH that chunk
  const int cg = GLDS ? (cc ^ ((r0 >> 1) & 7)) : cc;
  const cn_buf_t xbuf = cn_make_buf(p.x, p.x_bytes);
  const cn_buf_t x2buf = cn_make_buf(XF ? p.x2 : p.x, p.x_bytes);   // XF mode 2: second source, same extent
  const cn_buf_t wbuf = cn_make_buf(p.w, p.w_bytes);
  int phin[NPR], pwin[NPR];
  unsigned int prow[NPR], wrow[NWR];   // byte offsets of the row starts (CN_OOB = row not valid)
#pragma unroll
  for (int i = 0; i < NPR; ++i) {
    int m = m0 + r0 + RS * i;
    const bool valid = m < p.M;
    int mm = valid ? m : 0;
    int n = (int)cn_fastdiv((unsigned)mm, p.div_hw);
    int rem = mm - n * HgWg;
    int hg = (int)cn_fastdiv((unsigned)rem, p.div_w);
    int wg = rem - hg * p.Wg;
    phin[i] = valid ? hg * p.a_h : -0x4000;   // invalid rows fail every bounds test below
    pwin[i] = wg * p.a_w;
    prow[i] = valid ? (unsigned int)(((n * p.Hi + hg * p.a_h) * p.Wi + wg * p.a_w) * p.Ci * EB) : CN_OOB;
  }
#pragma unroll
  for (int i = 0; i < NWR; ++i) {
    const int co = n0 + r0 + RS * i;
    wrow[i] = co < p.Co ? (unsigned int)(co * (int)p.w_row * EB) : CN_OOB;
  }
  // loop-invariant LDS addresses: staging stores and fragment reads
  const int st0 = ig_slot(r0, cc);   // rows r0 + RS*i share the swizzle: slot(i) = st0 + i*RS*128
  const int wc = wave % WC;
  const int wp = wave / WC;
  const int lrow = lane & 31;
  const int swz = (lrow >> 1) & 7;   // tile bases are multiples of 32 rows: the swizzle is per lane
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = lrow * 128 + (((kk * 2 + (lane >> 5)) ^ swz) << 4);
  const int rd_w = wc * TI * 32 * 128;
  const int rd_p = BN * 128 + wp * TJ * 32 * 128;
  __syncthreads();

  constexpr int OEBc = OUTF32 ? 4 : EB;
  const int OEB = OEBc;
  const int pitch = BN * OEB + 16;
  typedef typename std::conditional<OUTF32, float, T>::type TO;   // element type of the stored output
  constexpr int EPC = 16 / OEBc;       // elements per 16-byte chunk of the output
  constexpr int CPR = BN / EPC;        // chunks per tile row
  constexpr int NPASS = BM * CPR / NT; // store passes: pass k handles tile row tid / CPR + k * (NT / CPR)
  constexpr int PB = !EPI ? 1 : (NPASS < 8 ? NPASS : 8);
  static_assert(NT % CPR == 0 && NPASS % PB == 0, "a thread keeps one chunk column for the whole store loop");
  const int epc = EPC, cpr = CPR;
  const int ecol = tid % CPR, erow0 = tid / CPR;
  const int c_first = n0 + ecol * EPC;
  const bool vec_ok = ((p.Co * OEB) & 15) == 0 && c_first + EPC <= p.Co;
  const bool bnb = EPI && p.bn_y != nullptr;
  // EPI instantiations only (dgrad with a residual-branch addend and / or the fused BN-backward
  // reduction; the plain kernel keeps its register budget).  Global-side epilogue operands (residual-branch gradient, BN input, ReLU bits) are fetched a batch of
  // passes at a time, the first batch *before* the accumulators are staged through LDS, so their
  // latency is overlapped instead of being exposed once per store pass.
  const bool pre = EPI && vec_ok && (p.addend != nullptr || bnb);
  u32x4 adv[PB], yvv[PB];
  unsigned int bitv[PB];
  auto preload_y = [&](int k0) {    // BN input tiles
#pragma unroll
    for (int kk = 0; kk < PB; ++kk) {
      const int pix = s_outpix[erow0 + (k0 + kk) * (NT / CPR)];
      const size_t goff = ((size_t)(pix < 0 ? 0 : pix) * (size_t)p.Co + (size_t)c_first) * OEBc;
      const u32x4 zero = {0u, 0u, 0u, 0u};
      yvv[kk] = (pix >= 0 && bnb) ? cn_ld16(p.bn_y + goff) : zero;
    }
  };
  auto preload_ab = [&](int k0) {   // residual-branch gradient and ReLU bits
#pragma unroll
    for (int kk = 0; kk < PB; ++kk) {
      const int pix = s_outpix[erow0 + (k0 + kk) * (NT / CPR)];
      const size_t goff = ((size_t)(pix < 0 ? 0 : pix) * (size_t)p.Co + (size_t)c_first) * OEBc;
      const u32x4 zero = {0u, 0u, 0u, 0u};
      // a subsampled addend (the gradient of a stride-2 1x1 projection: zero off the even pixels) is read where it exists
      const int apx = p.addend_sub == 2 ? s_addpix[erow0 + (k0 + kk) * (NT / CPR)] : pix;
      const size_t aoff = ((size_t)(apx < 0 ? 0 : apx) * (size_t)p.Co + (size_t)c_first) * OEBc;
      adv[kk] = (pix >= 0 && apx >= 0 && p.addend != nullptr) ? cn_ld16(p.addend + aoff) : zero;
      bitv[kk] = (pix >= 0 && bnb && p.bn_mask != nullptr) ? (unsigned int)p.bn_mask[goff >> 4] : 0u;
    }
  };
  // the BN-input tile is requested before the reduction loop (for the short reductions of the 1x1 layers it
  // is then in flight together with the GEMM operands); the addend / mask bits follow once the K loop's
  // staging registers are free (two waves per SIMD fit in the register file that way)
  if (pre && bnb) preload_y(0);

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  u32x4 preg[NPR], wreg[NWR];
  u32x4 preg2[XF ? NPR : 1];   // XF mode 2: the second source's chunks
  const int nkt = (p.nchunks + 7) >> 3;
  const bool simple = p.simple != 0;   // every tap of every valid row is inside the image

  unsigned int xf_ok = 0;   // XF: which of this thread's staged pixel rows hold real data (bit i), and their channel chunk
  int xf_chunk = 0;
  // per-K-tile addressing state of this thread (set by tile_addr, consumed by issue_p / issue_w)
  unsigned int t_kb = 0, t_wofs = 0, t_xofs = 0;
  int t_dh = 0, t_dw = 0;
  auto tile_addr = [&](int kt) {
    const int kc = kt * 8 + cg;
    const bool kvalid = kc < p.nchunks;
    int tap = 0, cchunk = kc;
    if (p.ntaps > 1) {
      tap = kvalid ? (int)cn_fastdiv((unsigned)kc, p.div_cpt) : 0;
      cchunk = kc - tap * p.cpt;
    }
    const int dhdw = s_taps[4 * tap];
    t_kb = kvalid ? (unsigned int)(cchunk * 16) : CN_OOB;   // 16 bytes per chunk
    t_wofs = (unsigned int)s_taps[4 * tap + 1] + t_kb;
    t_xofs = (unsigned int)s_taps[4 * tap + 2] + t_kb;
    t_dh = (int)(short)(dhdw & 0xffff);
    t_dw = dhdw >> 16;
    if (XF) { xf_ok = 0; xf_chunk = cchunk; }
  };
  auto issue_p = [&](int i, int buf) {   // pixel row r0 + RS*i of the tile
    char* dp_ = lds + buf * STAGE + BN * 128 + (8 * wave) * 128;  // this wave's KiB of pixel rows
    unsigned int o;
    if (!ILV && simple) {   // (the interleaved form keeps ONE straight-line body: no branch between its MFMAs)
      o = (prow[i] | t_kb) >= CN_OOB ? CN_OOB : prow[i] + t_xofs;
    } else {
      // (bitwise &: one straight-line predicate, no short-circuit branches between the MFMAs of the interleaved form)
      const bool ok = ((unsigned)(phin[i] + t_dh) < (unsigned)p.Hi) & ((unsigned)(pwin[i] + t_dw) < (unsigned)p.Wi) &
                      (t_kb < CN_OOB);
      o = ok ? prow[i] + t_xofs : CN_OOB;
    }
    if (XF) xf_ok |= (o < CN_OOB ? 1u : 0u) << i;
    if (GLDS) cn_buf_ld16_lds(xbuf, o, dp_ + i * RS * 128);
    else preg[i] = cn_buf_ld16(xbuf, o);
    if (XF == 3) preg2[i] = cn_buf_ld16(x2buf, o);
    else if (XF) preg2[i] = cn_buf_ld16(x2buf, o);
  };
  auto issue_w = [&](int i, int buf) {   // filter row r0 + RS*i of the tile
    char* dw_ = lds + buf * STAGE + (8 * wave) * 128;             // this wave's KiB of filter rows
    const unsigned int o = (wrow[i] | t_kb) >= CN_OOB ? CN_OOB : wrow[i] + t_wofs;
    if (GLDS) cn_buf_ld16_lds(wbuf, o, dw_ + i * RS * 128);
    else wreg[i] = cn_buf_ld16(wbuf, o);
  };
  auto load_tile = [&](int kt, int buf) {
    tile_addr(kt);
#pragma unroll
    for (int i = 0; i < NPR; ++i) issue_p(i, buf);
#pragma unroll
    for (int i = 0; i < NWR; ++i) issue_w(i, buf);
  };
  auto store_tile = [&](int buf) {
    char* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NWR; ++i) cn_st16(base + st0 + i * RS * 128, wreg[i]);
    if (XF == 3) {   // lazy z: the junction's apply (bn_apply_kernel's order and rounding), stored to xz as well
      constexpr int CH = ElemTraits<T>::kChunk;
      const bool dual = p.xf2 != nullptr;
      float sc[CH], sh[CH], rs[CH], rb[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        sc[e] = s_xf[xf_chunk * CH + e];
        sh[e] = s_xf[p.Ci + xf_chunk * CH + e];
        rs[e] = dual ? s_xf[2 * p.Ci + xf_chunk * CH + e] : 1.f;
        rb[e] = dual ? s_xf[3 * p.Ci + xf_chunk * CH + e] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NPR; ++i) {
        float f[CH], r[CH];
        Chunk<T>::unpack(preg[i], f);
        Chunk<T>::unpack(preg2[i], r);
#pragma unroll
        for (int e = 0; e < CH; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
        if (dual) {
#pragma unroll
          for (int e = 0; e < CH; ++e) r[e] = fmaf(r[e], rs[e], rb[e]);
          Chunk<T>::unpack(Chunk<T>::pack(r), r);
        }
#pragma unroll
        for (int e = 0; e < CH; ++e) f[e] += r[e];
        unsigned int bits = 0;
        if (p.xf_relu) {
#pragma unroll
          for (int e = 0; e < CH; ++e) {
            bits |= (f[e] > 0.f ? 1u : 0u) << e;
            f[e] = f[e] > 0.f ? f[e] : 0.f;
          }
        }
        const u32x4 v = Chunk<T>::pack(f);
        const bool ok = ((xf_ok >> i) & 1u) != 0;
        if (ok) {   // (1x1, stride 1: the element's offset in z is its offset in x)
          const unsigned int o = prow[i] + t_xofs;
          cn_st16(p.xz + o, v);
          if (p.xf_relu && p.xz_mask != nullptr) p.xz_mask[o >> 4] = (unsigned char)bits;
        }
        preg[i] = ok ? v : cn_zero16();
      }
    } else if (XF) {
      constexpr int CH = ElemTraits<T>::kChunk;
      {   // XF = 1 is mode 2: dy = c1*g + c2*y + c3, the operation order of bn_bwd_apply_kernel (bit-identical operand)
        float c1[CH], c2[CH], c3[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          c1[e] = s_xf[xf_chunk * CH + e];
          c2[e] = s_xf[p.Ci + xf_chunk * CH + e];
          c3[e] = s_xf[2 * p.Ci + xf_chunk * CH + e];
        }
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
          float g[CH], v[CH];
          Chunk<T>::unpack(preg[i], g);
          Chunk<T>::unpack(preg2[i], v);
#pragma unroll
          for (int e = 0; e < CH; ++e) g[e] = fmaf(c1[e], g[e], fmaf(c2[e], v[e], c3[e]));
          const u32x4 o = Chunk<T>::pack(g);
          preg[i] = ((xf_ok >> i) & 1u) ? o : cn_zero16();
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NPR; ++i) cn_st16(base + st0 + BN * 128 + i * RS * 128, preg[i]);
  };
  auto compute = [&](int buf) {
    const char* wt = lds + buf * STAGE + rd_w;
    const char* pt = lds + buf * STAGE + rd_p;
    if (FRAGDB) {
      // fragments double-buffered in registers: the ds_read_b128 of k-step kk+1 are issued (and kept
      // there by a scheduling fence) ahead of the MFMAs of k-step kk
      u32x4 af[2][TI], bfr[2][TJ];
#pragma unroll
      for (int a = 0; a < TI; ++a) af[0][a] = cn_ld16(wt + koff[0] + a * 32 * 128);
#pragma unroll
      for (int b = 0; b < TJ; ++b) bfr[0][b] = cn_ld16(pt + koff[0] + b * 32 * 128);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk < 3) {
#pragma unroll
          for (int a = 0; a < TI; ++a) af[nxt][a] = cn_ld16(wt + koff[kk + 1] + a * 32 * 128);
#pragma unroll
          for (int b = 0; b < TJ; ++b) bfr[nxt][b] = cn_ld16(pt + koff[kk + 1] + b * 32 * 128);
        }
        cn_sched_fence();
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int b = 0; b < TJ; ++b) ig_mma<T>(af[cur][a], bfr[cur][b], acc[a][b]);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4 af[TI], bfr[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) af[a] = cn_ld16(wt + koff[kk] + a * 32 * 128);
#pragma unroll
        for (int b = 0; b < TJ; ++b) bfr[b] = cn_ld16(pt + koff[kk] + b * 32 * 128);
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int b = 0; b < TJ; ++b) ig_mma<T>(af[a], bfr[b], acc[a][b]);
      }
    }
  };

  // ILV: compute(buf) with the DMA instructions of the next K tile (addresses set by tile_addr) issued one after every
  // IEVERY-th MFMA, from the first MFMA on: all NPR+NWR of them are out by the middle of the tile, so they have half
  // a tile of MFMA time to land before the barrier, and none is issued while the matrix pipe idles.
  auto compute_ilv = [&](int buf, int nbuf) {
    constexpr int IEVERY = (TI * TJ >= 8) ? 2 : 1;
    static_assert(!ILV || (NPR + NWR) * IEVERY <= 3 * TI * TJ, "the DMA instructions must all be issued in the first three k-steps");
    const char* wt = lds + buf * STAGE + rd_w;
    const char* pt = lds + buf * STAGE + rd_p;
    constexpr int NFB = FRAGDB ? 2 : 1;   // fragment buffers: double (reads of k-step kk+1 ahead of the MFMAs of kk) or single
    u32x4 af[NFB][TI], bfr[NFB][TJ];
    if (FRAGDB) {
#pragma unroll
      for (int a = 0; a < TI; ++a) af[0][a] = cn_ld16(wt + koff[0] + a * 32 * 128);
#pragma unroll
      for (int b = 0; b < TJ; ++b) bfr[0][b] = cn_ld16(pt + koff[0] + b * 32 * 128);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = FRAGDB ? (kk & 1) : 0, nxt = FRAGDB ? (cur ^ 1) : 0;
      if (FRAGDB) {
        if (kk < 3) {
#pragma unroll
          for (int a = 0; a < TI; ++a) af[nxt][a] = cn_ld16(wt + koff[kk + 1] + a * 32 * 128);
#pragma unroll
          for (int b = 0; b < TJ; ++b) bfr[nxt][b] = cn_ld16(pt + koff[kk + 1] + b * 32 * 128);
        }
        cn_sched_fence();
      } else {
#pragma unroll
        for (int a = 0; a < TI; ++a) af[0][a] = cn_ld16(wt + koff[kk] + a * 32 * 128);
#pragma unroll
        for (int b = 0; b < TJ; ++b) bfr[0][b] = cn_ld16(pt + koff[kk] + b * 32 * 128);
      }
#pragma unroll
      for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) {
          ig_mma<T>(af[cur][a], bfr[cur][b], acc[a][b]);
          const int m = kk * TI * TJ + a * TJ + b;      // compile-time after unrolling
          if (m % IEVERY == 0 && m / IEVERY < NPR + NWR) {
            const int slot = m / IEVERY;
            if (slot < NPR) issue_p(slot, nbuf);
            else issue_w(slot - NPR, nbuf);
            cn_sched_fence();
          }
        }
    }
  };

  if (nkt > 0) {
    load_tile(0, 0);
    if (!GLDS) store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      if (GLDS && ILV) {
        // ONE loop body for every tile (a second, DMA-free body for the last tile made the register allocator keep two
        // accumulator sets and copy between them): past the end tile_addr() yields out-of-range offsets, so the last
        // tile's DMA instructions fetch nothing and write zeros into the buffer nobody reads again; they have landed
        // before the barrier below, i.e. before the epilogue reuses the LDS.
        const int buf = kt & 1;
        tile_addr(kt + 1);
        compute_ilv(buf, buf ^ 1);                      // DMA issue interleaved with this tile's MFMAs
        __syncthreads();
      } else if (GLDS) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1, buf ^ 1);   // DMA into the other buffer while we compute
        compute(buf);
        __syncthreads();                                // (hipcc drains vmcnt before the barrier)
      } else {
        if (kt + 1 < nkt) load_tile(kt + 1, 0);
        compute(0);
        __syncthreads();
        if (kt + 1 < nkt) {
          store_tile(0);
          __syncthreads();
        }
      }
    }
  }

  if (pre) preload_ab(0);
  // ---- epilogue: (bias, ReLU) -> LDS out tile [BM pixels][BN channels] -> coalesced global store
  if (p.bias != nullptr) {   // uniform branch; a lane's 4 consecutive channels = one 16-byte bias load
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + (wc * TI + a) * 32 + 8 * q + 4 * (lane >> 5);
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < p.Co) {
          const f32x4 t = *(const f32x4*)(p.bias + c);
          bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
        } else {
          for (int e = 0; e < 4; ++e)
            if (c + e < p.Co) bv[e] = p.bias[c + e];
        }
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][b][q * 4 + e] += bv[e];
      }
  }
  if (p.relu) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = acc[a][b][r] > 0.f ? acc[a][b][r] : 0.f;
  }
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int prow_l = (wp * TJ + b) * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = (wc * TI + a) * 32 + 8 * q + 4 * (lane >> 5);
        char* dst = lds + prow_l * pitch + c * OEB;
        if (OEB == 4) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[a][b][q * 4 + e];
          *(f32x4*)dst = v;
        } else {
          u32x2 pk;
          pk[0] = cn_pack2<T>(acc[a][b][q * 4], acc[a][b][q * 4 + 1]);
          pk[1] = cn_pack2<T>(acc[a][b][q * 4 + 2], acc[a][b][q * 4 + 3]);
          *(u32x2*)dst = pk;
        }
      }
    }
  __syncthreads();
  // ---- optional BatchNorm statistics of this tile (what bn_stats_kernel would re-read from HBM):
  // per-channel sum and sum of squares of the *stored* (rounded) outputs, one partial row per pixel tile
  constexpr int NCOL = BN * OEBc / 4;   // dword columns of the out tile (2 channels each for bf16)
  constexpr int NG = NT / NCOL;         // row groups
  constexpr int RPG = BM / NG;
  float st[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.stats != nullptr) {
    const int col = tid % NCOL, g = tid / NCOL;
    int nrows = p.M - m0;
    if (nrows > BM) nrows = BM;
    const char* src = lds + col * 4 + g * RPG * pitch;
    // centred sums: with a per-channel pivot near the mean (the consumer BatchNorm's running mean) the fp32
    // sum of squares no longer cancels against mean^2 when |mean| >> sigma
    float pv0 = 0.f, pv1 = 0.f;
    if (p.stats_pivot != nullptr) {
      const int c = n0 + (OEBc == 4 ? col : 2 * col);
      if (c < p.Co) pv0 = cn_pivot(p.stats_pivot[c]);
      if (OEBc != 4 && c + 1 < p.Co) pv1 = cn_pivot(p.stats_pivot[c + 1]);
    }
    auto acc_row = [&](int r) {
      const unsigned int v = *(const unsigned int*)(src + r * pitch);
      if (OEBc == 4) {
        const float f = __builtin_bit_cast(float, v) - pv0;
        st[0] += f;
        st[2] = fmaf(f, f, st[2]);
      } else {
        float lo, hi;
        cn_unpack2<T>(v, lo, hi);
        lo -= pv0;
        hi -= pv1;
        st[0] += lo;
        st[1] += hi;
        st[2] = fmaf(lo, lo, st[2]);
        st[3] = fmaf(hi, hi, st[3]);
      }
    };
    if (nrows == BM) {
#pragma unroll 8
      for (int r = 0; r < RPG; ++r) acc_row(r);
    } else {
      const int cnt = nrows - g * RPG;
      for (int r = 0; r < RPG && r < cnt; ++r) acc_row(r);
    }
  }
  // per-channel BN coefficients of this thread's chunk column (loaded after the staging pass: kept out of its register peak)
  float bs1[EPC], bs2[EPC], bmu[EPC], bis[EPC], bsc[EPC], bsh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { bs1[e] = 0.f; bs2[e] = 0.f; bmu[e] = 0.f; bis[e] = 0.f; bsc[e] = 0.f; bsh[e] = 0.f; }
  if (bnb && vec_ok) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      bmu[e] = p.bn_coef[c_first + e];
      bis[e] = p.bn_coef[p.Co + c_first + e];
      bsc[e] = p.bn_coef[2 * p.Co + c_first + e];
      bsh[e] = p.bn_coef[3 * p.Co + c_first + e];
    }
  }
  if (c_first < p.Co) {
    for (int k0 = 0; k0 < NPASS; k0 += PB) {
      if (pre && k0 > 0) { preload_y(k0); preload_ab(k0); }
#pragma unroll
      for (int kk = 0; kk < PB; ++kk) {
        const int row = erow0 + (k0 + kk) * (NT / CPR);
        const int pix = s_outpix[row];
        if (pix < 0) continue;
        const char* src = lds + row * pitch + ecol * 16;
        const size_t goff = ((size_t)pix * (size_t)p.Co + (size_t)c_first) * OEB;
        char* dst = p.y + goff;
        const int apx = (EPI && p.addend_sub == 2) ? s_addpix[row] : pix;   // element-wise tail below
        const size_t aoff = ((size_t)(apx < 0 ? 0 : apx) * (size_t)p.Co + (size_t)c_first) * OEB;
        if (vec_ok) {
          u32x4 v = cn_ld16(src);
          if (pre) {
            float fv[EPC];
            Chunk<TO>::unpack(v, fv);
            if (p.addend != nullptr) {   // e.g. the residual-branch gradient folded into dgrad
              float fa[EPC];
              Chunk<TO>::unpack(adv[kk], fa);
#pragma unroll
              for (int e = 0; e < EPC; ++e) fv[e] += fa[e];
            }
            if (bnb) {
              float yv[EPC];
              Chunk<TO>::unpack(yvv[kk], yv);
              if (p.bn_mask != nullptr) {
                const unsigned int bits = bitv[kk];
#pragma unroll
                for (int e = 0; e < EPC; ++e) fv[e] = ((bits >> e) & 1u) ? fv[e] : 0.f;
              } else if (p.bn_relu) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) fv[e] = fmaf(yv[e], bsc[e], bsh[e]) > 0.f ? fv[e] : 0.f;
              }
              v = Chunk<TO>::pack(fv);
              // statistics of the values as stored: exact already unless an addend made fv wider than T
              if (p.addend != nullptr) Chunk<TO>::unpack(v, fv);
#pragma unroll
              for (int e = 0; e < EPC; ++e) {   // sum g and sum g*y; the mean / invstd fix-up is per thread, below
                bs1[e] += fv[e];
                bs2[e] = fmaf(fv[e], yv[e], bs2[e]);
              }
            } else {
              v = Chunk<TO>::pack(fv);
            }
          }
          cn_st16(dst, v);   // plain store: a non-temporal one measured neutral here (profiles/README.md)
        } else {   // ragged channel count: element-wise tail (never combined with the BN reduction)
          for (int e = 0; e < epc && c_first + e < p.Co; ++e) {
            if (OEB == 4) {
              float f = ((const float*)src)[e];
              if (p.addend != nullptr && apx >= 0) f += ((const float*)(p.addend + aoff))[e];
              ((float*)dst)[e] = f;
            } else {
              float f = cn_load_elem<T>((const T*)src + e);
              if (p.addend != nullptr && apx >= 0) f += cn_load_elem<T>((const T*)(p.addend + aoff) + e);
              cn_store_elem<T>((T*)dst + e, f);
            }
          }
        }
      }
    }
  }
  (void)cpr;
  if (bnb) {
    __syncthreads();   // the out tile has been consumed: its first 16 KiB take the per-thread partials
    float* red = (float*)lds;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      // sum g*xhat = invstd * (sum g*y - mean * sum g) over this thread's <= 16 rows (fp32: the cancellation
      // costs ~1e-7 * |mean|/sigma relative, far inside the stated tolerances)
      red[tid * 2 * EPC + e] = bs1[e];
      red[tid * 2 * EPC + EPC + e] = bis[e] * (bs2[e] - bmu[e] * bs1[e]);
    }
    __syncthreads();
    if (tid < BN) {
      const int cchunk = tid / EPC, e = tid % EPC;
      float a1 = 0.f, a2 = 0.f;
      for (int j = 0; j < NT / CPR; ++j) {   // fixed order: deterministic
        const float* o = red + (j * CPR + cchunk) * 2 * EPC;
        a1 += o[e];
        a2 += o[EPC + e];
      }
      const int c = n0 + tid;
      if (c < p.Co) {
        float* dst = p.bn_partial + (size_t)(p.bn_row0 + mt) * 2 * (size_t)p.Co;
        dst[c] = a1;
        dst[p.Co + c] = a2;
      }
    }
  }
  if (p.stats != nullptr) {
    __syncthreads();   // the out tile has been consumed: reuse its first bytes for the row-group partials
    f32x4* red = (f32x4*)lds;
    f32x4 mine;
    mine[0] = st[0]; mine[1] = st[1]; mine[2] = st[2]; mine[3] = st[3];
    red[tid] = mine;
    __syncthreads();
    if (tid < NCOL) {
      f32x4 a = red[tid];
#pragma unroll
      for (int g = 1; g < NG; ++g) {   // fixed order: deterministic
        const f32x4 b = red[g * NCOL + tid];
        a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
      }
      // the statistics buffer has one row per 128 pixels (cn_conv2d_bnstats_rows): a 256-pixel tile fills the
      // first of its two rows and zeroes the second
      constexpr int RPT = BM / 128 > 0 ? BM / 128 : 1;
      float* dst = p.stats + (size_t)mt * RPT * 2 * (size_t)p.Co;
      if (OEBc == 4) {
        const int c = n0 + tid;
        if (c < p.Co) { dst[c] = a[0]; dst[p.Co + c] = a[2]; }
      } else {
        const int c = n0 + 2 * tid;
        if (c < p.Co) { dst[c] = a[0]; dst[p.Co + c] = a[2]; }
        if (c + 1 < p.Co) { dst[c + 1] = a[1]; dst[p.Co + c + 1] = a[3]; }
      }
      if (RPT > 1) {
#pragma unroll
        for (int x = 1; x < RPT; ++x) {
          if (mt * RPT + x >= p.stats_rows) break;
          float* dz = dst + (size_t)x * 2 * (size_t)p.Co;
          const int c = n0 + (OEBc == 4 ? tid : 2 * tid);
          if (c < p.Co) { dz[c] = 0.f; dz[p.Co + c] = 0.f; }
          if (OEBc != 4 && c + 1 < p.Co) { dz[c + 1] = 0.f; dz[p.Co + c + 1] = 0.f; }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pixels per partial row of the fused BN-backward reduction
static int ig_epi_rows_bm(int) { return 128; }

template <typename T, bool OUTF32>
static int ig_launch(IgemmParams& p, hipStream_t stream) {
  const int nkt = (p.nchunks + 7) / 8;
  // variant: 1 = register-staged single buffer (short reductions: more workgroups per CU), 3 = LDS-DMA double buffer
  // (from "igemm_dma_min_nkt" K tiles on; measured per layer, profiles/r01_conv_layers_b256_bf16.txt); knob
  // "igemm_variant" forces one (tests / tools/bench_layers.py), 0 / unset = the heuristic
  int variant = cn_get_option("igemm_variant", 0);
  if (variant != 1 && variant != 3) variant = nkt < cn_get_option("igemm_dma_min_nkt", 16) ? 1 : 3;
  const bool epi = p.addend != nullptr || p.bn_y != nullptr;   // epilogue with global-side operands
  const int BM = 128, BN = p.Co <= 64 ? 64 : 128;
  p.n_ntiles = (p.Co + BN - 1) / BN;
  const int n_mtiles = (p.M + BM - 1) / BM;
  p.n_mtiles = n_mtiles;
  dim3 grid((unsigned)(p.n_ntiles * n_mtiles));
  const char* tname = std::is_same<T, float>::value ? "float" : (std::is_same<T, f16_t>::value ? "f16_t" : "bf16_t");
  // EPI: epilogue with global-side operands (residual-branch addend, fused BN-backward reduction)
#define IG_GO2(WC, WP, TI, TJ, EP)                                                                              \
  do {                                                                                                         \
    cn_set_last_kernel("igemm_kernel<%s, %d, %d, %d, %d, %d, %s, %s, false, %s, false, false>", tname, WC, WP, TI, TJ, \
                       variant == 1 ? 1 : 2, OUTF32 ? "true" : "false", variant >= 3 ? "true" : "false",      \
                       EP ? "true" : "false");                                                                 \
    if (variant == 1) CN_LAUNCH((igemm_kernel<T, WC, WP, TI, TJ, 1, OUTF32, false, false, EP>), grid, dim3(256), stream, p); \
    else CN_LAUNCH((igemm_kernel<T, WC, WP, TI, TJ, 2, OUTF32, true, false, EP>), grid, dim3(256), stream, p);        \
  } while (0)
#define IG_GO(WC, WP, TI, TJ) do { if (epi) IG_GO2(WC, WP, TI, TJ, true); else IG_GO2(WC, WP, TI, TJ, false); } while (0)
  if (p.xf != nullptr && p.xf_mode == 3) {   // lazy z: the junction apply on the operand load, z stored by this kernel
    if (epi || p.Ci > IG_XF_MAX || !p.simple || p.ntaps != 1 || p.n_ntiles != 1 || p.xz == nullptr || p.x2 == nullptr) {
      cn_set_error("igemm: lazy z needs a 1x1 / stride-1 gather, one channel tile (Co <= 128), <= %d input channels", IG_XF_MAX);
      return CN_EINVAL;
    }
    // the tile shapes the plain convolution would run on (same statistics-partial association => same bits downstream):
    // 64-channel tile, else the 128 x 128 tile on eight waves (16-bit storage, short reductions) or on four
    bool w8 = false;
    if constexpr (sizeof(T) == 2 && !OUTF32) w8 = p.Co > 64 && nkt <= cn_get_option("igemm_8w", 16);
    cn_set_last_kernel("igemm_kernel<%s, %s, 1, %s, false, false, false, 3, false>", tname,
                       p.Co <= 64 ? "1, 4, 2, 1" : (w8 ? "2, 4, 2, 1" : "2, 2, 2, 2"), OUTF32 ? "true" : "false");
    if (p.Co <= 64) CN_LAUNCH((igemm_kernel<T, 1, 4, 2, 1, 1, OUTF32, false, false, false, 3>), grid, dim3(256), stream, p);
    else if (w8) {
      if constexpr (sizeof(T) == 2 && !OUTF32)
        CN_LAUNCH((igemm_kernel<T, 2, 4, 2, 1, 1, false, false, false, false, 3>), grid, dim3(512), stream, p);
    } else CN_LAUNCH((igemm_kernel<T, 2, 2, 2, 2, 1, OUTF32, false, false, false, 3>), grid, dim3(256), stream, p);
    return cn_check_launch("igemm");
  }
  if (p.xf != nullptr) {   // operand transform: register-staged single buffer only
    if (epi || p.Ci > IG_XF_MAX) { cn_set_error("igemm: operand transform with an epilogue operand / more than %d channels", IG_XF_MAX); return CN_EINVAL; }
    if (p.xf_mode != 2) { cn_set_error("igemm: unknown operand transform %d", p.xf_mode); return CN_EINVAL; }
    cn_set_last_kernel("igemm_kernel<%s, %s, 1, %s, false, false, false, true, false>", tname,
                       p.Co <= 64 ? "1, 4, 2, 1" : "2, 2, 2, 2", OUTF32 ? "true" : "false");
    if (p.Co <= 64) CN_LAUNCH((igemm_kernel<T, 1, 4, 2, 1, 1, OUTF32, false, false, false, 1>), grid, dim3(256), stream, p);
    else CN_LAUNCH((igemm_kernel<T, 2, 2, 2, 2, 1, OUTF32, false, false, false, 1>), grid, dim3(256), stream, p);
    return cn_check_launch("igemm");
  }
  if constexpr (sizeof(T) == 2 && !OUTF32) {
    // 256 pixels x 256 channels, 8 waves of 128 pixels x 64 channels, LDS-DMA double buffer, fragments double-buffered
    // in registers, one workgroup per CU: half the L2->LDS bytes per flop of the 128x128 tile.  It wins where the
    // reduction is long enough to amortise the big tile's ramp (>= 16 K tiles) and the launch still has about a
    // tile per CU (measured per layer, profiles/r02c_conv_layers_256sq_tile.txt: 256-wide 3x3 at 14x14 690 -> 900
    // TFLOP/s, 1024 -> 256 1x1 550 -> 760); short reductions and the 7x7 maps (98 tiles) stay on 128x128.
    const int big = cn_get_option("igemm_256sq", -1);   // -1 heuristic, 0 never, 1 whenever the shape allows it
    const long long tiles256 = (long long)((p.M + 255) / 256) * ((p.Co + 255) / 256);
    const bool shape_ok = p.Co >= 256 && p.Co % 128 == 0 && !epi;
    const bool want = big == 1 || (big < 0 && nkt >= cn_get_option("igemm_256sq_min_nkt", 16) &&
                                   tiles256 >= cn_get_option("igemm_256sq_min_tiles", 160));
    if (shape_ok && big != 0 && (cn_get_option("igemm_variant", 0) == 0 && want)) {
      p.n_ntiles = (p.Co + 255) / 256;
      p.n_mtiles = (p.M + 255) / 256;
      dim3 g2((unsigned)(p.n_ntiles * p.n_mtiles));
      // interleaved DMA issue, single fragment buffer (with register-double-buffered fragments it measured 1-2 % slower,
      // without the interleave 7-10 % slower: profiles/r03_ab_ilv_with_fragdb_rejected.txt, r03_ab_igemm_interleaved_dma_issue.txt)
      cn_set_last_kernel("igemm_kernel<%s, 4, 2, 2, 4, 2, false, true, false, false, false, true>", tname);
      CN_LAUNCH((igemm_kernel<T, 4, 2, 2, 4, 2, false, true, false, false, false, true>), g2, dim3(512), stream, p);
      return cn_check_launch("igemm");
    }
  }
  if constexpr (sizeof(T) == 2 && !OUTF32) {
    // 128 x 128 LDS-DMA double buffer with interleaved DMA issue (see ILV)
    if (variant == 3 && !epi && p.Co > 64) {
      cn_set_last_kernel("igemm_kernel<%s, 2, 2, 2, 2, 2, false, true, true, false, false, true>", tname);
      CN_LAUNCH((igemm_kernel<T, 2, 2, 2, 2, 2, false, true, true, false, false, true>), grid, dim3(256), stream, p);
      return cn_check_launch("igemm");
    }
  }
  // (a 64 x 256 tile for <= 64-channel layers - register-staged or LDS-DMA + interleaved issue - measured no faster /
  //  slower than the 64 x 128 tile: profiles/r03_ab_c64_tile_rejected.txt)
  if constexpr (sizeof(T) == 2 && !OUTF32) {
    // junction dgrads (EPI: residual-branch addend + BatchNorm-backward reduction in the epilogue) are epilogue-bound
    // streaming kernels with a short reduction: the same 128 x 128 tile on EIGHT waves (half the staging / epilogue
    // registers per thread -> four waves per SIMD instead of two) hides the epilogue operands' latency better.
    // Knob "igemm_epi_8w" (A/B; the stored outputs are identical - same tile, same accumulation order - the per-tile
    // partial sums of the fused reduction are associated over 512 instead of 256 threads).  Whole step: -0.45 %
    // (profiles/r03_ab_whole_step_knobs.txt).
    if (epi && p.Co > 64 && variant == 1 && cn_get_option("igemm_epi_8w", 1) != 0) {
      cn_set_last_kernel("igemm_kernel<%s, 2, 4, 2, 1, 1, false, false, false, true, false, false>", tname);
      CN_LAUNCH((igemm_kernel<T, 2, 4, 2, 1, 1, false, false, false, true>), grid, dim3(512), stream, p);
      return cn_check_launch("igemm");
    }
    // the same eight-wave form of the plain register-staged tile for short reductions (A/B knob "igemm_8w": number of
    // K tiles up to which it is used; 0 = off)
    if (!epi && p.Co > 64 && variant == 1 && nkt <= cn_get_option("igemm_8w", 16)) {
      cn_set_last_kernel("igemm_kernel<%s, 2, 4, 2, 1, 1, false, false, false, false, false, false>", tname);
      CN_LAUNCH((igemm_kernel<T, 2, 4, 2, 1, 1, false, false, false, false>), grid, dim3(512), stream, p);
      return cn_check_launch("igemm");
    }
  }
  if (p.Co <= 64) IG_GO(1, 4, 2, 1);
  else IG_GO(2, 2, 2, 2);
#undef IG_GO
#undef IG_GO2
  return cn_check_launch("igemm");
}

static int ig_dispatch(IgemmParams& p, int dtype, hipStream_t stream) {
  if (p.M <= 0 || p.Co <= 0) return CN_OK;
  if (p.x == nullptr || p.w == nullptr || p.y == nullptr) { cn_set_error("igemm: null operand"); return CN_EINVAL; }
  if (dtype == CN_BF16) return p.out_f32 ? ig_launch<bf16_t, true>(p, stream) : ig_launch<bf16_t, false>(p, stream);
  if (dtype == CN_F16) return p.out_f32 ? ig_launch<f16_t, true>(p, stream) : ig_launch<f16_t, false>(p, stream);
  if (dtype == CN_F32) { p.out_f32 = 1; return ig_launch<float, true>(p, stream); }
  cn_set_error("igemm: bad dtype %d", dtype);
  return CN_EINVAL;
}

static int ig_common(IgemmParams& p, int dtype, int Ci, int ntaps) {
  const int CH = cn_dtype_chunk(dtype);
  if (Ci % CH != 0) {
    cn_set_error("igemm: reduction channels %d not a multiple of the 16-byte chunk (%d elems)", Ci, CH);
    return CN_ESHAPE;
  }
  if (ntaps > IG_MAX_TAPS) {
    cn_set_error("igemm: %d taps > %d", ntaps, IG_MAX_TAPS);
    return CN_ESHAPE;
  }
  p.cpt = Ci / CH;
  p.div_cpt = cn_make_fastdiv((unsigned)p.cpt);
  p.ntaps = ntaps;
  p.nchunks = ntaps * p.cpt;
  p.M = p.N * p.Hg * p.Wg;
  p.div_hw = cn_make_fastdiv((unsigned)(p.Hg * p.Wg));
  p.div_w = cn_make_fastdiv((unsigned)p.Wg);
  const int EB = cn_dtype_bytes(dtype);
  const long long xb = (long long)p.N * p.Hi * p.Wi * p.Ci * EB;
  const long long wb = (long long)p.Co * p.w_row * EB;
  if (xb >= (1ll << 31) || wb >= (1ll << 31)) {
    cn_set_error("igemm: operand of %lld bytes exceeds the 2 GiB buffer-descriptor window", xb > wb ? xb : wb);
    return CN_ESHAPE;
  }
  p.x_bytes = (unsigned int)xb;
  p.w_bytes = (unsigned int)wb;
  return CN_OK;
}

// 1 when every tap of every enumerated output position reads inside the source image
static int ig_is_simple(const IgemmParams& p, const int* dhdw, int ntaps) {
  for (int t = 0; t < ntaps; ++t) {
    const int dh = (int)(short)(dhdw[t] & 0xffff), dw = dhdw[t] >> 16;
    if (dh < 0 || dw < 0) return 0;
    if ((p.Hg - 1) * p.a_h + dh >= p.Hi || (p.Wg - 1) * p.a_w + dw >= p.Wi) return 0;
  }
  return 1;
}

struct IgLazyZ {   // XF mode 3 operands (see IgemmParams::xf2)
  const void* res;
  const float* xf2;
  void* z;
  unsigned char* mask;
};

int cn_dense_smallm(const void* A, const void* B, void* C, const float* bias, int M, int N, int Kd, int dtype, int out_f32,
                    int relu, hipStream_t stream);

static int ig_conv_fwd(const void* x, const void* w_krsc, void* y, const float* bias, float* stats, int N, int H,
                       int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                       int dtype, int out_f32, int relu, void* stream, const float* xf = nullptr, int xf_relu = 0,
                       const float* stats_pivot = nullptr, const IgLazyZ* lz = nullptr) {
  const int P = (H + 2 * pad_h - R) / stride_h + 1;
  const int Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0 || N <= 0) { cn_set_error("conv2d_fwd: empty output"); return CN_ESHAPE; }
  // a dense layer on a batch (1 x 1 image, 1 x 1 filter: the classifier): 32 x 32 output tiles with a four-way split of the
  // reduction (dense.hip) instead of 16 workgroups of the 128 x 128 tile
  if (H == 1 && W == 1 && R == 1 && S == 1 && pad_h == 0 && pad_w == 0 && stats == nullptr && xf == nullptr && lz == nullptr &&
      x != nullptr && w_krsc != nullptr && y != nullptr) {
    const int rc = cn_dense_smallm(x, w_krsc, y, bias, N, K, C, dtype, out_f32, relu, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)x; p.w = (const char*)w_krsc; p.y = (char*)y; p.bias = bias; p.stats = stats;
  p.xf = xf; p.xf_relu = xf_relu;
  if (lz != nullptr) { p.xf_mode = 3; p.x2 = (const char*)lz->res; p.xf2 = lz->xf2; p.xz = (char*)lz->z; p.xz_mask = lz->mask; }
  p.stats_pivot = stats_pivot;
  p.stats_rows = (int)(((long long)N * P * Q + 127) / 128);
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
  p.simple = ig_is_simple(p, p.tap_dhdw, R * S);
  return ig_dispatch(p, dtype, (hipStream_t)stream);
}

extern "C" int cn_conv2d_fwd(const void* x, const void* w_krsc, void* y, const float* bias, int N, int H,
                             int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                             int pad_w, int dtype, int out_f32, int relu, void* stream) {
  return ig_conv_fwd(x, w_krsc, y, bias, nullptr, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype,
                     out_f32, relu, stream);
}

// Rows of the statistics partial buffer cn_conv2d_fwd_bnstats writes for M = N*P*Q output pixels.
extern "C" int cn_conv2d_bnstats_rows(long long M) { return (int)((M + 127) / 128); }

// Convolution forward that also emits, per 128-pixel tile, the per-channel sum and sum of squares of
// the outputs it stores: partial[row][0:K] = sum, partial[row][K:2K] = sum of squares, with
// cn_conv2d_bnstats_rows(N*P*Q) rows.  cn_bn_fwd_train_partials consumes them, which removes the
// statistics pass over y (one full HBM read of the conv output per BatchNorm).
extern "C" int cn_conv2d_fwd_bnstats(const void* x, const void* w_krsc, void* y, const float* bias, int N, int H,
                                     int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                                     int pad_w, int dtype, int relu, float* partial, int partial_rows,
                                     void* stream) {
  const long long P = (H + 2 * pad_h - R) / stride_h + 1, Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (partial == nullptr || partial_rows < cn_conv2d_bnstats_rows((long long)N * P * Q)) {
    cn_set_error("conv2d_fwd_bnstats: partial buffer of %d rows is too small", partial_rows);
    return CN_EWORKSPACE;
  }
  return ig_conv_fwd(x, w_krsc, y, bias, partial, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype,
                     0, relu, stream);
}

// The same with centred statistics: partial rows hold sum (y - pivot[c]) | sum (y - pivot[c])^2 (pivot: K floats,
// normally the running mean of the BatchNorm that consumes y; cn_bn_fwd_train_partials_centered un-centres them).
extern "C" int cn_conv2d_fwd_bnstats_centered(const void* x, const void* w_krsc, void* y, const float* bias, int N,
                                              int H, int W, int C, int K, int R, int S, int stride_h, int stride_w,
                                              int pad_h, int pad_w, int dtype, int relu, float* partial,
                                              int partial_rows, const float* pivot, void* stream) {
  const long long P = (H + 2 * pad_h - R) / stride_h + 1, Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (partial == nullptr || pivot == nullptr || partial_rows < cn_conv2d_bnstats_rows((long long)N * P * Q)) {
    cn_set_error("conv2d_fwd_bnstats_centered: needs a pivot and a partial buffer of enough rows (%d given)", partial_rows);
    return CN_EWORKSPACE;
  }
  return ig_conv_fwd(x, w_krsc, y, bias, partial, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype,
                     0, relu, stream, nullptr, 0, pivot);
}

// "Lazy z" forward (round 3): the convolution's input is the output of a residual junction that has not been applied
// yet,   z = relu?( bn_y*scale[c] + shift[c] + r ),   r = res   or   r = round_T(res*rscale[c] + rshift[c])  (res_stats
// given: `res` is the projection shortcut's BatchNorm input, as in cn_bn_apply_dual).  The 1x1 / stride-1 convolution
// with K <= 128 output channels forms z on its operand load - bn_apply_kernel's operation order and rounding -, stores
// it to `z` (and the ReLU bits to z_mask, optional) and multiplies: the same bits as the junction's apply pass followed
// by cn_conv2d_fwd_bnstats, without that pass (read y, read res, write z) and without this convolution's re-read of z.
// stats / res_stats: the 4*C floats of cn_bn_fwd_train* (z = NULL).  partial (optional) as cn_conv2d_fwd_bnstats
// (pivot optional: centred sums).
extern "C" int cn_conv2d_fwd_lazyz(const void* bn_y, const void* res, const float* stats, const float* res_stats,
                                   int relu, void* z, unsigned char* z_mask, const void* w_krsc, void* y, int N, int H,
                                   int W, int C, int K, int dtype, float* partial, int partial_rows, const float* pivot,
                                   void* stream) {
  if (bn_y == nullptr || res == nullptr || stats == nullptr || z == nullptr) { cn_set_error("conv2d_fwd_lazyz: null operand"); return CN_EINVAL; }
  if (K > 128) { cn_set_error("conv2d_fwd_lazyz: %d output channels > 128 (one channel tile)", K); return CN_ESHAPE; }
  if (partial != nullptr && partial_rows < cn_conv2d_bnstats_rows((long long)N * H * W)) {
    cn_set_error("conv2d_fwd_lazyz: partial buffer of %d rows is too small", partial_rows);
    return CN_EWORKSPACE;
  }
  IgLazyZ lz;
  lz.res = res; lz.xf2 = res_stats != nullptr ? res_stats + 2 * C : nullptr; lz.z = z; lz.mask = z_mask;
  return ig_conv_fwd(bn_y, w_krsc, y, nullptr, partial, N, H, W, C, K, 1, 1, 1, 1, 0, 0, dtype, 0, 0, stream, stats + 2 * C,
                     relu, pivot, &lz);
}

struct IgBnBwd {
  const void* y;
  const unsigned char* mask;
  const float* coef;
  float* partial;
  int relu, rows_cap;
};

static int ig_conv_dgrad(const void* dy, const void* w_crsk, void* dx, const void* addend, int N, int H,
                         int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                         int pad_w, int dtype, int out_f32, const IgBnBwd* bn, void* stream, int addend_sub = 1,
                         const void* lazy_y = nullptr, const float* lazy_coef = nullptr) {
  if (addend_sub != 1 && addend_sub != 2) { cn_set_error("conv2d_dgrad: addend subsampling %d (1 or 2)", addend_sub); return CN_EINVAL; }
  int bn_row = 0;
  const int P = (H + 2 * pad_h - R) / stride_h + 1;
  const int Q = (W + 2 * pad_w - S) / stride_w + 1;
  if (P <= 0 || Q <= 0 || N <= 0) { cn_set_error("conv2d_dgrad: empty output"); return CN_ESHAPE; }
  // the classifier's data gradient: dx[B][C] = dy[B][K] * w_crsk[C][K]^T (dense.hip)
  if (H == 1 && W == 1 && R == 1 && S == 1 && pad_h == 0 && pad_w == 0 && stride_h == 1 && stride_w == 1 && addend == nullptr &&
      bn == nullptr && lazy_y == nullptr && !out_f32 && dy != nullptr && w_crsk != nullptr && dx != nullptr) {
    const int rc = cn_dense_smallm(dy, w_crsk, dx, nullptr, N, C, K, dtype, 0, 0, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  // One launch per output-parity class (ph, pw): dX[n, hg*st+ph, wg*st+pw, :] receives exactly the
  // taps r with (ph + pad - r) % st == 0, reading dY at row hg + (ph + pad - r)/st.
  for (int ph = 0; ph < stride_h; ++ph)
    for (int pw = 0; pw < stride_w; ++pw) {
      IgemmParams p;
      memset(&p, 0, sizeof(p));
      p.x = (const char*)dy; p.w = (const char*)w_crsk; p.y = (char*)dx; p.bias = nullptr;
      if (lazy_y != nullptr) { p.x2 = (const char*)lazy_y; p.xf = lazy_coef; p.xf_mode = 2; }   // dy = c1*g + c2*y + c3 on load
      p.addend = (const char*)addend;
      p.addend_sub = addend != nullptr ? addend_sub : 1; p.add_H = (H + 1) / 2; p.add_W = (W + 1) / 2;
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
      p.simple = ig_is_simple(p, dhdw, nt);
      if (bn != nullptr) {
        const int rbm = ig_epi_rows_bm(p.Co);
        const int rows = (p.M + rbm - 1) / rbm;
        if (bn_row + rows > bn->rows_cap) {
          cn_set_error("conv2d_dgrad_bnbwd: partial buffer of %d rows is too small", bn->rows_cap);
          return CN_EWORKSPACE;
        }
        p.bn_y = (const char*)bn->y; p.bn_mask = bn->mask; p.bn_coef = bn->coef; p.bn_partial = bn->partial;
        p.bn_relu = bn->relu; p.bn_row0 = bn_row;
        bn_row += rows;
      }
      rc = ig_dispatch(p, dtype, (hipStream_t)stream);
      if (rc) return rc;
    }
  return CN_OK;
}

extern "C" int cn_conv2d_dgrad(const void* dy, const void* w_crsk, void* dx, const void* addend, int N, int H,
                               int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                               int pad_w, int dtype, int out_f32, void* stream) {
  return ig_conv_dgrad(dy, w_crsk, dx, addend, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype,
                       out_f32, nullptr, stream);
}

// "Lazy dy" data gradient (round 3): the upstream gradient dy is the result of a training-mode BatchNorm backward,
//     dy[m][k] = c1[k]*g[m][k] + c2[k]*y[m][k] + c3[k]      (cn_bn_bwd_partials with dy = NULL leaves coef = [c1 | c2 | c3]),
// and is formed on the operand load from g (the masked gradient w.r.t. the BatchNorm output) and y (the BatchNorm
// input), with the operation order and rounding of the apply kernel: the same bits as cn_bn_bwd_partials(dy) followed
// by cn_conv2d_dgrad(dy), without writing / re-reading dy.  K <= 512 channels, no epilogue operands.
extern "C" int cn_conv2d_dgrad_lazy(const void* g, const void* bn_y, const float* coef, const void* w_crsk, void* dx,
                                    int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                                    int pad_w, int dtype, void* stream) {
  if (g == nullptr || bn_y == nullptr || coef == nullptr) { cn_set_error("conv2d_dgrad_lazy: needs g, y and the coefficients"); return CN_EINVAL; }
  if (K > IG_XF_MAX) { cn_set_error("conv2d_dgrad_lazy: %d gradient channels > %d", K, IG_XF_MAX); return CN_ESHAPE; }
  return ig_conv_dgrad(g, w_crsk, dx, nullptr, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype, 0, nullptr,
                       stream, 1, bn_y, coef);
}

// The same with a SUBSAMPLED addend: `addend` is [N][(H+1)/2][(W+1)/2][C], the values at the even (h, w) pixels of a
// tensor that is zero everywhere else - the input gradient of a stride-2 1x1 projection (models/resnet.py:176-181),
// which is what meets this gradient at the block input.  Saves writing and re-reading the three quarters of zeros.
extern "C" int cn_conv2d_dgrad_sa(const void* dy, const void* w_crsk, void* dx, const void* addend, int addend_sub, int N,
                                  int H, int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                                  int pad_w, int dtype, int out_f32, void* stream) {
  return ig_conv_dgrad(dy, w_crsk, dx, addend, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype,
                       out_f32, nullptr, stream, addend_sub);
}

// Partial rows cn_conv2d_dgrad_bnbwd writes: one per 128-pixel tile of every
// output-parity class.
extern "C" int cn_conv2d_dgrad_bnbwd_rows(int N, int H, int W, int C, int stride_h, int stride_w) {
  int rows = 0;
  const int rbm = ig_epi_rows_bm(C);
  for (int ph = 0; ph < stride_h; ++ph)
    for (int pw = 0; pw < stride_w; ++pw) {
      const long long hg = (H - ph + stride_h - 1) / stride_h, wg = (W - pw + stride_w - 1) / stride_w;
      if (hg <= 0 || wg <= 0) continue;
      rows += (int)(((long long)N * hg * wg + rbm - 1) / rbm);
    }
  return rows;
}

// Data gradient fused with the reduction half of the BatchNorm backward of the layer that produced
// this convolution's input: x = act(BN(bn_y) [+ residual]).  Stores g = dx * relu_mask instead of dx
// (mask from `bn_mask` bits, or recomputed from bn_y*scale+shift > 0 when bn_relu and no mask) and
// writes partial[row] = [sum g | sum g*(bn_y-mean)*invstd] per 128-pixel tile
// (cn_conv2d_dgrad_bnbwd_rows rows of 2*C floats) for cn_bn_bwd_partials.  bn_coef = the 4*C floats
// [mean | invstd | scale | shift] cn_bn_fwd_train wrote.
static int dgrad_bnbwd_impl(const void* dy, const void* w_crsk, void* g, const void* addend, int addend_sub, int N,
                            int H, int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                            int dtype, const void* bn_y, const unsigned char* bn_mask, const float* bn_coef,
                            int bn_relu, float* partial, int partial_rows, void* stream) {
  const int CH = cn_dtype_chunk(dtype);
  if (bn_y == nullptr || bn_coef == nullptr || partial == nullptr || C % CH != 0) {
    cn_set_error("conv2d_dgrad_bnbwd: needs bn_y, bn_coef, partial and C (%d) a multiple of %d", C, CH);
    return CN_EINVAL;
  }
  IgBnBwd bn;
  bn.y = bn_y; bn.mask = bn_mask; bn.coef = bn_coef; bn.partial = partial; bn.relu = bn_relu;
  bn.rows_cap = partial_rows;
  return ig_conv_dgrad(dy, w_crsk, g, addend, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype, 0, &bn,
                       stream, addend_sub);
}

extern "C" int cn_conv2d_dgrad_bnbwd(const void* dy, const void* w_crsk, void* g, const void* addend, int N, int H,
                                     int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                                     int pad_w, int dtype, const void* bn_y, const unsigned char* bn_mask,
                                     const float* bn_coef, int bn_relu, float* partial, int partial_rows,
                                     void* stream) {
  return dgrad_bnbwd_impl(dy, w_crsk, g, addend, 1, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype, bn_y,
                          bn_mask, bn_coef, bn_relu, partial, partial_rows, stream);
}

// ... with a subsampled addend (see cn_conv2d_dgrad_sa)
extern "C" int cn_conv2d_dgrad_bnbwd_sa(const void* dy, const void* w_crsk, void* g, const void* addend, int addend_sub,
                                        int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w,
                                        int pad_h, int pad_w, int dtype, const void* bn_y,
                                        const unsigned char* bn_mask, const float* bn_coef, int bn_relu,
                                        float* partial, int partial_rows, void* stream) {
  return dgrad_bnbwd_impl(dy, w_crsk, g, addend, addend_sub, N, H, W, C, K, R, S, stride_h, stride_w, pad_h, pad_w, dtype,
                          bn_y, bn_mask, bn_coef, bn_relu, partial, partial_rows, stream);
}
