// bn.hip -- BatchNorm2d (training + inference) with fused residual-add and ReLU, NHWC, gfx950.
//
// Replaces nn.BatchNorm2d + nn.ReLU(inplace) + the `out += residual` of the reference blocks
// (/root/reference models/resnet.py:98-118,141-165,128-134,162-163) and their autograd backward.
// Semantics follow torch.nn.BatchNorm2d defaults as the reference instantiates them
// (eps=1e-5, momentum=0.1, affine, track_running_stats): normalise with the *biased* batch
// variance, update running_var with the *unbiased* one, num_batches_tracked += 1.
//
// All kernels are HBM-bound streaming kernels: channels are the fastest (NHWC) dimension, every
// lane owns one 16-byte channel chunk and walks over pixels, so loads are fully coalesced and the
// per-channel reductions need no cross-lane traffic until one small LDS step per workgroup.
// Cross-workgroup reduction is two-stage (partials + finalize) => deterministic, no atomics.
//
//   forward (train):  bn_stats -> bn_finalize -> bn_apply        (reads y twice, writes z once)
//   backward:         bn_bwd_reduce -> bn_bwd_finalize -> bn_bwd_apply
#include "cn_common.h"
#include "cn_api_internal.h"

static inline int bn_next_pow2_log2(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

struct BnMap {
  int tpr_log2;  // threads per row (power of two, <= 256)
  int gy;        // column groups
  int rpp;       // rows per pass of one workgroup
};
static BnMap bn_map(int cpr) {
  BnMap m;
  int l = bn_next_pow2_log2(cpr);
  if (l > 8) l = 8;
  m.tpr_log2 = l;
  m.gy = (cpr + (1 << l) - 1) >> l;
  m.rpp = 256 >> l;
  return m;
}
// 1 = the apply passes use non-temporal accesses (default; tensors below "bn_nt_min_mb" MB and
// "bn_nt" = 0 use the cached policy: whole-step A/B knobs)
static int bn_nt_flag(long long M, int C, int dtype) {
  if (cn_get_option("bn_nt", 1) == 0) return 0;
  const long long bytes = M * C * cn_dtype_bytes(dtype);
  return bytes < (long long)cn_get_option("bn_nt_min_mb", 0) * (1ll << 20) ? 0 : 1;
}
// Cache policy of the streaming passes is a COMPILE-TIME parameter of the kernels (NT): a run-time select
// between a plain and a non-temporal access of the same address is folded by LLVM into one plain access
// (round 1 shipped exactly that; tools/check_nt.sh now greps the code object for the `nt` accesses).
template <bool NT, int SITE>
__device__ __forceinline__ u32x4 bn_ld(const void* p) {
  if constexpr (NT) return cn_ld16_stream<SITE>(p);
  else return cn_ld16(p);
}
template <bool NT, int SITE>
__device__ __forceinline__ void bn_st(void* p, const u32x4& v) {
  if constexpr (NT) cn_st16_stream<SITE>(p, v);
  else cn_st16(p, v);
}
static int bn_row_blocks(long long M, const BnMap& m, int target_blocks) {
  long long passes = (M + m.rpp - 1) / m.rpp;
  long long nb = target_blocks / m.gy;
  if (nb < 1) nb = 1;
  long long cap = (passes + 3) / 4;  // at least ~4 passes per workgroup
  if (cap < 1) cap = 1;
  if (nb > cap) nb = cap;
  return (int)nb;
}

// Column sums of a 256-thread workgroup laid out as row slices of `tpr` chunk columns (thread tid owns column
// tid & (tpr-1) of row slice tid >> tpr_log2): the row slices that live in one wave are folded with wavefront
// shuffles (an xor butterfly over the lane bits above log2(tpr)), LDS then carries one value set per wave instead of
// one per thread, and the owner (row slice 0) adds the <= 4 of them in a fixed order.  Deterministic.
template <int N>
__device__ __forceinline__ void bn_block_colsum(float (&a)[N], float (&b)[N], float* red /*[256][2N]*/, int tpr_log2,
                                                int tid) {
  const int tpr = 1 << tpr_log2;
  for (int m = 32; m >= tpr; m >>= 1) {   // wave-uniform trip count (0 when a row slice fills a wave or more)
#pragma unroll
    for (int e = 0; e < N; ++e) { a[e] += cn_shfl_xor(a[e], m); b[e] += cn_shfl_xor(b[e], m); }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) { red[tid * 2 * N + e] = a[e]; red[tid * 2 * N + N + e] = b[e]; }
  __syncthreads();
  if ((tid >> tpr_log2) == 0) {
    const int stride = tpr > 64 ? tpr : 64;   // one representative per wave (or per row slice when slices span waves)
#pragma unroll
    for (int e = 0; e < N; ++e) { a[e] = 0.f; b[e] = 0.f; }
    for (int t = tid; t < 256; t += stride) {
      const float* o = red + t * 2 * N;
#pragma unroll
      for (int e = 0; e < N; ++e) { a[e] += o[e]; b[e] += o[N + e]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const char* y, float* partial, int M, int C,
                                                      int tpr_log2, const float* pivot) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  __shared__ float red[256 * 2 * CH];
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2, rpp = 256 >> tpr_log2;
  const int cpr = C / CH;
  const int tcol = tid & (tpr - 1), rsub = tid >> tpr_log2;
  const int col = blockIdx.y * tpr + tcol;
  float s[CH], q[CH], pv[CH];   // centred sums: sum (y - pivot), sum (y - pivot)^2 (pivot = 0 without one)
#pragma unroll
  for (int e = 0; e < CH; ++e) { s[e] = 0.f; q[e] = 0.f; pv[e] = 0.f; }
  if (col < cpr) {
    if (pivot != nullptr) {
#pragma unroll
      for (int e = 0; e < CH; ++e) pv[e] = cn_pivot(pivot[col * CH + e]);
    }
    const int step = gridDim.x * rpp;
    const size_t cb = (size_t)col * CH * EB, rb = (size_t)C * EB;
    int row = blockIdx.x * rpp + rsub;
    for (; row + 3 * step < M; row += 4 * step) {   // 4 independent 16-byte loads in flight per lane
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = cn_ld16(y + (size_t)(row + u * step) * rb + cb);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[CH];
        Chunk<T>::unpack(v[u], f);
#pragma unroll
        for (int e = 0; e < CH; ++e) { const float d = f[e] - pv[e]; s[e] += d; q[e] = fmaf(d, d, q[e]); }
      }
    }
    for (; row < M; row += step) {
      float f[CH];
      Chunk<T>::unpack(cn_ld16(y + (size_t)row * rb + cb), f);
#pragma unroll
      for (int e = 0; e < CH; ++e) { const float d = f[e] - pv[e]; s[e] += d; q[e] = fmaf(d, d, q[e]); }
    }
  }
  bn_block_colsum<CH>(s, q, red, tpr_log2, tid);
  if (rsub == 0 && col < cpr) {
    float* dst = partial + (size_t)blockIdx.x * 2 * C + col * CH;
#pragma unroll
    for (int e = 0; e < CH; ++e) { dst[e] = s[e]; dst[C + e] = q[e]; }
  }
}

// Fixed-order, latency-tolerant reduction of the per-workgroup partials: BN_FC channels per workgroup,
// BN_FP threads per channel each summing every BN_FP-th partial row with 16 independent loads in
// flight, then a fixed-order LDS combine.  (A single thread walking all partials is a chain of
// dependent L2 round trips: measured 0.5 ms per launch at 2048 partials; with 8 threads per channel
// 512 rows still took 4 dependent rounds, with 32 it is one or two.)
#define BN_FC 8     /* channels per workgroup */
#define BN_FP 32    /* row slices (threads per channel) */
__device__ __forceinline__ void bn_sum_partials(const float* partial, int nrb, int C, int c, int part,
                                                double* red /*[2][BN_FP][BN_FC]*/, double& s_out, double& q_out) {
  double s = 0.0, q = 0.0;
  if (c < C) {
    int r = part;
    for (; r + 15 * BN_FP < nrb; r += 16 * BN_FP) {   // 32 independent loads in flight: 512 rows = one round trip
      float a[16], b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a[u] = partial[(size_t)(r + BN_FP * u) * 2 * C + c];
        b[u] = partial[(size_t)(r + BN_FP * u) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        s += ((double)a[u] + (double)a[u + 1]) + ((double)a[u + 2] + (double)a[u + 3]);
        q += ((double)b[u] + (double)b[u + 1]) + ((double)b[u + 2] + (double)b[u + 3]);
      }
    }
    for (; r + 7 * BN_FP < nrb; r += 8 * BN_FP) {   // 16 independent loads in flight per thread
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = partial[(size_t)(r + BN_FP * u) * 2 * C + c];
        b[u] = partial[(size_t)(r + BN_FP * u) * 2 * C + C + c];
      }
      s += (((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3])) +
           (((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]));
      q += (((double)b[0] + (double)b[1]) + ((double)b[2] + (double)b[3])) +
           (((double)b[4] + (double)b[5]) + ((double)b[6] + (double)b[7]));
    }
    if (r < nrb) {   // < 8 rows left: requested together (one by one they are a chain of L2 round trips, ~1.4 us
      float a[7], b[7];   // each: 392 rows cost 5.4 us, 483 rows 9.6 us), added in row order; clamped index, not a
#pragma unroll            // branch around each load
      for (int u = 0; u < 7; ++u) {
        const int ru = r + BN_FP * u;
        const int rc = ru < nrb ? ru : nrb - 1;
        const float av = partial[(size_t)rc * 2 * C + c], bv = partial[(size_t)rc * 2 * C + C + c];
        a[u] = ru < nrb ? av : 0.f;
        b[u] = ru < nrb ? bv : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) { s += (double)a[u]; q += (double)b[u]; }
    }
  }
  const int lc = threadIdx.x % BN_FC;
  red[part * BN_FC + lc] = s;
  red[BN_FP * BN_FC + part * BN_FC + lc] = q;
  __syncthreads();
  s = 0.0;
  q = 0.0;
#pragma unroll
  for (int k = 0; k < BN_FP; ++k) { s += red[k * BN_FC + lc]; q += red[BN_FP * BN_FC + k * BN_FC + lc]; }
  s_out = s;
  q_out = q;
}

// out[r2][col] = sum of the G consecutive partial rows r2*G .. r2*G+G-1 (fixed order).  Brings the
// per-pixel-tile partials a convolution epilogue emitted (thousands of rows for the 56x56 layers) down
// to the few hundred rows bn_finalize_kernel walks.
__global__ __launch_bounds__(256) void bn_partials_compress_kernel(const float* in, float* out, int nrb, int G,
                                                                  int W) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= W) return;
  int r = blockIdx.y * G;
  const int re = r + G < nrb ? r + G : nrb;
  float acc = 0.f;
  for (; r + 7 < re; r += 8) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = in[(size_t)(r + u) * W + col];
    acc += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  if (r < re) {   // < 8 rows left: one batch of loads, added in row order
    float a[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int rc = r + u < re ? r + u : re - 1;
      const float v = in[(size_t)rc * W + col];
      a[u] = r + u < re ? v : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) acc += a[u];
  }
  out[(size_t)blockIdx.y * W + col] = acc;
}

// BN_FC channels per workgroup: statistics, running-stat update and the fused scale/shift the apply
// kernel consumes.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* partial, int nrb, int M, int C,
                                                         const float* gamma, const float* beta,
                                                         float* running_mean, float* running_var,
                                                         long long* num_batches_tracked, float momentum,
                                                         float eps, float* save_mean, float* save_invstd,
                                                         float* scale, float* shift, int centered) {
  __shared__ double red[512];
  const int c = blockIdx.x * BN_FC + (threadIdx.x % BN_FC);
  const int part = threadIdx.x / BN_FC;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
  // per-channel parameters are requested before the partial reduction (one memory round trip less on a
  // kernel that is nothing but latency)
  const bool owner = c < C && part == 0;
  const float g_pre = (owner && gamma != nullptr) ? gamma[c] : 1.f;
  const float b_pre = (owner && beta != nullptr) ? beta[c] : 0.f;
  const float rm_pre = (owner && running_mean != nullptr) ? running_mean[c] : 0.f;
  const float rv_pre = (owner && running_mean != nullptr) ? running_var[c] : 0.f;
  double s, q;
  bn_sum_partials(partial, nrb, C, c, part, red, s, q);
  if (c >= C || part != 0) return;
  // centred partials (sums of y - running_mean as it was before this update): the variance no longer comes out of
  // the difference of two numbers of size mean^2
  const double dmean = s / (double)M;
  const double mean = centered ? (double)cn_pivot(rm_pre) + dmean : dmean;
  double var = q / (double)M - dmean * dmean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (running_mean != nullptr) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)rm_pre + (double)momentum * mean);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)rv_pre + (double)momentum * unbiased);
  }
  const float sc = g_pre * invstd;
  scale[c] = sc;
  shift[c] = b_pre - (float)mean * sc;
}

// Inference coefficients from the running statistics.
__global__ __launch_bounds__(256) void bn_infer_coeffs_kernel(int C, const float* gamma, const float* beta,
                                                             const float* running_mean,
                                                             const float* running_var, float eps,
                                                             float* scale, float* shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(running_var[c] + eps);
  const float g = gamma != nullptr ? gamma[c] : 1.f;
  const float b = beta != nullptr ? beta[c] : 0.f;
  scale[c] = g * invstd;
  shift[c] = b - running_mean[c] * g * invstd;
}

// z = act(y*scale[c] + shift[c] (+ residual)).  When `mask` is given (ReLU after a residual add) one
// byte per 16-byte chunk records which outputs were positive, so backward reads M*C/CH bytes instead
// of re-reading z (the mask cannot be recomputed from y alone once a residual was added).
// DUAL (the junction behind a projection shortcut): `res` is the shortcut BatchNorm's INPUT and that BatchNorm's apply,
// r = round_T(res*rscale[c] + rshift[c]) - exactly what its own apply pass would have stored -, happens here, so the
// normalised shortcut tensor is never written or re-read (one write + one read of the junction-sized tensor less).
template <typename T, bool NT, bool DUAL = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const char* y, const char* res, char* z,
                                                      unsigned char* mask, const float* scale,
                                                      const float* shift, int M, int C, int relu,
                                                      int tpr_log2, int rev, const float* rscale,
                                                      const float* rshift) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2, rpp = 256 >> tpr_log2;
  const int cpr = C / CH;
  const int col = blockIdx.y * tpr + (tid & (tpr - 1)), rsub = tid >> tpr_log2;
  if (col >= cpr) return;
  float sc[CH], sh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { sc[e] = scale[col * CH + e]; sh[e] = shift[col * CH + e]; }
  float rs[DUAL ? CH : 1], rb[DUAL ? CH : 1];
  if constexpr (DUAL) {
#pragma unroll
    for (int e = 0; e < CH; ++e) { rs[e] = rscale[col * CH + e]; rb[e] = rshift[col * CH + e]; }
  }
  const int step = gridDim.x * rpp;
#pragma unroll 4
  for (int it = blockIdx.x * rpp + rsub; it < M; it += step) {
    const int row = (rev & 1) ? M - 1 - it : it;   // bit 0: sweep back to front
    const size_t off = ((size_t)row * C + (size_t)col * CH) * EB;
    float f[CH];
    Chunk<T>::unpack(bn_ld<NT, 1>(y + off), f);
#pragma unroll
    for (int e = 0; e < CH; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
    if (res != nullptr) {
      float r[CH];
      Chunk<T>::unpack(bn_ld<NT, 2>(res + off), r);
      if constexpr (DUAL) {   // the shortcut BatchNorm's apply (no activation, no residual), rounded to T as stored
#pragma unroll
        for (int e = 0; e < CH; ++e) r[e] = fmaf(r[e], rs[e], rb[e]);
        Chunk<T>::unpack(Chunk<T>::pack(r), r);
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) f[e] += r[e];
    }
    if (relu) {
      unsigned int bits = 0;
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        bits |= (f[e] > 0.f ? 1u : 0u) << e;
        f[e] = f[e] > 0.f ? f[e] : 0.f;
      }
      if (mask != nullptr) mask[(size_t)row * cpr + col] = (unsigned char)bits;
    }
    cn_st16(z + off, Chunk<T>::pack(f));   // plain: the next conv reads z straight away (NT store measured -1.7 %)
  }
}

// Per-channel sum(g) and sum(g * xhat), g = dz * relu_mask.
//   mask source: the byte mask written by bn_apply (needed when a residual was added) or, when
//   mask == nullptr and relu != 0, recomputed from y*scale+shift > 0.
template <typename T, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const char* dz, const char* y, const unsigned char* zmask,
                                                           const float* mean, const float* invstd,
                                                           const float* scale, const float* shift,
                                                           float* partial, int M, int C, int relu,
                                                           int tpr_log2, int rev) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  __shared__ float red[256 * 2 * CH];
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2, rpp = 256 >> tpr_log2;
  const int cpr = C / CH;
  const int tcol = tid & (tpr - 1), rsub = tid >> tpr_log2;
  const int col = blockIdx.y * tpr + tcol;
  float s1[CH], s2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (col < cpr) {
    float mu[CH], is[CH], sc[CH], sh[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      mu[e] = mean[col * CH + e];
      is[e] = invstd[col * CH + e];
      sc[e] = scale[col * CH + e];
      sh[e] = shift[col * CH + e];
    }
    const int step = gridDim.x * rpp;
    const size_t cb = (size_t)col * CH * EB, rb = (size_t)C * EB;
    auto accum = [&](const u32x4& gz, const u32x4& vy, unsigned int bits) {
      float g[CH], v[CH];
      Chunk<T>::unpack(gz, g);
      Chunk<T>::unpack(vy, v);
      if (relu) {
        if (zmask != nullptr) {
#pragma unroll
          for (int e = 0; e < CH; ++e) g[e] = ((bits >> e) & 1u) ? g[e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < CH; ++e) g[e] = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        s1[e] += g[e];
        s2[e] = fmaf(g[e], (v[e] - mu[e]) * is[e], s2[e]);
      }
    };
    int it = blockIdx.x * rpp + rsub;
    for (; it + 3 * step < M; it += 4 * step) {   // 8-12 independent loads in flight per lane
      u32x4 gz[4], vy[4];
      unsigned int bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (rev & 1) ? M - 1 - (it + u * step) : it + u * step;
        gz[u] = bn_ld<NT, 5>(dz + (size_t)row * rb + cb);
        vy[u] = bn_ld<NT, 6>(y + (size_t)row * rb + cb);
        if (relu && zmask != nullptr) bits[u] = zmask[(size_t)row * cpr + col];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accum(gz[u], vy[u], bits[u]);
    }
    for (; it < M; it += step) {
      const int row = (rev & 1) ? M - 1 - it : it;
      unsigned int bits = 0u;
      if (relu && zmask != nullptr) bits = zmask[(size_t)row * cpr + col];
      accum(bn_ld<NT, 5>(dz + (size_t)row * rb + cb), bn_ld<NT, 6>(y + (size_t)row * rb + cb), bits);
    }
  }
  bn_block_colsum<CH>(s1, s2, red, tpr_log2, tid);
  if (rsub == 0 && col < cpr) {
    float* dst = partial + (size_t)blockIdx.x * 2 * C + col * CH;
#pragma unroll
    for (int e = 0; e < CH; ++e) { dst[e] = s1[e]; dst[C + e] = s2[e]; }
  }
}

// dgamma/dbeta (optionally accumulated) and the three per-channel coefficients of
//   dy = c1*g + c2*y + c3     ( = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) )
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partial, int nrb, int M, int C,
                                                             const float* gamma, const float* mean,
                                                             const float* invstd, float* dgamma,
                                                             float* dbeta, float beta_acc, float gscale,
                                                             float* coef) {
  __shared__ double red[512];
  const int c = blockIdx.x * BN_FC + (threadIdx.x % BN_FC);
  const int part = threadIdx.x / BN_FC;
  const bool owner = c < C && part == 0;   // operands requested before the partial reduction (latency)
  const float g = (owner && gamma != nullptr) ? gamma[c] : 1.f;
  const float is_pre = owner ? invstd[c] : 0.f;
  const float mu_pre = owner ? mean[c] : 0.f;
  const float dg_pre = (owner && dgamma != nullptr && beta_acc != 0.f) ? dgamma[c] : 0.f;
  const float db_pre = (owner && dbeta != nullptr && beta_acc != 0.f) ? dbeta[c] : 0.f;
  double s1, s2;
  bn_sum_partials(partial, nrb, C, c, part, red, s1, s2);
  if (c >= C || part != 0) return;
  if (dgamma != nullptr) dgamma[c] = (beta_acc != 0.f ? beta_acc * dg_pre : 0.f) + (float)s2 * gscale;
  if (dbeta != nullptr) dbeta[c] = (beta_acc != 0.f ? beta_acc * db_pre : 0.f) + (float)s1 * gscale;
  const double k = (double)g * (double)is_pre;
  const double a2 = k * (double)is_pre * s2 / (double)M;
  coef[c] = (float)k;
  coef[C + c] = (float)(-a2);
  coef[2 * C + c] = (float)(a2 * (double)mu_pre - k * s1 / (double)M);
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const char* dz, const char* y, const unsigned char* zmask,
                                                          const float* scale, const float* shift,
                                                          const float* coef, char* dy, char* dres, int M,
                                                          int C, int relu, int tpr_log2, int rev) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2, rpp = 256 >> tpr_log2;
  const int cpr = C / CH;
  const int col = blockIdx.y * tpr + (tid & (tpr - 1)), rsub = tid >> tpr_log2;
  if (col >= cpr) return;
  float c1[CH], c2[CH], c3[CH], sc[CH], sh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    c1[e] = coef[col * CH + e];
    c2[e] = coef[C + col * CH + e];
    c3[e] = coef[2 * C + col * CH + e];
    sc[e] = scale[col * CH + e];
    sh[e] = shift[col * CH + e];
  }
  const int step = gridDim.x * rpp;
#pragma unroll 2
  for (int it = blockIdx.x * rpp + rsub; it < M; it += step) {
    const int row = (rev & 1) ? M - 1 - it : it;
    const size_t off = ((size_t)row * C + (size_t)col * CH) * EB;
    float g[CH], v[CH];
    Chunk<T>::unpack(bn_ld<NT, 3>(dz + off), g);
    Chunk<T>::unpack(bn_ld<NT, 4>(y + off), v);
    if (relu) {
      if (zmask != nullptr) {
        const unsigned int bits = zmask[(size_t)row * cpr + col];
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = ((bits >> e) & 1u) ? g[e] : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
      }
    }
    if (dres != nullptr) bn_st<NT, 3>(dres + off, Chunk<T>::pack(g));
    float o[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) o[e] = fmaf(c1[e], g[e], fmaf(c2[e], v[e], c3[e]));
    bn_st<NT, 3>(dy + off, Chunk<T>::pack(o));
  }
}

// ------------------------------------------------------------------------------------------------
// Stem variant: the upstream gradient is the gradient of the max-pooled map (models/resnet.py:228-230:
// bn1 -> relu -> maxpool).  dz[n,h,w,c] is gathered on the fly from the pooled gradient and the winning
// taps the fused forward stored (cn_maxpool_fwd_bnrelu), so the dense 112x112 dz is never written or
// re-read: both passes read y plus (L2-resident) pooled data.
struct BnPoolGeom {
  const char* dpool;            // [N,P,Q,C] gradient of the pooled map
  const unsigned char* idx;     // [N,P,Q,C] winning tap (kh*k + kw) per pooled element
  int H, W, P, Q, k, st, pad;
  FastDiv div_hw, div_w, div_st;
};

// At most 2 x 2 windows cover an input pixel when k <= 2*stride (checked by the host): fixed-trip loops
// with predicated loads, so the 4 (gradient, tap) pairs of a pixel are requested back to back.
template <typename T>
__device__ __forceinline__ void bn_pool_gather(const BnPoolGeom& g, int row, int col, int C, float* out) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int n = (int)cn_fastdiv((unsigned)row, g.div_hw);
  const int rem = row - n * g.H * g.W;
  const int h = (int)cn_fastdiv((unsigned)rem, g.div_w);
  const int w = rem - h * g.W;
  // window ranges with the host-precomputed fast division by the stride (runtime `/` costs ~30 VALU each)
  int p_lo = h + g.pad - g.k + 1;
  p_lo = p_lo > 0 ? (int)cn_fastdiv((unsigned)(p_lo + g.st - 1), g.div_st) : 0;
  int p_hi = (int)cn_fastdiv((unsigned)(h + g.pad), g.div_st);
  if (p_hi > g.P - 1) p_hi = g.P - 1;
  int q_lo = w + g.pad - g.k + 1;
  q_lo = q_lo > 0 ? (int)cn_fastdiv((unsigned)(q_lo + g.st - 1), g.div_st) : 0;
  int q_hi = (int)cn_fastdiv((unsigned)(w + g.pad), g.div_st);
  if (q_hi > g.Q - 1) q_hi = g.Q - 1;
  u32x4 gv[4];
  unsigned long long pk[4];
  int tap[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pp = p_hi - (i >> 1), q = q_hi - (i & 1);
    const bool ok = pp >= p_lo && q >= q_lo;
    const size_t o = ok ? ((size_t)(n * g.P + pp) * g.Q + q) * C + (size_t)col * CH : 0;
    gv[i] = cn_ld16(g.dpool + o * EB);
    if (CH == 8) pk[i] = *(const unsigned long long*)(g.idx + o);
    else pk[i] = *(const unsigned int*)(g.idx + o);
    tap[i] = ok ? (h - (pp * g.st - g.pad)) * g.k + (w - (q * g.st - g.pad)) : -1;   // -1 never matches a stored tap
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) out[e] = 0.f;
#pragma unroll
  for (int i = 3; i >= 0; --i) {   // ascending (p, q): the summation order of the unfused max-pool backward
    float v[CH];
    Chunk<T>::unpack(gv[i], v);
#pragma unroll
    for (int e = 0; e < CH; ++e)
      if ((int)((pk[i] >> (8 * e)) & 0xffull) == tap[i]) out[e] += v[e];
  }
  Chunk<T>::unpack(Chunk<T>::pack(out), out);   // what the unfused chain's max-pool backward would have stored
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(BnPoolGeom geo, const char* y, const float* mean,
                                                                const float* invstd, const float* scale,
                                                                const float* shift, float* partial, int M, int C,
                                                                int tpr_log2) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  __shared__ float red[256 * 2 * CH];
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2, rpp = 256 >> tpr_log2;
  const int cpr = C / CH;
  const int tcol = tid & (tpr - 1), rsub = tid >> tpr_log2;
  const int col = blockIdx.y * tpr + tcol;
  float s1[CH], s2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (col < cpr) {
    float mu[CH], is[CH], sc[CH], sh[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      mu[e] = mean[col * CH + e];
      is[e] = invstd[col * CH + e];
      sc[e] = scale[col * CH + e];
      sh[e] = shift[col * CH + e];
    }
    const int step = gridDim.x * rpp;
    auto accum = [&](const float* g, const float* v) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const float gm = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
        s1[e] += gm;
        s2[e] = fmaf(gm, (v[e] - mu[e]) * is[e], s2[e]);
      }
    };
    int row = blockIdx.x * rpp + rsub;
    for (; row + step < M; row += 2 * step) {   // two pixels (their y chunk + 4 gathered pairs each) in flight
      float g0[CH], g1[CH], v0[CH], v1[CH];
      const u32x4 y0 = cn_ld16(y + ((size_t)row * C + (size_t)col * CH) * EB);
      const u32x4 y1 = cn_ld16(y + ((size_t)(row + step) * C + (size_t)col * CH) * EB);
      bn_pool_gather<T>(geo, row, col, C, g0);
      bn_pool_gather<T>(geo, row + step, col, C, g1);
      Chunk<T>::unpack(y0, v0);
      Chunk<T>::unpack(y1, v1);
      accum(g0, v0);
      accum(g1, v1);
    }
    for (; row < M; row += step) {
      float g[CH], v[CH];
      Chunk<T>::unpack(cn_ld16(y + ((size_t)row * C + (size_t)col * CH) * EB), v);
      bn_pool_gather<T>(geo, row, col, C, g);
      accum(g, v);
    }
  }
  bn_block_colsum<CH>(s1, s2, red, tpr_log2, tid);
  if (rsub == 0 && col < cpr) {
    float* dst = partial + (size_t)blockIdx.x * 2 * C + col * CH;
#pragma unroll
    for (int e = 0; e < CH; ++e) { dst[e] = s1[e]; dst[C + e] = s2[e]; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(BnPoolGeom geo, const char* y, const float* scale,
                                                               const float* shift, const float* coef, char* dy,
                                                               int M, int C, int tpr_log2) {
  constexpr int CH = ElemTraits<T>::kChunk;
  constexpr int EB = ElemTraits<T>::kBytes;
  const int tid = threadIdx.x;
  const int tpr = 1 << tpr_log2, rpp = 256 >> tpr_log2;
  const int cpr = C / CH;
  const int col = blockIdx.y * tpr + (tid & (tpr - 1)), rsub = tid >> tpr_log2;
  if (col >= cpr) return;
  float c1[CH], c2[CH], c3[CH], sc[CH], sh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    c1[e] = coef[col * CH + e];
    c2[e] = coef[C + col * CH + e];
    c3[e] = coef[2 * C + col * CH + e];
    sc[e] = scale[col * CH + e];
    sh[e] = shift[col * CH + e];
  }
  const int step = gridDim.x * rpp;
  auto emit = [&](size_t off, const float* g, const float* v) {
    float o[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      const float gm = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
      o[e] = fmaf(c1[e], gm, fmaf(c2[e], v[e], c3[e]));
    }
    cn_st16_stream(dy + off, Chunk<T>::pack(o));
  };
  int row = blockIdx.x * rpp + rsub;
  for (; row + step < M; row += 2 * step) {
    const size_t off0 = ((size_t)row * C + (size_t)col * CH) * EB;
    const size_t off1 = ((size_t)(row + step) * C + (size_t)col * CH) * EB;
    float g0[CH], g1[CH], v0[CH], v1[CH];
    const u32x4 y0 = cn_ld16(y + off0);
    const u32x4 y1 = cn_ld16(y + off1);
    bn_pool_gather<T>(geo, row, col, C, g0);
    bn_pool_gather<T>(geo, row + step, col, C, g1);
    Chunk<T>::unpack(y0, v0);
    Chunk<T>::unpack(y1, v1);
    emit(off0, g0, v0);
    emit(off1, g1, v1);
  }
  for (; row < M; row += step) {
    const size_t off = ((size_t)row * C + (size_t)col * CH) * EB;
    float g[CH], v[CH];
    Chunk<T>::unpack(cn_ld16(y + off), v);
    bn_pool_gather<T>(geo, row, col, C, g);
    emit(off, g, v);
  }
}

// Inference-mode backward is not part of the reference hot path (validate() runs under no_grad).

// ------------------------------------------------------------------------------------------------
// (dtype, cache policy) -> kernel instantiation
#define BN_DISPATCH(kern, dtype, nt, grid, stream, ...)                                            \
  do {                                                                                             \
    if ((dtype) == CN_BF16) {                                                                      \
      if (nt) CN_LAUNCH((kern<bf16_t, true>), grid, dim3(256), stream, __VA_ARGS__);                \
      else CN_LAUNCH((kern<bf16_t, false>), grid, dim3(256), stream, __VA_ARGS__);                  \
    } else if ((dtype) == CN_F16) {                                                                \
      if (nt) CN_LAUNCH((kern<f16_t, true>), grid, dim3(256), stream, __VA_ARGS__);                 \
      else CN_LAUNCH((kern<f16_t, false>), grid, dim3(256), stream, __VA_ARGS__);                   \
    } else {                                                                                       \
      if (nt) CN_LAUNCH((kern<float, true>), grid, dim3(256), stream, __VA_ARGS__);                 \
      else CN_LAUNCH((kern<float, false>), grid, dim3(256), stream, __VA_ARGS__);                   \
    }                                                                                              \
  } while (0)

// (the reduction passes: cached loads only - non-temporal loads there measured slower, profiles/README.md)
#define BN_DISPATCH_PLAIN(kern, dtype, grid, stream, ...)                                          \
  do {                                                                                             \
    if ((dtype) == CN_BF16) CN_LAUNCH((kern<bf16_t, false>), grid, dim3(256), stream, __VA_ARGS__); \
    else if ((dtype) == CN_F16) CN_LAUNCH((kern<f16_t, false>), grid, dim3(256), stream, __VA_ARGS__); \
    else CN_LAUNCH((kern<float, false>), grid, dim3(256), stream, __VA_ARGS__);                     \
  } while (0)

#define BN_TARGET_BLOCKS 512   /* partial rows per channel the finalize kernels take without a compression launch */
/* Row blocks of the standalone reduction passes (bn_stats / bn_bwd_reduce; knob "bn_reduce_blocks").  256: the kernels keep
 * 8-12 sixteen-byte loads in flight per lane, so 256 workgroups already saturate HBM, and fewer, longer workgroups leave the
 * chain's other kernels and the side stream more of the chip: 14.93k img/s vs 14.76k at 512, 14.59k at 128, 13.76k at 64
 * (round 4, profiles/README.md). */
#define BN_REDUCE_BLOCKS 256
#define BN_APPLY_BLOCKS 2048   /* pure streaming kernels */
/* Sweep direction of the streaming kernels, bit 0: forward apply, bit 1: backward reduce, bit 2: backward
 * apply.  A kernel that sweeps in the opposite direction to the one that last touched its input finds
 * the freshest part of that tensor still in the 256 MB Infinity Cache (tuning knob "bn_reverse"). */
#define BN_REVERSE_DEFAULT 0

extern "C" size_t cn_bn_workspace(int M, int C, int dtype) {
  const int CH = cn_dtype_chunk(dtype);
  if (C % CH != 0 || M <= 0) return 0;
  BnMap m = bn_map(C / CH);
  int nrb = bn_row_blocks(M, m, 2048);   // upper bound over the tunable reduce-grid sizes
  // ... and room for the row-compressed form of any number of caller-supplied partial rows
  // (cn_bn_fwd_train_partials / cn_bn_bwd_partials compress > BN_TARGET_BLOCKS rows into <= BN_TARGET_BLOCKS)
  if (nrb < BN_TARGET_BLOCKS) nrb = BN_TARGET_BLOCKS;
  return (size_t)nrb * 2 * C * sizeof(float);
}

static int bn_check(const char* who, int M, int C, int dtype) {
  if (!cn_dtype_ok(dtype)) { cn_set_error("%s: bad dtype %d", who, dtype); return CN_EINVAL; }
  const int CH = cn_dtype_chunk(dtype);
  if (M <= 0 || C <= 0 || C % CH != 0) {
    cn_set_error("%s: need M>0 and C (%d) a multiple of %d", who, C, CH);
    return CN_ESHAPE;
  }
  return CN_OK;
}

// finalize + apply shared by both training-forward entry points
static int bn_fwd_tail(const float* partial, int nrb, const void* y, const void* residual, void* z,
                       unsigned char* relu_mask, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, long long* num_batches_tracked, float momentum, float eps,
                       float* stats_out, int M, int C, int relu, int dtype, const BnMap& m, hipStream_t stream,
                       int centered) {
  CN_LAUNCH(bn_finalize_kernel, dim3((unsigned)((C + BN_FC - 1) / BN_FC)), dim3(256), stream, partial, nrb,
            M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, stats_out,
            stats_out + C, stats_out + 2 * C, stats_out + 3 * C, centered);
  if (z == nullptr) return cn_check_launch("bn_fwd_train");   // statistics only (the consumer applies them itself)
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  const int rev = (cn_get_option("bn_reverse", BN_REVERSE_DEFAULT) & 1);
  BN_DISPATCH(bn_apply_kernel, dtype, bn_nt_flag(M, C, dtype), agrid, stream, (const char*)y, (const char*)residual, (char*)z, relu_mask, (const float*)(stats_out + 2 * C), (const float*)(stats_out + 3 * C), M, C, relu, m.tpr_log2, rev, (const float*)nullptr, (const float*)nullptr);
  return cn_check_launch("bn_fwd_train");
}

// Training forward.  stats_out = [save_mean | save_invstd | scale | shift] (4*C floats).
extern "C" int cn_bn_fwd_train(const void* y, const void* residual, void* z, unsigned char* relu_mask,
                               const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               long long* num_batches_tracked, float momentum, float eps, float* stats_out,
                               int M, int C, int relu, int dtype, void* workspace, size_t ws_bytes,
                               void* stream_) {
  int rc = bn_check("bn_fwd_train", M, C, dtype);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  int nrb = bn_row_blocks(M, m, cn_get_option("bn_reduce_blocks", BN_REDUCE_BLOCKS));
  if (workspace == nullptr || ws_bytes < (size_t)nrb * 2 * C * sizeof(float)) {
    cn_set_error("bn_fwd_train: workspace too small");
    return CN_EWORKSPACE;
  }
  float* partial = (float*)workspace;
  dim3 grid((unsigned)nrb, (unsigned)m.gy);
  // statistics centred on the running mean whenever there is one (see bn_finalize_kernel)
  const float* pivot = running_mean;
  CN_DISPATCH_T(dtype, CN_LAUNCH(bn_stats_kernel<TT>, grid, dim3(256), stream, (const char*)y, partial, M, C, m.tpr_log2, pivot));
  return bn_fwd_tail(partial, nrb, y, residual, z, relu_mask, gamma, beta, running_mean, running_var,
                     num_batches_tracked, momentum, eps, stats_out, M, C, relu, dtype, m, stream, pivot != nullptr);
}

// Training forward from statistics partials a producer already reduced (cn_conv2d_fwd_bnstats):
// partial = [nrb][2*C] floats (sum | sum of squares per row).  Skips the statistics read of y.
static int bn_fwd_train_partials_impl(const void* y, const void* residual, void* z, unsigned char* relu_mask,
                                        const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, long long* num_batches_tracked, float momentum,
                                        float eps, float* stats_out, int M, int C, int relu, int dtype,
                                        const float* partial, int nrb, void* workspace, size_t ws_bytes,
                                        void* stream_, int centered) {
  int rc = bn_check("bn_fwd_train_partials", M, C, dtype);
  if (rc) return rc;
  if (centered && running_mean == nullptr) { cn_set_error("bn_fwd_train_partials_centered: the pivot is running_mean"); return CN_EINVAL; }
  if (partial == nullptr || nrb <= 0) { cn_set_error("bn_fwd_train_partials: no partials"); return CN_EINVAL; }
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  if (nrb > BN_TARGET_BLOCKS) {
    const int G = (nrb + BN_TARGET_BLOCKS - 1) / BN_TARGET_BLOCKS;
    const int nr2 = (nrb + G - 1) / G;
    if (workspace == nullptr || ws_bytes < (size_t)nr2 * 2 * C * sizeof(float)) {
      cn_set_error("bn_fwd_train_partials: workspace too small");
      return CN_EWORKSPACE;
    }
    CN_LAUNCH(bn_partials_compress_kernel, dim3((unsigned)((2 * C + 255) / 256), (unsigned)nr2), dim3(256), stream,
              partial, (float*)workspace, nrb, G, 2 * C);
    partial = (const float*)workspace;
    nrb = nr2;
  }
  return bn_fwd_tail(partial, nrb, y, residual, z, relu_mask, gamma, beta, running_mean, running_var,
                     num_batches_tracked, momentum, eps, stats_out, M, C, relu, dtype, m, stream, centered);
}

extern "C" int cn_bn_fwd_train_partials(const void* y, const void* residual, void* z, unsigned char* relu_mask,
                                        const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, long long* num_batches_tracked, float momentum,
                                        float eps, float* stats_out, int M, int C, int relu, int dtype,
                                        const float* partial, int nrb, void* workspace, size_t ws_bytes,
                                        void* stream_) {
  return bn_fwd_train_partials_impl(y, residual, z, relu_mask, gamma, beta, running_mean, running_var,
                                    num_batches_tracked, momentum, eps, stats_out, M, C, relu, dtype, partial, nrb,
                                    workspace, ws_bytes, stream_, 0);
}

// Partials that cn_conv2d_fwd_bnstats_centered emitted with pivot = running_mean (as it is now, before this call
// updates it): sum (y - running_mean) | sum (y - running_mean)^2 per row.
extern "C" int cn_bn_fwd_train_partials_centered(const void* y, const void* residual, void* z,
                                                 unsigned char* relu_mask, const float* gamma, const float* beta,
                                                 float* running_mean, float* running_var,
                                                 long long* num_batches_tracked, float momentum, float eps,
                                                 float* stats_out, int M, int C, int relu, int dtype,
                                                 const float* partial, int nrb, void* workspace, size_t ws_bytes,
                                                 void* stream_) {
  return bn_fwd_train_partials_impl(y, residual, z, relu_mask, gamma, beta, running_mean, running_var,
                                    num_batches_tracked, momentum, eps, stats_out, M, C, relu, dtype, partial, nrb,
                                    workspace, ws_bytes, stream_, 1);
}

// Inference forward from running statistics.  coeffs = scratch of 2*C floats.
extern "C" int cn_bn_fwd_infer(const void* y, const void* residual, void* z, const float* gamma,
                               const float* beta, const float* running_mean, const float* running_var,
                               float eps, float* coeffs, int M, int C, int relu, int dtype, void* stream_) {
  int rc = bn_check("bn_fwd_infer", M, C, dtype);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  CN_LAUNCH(bn_infer_coeffs_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), stream, C, gamma, beta,
            running_mean, running_var, eps, coeffs, coeffs + C);
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  const int rev = (cn_get_option("bn_reverse", BN_REVERSE_DEFAULT) & 1);
  BN_DISPATCH(bn_apply_kernel, dtype, bn_nt_flag(M, C, dtype), agrid, stream, (const char*)y, (const char*)residual, (char*)z, (unsigned char*)nullptr, (const float*)coeffs, (const float*)(coeffs + C), M, C, relu, m.tpr_log2, rev, (const float*)nullptr, (const float*)nullptr);
  return cn_check_launch("bn_fwd_infer");
}

// The apply pass of a residual junction whose shortcut is a projection (conv + BatchNorm) with BOTH BatchNorms
// already finalised (cn_bn_fwd_train* called with z = NULL: statistics, running statistics, scale / shift only):
//   z = relu?( y*scale[c] + shift[c] + round_T(res_y*rscale[c] + rshift[c]) )
// stats / res_stats = the 4*C floats [save_mean | save_invstd | scale | shift] of the junction / shortcut BatchNorm.
// Bit-identical to the shortcut BatchNorm's own apply followed by the junction's (tests/test_ops.py), without the
// write and the re-read of the normalised shortcut tensor.
extern "C" int cn_bn_apply_dual(const void* y, const void* res_y, void* z, unsigned char* relu_mask,
                                const float* stats, const float* res_stats, int M, int C, int relu, int dtype,
                                void* stream_) {
  int rc = bn_check("bn_apply_dual", M, C, dtype);
  if (rc) return rc;
  if (y == nullptr || res_y == nullptr || z == nullptr || stats == nullptr || res_stats == nullptr) {
    cn_set_error("bn_apply_dual: null operand");
    return CN_EINVAL;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  const int rev = (cn_get_option("bn_reverse", BN_REVERSE_DEFAULT) & 1);
  const int nt = bn_nt_flag(M, C, dtype);
#define BN_DUAL_LAUNCH(TT)                                                                                             \
  do {                                                                                                                 \
    if (nt) CN_LAUNCH((bn_apply_kernel<TT, true, true>), agrid, dim3(256), stream, (const char*)y, (const char*)res_y,  \
                      (char*)z, relu_mask, stats + 2 * C, stats + 3 * C, M, C, relu, m.tpr_log2, rev,                  \
                      res_stats + 2 * C, res_stats + 3 * C);                                                           \
    else CN_LAUNCH((bn_apply_kernel<TT, false, true>), agrid, dim3(256), stream, (const char*)y, (const char*)res_y,    \
                   (char*)z, relu_mask, stats + 2 * C, stats + 3 * C, M, C, relu, m.tpr_log2, rev,                     \
                   res_stats + 2 * C, res_stats + 3 * C);                                                              \
  } while (0)
  if (dtype == CN_BF16) BN_DUAL_LAUNCH(bf16_t);
  else if (dtype == CN_F16) BN_DUAL_LAUNCH(f16_t);
  else BN_DUAL_LAUNCH(float);
#undef BN_DUAL_LAUNCH
  return cn_check_launch("bn_apply_dual");
}

// Training backward.  stats = the 4*C floats written by cn_bn_fwd_train; coef_scratch = 3*C floats.
// dgamma/dbeta are written (beta_acc = 0) or accumulated (beta_acc = 1).  dres (optional) receives
// the masked upstream gradient for the residual branch.
extern "C" int cn_bn_bwd(const void* dz, const void* y, const unsigned char* relu_mask, const float* gamma,
                         const float* stats, void* dy, void* dres, float* dgamma, float* dbeta,
                         float beta_acc, float gscale, float* coef_scratch, int M, int C, int relu, int dtype,
                         void* workspace, size_t ws_bytes, void* stream_) {
  int rc = bn_check("bn_bwd", M, C, dtype);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  int nrb = bn_row_blocks(M, m, cn_get_option("bn_reduce_blocks", BN_REDUCE_BLOCKS));
  if (workspace == nullptr || ws_bytes < (size_t)nrb * 2 * C * sizeof(float)) {
    cn_set_error("bn_bwd: workspace too small");
    return CN_EWORKSPACE;
  }
  float* partial = (float*)workspace;
  const float* mean = stats;
  const float* invstd = stats + C;
  const float* scale = stats + 2 * C;
  const float* shift = stats + 3 * C;
  dim3 grid((unsigned)nrb, (unsigned)m.gy);
  const int revopt = cn_get_option("bn_reverse", BN_REVERSE_DEFAULT);
  const int rev_r = (revopt >> 1) & 1, rev_a = ((revopt >> 2) & 1);
  CnMarkLast last;   // an armed completion mark goes on the apply kernel only
  BN_DISPATCH_PLAIN(bn_bwd_reduce_kernel, dtype, grid, stream, (const char*)dz, (const char*)y, relu_mask, mean, invstd, scale, shift, partial, M, C, relu, m.tpr_log2, rev_r);
  if (dy == nullptr) {
    // "lazy dy": reduce + finalize only, the consumers form c1*dz + c2*y + c3 themselves.  Only where dz needs no mask
    // (no ReLU behind this BatchNorm) and no residual-branch copy is wanted.
    if (relu != 0 || dres != nullptr) { cn_set_error("bn_bwd: dy = NULL needs relu = 0 and no dres"); return CN_EINVAL; }
    last.release();
  }
  CN_LAUNCH(bn_bwd_finalize_kernel, dim3((unsigned)((C + BN_FC - 1) / BN_FC)), dim3(256), stream, (const float*)partial,
            nrb, M, C, gamma, mean, invstd, dgamma, dbeta, beta_acc, gscale, coef_scratch);
  if (dy == nullptr) return cn_check_launch("bn_bwd");
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  last.release();
  BN_DISPATCH(bn_bwd_apply_kernel, dtype, bn_nt_flag(M, C, dtype), agrid, stream, (const char*)dz, (const char*)y, relu_mask, scale, shift, (const float*)coef_scratch, (char*)dy, (char*)dres, M, C, relu, m.tpr_log2, rev_a);
  return cn_check_launch("bn_bwd");
}

// Training backward when the producer of the upstream gradient already masked it and reduced it
// (cn_conv2d_dgrad_bnbwd): g = dz * relu_mask, partial = [nrb][2*C] floats of sum g | sum g*xhat.
// Runs finalize + apply only (no second pass over g and y for the sums, no dres: it is g itself).
extern "C" int cn_bn_bwd_partials(const void* g, const void* y, const float* gamma, const float* stats, void* dy,
                                  float* dgamma, float* dbeta, float beta_acc, float gscale,
                                  float* coef_scratch, int M, int C, int dtype, const float* partial, int nrb,
                                  void* workspace, size_t ws_bytes, void* stream_) {
  int rc = bn_check("bn_bwd_partials", M, C, dtype);
  if (rc) return rc;
  if (partial == nullptr || nrb <= 0) { cn_set_error("bn_bwd_partials: no partials"); return CN_EINVAL; }
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  CnMarkLast last;   // an armed completion mark goes on the apply kernel only
  if (nrb > BN_TARGET_BLOCKS) {
    const int G = (nrb + BN_TARGET_BLOCKS - 1) / BN_TARGET_BLOCKS;
    const int nr2 = (nrb + G - 1) / G;
    if (workspace == nullptr || ws_bytes < (size_t)nr2 * 2 * C * sizeof(float)) {
      cn_set_error("bn_bwd_partials: workspace too small");
      return CN_EWORKSPACE;
    }
    CN_LAUNCH(bn_partials_compress_kernel, dim3((unsigned)((2 * C + 255) / 256), (unsigned)nr2), dim3(256), stream,
              partial, (float*)workspace, nrb, G, 2 * C);
    partial = (const float*)workspace;
    nrb = nr2;
  }
  const float* mean = stats;
  const float* invstd = stats + C;
  const float* scale = stats + 2 * C;
  const float* shift = stats + 3 * C;
  if (dy == nullptr) last.release();   // finalize only: that kernel is the call's last
  CN_LAUNCH(bn_bwd_finalize_kernel, dim3((unsigned)((C + BN_FC - 1) / BN_FC)), dim3(256), stream, partial, nrb, M, C, gamma,
            mean, invstd, dgamma, dbeta, beta_acc, gscale, coef_scratch);
  // dy == NULL ("lazy dy"): the consumers (cn_conv2d_dgrad_lazy / cn_conv2d_wgrad_lazy) form c1*g + c2*y + c3 on
  // their operand loads from coef_scratch; no apply pass
  if (dy == nullptr) return cn_check_launch("bn_bwd_partials");
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  const int rev_a = ((cn_get_option("bn_reverse", BN_REVERSE_DEFAULT) >> 2) & 1);
  last.release();
  BN_DISPATCH(bn_bwd_apply_kernel, dtype, bn_nt_flag(M, C, dtype), agrid, stream, (const char*)g, (const char*)y, (const unsigned char*)nullptr, scale, shift, (const float*)coef_scratch, (char*)dy, (char*)nullptr, M, C, 0, m.tpr_log2, rev_a);
  return cn_check_launch("bn_bwd_partials");
}

// ------------------------------------------------------------------------------------------------
// Cross-rank (SyncBatchNorm) building blocks, /root/reference main.py:190-191
// (nn.SyncBatchNorm.convert_sync_batchnorm).  The library holds no communicator: each rank reduces
// its own partials to 2*C double-precision sums, the caller all-reduces that small buffer
// (torch.distributed over RCCL) and hands the global sums + global row count back.

// out[col] = sum_r partial[r][col]  (double accumulation, fixed order, 8 threads per column)
__global__ __launch_bounds__(256) void bn_partials_total_kernel(const float* partial, int nrb, int W, double* out) {
  __shared__ double red[256];
  const int lc = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + lc;
  double s = 0.0;
  if (col < W) {
    int r = part;
    for (; r + 56 < nrb; r += 64) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = partial[(size_t)(r + 8 * u) * W + col];
      s += (((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3])) +
           (((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]));
    }
    for (; r < nrb; r += 8) s += (double)partial[(size_t)r * W + col];
  }
  red[part * 32 + lc] = s;
  __syncthreads();
  if (part == 0 && col < W) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k * 32 + lc];
    out[col] = t;
  }
}

// statistics from global sums: sums = [sum x | sum x^2] over m_total rows of all ranks
__global__ __launch_bounds__(256) void bn_finalize_sums_kernel(const double* sums, long long m_total, int C,
                                                              const float* gamma, const float* beta,
                                                              float* running_mean, float* running_var,
                                                              long long* num_batches_tracked, float momentum,
                                                              float eps, float* save_mean, float* save_invstd,
                                                              float* scale, float* shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
  if (c >= C) return;
  const double mean = sums[c] / (double)m_total;
  double var = sums[C + c] / (double)m_total - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (running_mean != nullptr) {
    const double unbiased = m_total > 1 ? var * (double)m_total / (double)(m_total - 1) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
  }
  const float g = gamma != nullptr ? gamma[c] : 1.f;
  const float b = beta != nullptr ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
}

// dgamma / dbeta from this rank's sums (the data-parallel gradient all-reduce averages them like every
// other parameter gradient), input-gradient coefficients from the global sums over m_total rows
__global__ __launch_bounds__(256) void bn_bwd_finalize_sums_kernel(const double* local, const double* global,
                                                                  long long m_total, int C, const float* gamma,
                                                                  const float* mean, const float* invstd,
                                                                  float* dgamma, float* dbeta, float beta_acc,
                                                                  float gscale, float* coef) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float g = gamma != nullptr ? gamma[c] : 1.f;
  if (dgamma != nullptr) dgamma[c] = (beta_acc != 0.f ? beta_acc * dgamma[c] : 0.f) + (float)local[C + c] * gscale;
  if (dbeta != nullptr) dbeta[c] = (beta_acc != 0.f ? beta_acc * dbeta[c] : 0.f) + (float)local[c] * gscale;
  const double k = (double)g * (double)invstd[c];
  const double a2 = k * (double)invstd[c] * global[C + c] / (double)m_total;
  coef[c] = (float)k;
  coef[C + c] = (float)(-a2);
  coef[2 * C + c] = (float)(a2 * (double)mean[c] - k * global[c] / (double)m_total);
}

// This rank's [sum y | sum y^2] (2*C doubles).  partial/nrb: rows a conv epilogue already produced
// (cn_conv2d_fwd_bnstats), or NULL/0 to run the statistics pass over y here.
extern "C" int cn_bn_local_sums(const void* y, int M, int C, int dtype, const float* partial, int nrb, double* sums,
                                void* workspace, size_t ws_bytes, void* stream_) {
  int rc = bn_check("bn_local_sums", M, C, dtype);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (partial == nullptr) {
    const int CH = cn_dtype_chunk(dtype);
    BnMap m = bn_map(C / CH);
    nrb = bn_row_blocks(M, m, cn_get_option("bn_reduce_blocks", BN_REDUCE_BLOCKS));
    if (workspace == nullptr || ws_bytes < (size_t)nrb * 2 * C * sizeof(float)) {
      cn_set_error("bn_local_sums: workspace too small");
      return CN_EWORKSPACE;
    }
    dim3 grid((unsigned)nrb, (unsigned)m.gy);
    CN_DISPATCH_T(dtype, CN_LAUNCH(bn_stats_kernel<TT>, grid, dim3(256), stream, (const char*)y, (float*)workspace, M, C, m.tpr_log2, (const float*)nullptr));
    partial = (const float*)workspace;
  }
  CN_LAUNCH(bn_partials_total_kernel, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), stream, partial, nrb, 2 * C, sums);
  return cn_check_launch("bn_local_sums");
}

// Training forward from (all-reduced) sums over m_total rows; M = this rank's rows.
extern "C" int cn_bn_fwd_train_sums(const void* y, const void* residual, void* z, unsigned char* relu_mask,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float momentum, float eps, float* stats_out,
                                    int M, int C, int relu, int dtype, const double* sums, long long m_total,
                                    void* stream_) {
  int rc = bn_check("bn_fwd_train_sums", M, C, dtype);
  if (rc) return rc;
  if (sums == nullptr || m_total < M) { cn_set_error("bn_fwd_train_sums: bad sums / m_total"); return CN_EINVAL; }
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  CN_LAUNCH(bn_finalize_sums_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), stream, sums, m_total, C, gamma,
            beta, running_mean, running_var, num_batches_tracked, momentum, eps, stats_out, stats_out + C,
            stats_out + 2 * C, stats_out + 3 * C);
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  BN_DISPATCH(bn_apply_kernel, dtype, bn_nt_flag(M, C, dtype), agrid, stream, (const char*)y, (const char*)residual, (char*)z, relu_mask, (const float*)(stats_out + 2 * C), (const float*)(stats_out + 3 * C), M, C, relu, m.tpr_log2, 0, (const float*)nullptr, (const float*)nullptr);
  return cn_check_launch("bn_fwd_train_sums");
}

// This rank's [sum g | sum g*xhat] (2*C doubles), g = dz * relu_mask.  partial/nrb: rows a dgrad
// epilogue already produced (then dz is g), or NULL/0 to run the reduction pass here.
extern "C" int cn_bn_bwd_local_sums(const void* dz, const void* y, const unsigned char* relu_mask,
                                    const float* stats, int M, int C, int relu, int dtype, const float* partial,
                                    int nrb, double* sums, void* workspace, size_t ws_bytes, void* stream_) {
  int rc = bn_check("bn_bwd_local_sums", M, C, dtype);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (partial == nullptr) {
    const int CH = cn_dtype_chunk(dtype);
    BnMap m = bn_map(C / CH);
    nrb = bn_row_blocks(M, m, cn_get_option("bn_reduce_blocks", BN_REDUCE_BLOCKS));
    if (workspace == nullptr || ws_bytes < (size_t)nrb * 2 * C * sizeof(float)) {
      cn_set_error("bn_bwd_local_sums: workspace too small");
      return CN_EWORKSPACE;
    }
    dim3 grid((unsigned)nrb, (unsigned)m.gy);
    BN_DISPATCH_PLAIN(bn_bwd_reduce_kernel, dtype, grid, stream, (const char*)dz, (const char*)y, relu_mask, stats, stats + C, stats + 2 * C, stats + 3 * C, (float*)workspace, M, C, relu, m.tpr_log2, 0);
    partial = (const float*)workspace;
  }
  CN_LAUNCH(bn_partials_total_kernel, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), stream, partial, nrb, 2 * C, sums);
  return cn_check_launch("bn_bwd_local_sums");
}

// Training backward from sums: dgamma/dbeta from `local_sums`, dy from `global_sums` / m_total.
// pre_masked != 0: dz is already g (masked by the dgrad epilogue).
extern "C" int cn_bn_bwd_sums(const void* dz, const void* y, const unsigned char* relu_mask, const float* gamma,
                              const float* stats, void* dy, void* dres, float* dgamma, float* dbeta, float beta_acc,
                              float gscale, float* coef_scratch, int M, int C, int relu, int pre_masked, int dtype,
                              const double* local_sums, const double* global_sums, long long m_total,
                              void* stream_) {
  int rc = bn_check("bn_bwd_sums", M, C, dtype);
  if (rc) return rc;
  if (local_sums == nullptr || global_sums == nullptr || m_total < M) {
    cn_set_error("bn_bwd_sums: bad sums / m_total");
    return CN_EINVAL;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  const float* mean = stats;
  const float* invstd = stats + C;
  const float* scale = stats + 2 * C;
  const float* shift = stats + 3 * C;
  CN_LAUNCH(bn_bwd_finalize_sums_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), stream, local_sums,
            global_sums, m_total, C, gamma, mean, invstd, dgamma, dbeta, beta_acc, gscale, coef_scratch);
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  const int arelu = pre_masked ? 0 : relu;
  const unsigned char* amask = pre_masked ? nullptr : relu_mask;
  BN_DISPATCH(bn_bwd_apply_kernel, dtype, bn_nt_flag(M, C, dtype), agrid, stream, (const char*)dz, (const char*)y, amask, scale, shift, (const float*)coef_scratch, (char*)dy, (char*)dres, M, C, arelu, m.tpr_log2, 0);
  return cn_check_launch("bn_bwd_sums");
}

// Backward of bn -> relu -> maxpool(k, stride, pad) given the gradient of the pooled map: the pool's
// gather backward is folded into both BatchNorm-backward passes (no dense dz tensor).
// y = BN input [N,H,W,C]; dpool / idx = [N,P,Q,C]; stats = the 4*C floats of the forward.
static int bn_bwd_maxpool_impl(const void* dpool, const unsigned char* idx, const void* y, const void* xmax,
                               const float* gamma, const float* stats, void* dy, float* dgamma, float* dbeta,
                               float beta_acc, float gscale, float* coef_scratch, int N, int H, int W, int C, int k,
                               int stride, int pad, int dtype, void* workspace, size_t ws_bytes, void* stream_) {
  const long long Ml = (long long)N * H * W;
  if (Ml >= (1ll << 31)) { cn_set_error("bn_bwd_maxpool: too many rows"); return CN_ESHAPE; }
  const int M = (int)Ml;
  int rc = bn_check("bn_bwd_maxpool", M, C, dtype);
  if (rc) return rc;
  if (k * k > 255 || pad * 2 > k || stride <= 0 || k > 2 * stride) {
    cn_set_error("bn_bwd_maxpool: unsupported window (needs k <= 2*stride, k*k <= 255, 2*pad <= k)");
    return CN_ESHAPE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = cn_dtype_chunk(dtype);
  BnMap m = bn_map(C / CH);
  int nrb = bn_row_blocks(M, m, cn_get_option("bn_reduce_blocks", BN_REDUCE_BLOCKS));
  if (workspace == nullptr || ws_bytes < (size_t)nrb * 2 * C * sizeof(float)) {
    cn_set_error("bn_bwd_maxpool: workspace too small");
    return CN_EWORKSPACE;
  }
  BnPoolGeom geo;
  geo.dpool = (const char*)dpool; geo.idx = idx;
  geo.H = H; geo.W = W; geo.k = k; geo.st = stride; geo.pad = pad;
  geo.P = (H + 2 * pad - k) / stride + 1;
  geo.Q = (W + 2 * pad - k) / stride + 1;
  geo.div_hw = cn_make_fastdiv((unsigned)(H * W));
  geo.div_w = cn_make_fastdiv((unsigned)W);
  geo.div_st = cn_make_fastdiv((unsigned)stride);
  float* partial = (float*)workspace;
  const float* mean = stats;
  const float* invstd = stats + C;
  const float* scale = stats + 2 * C;
  const float* shift = stats + 3 * C;
  dim3 grid((unsigned)nrb, (unsigned)m.gy);
  CnMarkLast last;   // an armed completion mark goes on the apply kernel only
  if (xmax != nullptr) {
    // sums over the pooled map: every pooled element sends its gradient to exactly one input pixel, whose
    // pre-BatchNorm value the forward kept (xmax), so sum g and sum g*xhat need neither the gather nor the big map
    const long long Mp = (long long)N * geo.P * geo.Q;
    nrb = bn_row_blocks((int)Mp, m, cn_get_option("bn_reduce_blocks", BN_REDUCE_BLOCKS));
    dim3 pgrid((unsigned)nrb, (unsigned)m.gy);
    BN_DISPATCH_PLAIN(bn_bwd_reduce_kernel, dtype, pgrid, stream, (const char*)dpool, (const char*)xmax, (const unsigned char*)nullptr, mean, invstd, scale, shift, partial, (int)Mp, C, 1, m.tpr_log2, 0);
  } else {
    CN_DISPATCH_T(dtype, CN_LAUNCH(bn_bwd_reduce_pool_kernel<TT>, grid, dim3(256), stream, geo, (const char*)y, mean, invstd, scale,
                shift, partial, M, C, m.tpr_log2));
  }
  CN_LAUNCH(bn_bwd_finalize_kernel, dim3((unsigned)((C + BN_FC - 1) / BN_FC)), dim3(256), stream, (const float*)partial,
            nrb, M, C, gamma, mean, invstd, dgamma, dbeta, beta_acc, gscale, coef_scratch);
  int nab = bn_row_blocks(M, m, cn_get_option("bn_apply_blocks", BN_APPLY_BLOCKS));
  dim3 agrid((unsigned)nab, (unsigned)m.gy);
  last.release();
  CN_DISPATCH_T(dtype, CN_LAUNCH(bn_bwd_apply_pool_kernel<TT>, agrid, dim3(256), stream, geo, (const char*)y, scale, shift,
              (const float*)coef_scratch, (char*)dy, M, C, m.tpr_log2));
  return cn_check_launch("bn_bwd_maxpool");
}

extern "C" int cn_bn_bwd_maxpool(const void* dpool, const unsigned char* idx, const void* y, const float* gamma,
                                 const float* stats, void* dy, float* dgamma, float* dbeta, float beta_acc,
                                 float gscale, float* coef_scratch, int N, int H, int W, int C, int k, int stride,
                                 int pad, int dtype, void* workspace, size_t ws_bytes, void* stream_) {
  return bn_bwd_maxpool_impl(dpool, idx, y, nullptr, gamma, stats, dy, dgamma, dbeta, beta_acc, gscale, coef_scratch, N,
                             H, W, C, k, stride, pad, dtype, workspace, ws_bytes, stream_);
}

// With the winning taps' pre-BatchNorm values (cn_maxpool_fwd_bnrelu_xmax): the reduction reads dpool and xmax only.
extern "C" int cn_bn_bwd_maxpool_xmax(const void* dpool, const unsigned char* idx, const void* y, const void* xmax,
                                      const float* gamma, const float* stats, void* dy, float* dgamma, float* dbeta,
                                      float beta_acc, float gscale, float* coef_scratch, int N, int H, int W, int C,
                                      int k, int stride, int pad, int dtype, void* workspace, size_t ws_bytes,
                                      void* stream_) {
  if (xmax == nullptr) { cn_set_error("bn_bwd_maxpool_xmax: no xmax tensor"); return CN_EINVAL; }
  return bn_bwd_maxpool_impl(dpool, idx, y, xmax, gamma, stats, dy, dgamma, dbeta, beta_acc, gscale, coef_scratch, N, H,
                             W, C, k, stride, pad, dtype, workspace, ws_bytes, stream_);
}
