// quant.hip -- the simulated-8-bit training operators of BASELINE config 5 (ResNet {'quantize': True}), gfx950.
//
// Replaces (reference, /root/reference models/modules/quantize.py): calculate_qparams (:19-38),
// UniformQuantize.forward (:41-76), UniformQuantizeGrad.backward (:101-112), QuantMeasure (:140-182), the
// per-output-channel weight quantiser of QConv2d / QLinear (:201-203, :239-240) and RangeBN (:256-330) with
// the gradient autograd derives for it (mean path + max / min routing).  Like the reference this is
// *simulated* integer arithmetic: tensors are snapped to the 2^bits-level grid and stay floating point, the
// convolutions themselves run on the ordinary MFMA kernels (igemm.hip / wgrad.hip).
//
// All of these are streaming (HBM-bound) passes: 16-byte coalesced accesses, grid-stride loops, two-stage
// fixed-order reductions (deterministic).  Arithmetic follows the reference's operation order in fp32 and is
// compiled without FMA contraction, so an fp32 run lands on the same quantisation levels as the CPU oracle.
#pragma clang fp contract(off)
#include "cn_common.h"
#include "cn_api_internal.h"
#include <math.h>

#define Q_NT 256

// ------------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ void q_block_minmax(float& mn, float& mx, float* red /* [8] */) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    mn = fminf(mn, cn_shfl_xor(mn, m));
    mx = fmaxf(mx, cn_shfl_xor(mx, m));
  }
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = mn; red[4 + (tid >> 6)] = mx; }
  __syncthreads();
  mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
}

// v as a T stores it (round to nearest even for the 16-bit types)
template <typename T> __device__ __forceinline__ float q_round_to(float v) { T t; cn_store_elem<T>(&t, v); return cn_load_elem<T>(&t); }

// Stage 1: block (split s, row r) scans its slice of row r.  partial[(r*splits + s)*2] = {min, max}.
template <typename T>
__global__ __launch_bounds__(Q_NT) void minmax_partial_kernel(const T* x, long long row_len, int splits, int vec,
                                                             float* partial) {
  __shared__ float red[8];
  constexpr int CH = ElemTraits<T>::kChunk;
  const int s = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const T* row = x + (size_t)r * (size_t)row_len;
  float mn = INFINITY, mx = -INFINITY;
  if (vec) {   // rows are whole 16-byte chunks
    const long long nch = row_len / CH;
    const long long per = (nch + splits - 1) / splits;
    const long long c0 = (long long)s * per, c1 = c0 + per < nch ? c0 + per : nch;
    auto visit = [&](const u32x4& v) {
      float f[CH];
      Chunk<T>::unpack(v, f);
#pragma unroll
      for (int e = 0; e < CH; ++e) { mn = fminf(mn, f[e]); mx = fmaxf(mx, f[e]); }
    };
    long long c = c0 + tid;
    for (; c + 3 * Q_NT < c1; c += 4 * Q_NT) {   // four loads in flight per thread (one alone leaves HBM latency exposed)
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = cn_ld16((const char*)row + (c + u * Q_NT) * 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) visit(v[u]);
    }
    for (; c < c1; c += Q_NT) visit(cn_ld16((const char*)row + c * 16));
  } else {
    const long long per = (row_len + splits - 1) / splits;
    const long long e0 = (long long)s * per, e1 = e0 + per < row_len ? e0 + per : row_len;
    for (long long e = e0 + tid; e < e1; e += Q_NT) {
      const float v = cn_load_elem<T>(row + e);
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
  }
  q_block_minmax(mn, mx, red);
  if (tid == 0) {
    partial[((size_t)r * splits + s) * 2] = mn;
    partial[((size_t)r * splits + s) * 2 + 1] = mx;
  }
}

// qp_extreme (optional, single-block launches only: rows <= Q_NT): additionally [zero_point, range] of the 'extreme' reduction
// over the rows - cn_qparams(minmax, rows, mode 1)'s output, same values - for a consumer that is known to be a gradient
// quantiser (round 6: 105 qparams launches per config-5 step less).
__global__ __launch_bounds__(Q_NT) void minmax_final_kernel(const float* partial, int rows, int splits, float* minmax,
                                                           float* qp_extreme) {
  __shared__ float red[8];
  const int r = blockIdx.x * Q_NT + threadIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  if (r < rows) {
    for (int s = 0; s < splits; ++s) {
      mn = fminf(mn, partial[((size_t)r * splits + s) * 2]);
      mx = fmaxf(mx, partial[((size_t)r * splits + s) * 2 + 1]);
    }
    minmax[2 * r] = mn;
    minmax[2 * r + 1] = mx;
  }
  if (qp_extreme != nullptr) {      // (uniform: every thread of the one block takes part)
    q_block_minmax(mn, mx, red);
    if (threadIdx.x == 0) {
      float range = mx - mn;
      if (range == 0.f) range = 1.f;
      qp_extreme[0] = mn;
      qp_extreme[1] = range;
    }
  }
}

static int q_splits(int rows, long long row_len) {
  long long s = (row_len + 16383) / 16384;          // >= 16 Ki elements per block
  const long long cap = rows >= 1024 ? 1 : (1024 + rows - 1) / rows;
  if (s > cap) s = cap;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" size_t cn_minmax_workspace(int rows, long long row_len) {
  return (size_t)rows * (size_t)q_splits(rows, row_len) * 2 * sizeof(float);
}

// Per-row minimum / maximum of `rows` contiguous rows of `row_len` elements: minmax[r] = {min, max}.
// (calculate_qparams' x.flatten(1).min(-1) / .max(-1), quantize.py:21-27.)
extern "C" int cn_minmax_rows(const void* x, int rows, long long row_len, int dtype, float* minmax, float* ws,
                              size_t ws_bytes, void* stream_) {
  if (rows <= 0 || row_len <= 0 || x == nullptr || minmax == nullptr) { cn_set_error("minmax_rows: bad arguments"); return CN_EINVAL; }
  if (rows > 65535) { cn_set_error("minmax_rows: %d rows > 65535", rows); return CN_ESHAPE; }
  if (ws == nullptr || ws_bytes < cn_minmax_workspace(rows, row_len)) { cn_set_error("minmax_rows: workspace too small"); return CN_EWORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  const int splits = q_splits(rows, row_len);
  dim3 grid((unsigned)splits, (unsigned)rows);
  if (dtype == CN_BF16) {
    const int vec = row_len % 8 == 0 && ((uintptr_t)x & 15) == 0;
    CN_LAUNCH(minmax_partial_kernel<bf16_t>, grid, dim3(Q_NT), stream, (const bf16_t*)x, row_len, splits, vec, ws);
  } else if (dtype == CN_F32) {
    const int vec = row_len % 4 == 0 && ((uintptr_t)x & 15) == 0;
    CN_LAUNCH(minmax_partial_kernel<float>, grid, dim3(Q_NT), stream, (const float*)x, row_len, splits, vec, ws);
  } else { cn_set_error("minmax_rows: bad dtype %d", dtype); return CN_EINVAL; }
  CN_LAUNCH(minmax_final_kernel, dim3((unsigned)((rows + Q_NT - 1) / Q_NT)), dim3(Q_NT), stream, (const float*)ws, rows,
            splits, minmax, (float*)nullptr);
  return cn_check_launch("minmax_rows");
}

// ------------------------------------------------------------------------------------------------ qparams
// mode 0: zero_point = mean_r min_r, range = mean_r max_r - mean_r min_r   (reduce_type 'mean', activations)
// mode 1: zero_point = min_r min_r,  range = max_r max_r - min_r min_r     (reduce_type 'extreme', gradients)
// A zero range is reported as 1 (the quantiser is then the identity on that constant tensor; the reference
// divides by zero there -- documented repair, oracle/make_golden_quant.py).  Optional running update
// (QuantMeasure, quantize.py:169-172): running = running * momentum + new * (1 - momentum).
__global__ __launch_bounds__(Q_NT) void qparams_kernel(const float* minmax, int rows, int mode, float* qp,
                                                      float* running_zp, float* running_range, float momentum) {
  __shared__ double sred[2 * Q_NT];
  const int tid = threadIdx.x;
  double a = mode == 0 ? 0.0 : (double)INFINITY, b = mode == 0 ? 0.0 : -(double)INFINITY;
  for (int r = tid; r < rows; r += Q_NT) {
    const double mn = minmax[2 * r], mx = minmax[2 * r + 1];
    if (mode == 0) { a += mn; b += mx; } else { a = mn < a ? mn : a; b = mx > b ? mx : b; }
  }
  sred[tid] = a;
  sred[Q_NT + tid] = b;
  __syncthreads();
  // fixed-order tree over the Q_NT partials (round 6: thread 0 used to add them one by one - 255 dependent LDS reads,
  // 18 us per call, 214 calls per config-5 step = 4 ms; the double sums round to the same float either way)
  for (int h = Q_NT / 2; h > 0; h >>= 1) {
    if (tid < h) {
      if (mode == 0) { sred[tid] += sred[tid + h]; sred[Q_NT + tid] += sred[Q_NT + tid + h]; }
      else {
        sred[tid] = sred[tid + h] < sred[tid] ? sred[tid + h] : sred[tid];
        sred[Q_NT + tid] = sred[Q_NT + tid + h] > sred[Q_NT + tid] ? sred[Q_NT + tid + h] : sred[Q_NT + tid];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    a = sred[0];
    b = sred[Q_NT];
    float zp, mxv;
    if (mode == 0) { zp = (float)(a / rows); mxv = (float)(b / rows); } else { zp = (float)a; mxv = (float)b; }
    float range = mxv - zp;
    if (range == 0.f) range = 1.f;
    qp[0] = zp;
    qp[1] = range;
    if (running_zp != nullptr) {
      const float keep = 1.f - momentum;
      running_zp[0] = running_zp[0] * momentum + zp * keep;
      running_range[0] = running_range[0] * momentum + range * keep;
    }
  }
}

extern "C" int cn_qparams(const float* minmax, int rows, int mode, float* qp, float* running_zp, float* running_range,
                          float momentum, void* stream) {
  if (minmax == nullptr || qp == nullptr || rows <= 0 || (mode != 0 && mode != 1)) { cn_set_error("qparams: bad arguments"); return CN_EINVAL; }
  if ((running_zp == nullptr) != (running_range == nullptr)) { cn_set_error("qparams: both running buffers or none"); return CN_EINVAL; }
  CN_LAUNCH(qparams_kernel, dim3(1), dim3(Q_NT), (hipStream_t)stream, minmax, rows, mode, qp, running_zp, running_range,
            momentum);
  return cn_check_launch("qparams");
}

// ------------------------------------------------------------------------------------------------ quantise
// Counter-based uniform noise in (-0.5, 0.5) for stochastic rounding when no noise tensor is supplied: a 32-bit
// integer hash (two multiply / xor-shift rounds) of the element index under a per-launch key.  (The first version used
// a 64-bit splitmix round per element: ~30 VALU instructions each, which made the gradient quantiser issue-bound at
// ~3 TB/s instead of HBM-bound.)
__host__ __device__ __forceinline__ unsigned int q_noise_key(unsigned long long seed) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned int)(z ^ (z >> 31));
}
__host__ __device__ __forceinline__ float q_hash_noise(unsigned int key, unsigned long long idx) {
  unsigned int h = ((unsigned int)idx ^ key) + (unsigned int)(idx >> 32) * 0x9E3779B1u;
  h ^= h >> 16; h *= 0x7feb352du;
  h ^= h >> 15; h *= 0x846ca68bu;
  h ^= h >> 16;
  return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f) - 0.5f;
}

// UniformQuantize.forward, unsigned, dequantised (quantize.py:55-76), operation for operation:
//   t = (x + (-zp)) / scale  [+ noise];  t = round_half_even(clamp(t, 0, qmax));  y = t * scale + zp
// (split into the LEVEL t and its de-quantisation so that the kernels which store 8-bit levels instead of snapped values -
// round 6 - run the very same arithmetic as the ones that store the values)
__host__ __device__ __forceinline__ float q_level(float x, float zp, float scale, float qmax, float noise) {
  float t = (x + (-zp)) / scale;
  t = t + noise;
  t = fminf(fmaxf(t, 0.f), qmax);
  return rintf(t);
}
__host__ __device__ __forceinline__ float q_dequant(float t, float zp, float scale) { return t * scale + zp; }
__host__ __device__ __forceinline__ float q_snap(float x, float zp, float scale, float qmax, float noise) {
  return q_dequant(q_level(x, zp, scale, qmax, noise), zp, scale);
}

// q_level for a whole chunk with ONE division-free pass where that is provably exact (round 6).  The level is
// rint(clamp(fl(fl(d / scale) + noise))); with ta = fl(d * fl(1 / scale)) in place of the quotient the sum moves by at most
// 5u|Q| + u (u = 2^-24: one rounding each in the reciprocal, the product and the two sums), i.e. < 4e-7 |ta| + 1e-7 - so
// wherever fl(ta + noise) lies further than that from a half-integer, both forms round to the same integer (and clamp alike:
// the bounds are integers).  A chunk holding ANY element inside that band (probability ~1e-4 per element), a NaN or an
// overflowing product is redone with the division: the result is q_level's bit for bit, at ~5 fewer VALU issue slots per element
// (the division is 10-11 of the 17 a deterministic quantiser spends per element, of ~32 with the noise hash).
template <int CH>
__host__ __device__ __forceinline__ void q_levels_chunk(const float* f, const float* nz, float zp, float scale, float inv,
                                                        float qmax, float* lv) {
#ifdef CN_EXACT_DIV          // (A/B build: csrc/build.sh with CN_EXTRA_FLAGS=-DCN_EXACT_DIV)
  bool unsure = true;
#else
  bool unsure = false;
#endif
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    const float ta = (f[e] + (-zp)) * inv;
    const float sa = ta + nz[e];
    const float r = rintf(sa);
    const float tol = fmaf(fabsf(ta), 4e-7f, 1e-7f);
    unsure = unsure || !(0.5f - fabsf(sa - r) > tol);      // (also true for NaN / Inf: every comparison with them is false)
    lv[e] = fminf(fmaxf(r, 0.f), qmax);
  }
  if (unsure) {
#pragma unroll
    for (int e = 0; e < CH; ++e) lv[e] = q_level(f[e], zp, scale, qmax, nz[e]);
  }
}

// ---- 8-bit level storage (round 6).  A tensor snapped to a <= 256-level grid is kept as one byte per element (CH bytes per
// chunk of CH elements, same chunk order) and de-quantised on load: value = T-rounded q_dequant(level) - exactly what the
// kernels that store snapped values write.  `raw` carries either the 16 bytes of a value chunk or the CH level bytes.
template <typename T>
__device__ __forceinline__ void q_st_levels(unsigned char* base, long long chunk, const float* lv) {
  constexpr int CH = ElemTraits<T>::kChunk;
  unsigned int w0 = 0, w1 = 0;
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    const unsigned int b = (unsigned int)lv[e] & 255u;
    if (e < 4) w0 |= b << (8 * e); else w1 |= b << (8 * (e - 4));
  }
  if (CH == 8) { u32x2 v; v.x = w0; v.y = w1; *(u32x2*)(base + chunk * 8) = v; }
  else *(unsigned int*)(base + chunk * 4) = w0;
}
template <typename T, bool Q8>
__device__ __forceinline__ u32x4 q_ld_raw(const void* base, long long chunk) {
  constexpr int CH = ElemTraits<T>::kChunk;
  if (!Q8) return cn_ld16((const char*)base + chunk * 16);
  u32x4 r = cn_zero16();
  if (CH == 8) { const u32x2 v = *(const u32x2*)((const char*)base + chunk * 8); r.x = v.x; r.y = v.y; }
  else r.x = *(const unsigned int*)((const char*)base + chunk * 4);
  return r;
}
struct QOp {        // how an operand is stored: values (q8 == 0) or levels of the grid (zp, scale)
  int q8;
  float zp, scale;
};
__device__ __forceinline__ QOp q_op_make(const float* qp, float qmax) {
  QOp o;
  o.q8 = qp != nullptr;
  o.zp = o.q8 ? qp[0] : 0.f;
  const float range = o.q8 ? qp[1] : 1.f;
  o.scale = (range == 0.f ? 1.f : range) / qmax;
  return o;
}
template <typename T, bool Q8>
__device__ __forceinline__ void q_unpack_raw(const u32x4& raw, const QOp& o, float* f) {
  constexpr int CH = ElemTraits<T>::kChunk;
  if (!Q8) { Chunk<T>::unpack(raw, f); return; }
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    const unsigned int w = e < 4 ? raw.x : raw.y;
    f[e] = q_round_to<T>(q_dequant((float)((w >> (8 * (e & 3))) & 255u), o.zp, o.scale));
  }
}

template <typename T, bool Y8>
__global__ __launch_bounds__(Q_NT) void quantize_kernel(const T* x, T* y, long long n, const float* zero_point,
                                                       const float* range, float qmax, const float* noise,
                                                       int stochastic, unsigned long long seed,
                                                       const unsigned long long* step, unsigned char* y8) {
  constexpr int CH = ElemTraits<T>::kChunk;
  // `step` (optional): a device counter the caller advances once per training step, mixed into the seed - a launch
  // replayed from a captured HIP graph (frozen kernel arguments) still draws fresh rounding noise every step
  if (step != nullptr) seed += step[0] * 0xD1B54A32D192ED03ull;
  const unsigned int key = q_noise_key(seed);
  const float zp = zero_point[0];
  const float scale = (range[0] == 0.f ? 1.f : range[0]) / qmax;
  const float inv = 1.0f / scale;
  const long long nch = n / CH;
  auto snap_chunk = [&](const u32x4& v, long long c) {
    float f[CH], nz[CH], lv[CH];
    Chunk<T>::unpack(v, f);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      const long long i = c * CH + e;
      nz[e] = noise != nullptr ? noise[i] : (stochastic ? q_hash_noise(key, (unsigned long long)i) : 0.f);
    }
    q_levels_chunk<CH>(f, nz, zp, scale, inv, qmax, lv);      // = q_level per element, division-free where provably exact
    if (Y8) q_st_levels<T>(y8, c, lv);      // (8-bit LEVELS: the consumer de-quantises on load, q_unpack_raw)
    else {
#pragma unroll
      for (int e = 0; e < CH; ++e) f[e] = q_dequant(lv[e], zp, scale);
      cn_st16((char*)y + c * 16, Chunk<T>::pack(f));
    }
  };
  const long long stride = (long long)gridDim.x * Q_NT;
  long long c = (long long)blockIdx.x * Q_NT + threadIdx.x;
  for (; c + stride < nch; c += 2 * stride) {   // two loads in flight per thread
    const u32x4 v0 = cn_ld16((const char*)x + c * 16);
    const u32x4 v1 = cn_ld16((const char*)x + (c + stride) * 16);
    snap_chunk(v0, c);
    snap_chunk(v1, c + stride);
  }
  for (; c < nch; c += stride) snap_chunk(cn_ld16((const char*)x + c * 16), c);
  // ragged tail (tensors that are not whole chunks: biases, the 3-channel image)
  for (long long i = nch * CH + (long long)blockIdx.x * Q_NT + threadIdx.x; i < n; i += (long long)gridDim.x * Q_NT) {
    const float nz = noise != nullptr ? noise[i] : (stochastic ? q_hash_noise(key, (unsigned long long)i) : 0.f);
    cn_store_elem<T>(y + i, q_snap(cn_load_elem<T>(x + i), zp, scale, qmax, nz));
  }
}

// q_grid rounded so that gridDim.x * Q_NT is a multiple of `cols` chunk columns (cols <= 2 * Q_NT in every model here)
static unsigned q_grid(long long work_items);
static unsigned q_grid_cols(long long work_items, int cols) {
  unsigned nb = q_grid(work_items);
  if (cols > Q_NT && cols % Q_NT == 0) { const unsigned m = (unsigned)(cols / Q_NT); nb = (nb + m - 1) / m * m; }
  return nb;
}
static unsigned q_grid(long long work_items) {
  long long nb = (work_items + Q_NT - 1) / Q_NT;
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

// y = quantise-dequantise(x) with the two scalars zero_point / range read on the device; 2^num_bits levels.
// Rounding noise: `noise` (fp32, one value per element, U(-0.5, 0.5)) when given, else the counter-based
// generator keyed by (seed, element index) when stochastic != 0, else none (deterministic rounding).
static int quantize_impl(const void* x, void* y, long long n, int dtype, const float* zero_point, const float* range,
                         int num_bits, const float* noise, int stochastic, unsigned long long seed,
                         const unsigned long long* step, void* stream_, unsigned char* y8 = nullptr) {
  if (n <= 0) return CN_OK;
  if (x == nullptr || (y == nullptr && y8 == nullptr) || zero_point == nullptr || range == nullptr || num_bits < 1 || num_bits > 23) { cn_set_error("quantize: bad arguments"); return CN_EINVAL; }
  if (((uintptr_t)x & 15) != 0 || ((uintptr_t)y & 15) != 0 || ((uintptr_t)y8 & 7) != 0) { cn_set_error("quantize: buffers must be 16-byte aligned"); return CN_EINVAL; }
  if (y8 != nullptr && (num_bits > 8 || n % (dtype == CN_F32 ? 4 : 8) != 0)) { cn_set_error("quantize_levels: <= 8 bits, whole chunks"); return CN_EINVAL; }
  hipStream_t stream = (hipStream_t)stream_;
  const float qmax = (float)((1 << num_bits) - 1);
#define QK(T, Y8, CHN) CN_LAUNCH((quantize_kernel<T, Y8>), dim3(q_grid((n + CHN - 1) / CHN)), dim3(Q_NT), stream, (const T*)x, (T*)y, n, \
                                zero_point, range, qmax, noise, stochastic, seed, step, y8)
  if (dtype == CN_BF16) { if (y8 != nullptr) QK(bf16_t, true, 8); else QK(bf16_t, false, 8); }
  else if (dtype == CN_F32) { if (y8 != nullptr) QK(float, true, 4); else QK(float, false, 4); }
#undef QK
  else { cn_set_error("quantize: bad dtype %d", dtype); return CN_EINVAL; }
  return cn_check_launch("quantize");
}
// cn_quantize_s storing the 8-bit LEVELS (one byte per element, element order kept) instead of the snapped values: the
// consumer de-quantises on load with the same (zero_point, range) - cn_rangebn_bwd_q8.  value = T(level * scale + zp).
extern "C" int cn_quantize_levels(const void* x, unsigned char* y8, long long n, int dtype, const float* zero_point,
                                  const float* range, int num_bits, const float* noise, int stochastic,
                                  unsigned long long seed, const unsigned long long* step_counter, void* stream_) {
  if (y8 == nullptr) { cn_set_error("quantize_levels: null output"); return CN_EINVAL; }
  return quantize_impl(x, nullptr, n, dtype, zero_point, range, num_bits, noise, stochastic, seed, step_counter, stream_, y8);
}
extern "C" int cn_quantize(const void* x, void* y, long long n, int dtype, const float* zero_point, const float* range,
                           int num_bits, const float* noise, int stochastic, unsigned long long seed, void* stream_) {
  return quantize_impl(x, y, n, dtype, zero_point, range, num_bits, noise, stochastic, seed, nullptr, stream_);
}
// cn_quantize whose generator seed is `seed` mixed with a device-resident step counter (advanced by cn_counter_inc once
// per training step): the launch can be captured into a HIP graph and still rounds with fresh noise on every replay.
extern "C" int cn_quantize_s(const void* x, void* y, long long n, int dtype, const float* zero_point, const float* range,
                             int num_bits, const float* noise, int stochastic, unsigned long long seed,
                             const unsigned long long* step_counter, void* stream_) {
  return quantize_impl(x, y, n, dtype, zero_point, range, num_bits, noise, stochastic, seed, step_counter, stream_);
}
__global__ void counter_inc_kernel(unsigned long long* p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1ull;
}
extern "C" int cn_counter_inc(unsigned long long* counter, void* stream_) {
  if (counter == nullptr) { cn_set_error("counter_inc: null"); return CN_EINVAL; }
  CN_LAUNCH(counter_inc_kernel, dim3(1), dim3(64), (hipStream_t)stream_, counter);
  return cn_check_launch("counter_inc");
}

// Per-row quantisation of an fp32 matrix [rows][row_len] with that row's own min / max (the weight
// quantiser: one row per output channel, quantize.py:201-203).  One workgroup per row.
__global__ __launch_bounds__(Q_NT) void quantize_rows_kernel(const float* x, float* y, int row_len, float qmax) {
  __shared__ float red[8];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* row = x + (size_t)r * row_len;
  float mn = INFINITY, mx = -INFINITY;
  for (int e = tid; e < row_len; e += Q_NT) { mn = fminf(mn, row[e]); mx = fmaxf(mx, row[e]); }
  q_block_minmax(mn, mx, red);
  float range = mx - mn;
  if (range == 0.f) range = 1.f;
  const float scale = range / qmax;
  for (int e = tid; e < row_len; e += Q_NT) y[(size_t)r * row_len + e] = q_snap(row[e], mn, scale, qmax, 0.f);
}

extern "C" int cn_quantize_rows(const float* x, float* y, int rows, int row_len, int num_bits, void* stream) {
  if (rows <= 0 || row_len <= 0) return CN_OK;
  if (x == nullptr || y == nullptr || num_bits < 1 || num_bits > 23) { cn_set_error("quantize_rows: bad arguments"); return CN_EINVAL; }
  CN_LAUNCH(quantize_rows_kernel, dim3((unsigned)rows), dim3(Q_NT), (hipStream_t)stream, x, y, row_len,
            (float)((1 << num_bits) - 1));
  return cn_check_launch("quantize_rows");
}

// The same for EVERY filter of a model in one launch (round 6: 54 launches of the kernel above + 54 single-layer
// weight preparations per ResNet-50 step became one launch + the arena's two): x / y are the flat fp32 parameter arena and
// its quantised shadow, rowtab holds {element offset, row length, 2^bits - 1} per output channel of every layer.
__global__ __launch_bounds__(Q_NT) void quantize_rows_multi_kernel(const float* x, float* y, const long long* rowtab) {
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const long long off = rowtab[3 * (size_t)blockIdx.x];
  const int row_len = (int)rowtab[3 * (size_t)blockIdx.x + 1];
  const float qmax = (float)rowtab[3 * (size_t)blockIdx.x + 2];
  const float* row = x + off;
  float mn = INFINITY, mx = -INFINITY;
  for (int e = tid; e < row_len; e += Q_NT) { mn = fminf(mn, row[e]); mx = fmaxf(mx, row[e]); }
  q_block_minmax(mn, mx, red);
  float range = mx - mn;
  if (range == 0.f) range = 1.f;
  const float scale = range / qmax;
  for (int e = tid; e < row_len; e += Q_NT) y[off + e] = q_snap(row[e], mn, scale, qmax, 0.f);
}

extern "C" int cn_quantize_rows_multi(const float* x, float* y, const long long* rowtab, int rows, void* stream) {
  if (rows <= 0) return CN_OK;
  if (x == nullptr || y == nullptr || rowtab == nullptr) { cn_set_error("quantize_rows_multi: bad arguments"); return CN_EINVAL; }
  CN_LAUNCH(quantize_rows_multi_kernel, dim3((unsigned)rows), dim3(Q_NT), (hipStream_t)stream, x, y, rowtab);
  return cn_check_launch("quantize_rows_multi");
}

// Input quantiser folded into RangeBN's STATISTICS pass (round 4): RangeBN's QuantMeasure (quantize.py:270,308) snaps its
// input to the 8-bit grid before anything else; with xqp = [zero_point, range] (cn_qparams' output) rangebn_stats_kernel
// reads the RAW convolution output, snaps each element on load - quantize_kernel's arithmetic, rounded to T - takes its
// statistics of the snapped values and STORES them (qx_out): the separate quantiser pass (one read of every RangeBN input)
// disappears, the apply pass and the backward kernels read qx_out.  (Round 3 snapped on load in all three kernels and
// never stored: the division three times per element made them VALU-bound and the step slower.)  xqp == nullptr: x is
// already quantised.
struct RbnSnap {
  float zp, scale, qmax, inv;
  int on;
};
__device__ __forceinline__ RbnSnap rbn_snap_make(const float* xqp, float qmax) {
  RbnSnap q;
  q.on = xqp != nullptr;
  q.zp = q.on ? xqp[0] : 0.f;
  const float range = q.on ? xqp[1] : 1.f;
  q.scale = (range == 0.f ? 1.f : range) / qmax;
  q.inv = 1.0f / q.scale;
  q.qmax = qmax;
  return q;
}
template <typename T>
__device__ __forceinline__ void rbn_snap_chunk(const RbnSnap& q, float* f) {
  constexpr int CH = ElemTraits<T>::kChunk;
  if (!q.on) return;
  float nz[CH], lv[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) nz[e] = 0.f;
  q_levels_chunk<CH>(f, nz, q.zp, q.scale, q.inv, q.qmax, lv);
#pragma unroll
  for (int e = 0; e < CH; ++e) f[e] = q_dequant(lv[e], q.zp, q.scale);
  Chunk<T>::unpack(Chunk<T>::pack(f), f);   // as stored in T
}

// ------------------------------------------------------------------------------------------------ RangeBN
// Activations NHWC [M][C]; the reference's view(C, chunks, M/chunks) of the (b, h, w)-ordered values is the
// split of the M pixels into `chunks` consecutive ranges.  Each range is cut further into `sub` slices so the
// launch fills the chip; slices are merged in order, so ties resolve to the FIRST maximal / minimal element
// exactly as torch.max / torch.min do.
struct RbnPartial {   // one per (chunk, slice, channel)
  float mx, mn, sum;
  int imx, imn;
};

template <typename T, bool Q8OUT>
__global__ __launch_bounds__(Q_NT) void rangebn_stats_kernel(const T* x, int M, int C, int chunks, int sub, int cols,
                                                            RbnPartial* part, const float* xqp, float qmax, T* qx_out,
                                                            unsigned char* q8_out) {
  constexpr int CH = ElemTraits<T>::kChunk;
  __shared__ float s_mx[Q_NT * CH], s_mn[Q_NT * CH], s_sum[Q_NT * CH];
  __shared__ int s_imx[Q_NT * CH], s_imn[Q_NT * CH];
  const int tid = threadIdx.x;
  const int lanes = Q_NT / cols;            // pixel lanes per chunk column
  const int col = tid % cols, lane = tid / cols;
  const int CC = C / CH;
  const int cc = blockIdx.x * cols + col;   // chunk column of this thread
  const int slice = blockIdx.y;             // chunk * sub + s
  const int chunk = slice / sub, s = slice - chunk * sub;
  const int L = M / chunks;
  const int per = (L + sub - 1) / sub;
  const int p0 = chunk * L + s * per;
  int p1 = p0 + per;
  if (p1 > (chunk + 1) * L) p1 = (chunk + 1) * L;
  float mx[CH], mn[CH], sm[CH];
  int imx[CH], imn[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { mx[e] = -INFINITY; mn[e] = INFINITY; sm[e] = 0.f; imx[e] = 0x7fffffff; imn[e] = 0x7fffffff; }
  const RbnSnap snap = rbn_snap_make(xqp, qmax);
  if (cc < CC) {
    auto visit = [&](const u32x4& v, int p) {
      float f[CH];
      Chunk<T>::unpack(v, f);
      if (Q8OUT) {        // the snapped input kept as 8-bit levels; the statistics see the values a load gives back
        float lv[CH], nz[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) nz[e] = 0.f;
        q_levels_chunk<CH>(f, nz, snap.zp, snap.scale, snap.inv, snap.qmax, lv);
#pragma unroll
        for (int e = 0; e < CH; ++e) f[e] = q_round_to<T>(q_dequant(lv[e], snap.zp, snap.scale));
        q_st_levels<T>(q8_out, (long long)p * CC + cc, lv);
      } else {
        rbn_snap_chunk<T>(snap, f);
        if (qx_out != nullptr) cn_st16((char*)qx_out + ((size_t)p * CC + cc) * 16, Chunk<T>::pack(f));   // (every element is visited once)
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        if (f[e] > mx[e]) { mx[e] = f[e]; imx[e] = p; }
        if (f[e] < mn[e]) { mn[e] = f[e]; imn[e] = p; }
        sm[e] += f[e];
      }
    };
    int p = p0 + lane;
    for (; p + 3 * lanes < p1; p += 4 * lanes) {   // four loads in flight, visited in increasing pixel order
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = cn_ld16((const char*)x + ((size_t)(p + u * lanes) * CC + cc) * 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) visit(v[u], p + u * lanes);
    }
    for (; p < p1; p += lanes) visit(cn_ld16((const char*)x + ((size_t)p * CC + cc) * 16), p);
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    s_mx[tid * CH + e] = mx[e]; s_mn[tid * CH + e] = mn[e]; s_sum[tid * CH + e] = sm[e];
    s_imx[tid * CH + e] = imx[e]; s_imn[tid * CH + e] = imn[e];
  }
  __syncthreads();
  if (lane == 0 && cc < CC) {
    for (int e = 0; e < CH; ++e) {
      float bmx = -INFINITY, bmn = INFINITY, bs = 0.f;
      int bimx = 0x7fffffff, bimn = 0x7fffffff;
      for (int l = 0; l < lanes; ++l) {   // lanes hold interleaved pixels: ties go to the smaller index
        const int t = (l * cols + col) * CH + e;
        if (s_mx[t] > bmx || (s_mx[t] == bmx && s_imx[t] < bimx)) { bmx = s_mx[t]; bimx = s_imx[t]; }
        if (s_mn[t] < bmn || (s_mn[t] == bmn && s_imn[t] < bimn)) { bmn = s_mn[t]; bimn = s_imn[t]; }
        bs += s_sum[t];
      }
      RbnPartial o;
      o.mx = bmx; o.mn = bmn; o.sum = bs; o.imx = bimx; o.imn = bimn;
      part[(size_t)slice * C + cc * CH + e] = o;
    }
  }
}

// Per channel: merge the slices, then mean, scale = (mean of chunk maxima - mean of chunk minima) * scale_fix,
// running statistics (running = running * momentum + new * (1 - momentum), quantize.py:300-305).
// stats[c] = mean, stats[C + c] = scale + eps;  arg[c][2*chunks] = pixel index of each chunk's first max / min.
__global__ __launch_bounds__(Q_NT) void rangebn_finalize_kernel(const RbnPartial* part, int M, int C, int chunks, int sub,
                                                               float scale_fix, float eps, float momentum,
                                                               float* running_mean, float* running_var, float* stats,
                                                               int* arg) {
  const int c = blockIdx.x * Q_NT + threadIdx.x;
  if (c >= C) return;
  double total = 0.0;
  float smx = 0.f, smn = 0.f;
  for (int j = 0; j < chunks; ++j) {
    float bmx = -INFINITY, bmn = INFINITY;
    int bimx = 0, bimn = 0;
    for (int s = 0; s < sub; ++s) {
      const RbnPartial o = part[(size_t)(j * sub + s) * C + c];
      if (o.mx > bmx) { bmx = o.mx; bimx = o.imx; }
      if (o.mn < bmn) { bmn = o.mn; bimn = o.imn; }
      total += (double)o.sum;
    }
    smx += bmx;
    smn += bmn;
    arg[(size_t)c * 2 * chunks + j] = bimx;
    arg[(size_t)c * 2 * chunks + chunks + j] = bimn;
  }
  const float mean = (float)(total / (double)M);
  const float scale = (smx / (float)chunks - smn / (float)chunks) * scale_fix;
  stats[c] = mean;
  stats[C + c] = scale + eps;
  if (running_mean != nullptr) {
    const float keep = 1.f - momentum;
    running_mean[c] = running_mean[c] * momentum + mean * keep;
    running_var[c] = running_var[c] * momentum + scale * keep;
  }
}

// The same result from a workgroup of RF_C channels x RF_T threads per channel (chunks <= RF_T / 2): the one-thread-
// per-channel walk above is chunks * sub (400 on the 56x56 layers) dependent L2 round trips, ~0.2 ms per launch.
// Thread t of a channel merges half h = t / chunks of the slices of chunk j = t % chunks, eight partials requested
// at a time; the halves are merged in order (ties keep the first, i.e. smaller, index), the chunk maxima / minima
// are then summed in chunk order exactly as above.
#define RF_C 8
#define RF_T 32
__global__ __launch_bounds__(RF_C * RF_T) void rangebn_finalize_par_kernel(const RbnPartial* part, int M, int C, int chunks,
                                                                          int sub, float scale_fix, float eps,
                                                                          float momentum, float* running_mean,
                                                                          float* running_var, float* stats, int* arg) {
  __shared__ float h_mx[RF_C][RF_T], h_mn[RF_C][RF_T];
  __shared__ int h_imx[RF_C][RF_T], h_imn[RF_C][RF_T];
  __shared__ double h_sum[RF_C][RF_T];
  const int lc = threadIdx.x % RF_C, t = threadIdx.x / RF_C;
  const int c = blockIdx.x * RF_C + lc;
  const int j = t % chunks, h = t / chunks;          // t < 2 * chunks <= RF_T does work
  const int half = (sub + 1) / 2;
  float bmx = -INFINITY, bmn = INFINITY;
  int bimx = 0, bimn = 0;
  double tot = 0.0;
  if (c < C && t < 2 * chunks) {
    const int s0 = h * half, s1 = (s0 + half < sub) ? s0 + half : sub;
    for (int s = s0; s < s1; s += 8) {
      RbnPartial o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int su = s + u < s1 ? s + u : s1 - 1;      // clamped: a repeated partial changes neither max nor min
        o[u] = part[(size_t)(j * sub + su) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s + u >= s1) break;
        if (o[u].mx > bmx) { bmx = o[u].mx; bimx = o[u].imx; }
        if (o[u].mn < bmn) { bmn = o[u].mn; bimn = o[u].imn; }
        tot += (double)o[u].sum;
      }
    }
  }
  h_mx[lc][t] = bmx; h_mn[lc][t] = bmn; h_imx[lc][t] = bimx; h_imn[lc][t] = bimn; h_sum[lc][t] = tot;
  __syncthreads();
  if (c < C && t < chunks) {          // second half of chunk t merged behind the first (strict compares: ties keep the first)
    const int u = t + chunks;
    if (h_mx[lc][u] > bmx) { bmx = h_mx[lc][u]; bimx = h_imx[lc][u]; }
    if (h_mn[lc][u] < bmn) { bmn = h_mn[lc][u]; bimn = h_imn[lc][u]; }
    arg[(size_t)c * 2 * chunks + t] = bimx;
    arg[(size_t)c * 2 * chunks + chunks + t] = bimn;
  }
  __syncthreads();
  if (c < C && t < chunks) { h_mx[lc][t] = bmx; h_mn[lc][t] = bmn; }
  __syncthreads();
  if (c >= C || t != 0) return;
  double total = 0.0;
  float smx = 0.f, smn = 0.f;
  for (int k = 0; k < chunks; ++k) { smx += h_mx[lc][k]; smn += h_mn[lc][k]; }
  for (int k = 0; k < 2 * chunks; ++k) total += h_sum[lc][k];
  const float mean = (float)(total / (double)M);
  const float scale = (smx / (float)chunks - smn / (float)chunks) * scale_fix;
  stats[c] = mean;
  stats[C + c] = scale + eps;
  if (running_mean != nullptr) {
    const float keep = 1.f - momentum;
    running_mean[c] = running_mean[c] * momentum + mean * keep;
    running_var[c] = running_var[c] * momentum + scale * keep;
  }
}

// inference statistics: stats = {running_mean, running_var + eps}
__global__ __launch_bounds__(Q_NT) void rangebn_infer_stats_kernel(const float* running_mean, const float* running_var,
                                                                  float eps, int C, float* stats) {
  const int c = blockIdx.x * Q_NT + threadIdx.x;
  if (c >= C) return;
  stats[c] = running_mean[c];
  stats[C + c] = running_var[c] + eps;
}

// z = act(((x - mean) / (scale + eps)) * w + b [+ residual])   (quantize.py:312-325, then the block's add / ReLU)
// Grid (blocks per row, rows): `rows` consecutive equal parts of the nch chunks (the samples of the batch; 1 = the whole
// tensor).  mm_partial (optional): [rows][gridDim.x][2] = {min, max} of the values AS STORED per block - the per-sample
// extremes the next activation quantiser needs (QuantMeasure, quantize.py:158-182), so its min / max pass over z disappears.
template <typename T, bool X8>
__global__ __launch_bounds__(Q_NT) void rangebn_apply_kernel(const T* x, const T* residual, T* z, const float* stats,
                                                            const float* weight, const float* bias, long long nch,
                                                            int C, int relu, float* mm_partial, const float* x8qp, float x8qmax) {
  constexpr int CH = ElemTraits<T>::kChunk;
  __shared__ float red[8];
  const QOp xop = q_op_make(x8qp, x8qmax);      // x8qp != nullptr: x holds 8-bit levels of that grid
  const int CC = C / CH;
  const long long rch = nch / gridDim.y;                       // chunks per row (a multiple of CC)
  const long long row0 = (long long)blockIdx.y * rch, row1 = row0 + rch;
  const long long stride = (long long)gridDim.x * Q_NT;
  // the grid stride is a multiple of the chunk columns in every launch the host makes (q_grid_cols): a thread stays on
  // one chunk column, so its 4 x CH coefficients are loaded once (they were re-read from L1 for every chunk: 32 loads
  // beside the one 16-byte load that carries the data)
  const bool fixed_col = stride % CC == 0;
  float mean[CH], den[CH], w[CH], b[CH];
  float mn = INFINITY, mx = -INFINITY;
  auto load_coef = [&](long long id) {
    const int c0 = (int)(id % CC) * CH;
#pragma unroll
    for (int e = 0; e < CH; ++e) { mean[e] = stats[c0 + e]; den[e] = stats[C + c0 + e]; w[e] = weight[c0 + e]; b[e] = bias[c0 + e]; }
  };
  auto apply = [&](const u32x4& vx, const u32x4& vr, long long id) {
    float f[CH], r[CH];
    q_unpack_raw<T, X8>(vx, xop, f);
    if (residual != nullptr) Chunk<T>::unpack(vr, r);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      float v = (f[e] - mean[e]) / den[e];
      v = v * w[e];
      v = v + b[e];
      if (residual != nullptr) v = v + r[e];
      if (relu) v = v > 0.f ? v : 0.f;
      f[e] = v;
    }
    cn_st16((char*)z + id * 16, Chunk<T>::pack(f));
    if (mm_partial != nullptr) {   // (of the fp32 values: rounding to T is monotonic, so the extremes are rounded once at the end)
#pragma unroll
      for (int e = 0; e < CH; ++e) { mn = fminf(mn, f[e]); mx = fmaxf(mx, f[e]); }
    }
  };
  long long id = row0 + (long long)blockIdx.x * Q_NT + threadIdx.x;
  if (id < row1) load_coef(id);
  if (fixed_col) {
    for (; id + stride < row1; id += 2 * stride) {   // two (four with a residual) loads in flight
      const u32x4 v0 = q_ld_raw<T, X8>(x, id), v1 = q_ld_raw<T, X8>(x, id + stride);
      u32x4 r0 = cn_zero16(), r1 = cn_zero16();
      if (residual != nullptr) { r0 = cn_ld16((const char*)residual + id * 16); r1 = cn_ld16((const char*)residual + (id + stride) * 16); }
      apply(v0, r0, id);
      apply(v1, r1, id + stride);
    }
  }
  for (; id < row1; id += stride) {
    if (!fixed_col) load_coef(id);
    apply(q_ld_raw<T, X8>(x, id), residual != nullptr ? cn_ld16((const char*)residual + id * 16) : cn_zero16(), id);
  }
  if (mm_partial != nullptr) {
    q_block_minmax(mn, mx, red);
    if (threadIdx.x == 0) {
      mm_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2] = q_round_to<T>(mn);
      mm_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = q_round_to<T>(mx);
    }
  }
}

// Grid of a per-row elementwise pass: blocks per row (their stride a multiple of `cols` chunk columns), <= 8192 blocks in all
static unsigned q_grid_rows(long long nch, int rows, int cols) {
  if (rows < 1) rows = 1;
  // >= 8 chunks per thread where the row is long enough: a thread's set-up (its chunk column's coefficients, the routing
  // table entries) is 24-40 loads, as much as the data of several chunks
  long long bpr = (nch / rows + 8 * Q_NT - 1) / (8 * Q_NT);
  const long long cap = 8192 / rows > 1 ? 8192 / rows : 1;
  if (bpr > cap) bpr = cap;
  if (bpr < 1) bpr = 1;
  if (cols > Q_NT && cols % Q_NT == 0) { const long long m = cols / Q_NT; bpr = (bpr + m - 1) / m * m; }
  return (unsigned)bpr;
}

extern "C" size_t cn_rangebn_workspace(int M, int C, int chunks) {
  if (M <= 0 || C <= 0 || chunks <= 0) return 0;
  int sub = 64;   // upper bound over the tunable slice lengths (rbn_sub)
  const size_t stats = (size_t)chunks * sub * C * sizeof(RbnPartial);
  // backward: >= 256 pixels per partial row, the coefficients, and room behind them for the per-sample min / max partials
  // of the fused form when they do not fit in the (then dead) partial rows - few pixels per sample, few channels, a large
  // batch (rows x blocks-per-row <= 8192 + one rounding step per row: q_grid_rows)
  const size_t mmtail = (size_t)(8192 + 8192 * (size_t)(1 + C / Q_NT)) * 2;
  const size_t bwd = ((size_t)((M + 255) / 256) * 2 * C + 3 * (size_t)C + mmtail) * sizeof(float);
  return stats > bwd ? stats : bwd;
}
static int rbn_sub(int M, int chunks) {
  const int px = cn_get_option("rbn_slice_px", 512);   // pixels per slice of a chunk (knob)
  int sub = (M / chunks + px - 1) / px;
  if (sub < 1) sub = 1;
  if (sub > 64) sub = 64;
  return sub;
}
static int rbn_cols(int CC) {
  int cols = 1;
  while (cols * 2 <= CC && cols < 32) cols *= 2;
  return cols;
}

// RangeBN forward on an already input-quantised x [M][C].  training != 0: batch statistics (mean; scale from
// `chunks` chunk-wise max - min, M % chunks == 0 as the reference's view() requires), running statistics
// updated, stats[2C] = {mean | scale + eps} and arg[C][2*chunks] saved for the backward pass.  training == 0:
// running statistics.  z = act(affine(normalised x) [+ residual]).
// xqp / qx_out (training): x is the RAW input, snapped by the statistics pass and stored to qx_out (see RbnSnap).
// mm_rows / z_minmax: per-row {min, max} of the stored z over mm_rows equal row groups of the M pixels (the batch samples).
static int rangebn_fwd_impl(const void* x, const void* residual, void* z, const float* weight, const float* bias,
                              float* running_mean, float* running_var, float momentum, float eps, int chunks,
                              float scale_fix, float* stats, int* arg, int M, int C, int relu, int training, int dtype,
                              float* ws, size_t ws_bytes, void* stream_, const float* xqp, int x_bits, void* qx_out,
                              int mm_rows, float* z_minmax, unsigned char* qx8_out = nullptr) {
  // qx8_out (with xqp): the snapped input is kept as 8-bit LEVELS there instead of values in qx_out; the apply pass
  // de-quantises them on load (same numbers; 1 byte instead of 2 / 4 per element written once and read twice)
  if (qx8_out != nullptr && (xqp == nullptr || x_bits > 8 || !training)) { cn_set_error("rangebn_fwd: 8-bit level storage needs the folded <= 8-bit input quantiser"); return CN_EINVAL; }
  if (qx8_out != nullptr) qx_out = qx8_out;     // (non-null marker for the checks below; never written as values)
  const float qmax = (float)((1 << (x_bits > 0 ? x_bits : 8)) - 1);
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = dtype == CN_BF16 ? 8 : 4;
  if (dtype != CN_BF16 && dtype != CN_F32) { cn_set_error("rangebn_fwd: bad dtype"); return CN_EINVAL; }
  if (M <= 0 || C <= 0 || C % CH != 0) { cn_set_error("rangebn_fwd: M=%d C=%d (C must be a multiple of %d)", M, C, CH); return CN_ESHAPE; }
  if (x == nullptr || z == nullptr || weight == nullptr || bias == nullptr || stats == nullptr) { cn_set_error("rangebn_fwd: null operand"); return CN_EINVAL; }
  if (xqp != nullptr && (!training || qx_out == nullptr)) { cn_set_error("rangebn_fwd: the folded input quantiser needs training statistics and qx_out"); return CN_EINVAL; }
  if (z_minmax != nullptr && (mm_rows < 1 || M % mm_rows != 0 || mm_rows > 8192)) { cn_set_error("rangebn_fwd: %d pixels do not split into %d rows", M, mm_rows); return CN_ESHAPE; }
  const int CC = C / CH;
  if (training) {
    if (chunks <= 0 || M % chunks != 0) { cn_set_error("rangebn_fwd: %d values per channel do not split into %d chunks", M, chunks); return CN_ESHAPE; }
    if (arg == nullptr) { cn_set_error("rangebn_fwd: training needs the arg buffer"); return CN_EINVAL; }
    if (ws == nullptr || ws_bytes < cn_rangebn_workspace(M, C, chunks)) { cn_set_error("rangebn_fwd: workspace too small"); return CN_EWORKSPACE; }
    const int sub = rbn_sub(M, chunks), cols = rbn_cols(CC);
    dim3 grid((unsigned)((CC + cols - 1) / cols), (unsigned)(chunks * sub));
#define RSK(T, Q8) CN_LAUNCH((rangebn_stats_kernel<T, Q8>), grid, dim3(Q_NT), stream, (const T*)x, M, C, chunks, sub, cols, (RbnPartial*)ws, xqp, qmax, (T*)qx_out, qx8_out)
    if (dtype == CN_BF16) { if (qx8_out != nullptr) RSK(bf16_t, true); else RSK(bf16_t, false); }
    else { if (qx8_out != nullptr) RSK(float, true); else RSK(float, false); }
#undef RSK
    if (2 * chunks <= RF_T)
      CN_LAUNCH(rangebn_finalize_par_kernel, dim3((unsigned)((C + RF_C - 1) / RF_C)), dim3(RF_C * RF_T), stream, (const RbnPartial*)ws, M,
              C, chunks, sub, scale_fix, eps, momentum, running_mean, running_var, stats, arg);
    else
      CN_LAUNCH(rangebn_finalize_kernel, dim3((unsigned)((C + Q_NT - 1) / Q_NT)), dim3(Q_NT), stream, (const RbnPartial*)ws, M,
              C, chunks, sub, scale_fix, eps, momentum, running_mean, running_var, stats, arg);
  } else {
    if (running_mean == nullptr || running_var == nullptr) { cn_set_error("rangebn_fwd: inference needs the running statistics"); return CN_EINVAL; }
    CN_LAUNCH(rangebn_infer_stats_kernel, dim3((unsigned)((C + Q_NT - 1) / Q_NT)), dim3(Q_NT), stream,
              (const float*)running_mean, (const float*)running_var, eps, C, stats);
  }
  const long long nch = (long long)M * CC;
  const void* xa = xqp != nullptr ? qx_out : x;       // the apply pass reads the snapped copy the statistics pass stored
  const int rows = z_minmax != nullptr ? mm_rows : 1;
  const unsigned bpr = z_minmax != nullptr ? q_grid_rows(nch, rows, CC) : q_grid_cols(nch, CC);
  float* mmp = nullptr;
  if (z_minmax != nullptr) {     // (the statistics partials in ws are dead once finalize has run, in stream order)
    if (ws == nullptr || ws_bytes < (size_t)rows * bpr * 2 * sizeof(float)) { cn_set_error("rangebn_fwd: workspace too small for the min / max partials"); return CN_EWORKSPACE; }
    mmp = ws;
  }
#define RAK(T, X8) CN_LAUNCH((rangebn_apply_kernel<T, X8>), dim3(bpr, (unsigned)rows), dim3(Q_NT), stream, (const T*)xa, (const T*)residual, \
                             (T*)z, (const float*)stats, weight, bias, nch, C, relu, mmp, X8 ? xqp : (const float*)nullptr, qmax)
  if (dtype == CN_BF16) { if (qx8_out != nullptr) RAK(bf16_t, true); else RAK(bf16_t, false); }
  else { if (qx8_out != nullptr) RAK(float, true); else RAK(float, false); }
#undef RAK
  if (z_minmax != nullptr)
    CN_LAUNCH(minmax_final_kernel, dim3((unsigned)((rows + Q_NT - 1) / Q_NT)), dim3(Q_NT), stream, (const float*)mmp, rows, (int)bpr, z_minmax, (float*)nullptr);
  return cn_check_launch("rangebn_fwd");
}

extern "C" int cn_rangebn_fwd(const void* x, const void* residual, void* z, const float* weight, const float* bias,
                              float* running_mean, float* running_var, float momentum, float eps, int chunks,
                              float scale_fix, float* stats, int* arg, int M, int C, int relu, int training, int dtype,
                              float* ws, size_t ws_bytes, void* stream_) {
  return rangebn_fwd_impl(x, residual, z, weight, bias, running_mean, running_var, momentum, eps, chunks, scale_fix, stats,
                          arg, M, C, relu, training, dtype, ws, ws_bytes, stream_, nullptr, 8, nullptr, 0, nullptr);
}
// Training forward on the RAW input with the producer-side fusions of round 4:
//   x_qparams = [zero_point, range] of the x_bits-bit input quantiser (cn_qparams): the statistics pass snaps every element
//     on load and stores the snapped tensor to qx_out (what cn_quantize would have written; the backward pass takes it);
//   z_minmax (optional, mm_rows rows = batch samples): per-sample {min, max} of the stored z for the next quantiser.
// Same qx, z, stats, arg bits as cn_quantize followed by cn_rangebn_fwd; z_minmax = cn_minmax_rows(z, mm_rows).
extern "C" int cn_rangebn_fwd_q(const void* x, const float* x_qparams, int x_bits, void* qx_out, const void* residual, void* z,
                                const float* weight, const float* bias, float* running_mean, float* running_var,
                                float momentum, float eps, int chunks, float scale_fix, float* stats, int* arg, int M,
                                int C, int relu, int dtype, int mm_rows, float* z_minmax, float* ws, size_t ws_bytes,
                                void* stream_) {
  if (x_qparams == nullptr || x_bits < 1 || x_bits > 23 || qx_out == nullptr) { cn_set_error("rangebn_fwd_q: needs the input quantiser's parameters and qx_out"); return CN_EINVAL; }
  return rangebn_fwd_impl(x, residual, z, weight, bias, running_mean, running_var, momentum, eps, chunks, scale_fix, stats,
                          arg, M, C, relu, 1, dtype, ws, ws_bytes, stream_, x_qparams, x_bits, qx_out, mm_rows, z_minmax);
}

// cn_rangebn_fwd_q with the snapped input kept as 8-bit LEVELS (qx8_out: one byte per element) instead of values: the same
// z / stats / arg / z_minmax bits; the backward pass takes the levels and x_qparams (cn_rangebn_bwd_q8).  x_bits <= 8.
extern "C" int cn_rangebn_fwd_q8(const void* x, const float* x_qparams, int x_bits, unsigned char* qx8_out, void* z,
                                 const float* weight, const float* bias, float* running_mean, float* running_var,
                                 float momentum, float eps, int chunks, float scale_fix, float* stats, int* arg, int M,
                                 int C, int relu, int dtype, int mm_rows, float* z_minmax, float* ws, size_t ws_bytes,
                                 void* stream_) {
  if (x_qparams == nullptr || x_bits < 1 || x_bits > 8 || qx8_out == nullptr) { cn_set_error("rangebn_fwd_q8: needs the <= 8-bit input quantiser's parameters and qx8_out"); return CN_EINVAL; }
  return rangebn_fwd_impl(x, nullptr, z, weight, bias, running_mean, running_var, momentum, eps, chunks, scale_fix, stats,
                          arg, M, C, relu, 1, dtype, ws, ws_bytes, stream_, x_qparams, x_bits, nullptr, mm_rows, z_minmax, qx8_out);
}

// ---- backward.  g = the (already quantised) gradient of the RangeBN output, x = its quantised input.
//   S1 = sum g, S2 = sum g * (x - mean), r = 1 / (scale + eps):
//   dbias += S1;  dweight += r * S2;  dx = g * (w r) - (w r) S1 / M
//   dL/dscale = -w r^2 S2 reaches x through the chunk maxima / minima: scale = fix/chunks * sum_j (max_j - min_j)
//   => dx[first argmax of chunk j] += dL/dscale * fix / chunks,  dx[first argmin of chunk j] -= the same.
template <typename T, bool G8, bool X8>
__global__ __launch_bounds__(Q_NT) void rangebn_bwd_reduce_kernel(const T* g, const T* x, const float* stats, int M, int C,
                                                                 int rows_per, int cols, float* partial, const float* g8qp,
                                                                 float g8qmax, const float* x8qp, float x8qmax) {
  constexpr int CH = ElemTraits<T>::kChunk;
  __shared__ float s1[Q_NT * CH], s2[Q_NT * CH];
  const QOp gop = q_op_make(g8qp, g8qmax), xop = q_op_make(x8qp, x8qmax);   // operands stored as 8-bit levels (or values)
  const int tid = threadIdx.x, CC = C / CH;
  const int lanes = Q_NT / cols, col = tid % cols, lane = tid / cols;
  const int cc = blockIdx.x * cols + col;
  const int p0 = blockIdx.y * rows_per;
  const int p1 = p0 + rows_per < M ? p0 + rows_per : M;
  float a1[CH], a2[CH], mean[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { a1[e] = 0.f; a2[e] = 0.f; mean[e] = cc < CC ? stats[cc * CH + e] : 0.f; }
  if (cc < CC) {
    auto visit = [&](const u32x4& vg, const u32x4& vx) {
      float fg[CH], fx[CH];
      q_unpack_raw<T, G8>(vg, gop, fg);
      q_unpack_raw<T, X8>(vx, xop, fx);
#pragma unroll
      for (int e = 0; e < CH; ++e) { a1[e] += fg[e]; a2[e] += fg[e] * (fx[e] - mean[e]); }
    };
    int p = p0 + lane;
    for (; p + 3 * lanes < p1; p += 4 * lanes) {   // eight loads in flight
      u32x4 vg[4], vx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        vg[u] = q_ld_raw<T, G8>(g, (long long)(p + u * lanes) * CC + cc);
        vx[u] = q_ld_raw<T, X8>(x, (long long)(p + u * lanes) * CC + cc);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) visit(vg[u], vx[u]);
    }
    for (; p < p1; p += lanes)
      visit(q_ld_raw<T, G8>(g, (long long)p * CC + cc), q_ld_raw<T, X8>(x, (long long)p * CC + cc));
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) { s1[tid * CH + e] = a1[e]; s2[tid * CH + e] = a2[e]; }
  __syncthreads();
  if (lane == 0 && cc < CC) {
    for (int e = 0; e < CH; ++e) {
      float t1 = 0.f, t2 = 0.f;
      for (int l = 0; l < lanes; ++l) { t1 += s1[(l * cols + col) * CH + e]; t2 += s2[(l * cols + col) * CH + e]; }
      partial[(size_t)blockIdx.y * 2 * C + cc * CH + e] = t1;
      partial[(size_t)blockIdx.y * 2 * C + C + cc * CH + e] = t2;
    }
  }
}

// coef[c] = w r, coef[C + c] = -(w r) S1 / M, coef[2C + c] = -w r^2 S2 * fix / chunks; parameter gradients accumulated.
// RF_C channels x RF_T row slices per workgroup, sixteen partial rows requested at a time, fixed-order combine (a single
// thread per channel walked all M / 1024 rows - 784 on the 56x56 layers - one dependent round trip after the other).
__global__ __launch_bounds__(RF_C * RF_T) void rangebn_bwd_finalize_kernel(const float* partial, int rows, int M, int C,
                                                                          const float* weight, const float* stats, float route,
                                                                          float* dweight, float* dbias, float* coef) {
  __shared__ double r1[RF_T][RF_C], r2[RF_T][RF_C];
  const int lc = threadIdx.x % RF_C, t = threadIdx.x / RF_C;
  const int c = blockIdx.x * RF_C + lc;
  double S1 = 0.0, S2 = 0.0;
  if (c < C) {
    for (int r = t; r < rows; r += 16 * RF_T) {
      float a[16], b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int ru = r + u * RF_T;
        const int rc = ru < rows ? ru : rows - 1;
        const float av = partial[(size_t)rc * 2 * C + c], bv = partial[(size_t)rc * 2 * C + C + c];
        a[u] = ru < rows ? av : 0.f;
        b[u] = ru < rows ? bv : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) { S1 += (double)a[u]; S2 += (double)b[u]; }
    }
  }
  r1[t][lc] = S1;
  r2[t][lc] = S2;
  __syncthreads();
  if (c >= C || t != 0) return;
  S1 = 0.0;
  S2 = 0.0;
  for (int k = 0; k < RF_T; ++k) { S1 += r1[k][lc]; S2 += r2[k][lc]; }
  const float r = 1.f / stats[C + c];
  const float w = weight[c];
  const float s1 = (float)S1, s2 = (float)S2;
  dbias[c] += s1;
  dweight[c] += r * s2;
  coef[c] = w * r;
  coef[C + c] = -(w * r * s1) / (float)M;
  coef[2 * C + c] = (-w * r * r * s2) * route;
}

// dx = g * a[c] + b[c], rounded to T, then the routing of dL/dscale: the chunk's first maximum gets +d[c], its first minimum
// -d[c] (each a separate rounded update, +d first: what a routing pass over the stored dx does - distinct chunks and
// distinct channels never share an element; maximum == minimum only in a constant chunk, where the two cancel).
// arg == nullptr: no routing here (rangebn_bwd_route_kernel follows).  Grid (blocks per row, rows) and mm_partial as
// rangebn_apply_kernel: per-sample {min, max} of the FINAL dx for the gradient quantiser of the convolution in front
// (UniformQuantizeGrad, quantize.py:101-112), so its min / max pass over dx disappears.
template <typename T, bool G8>
__global__ __launch_bounds__(Q_NT) void rangebn_bwd_apply_kernel(const T* g, T* dx, const float* coef, long long nch, int C,
                                                                const int* arg, int chunks, int L, float* mm_partial,
                                                                const float* g8qp, float g8qmax) {
  constexpr int CH = ElemTraits<T>::kChunk;
  __shared__ float red[8];
  const QOp gop = q_op_make(g8qp, g8qmax);
  const int CC = C / CH;
  const long long rch = nch / gridDim.y;
  const long long row0 = (long long)blockIdx.y * rch, row1 = row0 + rch;
  const long long stride = (long long)gridDim.x * Q_NT;
  const bool fixed_col = stride % CC == 0;   // see rangebn_apply_kernel
  const int dm = (int)(stride / CC);         // pixels a thread advances per step (fixed_col)
  float a[CH], b[CH], d[CH];
  int imx[CH], imn[CH];
  int c0 = 0, jend = 0, jcur = -1;           // the thread's channels; end pixel / index of the statistics chunk in `imx / imn`
  float mn = INFINITY, mx = -INFINITY, mnr = INFINITY, mxr = -INFINITY;
  auto load_coef = [&](long long id) {
    c0 = (int)(id % CC) * CH;
    jcur = -1;
    jend = 0;
#pragma unroll
    for (int e = 0; e < CH; ++e) { a[e] = coef[c0 + e]; b[e] = coef[C + c0 + e]; d[e] = arg != nullptr ? coef[2 * C + c0 + e] : 0.f; }
  };
  // (m = pixel of chunk `id`, carried by the caller: no 64-bit division per element)
  auto apply = [&](const u32x4& v, long long id, int m) {
    float f[CH];
    q_unpack_raw<T, G8>(v, gop, f);
#pragma unroll
    for (int e = 0; e < CH; ++e) f[e] = f[e] * a[e] + b[e];
    u32x4 o = Chunk<T>::pack(f);
    bool routed = false;
    if (arg != nullptr) {
      if (m >= jend || jcur < 0) {             // entered another statistics chunk (a thread's pixels only increase)
        jcur = m / L;
        jend = (jcur + 1) * L;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          imx[e] = arg[(size_t)(c0 + e) * 2 * chunks + jcur];
          imn[e] = arg[(size_t)(c0 + e) * 2 * chunks + chunks + jcur];
        }
      }
      int hit = 0;
#pragma unroll
      for (int e = 0; e < CH; ++e) hit |= (int)(m == imx[e]) | (int)(m == imn[e]);
      if (hit) {
        routed = true;
        Chunk<T>::unpack(o, f);                // the stored value, then +d and -d each rounded to T
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          if (m == imx[e]) f[e] = q_round_to<T>(f[e] + d[e]);
          if (m == imn[e]) f[e] = q_round_to<T>(f[e] - d[e]);
        }
        o = Chunk<T>::pack(f);
      }
    }
    cn_st16((char*)dx + id * 16, o);
    if (mm_partial != nullptr) {
      // plain chunks: the fp32 values (rounding to T is monotonic, the extremes are rounded once at the end); a chunk with a
      // routed element: its values as stored (f holds them)
      if (routed) {
#pragma unroll
        for (int e = 0; e < CH; ++e) { mnr = fminf(mnr, f[e]); mxr = fmaxf(mxr, f[e]); }
      } else {
#pragma unroll
        for (int e = 0; e < CH; ++e) { mn = fminf(mn, f[e]); mx = fmaxf(mx, f[e]); }
      }
    }
  };
  long long id = row0 + (long long)blockIdx.x * Q_NT + threadIdx.x;
  int m = 0;
  if (id < row1) { load_coef(id); m = (int)(id / CC); }
  if (fixed_col) {
    for (; id + stride < row1; id += 2 * stride, m += 2 * dm) {
      const u32x4 v0 = q_ld_raw<T, G8>(g, id), v1 = q_ld_raw<T, G8>(g, id + stride);
      apply(v0, id, m);
      apply(v1, id + stride, m + dm);
    }
    for (; id < row1; id += stride, m += dm) apply(q_ld_raw<T, G8>(g, id), id, m);
  } else {
    for (; id < row1; id += stride) {
      load_coef(id);
      apply(q_ld_raw<T, G8>(g, id), id, (int)(id / CC));
    }
  }
  if (mm_partial != nullptr) {
    mn = fminf(q_round_to<T>(mn), mnr);
    mx = fmaxf(q_round_to<T>(mx), mxr);
    q_block_minmax(mn, mx, red);
    if (threadIdx.x == 0) {
      mm_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2] = mn;
      mm_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = mx;
    }
  }
}

// one thread per (channel, chunk): the chunk's first maximum gets +d, its first minimum -d (the unfused form of the routing)
template <typename T>
__global__ __launch_bounds__(Q_NT) void rangebn_bwd_route_kernel(T* dx, const float* coef, const int* arg, int C, int chunks) {
  const int id = blockIdx.x * Q_NT + threadIdx.x;
  if (id >= C * chunks) return;
  const int c = id / chunks, j = id - c * chunks;
  const float d = coef[2 * C + c];
  const int imx = arg[(size_t)c * 2 * chunks + j], imn = arg[(size_t)c * 2 * chunks + chunks + j];
  T* pm = dx + (size_t)imx * C + c;
  T* pn = dx + (size_t)imn * C + c;
  const float vm = cn_load_elem<T>(pm), vn = cn_load_elem<T>(pn);
  if (imx == imn) {     // +d then -d on one element, each rounded to T like the two separate updates
    cn_store_elem<T>(pm, vm + d);
    cn_store_elem<T>(pm, cn_load_elem<T>(pm) - d);
  } else {
    cn_store_elem<T>(pm, vm + d);
    cn_store_elem<T>(pn, vn - d);
  }
}

// mm_rows / dx_minmax given: the routing is folded into the apply pass and the per-row extremes of the final dx come out
// of it (3 launches); else apply + routing pass (4 launches).  Same dx bits.
static int rangebn_bwd_impl(const void* g, const void* x, const float* weight, const float* stats, const int* arg,
                              void* dx, float* dweight, float* dbias, int M, int C, int chunks, float scale_fix,
                              int dtype, float* ws, size_t ws_bytes, void* stream_, int mm_rows, float* dx_minmax,
                              const float* g8qp = nullptr, int g_bits = 8, const float* x8qp = nullptr, int x_bits = 8,
                              float* dx_qp = nullptr) {
  // g8qp / x8qp: that operand holds 8-bit LEVELS of the grid (zero point, range) = g8qp[0..1] / x8qp[0..1] (cn_quantize_levels,
  // cn_rangebn_fwd_q8) instead of values
  if ((g8qp != nullptr && (g_bits < 1 || g_bits > 8)) || (x8qp != nullptr && (x_bits < 1 || x_bits > 8))) { cn_set_error("rangebn_bwd: level operands have <= 8 bits"); return CN_EINVAL; }
  const float gqmax = (float)((1 << g_bits) - 1), xqmax = (float)((1 << x_bits) - 1);
  hipStream_t stream = (hipStream_t)stream_;
  const int CH = dtype == CN_BF16 ? 8 : 4;
  if (dtype != CN_BF16 && dtype != CN_F32) { cn_set_error("rangebn_bwd: bad dtype"); return CN_EINVAL; }
  if (M <= 0 || C <= 0 || C % CH != 0 || chunks <= 0 || M % chunks != 0) { cn_set_error("rangebn_bwd: bad shape"); return CN_ESHAPE; }
  if (g == nullptr || x == nullptr || weight == nullptr || stats == nullptr || arg == nullptr || dx == nullptr ||
      dweight == nullptr || dbias == nullptr) { cn_set_error("rangebn_bwd: null operand"); return CN_EINVAL; }
  if (ws == nullptr || ws_bytes < cn_rangebn_workspace(M, C, chunks)) { cn_set_error("rangebn_bwd: workspace too small"); return CN_EWORKSPACE; }
  const bool fused = dx_minmax != nullptr;
  if (fused && (mm_rows < 1 || M % mm_rows != 0 || mm_rows > 8192)) { cn_set_error("rangebn_bwd: %d pixels do not split into %d rows", M, mm_rows); return CN_ESHAPE; }
  const int CC = C / CH, cols = rbn_cols(CC);
  int rpr = cn_get_option("rbn_bwd_row_px", 512);   // pixels per partial row of the backward reduction (knob)
  if (rpr < 256) rpr = 256;
  const int rows = (M + rpr - 1) / rpr;
  float* partial = ws;
  float* coef = ws + (size_t)rows * 2 * C;
  dim3 grid((unsigned)((CC + cols - 1) / cols), (unsigned)rows);
  const long long nch = (long long)M * CC;
  const float route = scale_fix / (float)chunks;
  const int arows = fused ? mm_rows : 1;
  const unsigned bpr = fused ? q_grid_rows(nch, arows, CC) : q_grid_cols(nch, CC);
  float* mmp = nullptr;
  if (fused) {   // the reduction partials are dead once finalize has run; coef lives behind them
    if ((size_t)arows * bpr * 2 <= (size_t)rows * 2 * C) mmp = partial;
    else if (((size_t)rows * 2 * C + 3 * (size_t)C + (size_t)arows * bpr * 2) * sizeof(float) <= ws_bytes) mmp = coef + 3 * (size_t)C;   // (HW = 1, C = 64, N > 64 ...)
    else { cn_set_error("rangebn_bwd: workspace too small for the min / max partials"); return CN_EWORKSPACE; }
  }
  const dim3 agrid(bpr, (unsigned)arows);
  const int L = M / chunks;
  const bool g8 = g8qp != nullptr, x8 = x8qp != nullptr;
#define RBW(T, G8, X8)                                                                                                                  \
  do {                                                                                                                                  \
    CN_LAUNCH((rangebn_bwd_reduce_kernel<T, G8, X8>), grid, dim3(Q_NT), stream, (const T*)g, (const T*)x, stats, M, C, rpr, cols, partial, g8qp, gqmax, x8qp, xqmax); \
    CN_LAUNCH(rangebn_bwd_finalize_kernel, dim3((unsigned)((C + RF_C - 1) / RF_C)), dim3(RF_C * RF_T), stream, (const float*)partial, rows, M, C, weight, stats, route, dweight, dbias, coef); \
    CN_LAUNCH((rangebn_bwd_apply_kernel<T, G8>), agrid, dim3(Q_NT), stream, (const T*)g, (T*)dx, (const float*)coef, nch, C, fused ? arg : (const int*)nullptr, chunks, L, mmp, g8qp, gqmax); \
    if (!fused) CN_LAUNCH(rangebn_bwd_route_kernel<T>, dim3((unsigned)((C * chunks + Q_NT - 1) / Q_NT)), dim3(Q_NT), stream, (T*)dx, (const float*)coef, arg, C, chunks); \
  } while (0)
  if (dtype == CN_BF16) {
    if (g8 && x8) RBW(bf16_t, true, true); else if (g8) RBW(bf16_t, true, false); else if (x8) RBW(bf16_t, false, true); else RBW(bf16_t, false, false);
  } else {
    if (g8 && x8) RBW(float, true, true); else if (g8) RBW(float, true, false); else if (x8) RBW(float, false, true); else RBW(float, false, false);
  }
#undef RBW
  if (fused)
    CN_LAUNCH(minmax_final_kernel, dim3((unsigned)((arows + Q_NT - 1) / Q_NT)), dim3(Q_NT), stream, (const float*)mmp, arows, (int)bpr, dx_minmax, arows <= Q_NT ? dx_qp : (float*)nullptr);
  return cn_check_launch("rangebn_bwd");
}

extern "C" int cn_rangebn_bwd(const void* g, const void* x, const float* weight, const float* stats, const int* arg,
                              void* dx, float* dweight, float* dbias, int M, int C, int chunks, float scale_fix,
                              int dtype, float* ws, size_t ws_bytes, void* stream_) {
  return rangebn_bwd_impl(g, x, weight, stats, arg, dx, dweight, dbias, M, C, chunks, scale_fix, dtype, ws, ws_bytes, stream_,
                          0, nullptr);
}
// cn_rangebn_bwd with the routing folded into the apply pass and dx_minmax[mm_rows][2] = cn_minmax_rows(dx, mm_rows) as a
// side output (mm_rows = batch samples).  Same dx, dweight, dbias bits.
extern "C" int cn_rangebn_bwd_mm(const void* g, const void* x, const float* weight, const float* stats, const int* arg,
                                 void* dx, float* dweight, float* dbias, int M, int C, int chunks, float scale_fix,
                                 int dtype, int mm_rows, float* dx_minmax, float* ws, size_t ws_bytes, void* stream_) {
  if (dx_minmax == nullptr) { cn_set_error("rangebn_bwd_mm: needs dx_minmax"); return CN_EINVAL; }
  return rangebn_bwd_impl(g, x, weight, stats, arg, dx, dweight, dbias, M, C, chunks, scale_fix, dtype, ws, ws_bytes, stream_,
                          mm_rows, dx_minmax);
}

// cn_rangebn_bwd_mm on operands stored as 8-bit LEVELS: g (the quantised output gradient: cn_quantize_levels with
// g_qparams, g_bits) and / or x (the snapped input: cn_rangebn_fwd_q8 with x_qparams, x_bits); a null qparams pointer
// means that operand holds values as before.  Same dx, dweight, dbias, dx_minmax bits as on the stored values.
// dx_qp_extreme (optional, mm_rows <= 256): [zero_point, range] = cn_qparams(dx_minmax, mm_rows, 1), for the gradient quantiser
// of the convolution in front.
extern "C" int cn_rangebn_bwd_q8(const void* g, const float* g_qparams, int g_bits, const void* x, const float* x_qparams,
                                 int x_bits, const float* weight, const float* stats, const int* arg, void* dx,
                                 float* dweight, float* dbias, int M, int C, int chunks, float scale_fix, int dtype,
                                 int mm_rows, float* dx_minmax, float* dx_qp_extreme, float* ws, size_t ws_bytes, void* stream_) {
  if (dx_minmax == nullptr) { cn_set_error("rangebn_bwd_q8: needs dx_minmax"); return CN_EINVAL; }
  if (dx_qp_extreme != nullptr && mm_rows > Q_NT) { cn_set_error("rangebn_bwd_q8: dx_qp_extreme needs mm_rows <= %d", Q_NT); return CN_ESHAPE; }
  return rangebn_bwd_impl(g, x, weight, stats, arg, dx, dweight, dbias, M, C, chunks, scale_fix, dtype, ws, ws_bytes, stream_,
                          mm_rows, dx_minmax, g_qparams, g_bits, x_qparams, x_bits, dx_qp_extreme);
}

// ------------------------------------------------------------------------------------------------ elementwise + min / max
// a = b * (c > 0) (op 2: the ReLU mask of a gradient) or a = relu(b + c) (op 4: the residual junction) - cn_eltwise's
// arithmetic - with minmax[rows][2] = cn_minmax_rows(a, rows) as a side output: the quantiser that consumes `a` next
// (the RangeBN gradient quantiser; the next block's activation quantiser) needs no min / max pass of its own.
template <typename T, int OP>
__global__ __launch_bounds__(Q_NT) void eltwise_mm_kernel(char* a, const char* b, const char* c, long long nch, float* mm_partial) {
  constexpr int CH = ElemTraits<T>::kChunk;
  __shared__ float red[8];
  const long long rch = nch / gridDim.y;
  const long long row0 = (long long)blockIdx.y * rch, row1 = row0 + rch;
  const long long stride = (long long)gridDim.x * Q_NT;
  float mn = INFINITY, mx = -INFINITY;
  auto one = [&](const u32x4& vb, const u32x4& vc, long long id) {
    float fb[CH], fc[CH], fa[CH];
    Chunk<T>::unpack(vb, fb);
    Chunk<T>::unpack(vc, fc);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      if (OP == 2) fa[e] = fc[e] > 0.f ? fb[e] : 0.f;
      else { const float v = fb[e] + fc[e]; fa[e] = v > 0.f ? v : 0.f; }
    }
    cn_st16(a + id * 16, Chunk<T>::pack(fa));
#pragma unroll
    for (int e = 0; e < CH; ++e) { mn = fminf(mn, fa[e]); mx = fmaxf(mx, fa[e]); }   // (rounded once at the end: monotonic)
  };
  long long id = row0 + (long long)blockIdx.x * Q_NT + threadIdx.x;
  for (; id + stride < row1; id += 2 * stride) {   // four loads in flight
    const u32x4 b0 = cn_ld16(b + id * 16), b1 = cn_ld16(b + (id + stride) * 16);
    const u32x4 c0 = cn_ld16(c + id * 16), c1 = cn_ld16(c + (id + stride) * 16);
    one(b0, c0, id);
    one(b1, c1, id + stride);
  }
  for (; id < row1; id += stride) one(cn_ld16(b + id * 16), cn_ld16(c + id * 16), id);
  q_block_minmax(mn, mx, red);
  if (threadIdx.x == 0) {
    mm_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2] = q_round_to<T>(mn);
    mm_partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = q_round_to<T>(mx);
  }
}

extern "C" size_t cn_eltwise_mm_workspace(long long n, int rows, int dtype) {
  const int CH = dtype == CN_F32 ? 4 : 8;
  if (n <= 0 || rows < 1) return 0;
  return (size_t)rows * q_grid_rows(n / CH, rows, 1) * 2 * sizeof(float);
}
static int eltwise_mm_impl(int op, void* a, const void* b, const void* c, long long n, int dtype, int rows, float* minmax,
                           float* qp_extreme, float* ws, size_t ws_bytes, void* stream_);
extern "C" int cn_eltwise_mm(int op, void* a, const void* b, const void* c, long long n, int dtype, int rows, float* minmax,
                             float* ws, size_t ws_bytes, void* stream_) {
  return eltwise_mm_impl(op, a, b, c, n, dtype, rows, minmax, nullptr, ws, ws_bytes, stream_);
}
// ... additionally qp_extreme[2] = cn_qparams(minmax, rows, 1) (rows <= 256): the consumer is a gradient quantiser
extern "C" int cn_eltwise_mm_qp(int op, void* a, const void* b, const void* c, long long n, int dtype, int rows, float* minmax,
                                float* qp_extreme, float* ws, size_t ws_bytes, void* stream_) {
  if (qp_extreme == nullptr || rows > Q_NT) { cn_set_error("eltwise_mm_qp: needs qp_extreme and rows <= %d", Q_NT); return CN_EINVAL; }
  return eltwise_mm_impl(op, a, b, c, n, dtype, rows, minmax, qp_extreme, ws, ws_bytes, stream_);
}
static int eltwise_mm_impl(int op, void* a, const void* b, const void* c, long long n, int dtype, int rows, float* minmax,
                           float* qp_extreme, float* ws, size_t ws_bytes, void* stream_) {
  if (dtype != CN_BF16 && dtype != CN_F32) { cn_set_error("eltwise_mm: bad dtype"); return CN_EINVAL; }
  if (op != 2 && op != 4) { cn_set_error("eltwise_mm: op %d (2 or 4)", op); return CN_EINVAL; }
  const int CH = dtype == CN_F32 ? 4 : 8;
  if (a == nullptr || b == nullptr || c == nullptr || minmax == nullptr) { cn_set_error("eltwise_mm: null operand"); return CN_EINVAL; }
  if (n <= 0 || rows < 1 || rows > 8192 || n % ((long long)rows * CH) != 0) { cn_set_error("eltwise_mm: n=%lld does not split into %d rows of whole chunks", n, rows); return CN_ESHAPE; }
  if (ws == nullptr || ws_bytes < cn_eltwise_mm_workspace(n, rows, dtype)) { cn_set_error("eltwise_mm: workspace too small"); return CN_EWORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  const long long nch = n / CH;
  const unsigned bpr = q_grid_rows(nch, rows, 1);
  const dim3 grid(bpr, (unsigned)rows);
#define ELTMM(T, OP) CN_LAUNCH((eltwise_mm_kernel<T, OP>), grid, dim3(Q_NT), stream, (char*)a, (const char*)b, (const char*)c, nch, ws)
  if (dtype == CN_BF16) { if (op == 2) ELTMM(bf16_t, 2); else ELTMM(bf16_t, 4); }
  else { if (op == 2) ELTMM(float, 2); else ELTMM(float, 4); }
#undef ELTMM
  CN_LAUNCH(minmax_final_kernel, dim3((unsigned)((rows + Q_NT - 1) / Q_NT)), dim3(Q_NT), stream, (const float*)ws, rows, (int)bpr, minmax, rows <= Q_NT ? qp_extreme : (float*)nullptr);
  return cn_check_launch("eltwise_mm");
}
