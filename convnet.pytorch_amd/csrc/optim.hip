// optim.hip -- fused multi-tensor SGD+momentum over flat fp32 buffers, gradient-norm / clipping,
// filter preparation (fp32 master KRSC -> compute-dtype KRSC + CRSK) and bias-gradient column sums.
//
// Replaces, for the reference's ResNet regime (/root/reference models/resnet.py:250-256 and the
// weight-decay filter at :34-40): optimizer.zero_grad/step as Trainer._step drives them
// (trainer.py:111-112,173), the per-parameter `p.grad.div_(loss_scale)` loop (trainer.py:165-169)
// and clip_grad_norm_ (trainer.py:171-172).  torch.optim.SGD semantics: g += wd*p;
// buf = mu*buf + g (a zero-initialised buf makes the first step buf = g exactly); p -= lr*buf.
// The reference issues ~4 tiny kernels per parameter tensor (161 tensors for ResNet-50); here the
// whole model is one launch per weight-decay group over a flat arena.
#include "cn_common.h"
#include "cn_api_internal.h"

// p, g, buf: fp32 flat; gscale folds 1/world_size and 1/loss_scale; clip_coef (optional, device
// scalar) folds the gradient-clipping factor computed by cn_clip_coef.
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float* buf, long long n, float lr,
                                                 float momentum, float wd, float gscale,
                                                 const float* clip_coef, const float* hyper) {
  if (hyper != nullptr) { lr = hyper[0]; momentum = hyper[1]; }   // device-resident schedule (graph replay)
  const float cs = clip_coef != nullptr ? gscale * clip_coef[0] : gscale;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pv = ((const f32x4*)p)[i], gv = ((const f32x4*)g)[i], bv = ((const f32x4*)buf)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = fmaf(wd, pv[e], gv[e] * cs);
      bv[e] = fmaf(momentum, bv[e], gg);
      pv[e] = fmaf(-lr, bv[e], pv[e]);
    }
    ((f32x4*)p)[i] = pv;
    ((f32x4*)buf)[i] = bv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float gg = fmaf(wd, p[i], g[i] * cs);
    buf[i] = fmaf(momentum, buf[i], gg);
    p[i] = fmaf(-lr, buf[i], p[i]);
  }
}

// Sum of squares: partial per workgroup, then cn_clip_coef reduces in fixed order.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long long n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    s = fmaf(g[i], g[i], s);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += cn_shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = total L2 norm * gscale (the reference's `grad` meter value), out[1] = clip coefficient
// min(1, max_norm / (norm + 1e-6)) as torch.nn.utils.clip_grad_norm_ computes it; meters (optional)
// accumulates norm*weight and weight.
__global__ void clip_coef_kernel(const float* partial, int nparts, float gscale, float max_norm, float* out,
                                 float* meters, float weight) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += (double)partial[i];
  const float norm = (float)sqrt(s) * gscale;
  out[0] = norm;
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = max_norm / (norm + 1e-6f);
    if (coef > 1.f) coef = 1.f;
  }
  out[1] = coef;
  if (meters != nullptr) { meters[0] += norm * weight; meters[1] += weight; }
}

// fp32 master filter [Co][taps][Creal] -> compute dtype [Co][taps][Cpad] (zero padded) and,
// optionally, the dgrad operand [Cpad? no: Creal==Cpad required][taps][Co].
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* w, T* krsc, T* crsk, int Co, int taps,
                                                         int Creal, int Cpad) {
  const long long total = (long long)Co * taps * Cpad;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    const int c = (int)(id % Cpad);
    const long long rest = id / Cpad;
    const int t = (int)(rest % taps);
    const int co = (int)(rest / taps);
    const float v = c < Creal ? w[(rest)*Creal + c] : 0.f;
    cn_store_elem<T>(krsc + id, v);
    if (crsk != nullptr) cn_store_elem<T>(crsk + ((size_t)c * taps + t) * Co + co, v);
  }
}

// Multi-tensor form: every filter of the model in ONE launch.  desc rows (int64 x 8):
//   [src_off (floats into the master arena), start (prefix sum of Co*taps*Cpad), krsc_off, crsk_off
//    (element offsets into the compute-dtype weight buffer, crsk_off < 0: none), Co, taps, Creal, Cpad]
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const float* master, T* wbuf, const long long* desc,
                                                               int nd, long long total) {
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
    int lo = 0, hi = nd - 1;
    while (lo < hi) {   // last descriptor with start <= id
      const int mid = (lo + hi + 1) >> 1;
      if (desc[(size_t)mid * 8 + 1] <= id) lo = mid; else hi = mid - 1;
    }
    const long long* d = desc + (size_t)lo * 8;
    const long long local = id - d[1];
    const int Co = (int)d[4], taps = (int)d[5], Creal = (int)d[6], Cpad = (int)d[7];
    const int c = (int)(local % Cpad);
    const long long rest = local / Cpad;
    const int t = (int)(rest % taps);
    const int co = (int)(rest / taps);
    const float v = c < Creal ? master[d[0] + rest * Creal + c] : 0.f;
    cn_store_elem<T>(wbuf + d[2] + local, v);
    if (d[3] >= 0) cn_store_elem<T>(wbuf + d[3] + ((size_t)c * taps + t) * Co + co, v);
  }
}

// Tiled variant for the regular filters (Cpad == Creal): one workgroup per 64 (co) x 64 (j = tap*C + c)
// tile of the fp32 master, staged through LDS so that both the KRSC copy (same order) and the CRSK
// copy (transposed: rows (c*taps + t), co contiguous) are written with coalesced stores.  The
// per-element kernel above spends a descriptor binary search and a scattered 2-byte store per
// weight (0.33 ms per ResNet-50 step, profiles/r01_*kernel_stats*); this one streams.
//   tiles: int[ntiles][4] = {descriptor index, co0, j0, 0}
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_tiled_kernel(const float* master, T* wbuf, const long long* desc,
                                                               const int* tiles) {
  __shared__ float tile[64][65];
  const int* tl = tiles + (size_t)blockIdx.x * 4;
  const long long* d = desc + (size_t)tl[0] * 8;
  const int co0 = tl[1], j0 = tl[2];
  const int Co = (int)d[4], taps = (int)d[5], C = (int)d[6];
  const int J = taps * C;
  const float* src = master + d[0];
  const int lane64 = threadIdx.x & 63, grp = threadIdx.x >> 6;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = grp + 4 * i, co = co0 + r, j = j0 + lane64;
    tile[r][lane64] = (co < Co && j < J) ? src[(size_t)co * J + j] : 0.f;
  }
  __syncthreads();
  T* krsc = wbuf + d[2];
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = grp + 4 * i, co = co0 + r, j = j0 + lane64;
    if (co < Co && j < J) cn_store_elem<T>(krsc + (size_t)co * J + j, tile[r][lane64]);
  }
  if (d[3] >= 0) {
    T* crsk = wbuf + d[3];
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int jj = grp + 4 * i, j = j0 + jj, co = co0 + lane64;
      if (j < J && co < Co) {
        const int t = j / C, c = j - t * C;
        cn_store_elem<T>(crsk + ((size_t)c * taps + t) * Co + co, tile[lane64][jj]);
      }
    }
  }
}

// Pixel-pair packing of a stem filter (see nchw_to_pairs_kernel): master [K][R][S][C] fp32 (C <= 4) ->
// out [K][R][S2][8] bf16, S2 = ceil(S/2), out[k][r][s2][j*4 + c] = master[k][r][2*s2 + j][c] (zero for
// 2*s2 + j >= S or c >= C); and the inverse scatter of the weight gradient computed in the packed layout.
__global__ __launch_bounds__(256) void weight_prep_pairs_kernel(const float* master, bf16_t* out, int K, int R, int S,
                                                               int C, int S2) {
  const int total = K * R * S2 * 8;
  for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
    const int e = id & 7, j = e >> 2, c = e & 3;
    int rest = id >> 3;
    const int s2 = rest % S2;
    rest /= S2;   // k * R + r
    const int s = 2 * s2 + j;
    const float v = (s < S && c < C) ? master[((size_t)rest * S + s) * C + c] : 0.f;
    cn_store_elem<bf16_t>(out + id, v);
  }
}
__global__ __launch_bounds__(256) void wgrad_unpack_pairs_kernel(const float* packed, float* dw, int K, int R, int S,
                                                                int C, int S2, float beta) {
  const int total = K * R * S * C;
  for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
    const int c = id % C;
    int rest = id / C;
    const int s = rest % S;
    rest /= S;   // k * R + r
    const float g = packed[((size_t)rest * S2 + (s >> 1)) * 8 + (s & 1) * 4 + c];
    dw[id] = beta != 0.f ? beta * dw[id] + g : g;
  }
}

// out[c] (+)= sum_m x[m][c], x fp32 or bf16 row-major [M][C]; one thread per column, rows strided
// over gridDim.y with a fixed-order second stage.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* x, float* partial, int M, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int m = blockIdx.y; m < M; m += gridDim.y) s += cn_load_elem<T>(x + (size_t)m * C + c);
  partial[(size_t)blockIdx.y * C + c] = s;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, float* out, int nparts, int C,
                                                          float beta, float scale) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int i = 0; i < nparts; ++i) s += partial[(size_t)i * C + c];
  out[c] = (beta != 0.f ? beta * out[c] : 0.f) + s * scale;
}

// Small dense layers whose width is not a multiple of the 16-byte chunk (the 10-way MNIST head of
// /root/reference models/mnist.py:30): one thread per output element, plain fp32 dot products.
//   fwd  : y[b][k]  = sum_c x[b][c] w[k][c] + bias[k]            (x in T, y fp32)
//   dgrad: dx[b][c] = sum_k dy[b][k] w[k][c]                      (dx in T)
//   wgrad: dw[k][c] += sum_b dy[b][k] x[b][c];  db[k] += sum_b dy[b][k]
template <typename T>
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const T* x, const float* w, const float* bias,
                                                              float* y, int B, int C, int K) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= B * K) return;
  const int b = id / K, k = id - b * K;
  float s = bias != nullptr ? bias[k] : 0.f;
  for (int c = 0; c < C; ++c) s = fmaf(cn_load_elem<T>(x + (size_t)b * C + c), w[(size_t)k * C + c], s);
  y[id] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void small_linear_dgrad_kernel(const float* dy, const float* w, T* dx, int B, int C,
                                                                int K) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= B * C) return;
  const int b = id / C, c = id - b * C;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(dy[(size_t)b * K + k], w[(size_t)k * C + c], s);
  cn_store_elem<T>(dx + id, s);
}
template <typename T>
__global__ __launch_bounds__(256) void small_linear_wgrad_kernel(const T* x, const float* dy, float* dw, float* db,
                                                                int B, int C, int K) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= K * (C + 1)) return;
  const int k = id / (C + 1), c = id - k * (C + 1);
  float s = 0.f;
  if (c < C) {
    for (int b = 0; b < B; ++b) s = fmaf(dy[(size_t)b * K + k], cn_load_elem<T>(x + (size_t)b * C + c), s);
    dw[(size_t)k * C + c] += s;
  } else if (db != nullptr) {
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * K + k];
    db[k] += s;
  }
}

// y = (T) x  (fp32 -> compute dtype), used for the fp32 logits gradient hand-off and tests
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* x, T* y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    cn_store_elem<T>(y + i, x[i]);
}

__global__ __launch_bounds__(256) void fill_kernel(float* x, long long n, float v) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) x[i] = v;
}

// ------------------------------------------------------------------------------------------------
static unsigned opt_grid(long long work, long long cap) {
  long long nb = (work + 255) / 256;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

extern "C" int cn_sgd_momentum(float* p, const float* g, float* buf, long long n, float lr, float momentum,
                               float weight_decay, float gscale, const float* clip_coef, const float* hyper_dev,
                               void* stream) {
  if (n <= 0) return CN_OK;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) != 0) {
    cn_set_error("sgd_momentum: buffers must be 16-byte aligned");
    return CN_EINVAL;
  }
  CN_LAUNCH(sgd_kernel, dim3(opt_grid(n / 4 + 1, 4096)), dim3(256), (hipStream_t)stream, p, g, buf, n, lr,
            momentum, weight_decay, gscale, clip_coef, hyper_dev);
  return cn_check_launch("sgd_momentum");
}

#define CN_NORM_PARTS 1024
extern "C" size_t cn_grad_norm_workspace(void) { return CN_NORM_PARTS * sizeof(float); }

extern "C" int cn_grad_norm_clip(const float* g, long long n, float gscale, float max_norm, float* out2,
                                 float* meters2, float meter_weight, float* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  unsigned nb = opt_grid(n, CN_NORM_PARTS);
  CN_LAUNCH(sumsq_kernel, dim3(nb), dim3(256), stream, g, n, workspace);
  CN_LAUNCH(clip_coef_kernel, dim3(1), dim3(64), stream, (const float*)workspace, (int)nb, gscale, max_norm, out2,
            meters2, meter_weight);
  return cn_check_launch("grad_norm_clip");
}

extern "C" int cn_weight_prep(const float* w_master, void* w_krsc, void* w_crsk, int Co, int taps, int Creal,
                              int Cpad, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (Cpad < Creal || (w_crsk != nullptr && Cpad != Creal)) {
    cn_set_error("weight_prep: bad channel padding (Creal=%d Cpad=%d)", Creal, Cpad);
    return CN_ESHAPE;
  }
  dim3 grid(opt_grid((long long)Co * taps * Cpad, 2048));
  if (!cn_dtype_ok(dtype)) { cn_set_error("weight_prep: bad dtype"); return CN_EINVAL; }
  CN_DISPATCH_T(dtype, CN_LAUNCH(weight_prep_kernel<TT>, grid, dim3(256), stream, w_master, (TT*)w_krsc, (TT*)w_crsk, Co,
              taps, Creal, Cpad));
  return cn_check_launch("weight_prep");
}

extern "C" int cn_weight_prep_multi(const float* master, void* wbuf, const long long* desc, int nd, long long total,
                                    int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (nd <= 0 || total <= 0) return CN_OK;
  dim3 grid(opt_grid(total, 8192));
  if (!cn_dtype_ok(dtype)) { cn_set_error("weight_prep_multi: bad dtype"); return CN_EINVAL; }
  CN_DISPATCH_T(dtype, CN_LAUNCH(weight_prep_multi_kernel<TT>, grid, dim3(256), stream, master, (TT*)wbuf, desc, nd, total));
  return cn_check_launch("weight_prep_multi");
}

extern "C" int cn_weight_prep_tiled(const float* master, void* wbuf, const long long* desc, const int* tiles,
                                    int ntiles, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (ntiles <= 0) return CN_OK;
  if (!cn_dtype_ok(dtype)) { cn_set_error("weight_prep_tiled: bad dtype"); return CN_EINVAL; }
  CN_DISPATCH_T(dtype, CN_LAUNCH(weight_prep_tiled_kernel<TT>, dim3((unsigned)ntiles), dim3(256), stream, master, (TT*)wbuf,
              desc, tiles));
  return cn_check_launch("weight_prep_tiled");
}

extern "C" int cn_weight_prep_pairs(const float* master_krsc, void* out, int K, int R, int S, int C, void* stream) {
  if (K <= 0 || R <= 0 || S <= 0 || C < 1 || C > 4) { cn_set_error("weight_prep_pairs: need 1 <= C <= 4"); return CN_ESHAPE; }
  const int S2 = (S + 1) / 2;
  CN_LAUNCH(weight_prep_pairs_kernel, dim3(opt_grid((long long)K * R * S2 * 8, 256)), dim3(256), (hipStream_t)stream,
            master_krsc, (bf16_t*)out, K, R, S, C, S2);
  return cn_check_launch("weight_prep_pairs");
}

extern "C" int cn_wgrad_unpack_pairs(const float* packed, float* dw_krsc, int K, int R, int S, int C, float beta,
                                     void* stream) {
  if (K <= 0 || R <= 0 || S <= 0 || C < 1 || C > 4) { cn_set_error("wgrad_unpack_pairs: need 1 <= C <= 4"); return CN_ESHAPE; }
  const int S2 = (S + 1) / 2;
  CN_LAUNCH(wgrad_unpack_pairs_kernel, dim3(opt_grid((long long)K * R * S * C, 256)), dim3(256), (hipStream_t)stream,
            packed, dw_krsc, K, R, S, C, S2, beta);
  return cn_check_launch("wgrad_unpack_pairs");
}

#define CN_COLSUM_PARTS 64
extern "C" size_t cn_colsum_workspace(int C) { return (size_t)CN_COLSUM_PARTS * C * sizeof(float); }

extern "C" int cn_colsum(const void* x, float* out, int M, int C, int dtype, float beta, float scale,
                         float* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || C <= 0) { cn_set_error("colsum: empty"); return CN_ESHAPE; }
  int parts = M < CN_COLSUM_PARTS ? M : CN_COLSUM_PARTS;
  dim3 grid((unsigned)((C + 255) / 256), (unsigned)parts);
  if (!cn_dtype_ok(dtype)) { cn_set_error("colsum: bad dtype"); return CN_EINVAL; }
  CN_DISPATCH_T(dtype, CN_LAUNCH(colsum_partial_kernel<TT>, grid, dim3(256), stream, (const TT*)x, workspace, M, C));
  CN_LAUNCH(colsum_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), stream, (const float*)workspace,
            out, parts, C, beta, scale);
  return cn_check_launch("colsum");
}

extern "C" int cn_small_linear(int mode, const void* x, const float* w, const float* bias, void* out, float* dw,
                               float* db, int B, int C, int K, int dtype, void* stream_) {
  // mode 0: out = y (fp32) from x;  1: out = dx (T) from x := dy (fp32);  2: dw/db += from x (T), out := dy (fp32)
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || C <= 0 || K <= 0) { cn_set_error("small_linear: empty"); return CN_ESHAPE; }
  if (!cn_dtype_ok(dtype)) { cn_set_error("small_linear: bad dtype"); return CN_EINVAL; }
#define SL(T)                                                                                                       \
  do {                                                                                                              \
    if (mode == 0)                                                                                                  \
      CN_LAUNCH(small_linear_fwd_kernel<T>, dim3((unsigned)((B * K + 255) / 256)), dim3(256), stream, (const T*)x, w, \
                bias, (float*)out, B, C, K);                                                                        \
    else if (mode == 1)                                                                                             \
      CN_LAUNCH(small_linear_dgrad_kernel<T>, dim3((unsigned)((B * C + 255) / 256)), dim3(256), stream,              \
                (const float*)x, w, (T*)out, B, C, K);                                                              \
    else                                                                                                            \
      CN_LAUNCH(small_linear_wgrad_kernel<T>, dim3((unsigned)((K * (C + 1) + 255) / 256)), dim3(256), stream,        \
                (const T*)x, (const float*)out, dw, db, B, C, K);                                                   \
  } while (0)
  if (dtype == CN_BF16) SL(bf16_t); else if (dtype == CN_F16) SL(f16_t); else SL(float);
#undef SL
  return cn_check_launch("small_linear");
}

extern "C" int cn_cast_from_f32(const float* x, void* y, long long n, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0) return CN_OK;
  dim3 grid(opt_grid(n, 4096));
  if (!cn_dtype_ok(dtype)) { cn_set_error("cast: bad dtype"); return CN_EINVAL; }
  CN_DISPATCH_T(dtype, CN_LAUNCH(cast_kernel<TT>, grid, dim3(256), stream, x, (TT*)y, n));
  return cn_check_launch("cast");
}

extern "C" int cn_fill_f32(float* x, long long n, float v, void* stream_) {
  if (n <= 0) return CN_OK;
  CN_LAUNCH(fill_kernel, dim3(opt_grid(n, 4096)), dim3(256), (hipStream_t)stream_, x, n, v);
  return cn_check_launch("fill");
}
