// cn_common.h -- shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.
//
// Everything hardware-specific that the kernels rely on is funnelled through the small set of
// cn_* wrappers below (MFMA, LDS transpose-read, wave shuffles) so that (a) the assumed lane maps
// are stated in exactly one place and (b) the TEST-ONLY SIMT emulator (cn_emul.h) can provide a
// host implementation of the same contract.  The maps are pinned on hardware by cn_probe_*.
#pragma once

#ifdef CN_EMULATE
#include "cn_emul.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stddef.h>

// ---------------------------------------------------------------- error codes (C ABI)
#define CN_OK 0
#define CN_EINVAL (-1)
#define CN_ESHAPE (-2)
#define CN_EHIP (-3)
#define CN_EWORKSPACE (-4)
#define CN_ERCCL (-5)

// dtype codes of the C ABI
#define CN_F32 0
#define CN_BF16 1
#define CN_F16 2   /* IEEE half storage (the reference's --dtype half), fp32 accumulation */
// storage-type facts by dtype code (the 16-bit types share chunk width and byte size)
static inline int cn_dtype_ok(int dtype) { return dtype == CN_F32 || dtype == CN_BF16 || dtype == CN_F16; }
static inline int cn_dtype_bytes(int dtype) { return dtype == CN_F32 ? 4 : 2; }
static inline int cn_dtype_chunk(int dtype) { return dtype == CN_F32 ? 4 : 8; }

void cn_set_error(const char* fmt, ...);
int cn_check_launch(const char* what);
int cn_get_option(const char* name, int dflt);
void cn_set_last_kernel(const char* fmt, ...);

// dtype code -> storage type: runs the statement(s) with `TT` bound to bf16_t / f16_t / float
#define CN_DISPATCH_T(dtype, ...)                                          \
  do {                                                                     \
    if ((dtype) == CN_BF16) { typedef bf16_t TT; __VA_ARGS__; }            \
    else if ((dtype) == CN_F16) { typedef f16_t TT; __VA_ARGS__; }         \
    else { typedef float TT; __VA_ARGS__; }                                \
  } while (0)

// ---------------------------------------------------------------- launch macro
#ifdef CN_EMULATE
struct CnMarkLast { void release() {} };
// (launch plans, plan.hip: while a plan is being recorded every launch is also kept as a closure)
extern int cn_plan_recording;
int cn_plan_rec_closure(const std::function<void()>& launch);
#define CN_LAUNCH(kern, grid, block, stream, ...)                                   \
  do {                                                                              \
    (void)(stream);                                                                 \
    const dim3 cn_g_ = (grid), cn_b_ = (block);                                     \
    auto cn_body_ = [=]() { kern(__VA_ARGS__); };                                   \
    if (cn_plan_recording) cn_plan_rec_closure([=]() { cn_emul::launch(cn_g_, cn_b_, cn_body_); }); \
    cn_emul::launch(cn_g_, cn_b_, cn_body_);                                        \
  } while (0)
#else
// While a mark is armed (cn_stream_arm, runtime.hip) every launch of this thread carries the mark's event as the
// kernel's own completion event (hipExtLaunchKernel's stopEvent): another stream can wait for that kernel without a
// marker packet in this stream's queue.
#include <hip/hip_ext.h>
#include <tuple>
#include <type_traits>
#include <utility>
template <typename Tup, size_t... I>
static inline void cn_tuple_ptrs(const Tup& t, const void** out, std::index_sequence<I...>) {
  const void* p[] = {(const void*)&std::get<I>(t)..., nullptr};
  for (size_t i = 0; i < sizeof...(I); ++i) out[i] = p[i];
}
extern thread_local hipEvent_t cn_tl_stop_event;
extern thread_local int cn_tl_stop_recorded;
extern thread_local int cn_tl_stop_hold;   // > 0: launches stay plain (CnMarkLast: only a call's last kernel takes the event)
struct CnMarkLast {   // scope guard of a multi-kernel entry point: release() right before its final launch
  CnMarkLast() { ++cn_tl_stop_hold; }
  ~CnMarkLast() { release(); }
  void release() { if (held) { --cn_tl_stop_hold; held = false; } }
  bool held = true;
};
// Launch plans (plan.hip): while one is being recorded every launch is logged with a private copy of its arguments
// (converted to the kernel's own parameter types) and issued plainly; an armed mark becomes an event recorded right
// behind the kernel - the recording normally runs under stream capture, where a kernel's stop event would be lost.
extern int cn_plan_recording;
extern thread_local int cn_tl_stop_handle;
int cn_plan_rec_kernel(const void* func, dim3 grid, dim3 block, hipStream_t stream, int nargs, const void* const* ptrs,
                       const size_t* sizes, const size_t* aligns, int mark_handle);
void cn_plan_rec_kernel_node(int op_index, hipStream_t stream);
template <typename... P, typename... A>
static inline void cn_plan_launch(void (*kern)(P...), dim3 grid, dim3 block, hipStream_t stream, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "CN_LAUNCH: argument count differs from the kernel's parameter count");
  std::tuple<typename std::decay<P>::type...> vals{static_cast<typename std::decay<P>::type>(a)...};
  const bool marked = cn_tl_stop_event != nullptr && cn_tl_stop_hold == 0;
  const size_t sizes[] = {sizeof(typename std::decay<P>::type)..., 0};
  const size_t aligns[] = {alignof(typename std::decay<P>::type)..., 1};
  const void* ptrs[sizeof...(P) + 1];
  cn_tuple_ptrs(vals, ptrs, std::index_sequence_for<P...>{});
  const int op = cn_plan_rec_kernel((const void*)kern, grid, block, stream, (int)sizeof...(P), ptrs, sizes, aligns,
                                    marked ? cn_tl_stop_handle : -1);
  void* args[sizeof...(P) + 1];
  for (size_t i = 0; i < sizeof...(P); ++i) args[i] = const_cast<void*>(ptrs[i]);
  (void)hipLaunchKernel((const void*)kern, grid, block, args, 0, stream);
  cn_plan_rec_kernel_node(op, stream);
  if (marked) {
    (void)hipEventRecord(cn_tl_stop_event, stream);
    cn_tl_stop_recorded = 1;
  }
}
#define CN_LAUNCH(kern, grid, block, stream, ...)                                                              \
  do {                                                                                                          \
    if (cn_plan_recording) {                                                                                    \
      cn_plan_launch(kern, dim3(grid), dim3(block), (hipStream_t)(stream), __VA_ARGS__);                        \
    } else if (cn_tl_stop_event != nullptr && cn_tl_stop_hold == 0) {                                           \
      hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, (stream), (hipEvent_t) nullptr, cn_tl_stop_event, \
                            0, __VA_ARGS__);                                                                    \
      cn_tl_stop_recorded = 1;                                                                                  \
    } else {                                                                                                    \
      hipLaunchKernelGGL(kern, (grid), (block), 0, (stream), __VA_ARGS__);                                      \
    }                                                                                                           \
  } while (0)
#endif

// ---------------------------------------------------------------- vector types
typedef short s16x8 __attribute__((ext_vector_type(8)));   // 8 x bf16 bit patterns (16 B)
typedef short s16x4 __attribute__((ext_vector_type(4)));   // 4 x bf16 bit patterns (8 B)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct bf16_t { unsigned short v; };
struct f16_t { unsigned short v; };

// ---------------------------------------------------------------- bf16 <-> f32 (RNE)
__host__ __device__ __forceinline__ float cn_bf16_to_f32(unsigned short h) {
  union { unsigned int u; float f; } c;
  c.u = ((unsigned int)h) << 16;
  return c.f;
}
__host__ __device__ __forceinline__ unsigned short cn_f32_to_bf16(float f) {
  union { unsigned int u; float f; } c;
  c.f = f;
  unsigned int u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// Two fp32 -> packed bf16 (round to nearest even).  On gfx950 this is ONE v_cvt_pk_bf16_f32 (hipcc
// emits it from the native __bf16 conversion); the software form (~5 VALU per element) made the
// igemm epilogue cost more VALU issue than the MFMAs of a short reduction (profiles/r01_pmc_sq*).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CN_EMULATE)
typedef __bf16 cn_bf16x2_native __attribute__((ext_vector_type(2)));
typedef float cn_f32x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cn_pack_bf16x2(float lo, float hi) {
  cn_f32x2_native f = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f, cn_bf16x2_native));
}
#else
__host__ __device__ __forceinline__ unsigned int cn_pack_bf16x2(float lo, float hi) {
  return (unsigned int)cn_f32_to_bf16(lo) | ((unsigned int)cn_f32_to_bf16(hi) << 16);
}
#endif

// ---------------------------------------------------------------- fp16 <-> f32 (RNE; v_cvt_f16_f32 / v_cvt_f32_f16)
__host__ __device__ __forceinline__ float cn_f16_to_f32(unsigned short h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__host__ __device__ __forceinline__ unsigned short cn_f32_to_f16(float f) {
  return __builtin_bit_cast(unsigned short, (_Float16)f);
}
__host__ __device__ __forceinline__ unsigned int cn_pack_f16x2(float lo, float hi) {
  return (unsigned int)cn_f32_to_f16(lo) | ((unsigned int)cn_f32_to_f16(hi) << 16);
}

// element traits: T = float, bf16_t or f16_t
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kBytes = 4;
  static constexpr int kChunk = 4;  // elements per 16-byte chunk
  static constexpr int kCode = CN_F32;
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int kBytes = 2;
  static constexpr int kChunk = 8;
  static constexpr int kCode = CN_BF16;
};
template <> struct ElemTraits<f16_t> {
  static constexpr int kBytes = 2;
  static constexpr int kChunk = 8;
  static constexpr int kCode = CN_F16;
};
// two fp32 -> one packed dword of T (16-bit storage types), and back
template <typename T> __host__ __device__ __forceinline__ unsigned int cn_pack2(float lo, float hi);
template <> __host__ __device__ __forceinline__ unsigned int cn_pack2<bf16_t>(float lo, float hi) { return cn_pack_bf16x2(lo, hi); }
template <> __host__ __device__ __forceinline__ unsigned int cn_pack2<f16_t>(float lo, float hi) { return cn_pack_f16x2(lo, hi); }
template <> __host__ __device__ __forceinline__ unsigned int cn_pack2<float>(float lo, float) { return __builtin_bit_cast(unsigned int, lo); }
template <typename T> __host__ __device__ __forceinline__ void cn_unpack2(unsigned int v, float& lo, float& hi);
template <> __host__ __device__ __forceinline__ void cn_unpack2<bf16_t>(unsigned int v, float& lo, float& hi) {
  lo = __builtin_bit_cast(float, v << 16);
  hi = __builtin_bit_cast(float, v & 0xffff0000u);
}
template <> __host__ __device__ __forceinline__ void cn_unpack2<f16_t>(unsigned int v, float& lo, float& hi) {
  lo = cn_f16_to_f32((unsigned short)(v & 0xffffu));
  hi = cn_f16_to_f32((unsigned short)(v >> 16));
}
template <> __host__ __device__ __forceinline__ void cn_unpack2<float>(unsigned int v, float& lo, float& hi) {
  lo = __builtin_bit_cast(float, v);
  hi = 0.f;
}

// Unpack a 16-byte chunk into floats / pack floats into a chunk.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __host__ __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
    union { unsigned int u; float f; } x;
#pragma unroll
    for (int i = 0; i < 4; ++i) { x.u = c[i]; f[i] = x.f; }
  }
  __host__ __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
    union { unsigned int u; float f; } x;
#pragma unroll
    for (int i = 0; i < 4; ++i) { x.f = f[i]; c[i] = x.u; }
    return c;
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  __host__ __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = cn_bf16_to_f32((unsigned short)(c[i] & 0xffffu));
      f[2 * i + 1] = cn_bf16_to_f32((unsigned short)(c[i] >> 16));
    }
  }
  __host__ __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = cn_pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return c;
  }
};

template <> struct Chunk<f16_t> {
  static constexpr int N = 8;
  __host__ __device__ static __forceinline__ void unpack(const u32x4& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = cn_f16_to_f32((unsigned short)(c[i] & 0xffffu));
      f[2 * i + 1] = cn_f16_to_f32((unsigned short)(c[i] >> 16));
    }
  }
  __host__ __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = cn_pack_f16x2(f[2 * i], f[2 * i + 1]);
    return c;
  }
};

template <typename T> __host__ __device__ __forceinline__ float cn_load_elem(const T* p);
template <> __host__ __device__ __forceinline__ float cn_load_elem<float>(const float* p) { return *p; }
template <> __host__ __device__ __forceinline__ float cn_load_elem<bf16_t>(const bf16_t* p) {
  return cn_bf16_to_f32(p->v);
}
template <> __host__ __device__ __forceinline__ float cn_load_elem<f16_t>(const f16_t* p) { return cn_f16_to_f32(p->v); }
template <typename T> __host__ __device__ __forceinline__ void cn_store_elem(T* p, float v);
template <> __host__ __device__ __forceinline__ void cn_store_elem<f16_t>(f16_t* p, float v) { p->v = cn_f32_to_f16(v); }
template <> __host__ __device__ __forceinline__ void cn_store_elem<float>(float* p, float v) { *p = v; }
template <> __host__ __device__ __forceinline__ void cn_store_elem<bf16_t>(bf16_t* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CN_EMULATE)
  p->v = __builtin_bit_cast(unsigned short, (__bf16)v);   // v_cvt_pk_bf16_f32, RNE
#else
  p->v = cn_f32_to_bf16(v);
#endif
}

// Pivot of the centred BatchNorm statistics (the running mean): any finite value gives the same batch statistics up
// to rounding, so a non-finite one (a diverged step, a bad checkpoint) is replaced by 0 in every kernel that reads it
// - the batch statistics must not depend on the running buffers (they do not in the reference).
__host__ __device__ __forceinline__ float cn_pivot(float v) {
  return (v == v && v < 3.0e38f && v > -3.0e38f) ? v : 0.f;
}

// ---------------------------------------------------------------- fast division (host-precomputed)
// q = n / d for 0 <= n < 2^31 using one mulhi + shift.
struct FastDiv {
  unsigned int mul, shr, d;
};
static inline FastDiv cn_make_fastdiv(unsigned int d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
  unsigned int lg = 0;
  while ((1ull << lg) < d) ++lg;  // ceil(log2(d))
  unsigned int p = 31 + lg;
  unsigned long long m = ((1ull << p) + d - 1) / d;
  f.mul = (unsigned int)m;
  f.shr = p - 32;
  return f;
}
__host__ __device__ __forceinline__ unsigned int cn_fastdiv(unsigned int n, const FastDiv& f) {
  if (f.d == 1) return n;
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(n, f.mul) >> f.shr;
#else
  return (unsigned int)(((unsigned long long)n * f.mul) >> 32) >> f.shr;
#endif
}

// ---------------------------------------------------------------- MFMA wrappers
// Assumed gfx950 lane maps (wave64, lane l), pinned by cn_probe_mfma():
//   32x32x16 bf16 : A[i][k]: i = l&31, k = 8*(l>>5)+e (e=0..7);  B[k][j]: j = l&31, same k.
//   32x32x2  f32  : A[i][k]: i = l&31, k = l>>5;                 B[k][j]: j = l&31, k = l>>5.
//   C/D (both)    : j = l&31, i = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
#ifndef CN_EMULATE
typedef __bf16 cn_bf16x8_native __attribute__((ext_vector_type(8)));
typedef __bf16 cn_bf16x4_native __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 cn_mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cn_bf16x8_native, a),
                                                 __builtin_bit_cast(cn_bf16x8_native, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 cn_mfma_32x32x2_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
typedef _Float16 cn_f16x8_native __attribute__((ext_vector_type(8)));
// same lane map as the bf16 form (pinned by cn_probe_mfma_f16)
__device__ __forceinline__ f32x16 cn_mfma_32x32x16_f16(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cn_f16x8_native, a),
                                                __builtin_bit_cast(cn_f16x8_native, b), c, 0, 0, 0);
}
// LDS transpose read (ds_read_b64_tr_b16).  Within each 16-lane group, lane L supplies the
// address of 4 consecutive 16-bit elements in[L][0..3]; lane i receives
//   out[i][j] = in[4*j + (i>>2)][i&3]      (i, L = lane & 15; j = 0..3)
__device__ __forceinline__ s16x4 cn_lds_read_tr16_b64(const void* lds_ptr) {
  typedef __attribute__((address_space(3))) cn_bf16x4_native* lds_p;
  cn_bf16x4_native t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(lds_ptr));
  return __builtin_bit_cast(s16x4, t);
}
__device__ __forceinline__ float cn_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float cn_shfl_down(float v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ int cn_shfl_xor_i(int v, int mask) { return __shfl_xor(v, mask, 64); }
#else
static inline f32x16 cn_mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
  struct P { s16x8 a, b; } mine{a, b};
  const void* const* all = cn_emul::wave_gather(&mine);
  int l = cn_emul::lane();
  int j = l & 31;
  f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = 0.f;
    for (int k = 0; k < 16; ++k) {
      const P* pa = (const P*)all[i + 32 * (k >> 3)];
      const P* pb = (const P*)all[j + 32 * (k >> 3)];
      acc += cn_bf16_to_f32((unsigned short)pa->a[k & 7]) * cn_bf16_to_f32((unsigned short)pb->b[k & 7]);
    }
    d[r] += acc;
  }
  cn_emul::wave_release();
  return d;
}
static inline f32x16 cn_mfma_32x32x16_f16(s16x8 a, s16x8 b, f32x16 c) {
  struct P { s16x8 a, b; } mine{a, b};
  const void* const* all = cn_emul::wave_gather(&mine);
  int l = cn_emul::lane();
  int j = l & 31;
  f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = 0.f;
    for (int k = 0; k < 16; ++k) {
      const P* pa = (const P*)all[i + 32 * (k >> 3)];
      const P* pb = (const P*)all[j + 32 * (k >> 3)];
      acc += cn_f16_to_f32((unsigned short)pa->a[k & 7]) * cn_f16_to_f32((unsigned short)pb->b[k & 7]);
    }
    d[r] += acc;
  }
  cn_emul::wave_release();
  return d;
}
static inline f32x16 cn_mfma_32x32x2_f32(float a, float b, f32x16 c) {
  struct P { float a, b; } mine{a, b};
  const void* const* all = cn_emul::wave_gather(&mine);
  int l = cn_emul::lane();
  int j = l & 31;
  f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = d[r];
    for (int k = 0; k < 2; ++k)
      acc = fmaf(((const P*)all[i + 32 * k])->a, ((const P*)all[j + 32 * k])->b, acc);
    d[r] = acc;
  }
  cn_emul::wave_release();
  return d;
}
static inline s16x4 cn_lds_read_tr16_b64(const void* lds_ptr) {
  s16x4 mine;
  memcpy(&mine, lds_ptr, 8);
  const void* const* all = cn_emul::wave_gather(&mine);
  int l = cn_emul::lane();
  int g = l & ~15, i = l & 15;
  s16x4 out;
  for (int j = 0; j < 4; ++j) out[j] = (*(const s16x4*)all[g + 4 * j + (i >> 2)])[i & 3];
  cn_emul::wave_release();
  return out;
}
static inline float cn_shfl_xor(float v, int mask) {
  const void* const* all = cn_emul::wave_gather(&v);
  float r = *(const float*)all[cn_emul::lane() ^ mask];
  cn_emul::wave_release();
  return r;
}
static inline float cn_shfl_down(float v, int d) {
  const void* const* all = cn_emul::wave_gather(&v);
  int src = cn_emul::lane() + d;
  float r = src < 64 ? *(const float*)all[src] : v;
  cn_emul::wave_release();
  return r;
}
static inline int cn_shfl_xor_i(int v, int mask) {
  const void* const* all = cn_emul::wave_gather(&v);
  int r = *(const int*)all[cn_emul::lane() ^ mask];
  cn_emul::wave_release();
  return r;
}
#endif

// ---------------------------------------------------------------- bounds-checked buffer loads
// A raw buffer descriptor over [p, p+nbytes): 16-byte loads at a 32-bit byte offset return zeros
// when the offset is out of range, which gives the im2col zero padding / tile tails for free
// (offset CN_OOB = "invalid") and keeps all address arithmetic in 32 bits.
#define CN_OOB 0x80000000u
#ifndef CN_EMULATE
typedef __amdgpu_buffer_rsrc_t cn_buf_t;
__device__ __forceinline__ cn_buf_t cn_make_buf(const void* p, unsigned int nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)nbytes, 0x00020000);
}
__device__ __forceinline__ u32x4 cn_buf_ld16(cn_buf_t b, unsigned int off) {
  return __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 0);
}
// same with the non-temporal cache policy (aux = 2): operands every element of which is read by one workgroup only
__device__ __forceinline__ u32x4 cn_buf_ld16_nt(cn_buf_t b, unsigned int off) {
  return __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 2);
}
#else
struct cn_buf_t { const char* p; unsigned int n; };
static inline cn_buf_t cn_make_buf(const void* p, unsigned int nbytes) {
  cn_buf_t b; b.p = (const char*)p; b.n = nbytes; return b;
}
static inline u32x4 cn_buf_ld16(cn_buf_t b, unsigned int off) {
  u32x4 z = {0u, 0u, 0u, 0u};
  if ((unsigned long long)off + 16ull <= (unsigned long long)b.n) return *(const u32x4*)(b.p + off);
  return z;
}
static inline u32x4 cn_buf_ld16_nt(cn_buf_t b, unsigned int off) { return cn_buf_ld16(b, off); }
#endif

// Direct-to-LDS DMA form (buffer_load_dwordx4 ... lds): the 64 lanes of the wave write one
// contiguous KiB at `lds_wave_base` (wave-uniform) + lane*16; no VGPR round trip, no ds_write.
// Completion is tracked by vmcnt; hipcc waits for it at the next __syncthreads().
#ifndef CN_EMULATE
__device__ __forceinline__ void cn_buf_ld16_lds(cn_buf_t b, unsigned int off, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)off,
                                           0, 0, 0);
}
__device__ __forceinline__ void cn_buf_ld16_lds_nt(cn_buf_t b, unsigned int off, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)off,
                                           0, 0, 2);
}
#else
static inline void cn_buf_ld16_lds(cn_buf_t b, unsigned int off, void* lds_wave_base) {
  *(u32x4*)((char*)lds_wave_base + cn_emul::lane() * 16) = cn_buf_ld16(b, off);
}
static inline void cn_buf_ld16_lds_nt(cn_buf_t b, unsigned int off, void* lds_wave_base) {
  cn_buf_ld16_lds(b, off, lds_wave_base);
}
#endif

// Counted wait on outstanding vector-memory operations (LDS-DMA included) and a raw workgroup barrier
// without the fence of __syncthreads(): lets DMA tiles stay in flight across barriers (multi-stage
// ring).  N must be a literal.  The emulator's DMA is synchronous, so the wait is a no-op there.
#ifndef CN_EMULATE
#define CN_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
__device__ __forceinline__ void cn_raw_barrier() { __builtin_amdgcn_s_barrier(); }
#else
#define CN_WAIT_VMCNT(N) do { } while (0)
static inline void cn_raw_barrier() { cn_emul::sync_threads(); }
#endif

// A value the program knows to be the same in every lane of the wave (e.g. threadIdx.x >> 6): tells the compiler so
// (v_readfirstlane -> SGPR), which keeps wave-uniform bases (the M0 operand of an LDS-DMA) out of VGPRs and avoids the
// waterfall loops hipcc otherwise builds around them.
#ifndef CN_EMULATE
__device__ __forceinline__ int cn_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
static inline int cn_uniform(int v) { return v; }
#endif

// Wave-level rendezvous for data exchanged between the lanes of ONE wave through LDS (a wave's LDS operations execute
// in order, so the hardware needs no instruction; the compiler must not move LDS accesses across it, and the emulator,
// which runs every lane as its own fiber, has to line the lanes up).
#ifndef CN_EMULATE
__device__ __forceinline__ void cn_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#else
static inline void cn_wave_sync() { int x = 0; cn_emul::wave_gather(&x); cn_emul::wave_release(); }
#endif

// Scheduling fence: nothing is moved across it by hipcc's machine scheduler (used to keep the
// fragment reads of the NEXT k-step ahead of the MFMAs of the current one).
#ifndef CN_EMULATE
__device__ __forceinline__ void cn_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#else
static inline void cn_sched_fence() {}
#endif

// 16-byte global / LDS accessors
__host__ __device__ __forceinline__ u32x4 cn_ld16(const void* p) { return *(const u32x4*)p; }
__host__ __device__ __forceinline__ void cn_st16(void* p, const u32x4& v) { *(u32x4*)p = v; }
// streaming 16-byte global store (big activation / gradient tensors written once and read back only after
// they have left the L2): a non-temporal store; -DCN_NO_NT_STORES makes it a plain store (A/B: +0.45 % on the
// ResNet-50 step, profiles/README.md).
template <int SITE = 0>
__host__ __device__ __forceinline__ void cn_st16_stream(void* p, const u32x4& v) {
#if !defined(CN_NO_NT_STORES) && defined(__HIP_DEVICE_COMPILE__) && !defined(CN_EMULATE)
#ifdef CN_PLAIN_STORE_SITE   /* A/B builds: one site back to a plain store */
  if (SITE == CN_PLAIN_STORE_SITE) { *(u32x4*)p = v; return; }
#endif
  __builtin_nontemporal_store(v, (u32x4*)p);
#else
  *(u32x4*)p = v;
#endif
}
// streaming 16-byte global load (read-once passes over big tensors: BatchNorm apply / reduce, epilogue
// operands): non-temporal, so the pass does not evict what the next kernels re-read from L2 / Infinity Cache;
// -DCN_NO_NT_LOADS makes it a plain load (A/B: +1.5 % on the ResNet-50 step, profiles/README.md).
template <int SITE = 0>
__host__ __device__ __forceinline__ u32x4 cn_ld16_stream(const void* p) {
#if !defined(CN_NO_NT_LOADS) && defined(__HIP_DEVICE_COMPILE__) && !defined(CN_EMULATE)
#ifdef CN_PLAIN_LOAD_SITE   /* A/B builds: one site back to a plain load */
  if (SITE == CN_PLAIN_LOAD_SITE) return *(const u32x4*)p;
#endif
  return __builtin_nontemporal_load((const u32x4*)p);
#else
  return *(const u32x4*)p;
#endif
}
__host__ __device__ __forceinline__ u32x4 cn_zero16() {
  u32x4 z = {0u, 0u, 0u, 0u};
  return z;
}

// XCD-aware (8 XCDs) bijective remap of a linear workgroup id so that consecutive logical tiles
// run on the same XCD (shared private L2).  Pure speed: any placement is still correct.
__host__ __device__ __forceinline__ unsigned int cn_xcd_remap(unsigned int bid, unsigned int nwg) {
  const unsigned int nx = 8;
  if (nwg < nx * 2) return bid;
  unsigned int xcd = bid % nx, slot = bid / nx;
  unsigned int q = nwg / nx, r = nwg % nx;
  unsigned int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}
