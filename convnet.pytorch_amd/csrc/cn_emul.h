// cn_emul.h -- TEST-ONLY SIMT emulator shim.
//
// This header lets the *unmodified* HIP kernel sources in this directory be compiled as plain
// host C++ (clang++ -DCN_EMULATE) and executed on a CPU, one workgroup at a time, with every
// GPU thread running as a ucontext fiber.  __syncthreads() and the wave-level collectives
// (MFMA, ds_read_tr, shuffles) are implemented as fiber rendezvous.  It exists because the
// development container has no GPU: the kernels' index arithmetic (im2col gather, LDS swizzles,
// MFMA fragment maps, split reductions) is exercised here against the oracle on tiny shapes
// before any GPU minute is spent.
//
// It is NOT a product path: `_lib.py` never loads the emulator library on its own, the `-m gpu`
// tests / bench.py / smoke() always go through libconvnet_hip.so, and the emulated MFMA / tr-read
// lane maps are *assumptions* that tests/test_gpu_probe.py pins on real gfx950 hardware.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;

namespace cn_emul {
extern dim3 g_tid, g_bid, g_bdim, g_gdim;
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_threads();
// Wave collective rendezvous: every lane of a wave calls wave_gather(ptr-to-its-payload) and
// receives the table of all 64 lanes' payload pointers; it must call wave_release() once it
// has finished reading the other lanes' payloads.
const void* const* wave_gather(const void* mine);
void wave_release();
inline int lane() { return (int)(g_tid.x & 63u); }
}  // namespace cn_emul

#define threadIdx (cn_emul::g_tid)
#define blockIdx (cn_emul::g_bid)
#define blockDim (cn_emul::g_bdim)
#define gridDim (cn_emul::g_gdim)
#define __syncthreads() cn_emul::sync_threads()

static inline void __threadfence() {}
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
