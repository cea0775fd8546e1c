// loss.hip -- softmax cross-entropy (forward + gradient in one pass), top-1/top-5 hit counting
// and on-device meter accumulation, gfx950.
//
// Replaces criterion(output, target) = CrossEntropyLoss(**{smooth_eps}) (/root/reference
// main.py:231-235, trainer.py:143; class lives in the un-vendored utils submodule and equals
// F.cross_entropy for smooth_eps = 0), its backward, and utils.meters.accuracy(output, target,
// topk=(1,5)) (trainer.py:224).  The reference pulls loss / prec@k to the host every step
// (trainer.py:153,225-229); here they are accumulated on the device and read when a report is due.
#include "cn_common.h"
#include "cn_api_internal.h"

__device__ __forceinline__ float ls_block_reduce(float v, float* red, bool is_max) {
  // 256 threads = 4 waves
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    float o = cn_shfl_xor(v, m);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// One workgroup per sample.  row_out[b] = {loss_b, top1_hit, top5_hit}
template <typename TG>
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, const long long* target,
                                                        TG* dlogits, float* row_out, int B, int K,
                                                        float gscale, const float* gscale_dev,
                                                        float smooth_eps) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (size_t)b * K;
  const int tgt = (int)target[b];
  float mx = -INFINITY;
  for (int k = tid; k < K; k += 256) mx = fmaxf(mx, row[k]);
  mx = ls_block_reduce(mx, red, true);
  float se = 0.f, sl = 0.f;
  const float lt = row[tgt];
  float rank = 0.f;
  for (int k = tid; k < K; k += 256) {
    const float v = row[k];
    se += expf(v - mx);
    sl += v;
    // rank of the target among the logits; ties resolved towards the lower index (topk order)
    if (v > lt || (v == lt && k < tgt)) rank += 1.f;
  }
  se = ls_block_reduce(se, red, false);
  sl = ls_block_reduce(sl, red, false);
  rank = ls_block_reduce(rank, red, false);
  const float lse = mx + logf(se);
  if (dlogits != nullptr) {
    if (gscale_dev != nullptr) gscale *= gscale_dev[0];  // upstream d(loss) scalar, read on device
    const float on = 1.f - smooth_eps, off = smooth_eps / (float)K;
    for (int k = tid; k < K; k += 256) {
      const float p = expf(row[k] - lse);
      const float t = (k == tgt ? on : 0.f) + off;
      cn_store_elem<TG>(dlogits + (size_t)b * K + k, (p - t) * gscale);
    }
  }
  if (tid == 0) {
    const float nll = lse - lt;
    const float smooth = lse - sl / (float)K;  // mean_k(-log p_k)
    row_out[3 * b] = (1.f - smooth_eps) * nll + smooth_eps * smooth;
    row_out[3 * b + 1] = rank < 1.f ? 1.f : 0.f;
    row_out[3 * b + 2] = rank < 5.f ? 1.f : 0.f;
  }
}

// Fixed-order reduction over the batch.
//   step_out[0..2] = mean loss, prec@1 (%), prec@5 (%) of this batch (reference meter .val)
//   meters[0..3]  += loss*B, prec1*B, prec5*B, B                 (reference meter .sum / .count)
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* row_out, int B, float* step_out,
                                                         float* meters) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float l = 0.f, c1 = 0.f, c5 = 0.f;
  for (int b = tid; b < B; b += 256) { l += row_out[3 * b]; c1 += row_out[3 * b + 1]; c5 += row_out[3 * b + 2]; }
  l = ls_block_reduce(l, red, false);
  c1 = ls_block_reduce(c1, red, false);
  c5 = ls_block_reduce(c5, red, false);
  if (tid == 0) {
    const float ml = l / (float)B, p1 = 100.f * c1 / (float)B, p5 = 100.f * c5 / (float)B;
    if (step_out != nullptr) { step_out[0] = ml; step_out[1] = p1; step_out[2] = p5; }
    if (meters != nullptr) { meters[0] += ml * (float)B; meters[1] += p1 * (float)B; meters[2] += p5 * (float)B; meters[3] += (float)B; }
  }
}

extern "C" int cn_softmax_ce(const float* logits, const long long* target, void* dlogits, int grad_dtype,
                             float* row_scratch, float* step_out, float* meters, int B, int K, float gscale,
                             const float* gscale_dev, float smooth_eps, void* stream_) {
  if (B <= 0 || K <= 0) { cn_set_error("softmax_ce: empty"); return CN_ESHAPE; }
  hipStream_t stream = (hipStream_t)stream_;
  if (dlogits != nullptr && grad_dtype == CN_BF16)
    CN_LAUNCH(softmax_ce_kernel<bf16_t>, dim3((unsigned)B), dim3(256), stream, logits, target, (bf16_t*)dlogits,
              row_scratch, B, K, gscale, gscale_dev, smooth_eps);
  else if (dlogits != nullptr && grad_dtype == CN_F16)
    CN_LAUNCH(softmax_ce_kernel<f16_t>, dim3((unsigned)B), dim3(256), stream, logits, target, (f16_t*)dlogits,
              row_scratch, B, K, gscale, gscale_dev, smooth_eps);
  else
    CN_LAUNCH(softmax_ce_kernel<float>, dim3((unsigned)B), dim3(256), stream, logits, target, (float*)dlogits,
              row_scratch, B, K, gscale, gscale_dev, smooth_eps);
  CN_LAUNCH(loss_reduce_kernel, dim3(1), dim3(256), stream, (const float*)row_scratch, B, step_out, meters);
  return cn_check_launch("softmax_ce");
}
