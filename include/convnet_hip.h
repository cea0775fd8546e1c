/* convnet_hip.h -- C ABI of libconvnet_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * data-parallel training hot path of eladhoffer/convNet.pytorch
 * (trainer.Trainer.train/forward/_step over models/resnet.py).
 *
 * The reference is 100 % Python and has no FFI: every entry point below replaces a *PyTorch op*
 * the reference calls (file:line citations are into the reference tree).  A maintainer binds the
 * library with ctypes; `convnet.pytorch_amd/_lib.py` is that binding.
 *
 * Conventions
 *   - Plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise.
 *   - The caller owns every buffer including workspaces; the library never allocates or frees
 *     device memory and never synchronises the device.  State it does keep, all of it process-side: (i) the table
 *     of tuning knobs behind cn_set_option (global, read at launch time: kernel-variant A/B only, results are the
 *     same for every value); (ii) the per-thread completion-event arming of cn_stream_arm / cn_stream_disarm, which
 *     changes HOW every launch of that thread is issued until disarmed (the last kernel of each entry point carries
 *     the armed event); (iii) one communicator handle per cn_comm_init, owned by the caller.
 *   - Size limit: gather sources and filters are addressed through 32-bit buffer descriptors, so a single conv /
 *     linear operand (N*H*W*C*elemsize, Co*R*S*Ci*elemsize) must be < 2 GiB; larger operands are refused with
 *     CN_ESHAPE (ResNet-50 b=256: the largest is 822 MB).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous and re-entrant.
 *   - Every function returns 0 (CN_OK) or a negative CN_E* code; cn_last_error() returns the
 *     thread-local reason.
 *   - Activations are NHWC, filters KRSC ([Co][kh][kw][Ci]); the channel count of every tensor a
 *     kernel vector-loads must be a multiple of the 16-byte chunk (8 bf16 / 4 fp32 elements).
 *   - dtype: CN_F32 = 0, CN_BF16 = 1, CN_F16 = 2 (16-bit storage, fp32 accumulation; the simulated-8-bit and int8
 *     entry points take CN_F32 / CN_BF16 only).
 */
#ifndef CONVNET_HIP_H
#define CONVNET_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CN_OK 0
#define CN_EINVAL (-1)
#define CN_ESHAPE (-2)
#define CN_EHIP (-3)
#define CN_EWORKSPACE (-4)
#define CN_ERCCL (-5)
#define CN_F32 0
#define CN_BF16 1
#define CN_F16 2 /* IEEE half storage, fp32 accumulation (the reference's --dtype half) */

const char* cn_last_error(void);
const char* cn_build_info(void);
/* name of the GEMM-class kernel instantiation the last cn_conv2d_* call of this thread launched (as rocprofv3
 * prints it, minus "void " and the parameter list); measurement code labels its timings with it */
const char* cn_last_kernel_name(void);
/* the names of ALL GEMM-class launches of this thread since the log was cleared, ';'-separated in launch order (one
 * entry point may launch several instantiations: a strided dgrad dispatches each output-parity class on its own
 * reduction length).  clear != 0 empties the log.  Lets measurement code count launches per kernel as rocprofv3 does. */
const char* cn_kernel_log(int clear);
int cn_is_emulator(void); /* 1 only in the TEST-ONLY CPU emulator build */
/* kernel-variant tuning knobs for A/B measurement (e.g. "igemm_stages" = 1|2); results never change */
int cn_set_option(const char* name, int value);
/* stream plumbing: `to` waits for everything queued on `from` so far (one device-scope event from a ring; replaces
 * torch's Stream.wait_stream between the backward chain and the weight-gradient side stream, trainer.py:151-159) */
int cn_stream_fork(void* from_stream, void* to_stream);
/* the same hand-off without a marker in the producer's queue: between cn_stream_arm() (returns a handle >= 0) and
 * cn_stream_disarm() (returns 1 if a kernel was launched in between) every kernel this thread launches signals the
 * handle's event on completion; cn_stream_wait_mark(handle, s) makes stream s wait for the last of them */
int cn_stream_arm(void);
int cn_stream_disarm(void);
int cn_stream_wait_mark(int handle, void* to_stream);
/* step timer (Trainer's graph = auto policy): mark = record the next ring event behind the stream's work (timing on, no
 * system-scope fence) together with the caller's tag; poll = time in ms between the oldest two marks once both have
 * completed, with their tags (returns 1; a negative period = the pair could not be timed), never waits (returns 0).
 * Watchers share the ring and keep the periods whose two tags are their own. */
int cn_step_timer_mark(void* stream, long long tag);
int cn_step_timer_poll(float* period_ms, long long* tag_prev, long long* tag_cur);

/* ---- launch plans (csrc/plan.hip): the per-batch host loop of trainer.py:106-177 issued from one C call ---------
 * cn_plan_begin starts a process-wide recording (the autograd engine's thread launches too): until cn_plan_end every
 * kernel launch of this library is logged with a private copy of its arguments and the stream it went to, every
 * cn_stream_fork / cn_stream_arm mark / cn_stream_wait_mark hand-off with its two ends, and every cn_comm_allreduce_bucket
 * / _join / _allreduce call is logged INSTEAD of issued.  The recording is meant to run under stream capture of
 * `main_stream` (nothing executes; the capturing allocator keeps the step's addresses for the plan): then
 * cn_plan_import_graph(plan, hipGraph_t) pairs every logged launch with its graph node and imports the nodes somebody
 * else put on the streams (returns their number; < 0: the step holds something a plan cannot re-issue).
 * cn_plan_replay issues the whole step: the same launches on the same streams in the same order, marks as kernel
 * completion events, RCCL calls live.  cn_plan_info: counts[8] = ops, own launches, imported nodes, events, hand-offs,
 * communicator calls, streams, replays.  A plan is tied to the buffers and streams it was recorded on. */
int cn_plan_begin(void** plan, void* main_stream);
int cn_plan_end(void* plan);
int cn_plan_import_graph(void* plan, void* hip_graph);
int cn_plan_replay(void* plan);
int cn_plan_info(void* plan, long long* counts);
/* rebindable batch: bind finds the argument words of the plan's own launches that point into [base, base + bytes) (returns
 * how many); set_input re-points them at another buffer of the same layout, so a replay reads the caller's batch in place */
int cn_plan_bind_input(void* plan, int slot, const void* base, size_t bytes);
int cn_plan_set_input(void* plan, int slot, const void* base);
const char* cn_plan_describe(void* plan);
int cn_plan_destroy(void* plan);

/* ---- nn.Conv2d / nn.Linear (models/resnet.py:75-78,126-132,178-179,226-227,242) ------------- */
/* y[N,P,Q,K] = conv(x[N,H,W,C], w[K,R,S,C]) (+bias[K]) (ReLU optional); out_f32 writes fp32
 * regardless of dtype (used for the classifier logits).  Linear = 1x1 conv on a 1x1 image. */
int cn_conv2d_fwd(const void* x, const void* w_krsc, void* y, const float* bias, int N, int H, int W, int C,
                  int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                  int out_f32, int relu, void* stream);
/* Same convolution, additionally emitting the BatchNorm statistics partials of its own output from the
 * epilogue: partial[row][0:K] = per-channel sum, partial[row][K:2K] = sum of squares of the stored
 * outputs of pixel tile `row` (cn_conv2d_bnstats_rows(N*P*Q) rows of 2*K floats).  Fuses the
 * nn.Conv2d -> nn.BatchNorm2d pairs of models/resnet.py:141-165 so the statistics pass never re-reads y. */
int cn_conv2d_bnstats_rows(long long M);
int cn_conv2d_fwd_bnstats(const void* x, const void* w_krsc, void* y, const float* bias, int N, int H, int W,
                          int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                          int dtype, int relu, float* partial, int partial_rows, void* stream);
/* centred form of the two calls above / below: the partial rows hold sum (y - pivot[k]) | sum (y - pivot[k])^2 with
 * pivot = the consumer BatchNorm's running_mean (K floats), so that the variance is not the difference of two numbers of
 * size mean^2 (ATen's batch_norm_stats uses Welford for the same reason, nn.BatchNorm2d of models/resnet.py:128-133);
 * cn_bn_fwd_train_partials_centered takes exactly such rows and the same running_mean before it updates it */
int cn_conv2d_fwd_bnstats_centered(const void* x, const void* w_krsc, void* y, const float* bias, int N, int H, int W,
                                   int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                                   int dtype, int relu, float* partial, int partial_rows, const float* pivot,
                                   void* stream);
/* 3x3 / stride-1 / pad-1 convolution with 64 -> 64 channels (the first stage's conv2, /root/reference
 * models/resnet.py:126-132) as a halo kernel (csrc/conv3x3.hip): a band's input rows are staged in LDS once and the MFMA
 * fragments of all nine taps are read straight out of that halo; filter in registers.  flip = 0: forward, w = KRSC
 * filter; flip = 1: data gradient, x = dy, w = CRSK filter.  partial (optional, forward): cn_conv3x3_c64_rows(N, H) rows
 * of 128 floats [sum | sum of squares] for cn_bn_fwd_train_partials.  Same output bits as cn_conv2d_fwd /
 * cn_conv2d_dgrad.  cn_conv3x3_c64_ok: W <= 56, 16-bit storage. */
int cn_conv3x3_c64_ok(int H, int W, int C, int K, int dtype);
int cn_conv3x3_c64_rows(int N, int H);
int cn_conv3x3_c64(const void* x, const void* w, void* y, int N, int H, int W, int dtype, int flip, float* partial,
                   int partial_rows, void* stream);
/* "Lazy a" for the 3x3 halo kernel (forward): the input is the INPUT bn_y of the BatchNorm in front of the convolution;
 * a = relu?(bn_y * scale + shift) is formed on the way into the halo (zero padding pads a) and written to a_out.  Replaces
 * the bn1 -> relu -> conv2 sequence of /root/reference models/resnet.py:122-128 in the first stage. */
int cn_conv3x3_c64_lazya(const void* bn_y, const float* stats, int relu, void* a_out, const void* w, void* y, int N, int H,
                         int W, int dtype, float* partial, int partial_rows, void* stream);
/* The 7x7 / stride-2 stem (/root/reference models/resnet.py:226) on the pixel-pair image of cn_nchw_to_pairs as a halo
 * kernel (csrc/stem.hip): y[n][oy][ox][k] = sum_{r<7, s2<4, e<8} xp[n][2*oy + r][ox + s2][e] * wp[k][r][s2][e], i.e.
 * cn_conv2d_fwd_bnstats on the pair image (R = 7, S = 4, stride (2, 1), no padding) with 64 output channels; the input
 * rows a band of output rows needs are staged in LDS once and the MFMA fragments are read straight out of that halo.
 * wp: cn_weight_prep_pairs.  partial: cn_stem_fwd_rows(N, P) rows of 128 floats [sum | sum of squares] of the stored
 * outputs.  Same output bits as the tiled kernel.  cn_stem_fwd_ok: shapes it is built for. */
int cn_stem_fwd_ok(int K, int R, int S2, int Jp, int dtype);
int cn_stem_fwd_rows(int N, int P);
int cn_stem_fwd(const void* xp, const void* wp, void* y, int N, int Hp, int Jp, int dtype, float* partial,
                int partial_rows, void* stream);
/* The stem's weight gradient on the pair image as a halo kernel: dwp [64][7][4][8] fp32 (what cn_wgrad_unpack_pairs takes)
 * = beta*dwp + scale * sum_{n,oy,ox} dy[n][oy][ox][k] * xp[n][2*oy + r][ox + s2][e]; both MFMA operands are LDS transpose
 * reads, the activation straight out of the band's halo.  cn_conv2d_wgrad on the pair image up to fp32 summation order.
 * Needs (Jp - 3) % 16 == 0 (cn_stem_wgrad_ok). */
int cn_stem_wgrad_ok(int K, int R, int S2, int Jp, int dtype);
size_t cn_stem_wgrad_workspace(int N, int Hp);
int cn_stem_wgrad(const void* xp, const void* dy, float* dwp, int N, int Hp, int Jp, int dtype, float beta, float scale,
                  void* workspace, size_t ws_bytes, void* stream);
/* dx[N,H,W,C] from dy[N,P,Q,K] and the transposed filter w_crsk[C][R][S][K]
 * (written by cn_weight_prep).  Strided convs run one launch per output-parity class. */
/* "Lazy z" forward: the input of this 1x1 / stride-1 convolution (K <= 128 output channels, C <= 512) is the output of
 * a residual junction that has been finalised (cn_bn_fwd_train* with z = NULL) but not applied:
 *   z = relu?( bn_y*scale[c] + shift[c] + r ),  r = res  or, with res_stats, r = round_T(res*rscale[c] + rshift[c])
 * (/root/reference models/resnet.py:141-165: bn3 -> += residual -> relu, then the next block's conv1).  The kernel forms z
 * on its operand load, stores it to z (ReLU bits to z_mask, optional) and convolves: same bits as the apply pass followed
 * by cn_conv2d_fwd_bnstats(_centered), minus that pass and this convolution's re-read of z.  stats / res_stats: the 4*C
 * floats of the junction / shortcut BatchNorm; partial / partial_rows / pivot as cn_conv2d_fwd_bnstats(_centered), optional. */
int cn_conv2d_fwd_lazyz(const void* bn_y, const void* res, const float* stats, const float* res_stats, int relu, void* z,
                        unsigned char* z_mask, const void* w_krsc, void* y, int N, int H, int W, int C, int K, int dtype,
                        float* partial, int partial_rows, const float* pivot, void* stream);
int cn_conv2d_dgrad(const void* dy, const void* w_crsk, void* dx, const void* addend /*optional: dx += addend,
                    the residual-branch gradient of models/resnet.py:162 folded into the epilogue*/, int N, int H,
                    int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                    int dtype, int out_f32, void* stream);
/* Data gradient fused with the reduction half of the BatchNorm backward of the layer that produced the
 * convolution's input x = act(BN(bn_y) [+ residual]) (models/resnet.py:141-165): stores g = dx * relu_mask
 * (mask bits from bn_mask, or recomputed from bn_y*scale+shift > 0 when bn_relu and bn_mask == NULL) and one
 * partial row [sum g | sum g*xhat] (2*C floats) per 128-pixel tile for cn_bn_bwd_partials.
 * bn_coef = the 4*C floats cn_bn_fwd_train wrote. */
int cn_conv2d_dgrad_bnbwd_rows(int N, int H, int W, int C, int stride_h, int stride_w);
int cn_conv2d_dgrad_bnbwd(const void* dy, const void* w_crsk, void* g, const void* addend, int N, int H, int W,
                          int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                          const void* bn_y, const unsigned char* bn_mask, const float* bn_coef, int bn_relu,
                          float* partial, int partial_rows, void* stream);
/* cn_conv2d_dgrad / cn_conv2d_dgrad_bnbwd with a SUBSAMPLED addend (addend_sub = 2): `addend` is
 * [N][(H+1)/2][(W+1)/2][C], the values at the even (h, w) pixels of a gradient that is zero everywhere else - the
 * input gradient of the stride-2 1x1 projection shortcut (models/resnet.py:176-181), computed as a stride-1 dgrad on
 * the coarse grid; the three quarters of zeros are neither written nor re-read */
int cn_conv2d_dgrad_sa(const void* dy, const void* w_crsk, void* dx, const void* addend, int addend_sub, int N, int H,
                       int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                       int out_f32, void* stream);
int cn_conv2d_dgrad_bnbwd_sa(const void* dy, const void* w_crsk, void* g, const void* addend, int addend_sub, int N,
                             int H, int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w,
                             int dtype, const void* bn_y, const unsigned char* bn_mask, const float* bn_coef,
                             int bn_relu, float* partial, int partial_rows, void* stream);
/* The block's last 1x1 convolution (conv3 / the stride-1 projection, C -> K = 64 / 128 -> 256, 128 -> 512 channels) as a
 * persistent streaming kernel with the statistics partials of cn_conv2d_fwd_bnstats kept in registers (one row per
 * workgroup: cn_conv1x1_stream_fwd_rows; partial may be NULL).  Output bits = cn_conv2d_fwd's. */
int cn_conv1x1_stream_fwd_ok(int C, int K, int dtype);
int cn_conv1x1_stream_fwd_rows(int N, int H, int W, int K);
int cn_conv1x1_stream_fwd(const void* x, const void* w_krsc, void* y, int N, int H, int W, int C, int K, int dtype,
                          float* partial, int partial_rows, void* stream);
/* "Lazy a": cn_conv1x1_stream_fwd reading the INPUT bn_y of the BatchNorm in front of the convolution (statistics
 * finalised, stats = [mean | invstd | scale | shift]); the kernel forms a = relu?(bn_y * scale + shift) on its operand path
 * (the bits of cn_bn_fwd_train's apply pass), writes it to a_out [M][C] and multiplies: the inner BatchNorm's apply pass disappears.  Replaces
 * the bn2 -> relu -> conv3 sequence of /root/reference models/resnet.py:126-132. */
int cn_conv1x1_stream_fwd_lazya(const void* bn_y, const float* stats, int relu, void* a_out, const void* w_krsc, void* y,
                                int N, int H, int W, int C, int K, int dtype, float* partial, int partial_rows,
                                void* stream);
/* The same operation for the LARGE junctions as one persistent streaming kernel (csrc/junction.hip): 1x1 / stride-1 /
 * unpadded convolution with K -> C channels of an instantiated shape (cn_conv2d_dgrad_junction_ok: 64 or 128 -> 256,
 * 128 -> 512; 16-bit storage), ReLU bits (bn_mask) and an addend required (addend_sub as cn_conv2d_dgrad_sa).  The
 * filter stays in registers, a workgroup owns a pixel range and all C channels, epilogue operands are requested a stage
 * ahead; partial: cn_conv2d_dgrad_junction_rows(N, H, W, C) rows of 2*C floats (one per workgroup) for
 * cn_bn_bwd_partials.  g: the bits of cn_conv2d_dgrad_bnbwd_sa; the partial sums differ by fp32 association. */
int cn_conv2d_dgrad_junction_ok(int C, int K, int dtype);
int cn_conv2d_dgrad_junction_rows(int N, int H, int W, int C);
int cn_conv2d_dgrad_junction_rows_k(int N, int H, int W, int C, int K);   /* ... for K -> C channels (K = 256: per channel slice) */
int cn_conv2d_dgrad_junction(const void* dy, const void* w_crsk, void* g, const void* addend, int addend_sub, int N, int H,
                             int W, int C, int K, int dtype, const void* bn_y, const unsigned char* bn_mask,
                             const float* bn_coef, float* partial, int partial_rows, void* stream);
/* dw[K,R,S,C_real] (fp32) = beta*dw + scale * sum_pixels dy (x) x ; split reduction through
 * `workspace` (cn_conv2d_wgrad_workspace bytes), fixed summation order. */
size_t cn_conv2d_wgrad_workspace(int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w,
                                 int pad_h, int pad_w, int dtype);
int cn_conv2d_wgrad(const void* x, const void* dy, float* dw_krsc, int C_real, int N, int H, int W, int C,
                    int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                    float beta, float scale, void* workspace, size_t ws_bytes, void* stream);
/* "Lazy dy" backward of a convolution that follows a training-mode BatchNorm in the backward direction (the last
 * convolution of a residual branch: models/resnet.py:141-165, bn3(conv3(.))): the gradient w.r.t. the BatchNorm input,
 *     dy[m][k] = c1[k]*g[m][k] + c2[k]*y[m][k] + c3[k],
 * is NOT materialised.  cn_bn_bwd_partials(dy = NULL) runs the finalize only and leaves coef = [c1 | c2 | c3] (3*K
 * floats); these two entry points form dy on their operand loads from g (masked gradient w.r.t. the BatchNorm output)
 * and bn_y (BatchNorm input) with the apply kernel's operation order and rounding, so the results carry the same bits
 * as cn_bn_bwd_partials(dy) + cn_conv2d_dgrad / cn_conv2d_wgrad on the register-staged kernels - minus one write and
 * two reads of dy.  K <= 512 gradient channels; 16-bit or fp32 storage. */
int cn_conv2d_dgrad_lazy(const void* g, const void* bn_y, const float* coef, const void* w_crsk, void* dx, int N, int H,
                         int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                         void* stream);
int cn_conv2d_wgrad_lazy(const void* x, const void* g, const void* bn_y, const float* coef, float* dw_krsc, int C_real,
                         int N, int H, int W, int C, int K, int R, int S, int stride_h, int stride_w, int pad_h,
                         int pad_w, int dtype, float beta, float scale, void* workspace, size_t ws_bytes, void* stream);
/* cn_conv2d_dgrad_lazy for a 1x1 / stride-1 convolution with K = 512 gradient channels and C = 128 input channels (conv3 of
 * the second stage) as a persistent streaming kernel (csrc/junction.hip: jdlazy_kernel).  Same output bits. */
int cn_conv2d_dgrad_lazy_stream_ok(int C, int K, int dtype);
int cn_conv2d_dgrad_lazy_stream(const void* g, const void* bn_y, const float* coef, const void* w_crsk, void* dx, int N,
                                int H, int W, int C, int K, int dtype, void* stream);
/* Junction pair: cn_conv2d_dgrad_lazy + cn_conv2d_wgrad_lazy of one 1x1 / stride-1 convolution in ONE pass over g and
 * bn_y (the two junction-sized reads of each are shared; /root/reference reaches both through loss.backward(),
 * trainer.py:162, for models/resnet.py:126-132's conv3 and :176-181's projection).  dx: the bits of
 * cn_conv2d_dgrad_lazy; dw_krsc = beta*dw + scale*wgrad, fp32 summation order of its own pixel ranges.  Instantiated
 * shapes: cn_conv2d_bwd1x1_lazy_ok(C, K, dtype) != 0 (K = 256 output, C = 64 input channels, 16-bit storage). */
int cn_conv2d_bwd1x1_lazy_ok(int C, int K, int dtype);
size_t cn_conv2d_bwd1x1_lazy_workspace(int N, int H, int W, int C, int K);
int cn_conv2d_bwd1x1_lazy(const void* x, const void* g, const void* bn_y, const float* coef, const void* w_crsk, void* dx,
                          float* dw_krsc, int N, int H, int W, int C, int K, int dtype, float beta, float scale,
                          void* workspace, size_t ws_bytes, void* stream);

/* ---- nn.BatchNorm2d (+ fused residual add + ReLU) (models/resnet.py:128-134,141-165) -------- */
size_t cn_bn_workspace(int M, int C, int dtype);
/* stats_out: 4*C floats = [batch mean | 1/sqrt(var+eps) | scale | shift]; M = N*H*W rows.
 * z == NULL: statistics / coefficients / running-stat update only (a fused consumer applies them). */
/* relu_mask (optional, M*C/chunk bytes): one bit per output recording z > 0, written when a residual is
 * added before the ReLU so that backward need not re-read z. */
int cn_bn_fwd_train(const void* y, const void* residual, void* z, unsigned char* relu_mask, const float* gamma,
                    const float* beta,
                    float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                    float eps, float* stats_out, int M, int C, int relu, int dtype, void* workspace,
                    size_t ws_bytes, void* stream);
/* cn_bn_fwd_train with the statistics partials supplied by the producer of y (cn_conv2d_fwd_bnstats):
 * partial = [nrb][2*C] floats. */
int cn_bn_fwd_train_partials(const void* y, const void* residual, void* z, unsigned char* relu_mask,
                             const float* gamma, const float* beta, float* running_mean, float* running_var,
                             long long* num_batches_tracked, float momentum, float eps, float* stats_out,
                             int M, int C, int relu, int dtype, const float* partial, int nrb, void* workspace,
                             size_t ws_bytes, void* stream);
int cn_bn_fwd_train_partials_centered(const void* y, const void* residual, void* z, unsigned char* relu_mask,
                             const float* gamma, const float* beta, float* running_mean, float* running_var,
                             long long* num_batches_tracked, float momentum, float eps, float* stats_out,
                             int M, int C, int relu, int dtype, const float* partial, int nrb, void* workspace,
                             size_t ws_bytes, void* stream);
int cn_bn_fwd_infer(const void* y, const void* residual, void* z, const float* gamma, const float* beta,
                    const float* running_mean, const float* running_var, float eps, float* coeffs /*2C*/,
                    int M, int C, int relu, int dtype, void* stream);
/* The apply pass of a residual junction behind a projection shortcut (conv + BatchNorm), both BatchNorms already
 * finalised by cn_bn_fwd_train* with z = NULL (statistics, running statistics, scale / shift only; the reference applies
 * them as two nn.BatchNorm2d calls and an add, /root/reference models/resnet.py:141-165):
 *   z = relu?( y*scale[c] + shift[c] + round_T(res_y*rscale[c] + rshift[c]) )
 * stats / res_stats: the 4*C floats cn_bn_fwd_train* wrote for the junction / shortcut BatchNorm.  Same bits as the
 * shortcut BatchNorm's own apply followed by the junction's, one write + one read of the shortcut tensor less. */
int cn_bn_apply_dual(const void* y, const void* res_y, void* z, unsigned char* relu_mask, const float* stats,
                     const float* res_stats, int M, int C, int relu, int dtype, void* stream);
/* relu_mask: the byte mask of cn_bn_fwd_train (needed when a residual was added), NULL => ReLU mask
 * recomputed from y.  dres (optional) receives the masked upstream gradient for the residual branch. */
int cn_bn_bwd(const void* dz, const void* y, const unsigned char* relu_mask, const float* gamma, const float* stats,
              void* dy, void* dres, float* dgamma, float* dbeta, float beta_acc, float gscale,
              float* coef_scratch /*3C*/, int M, int C, int relu, int dtype, void* workspace,
              size_t ws_bytes, void* stream);

/* cn_bn_bwd when the upstream gradient arrives already masked (g = dz * relu_mask) together with its
 * reduction partials ([nrb][2*C]: sum g | sum g*xhat) from cn_conv2d_dgrad_bnbwd: finalize + apply only. */
int cn_bn_bwd_partials(const void* g, const void* y, const float* gamma, const float* stats, void* dy,
                       float* dgamma, float* dbeta, float beta_acc, float gscale, float* coef_scratch /*3C*/,
                       int M, int C, int dtype, const float* partial, int nrb, void* workspace, size_t ws_bytes,
                       void* stream);

/* ---- nn.SyncBatchNorm (main.py:190-191, --sync-bn) ----------------------------------------------
 * Each rank reduces its own statistics to 2*C doubles, the caller all-reduces that buffer in-stream
 * (cn_comm_allreduce, dtype 2, on the rank's communicator handle) and passes the global sums and the global row
 * count back; dgamma/dbeta stay per-rank sums (averaged by the data-parallel gradient all-reduce). */
int cn_bn_local_sums(const void* y, int M, int C, int dtype, const float* partial /*optional conv-epilogue rows*/,
                     int nrb, double* sums /*[sum y | sum y^2]*/, void* workspace, size_t ws_bytes, void* stream);
int cn_bn_fwd_train_sums(const void* y, const void* residual, void* z, unsigned char* relu_mask, const float* gamma,
                         const float* beta, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float momentum, float eps, float* stats_out, int M, int C,
                         int relu, int dtype, const double* sums, long long m_total, void* stream);
int cn_bn_bwd_local_sums(const void* dz, const void* y, const unsigned char* relu_mask, const float* stats, int M,
                         int C, int relu, int dtype, const float* partial /*optional dgrad-epilogue rows*/, int nrb,
                         double* sums /*[sum g | sum g*xhat]*/, void* workspace, size_t ws_bytes, void* stream);
int cn_bn_bwd_sums(const void* dz, const void* y, const unsigned char* relu_mask, const float* gamma,
                   const float* stats, void* dy, void* dres, float* dgamma, float* dbeta, float beta_acc,
                   float gscale, float* coef_scratch /*3C*/, int M, int C, int relu, int pre_masked, int dtype,
                   const double* local_sums, const double* global_sums, long long m_total, void* stream);

/* ---- nn.MaxPool2d / nn.AdaptiveAvgPool2d(1) (models/resnet.py:230,241) ---------------------- */
int cn_maxpool_fwd(const void* x, void* y, unsigned char* argmax_tap, int N, int H, int W, int C, int k,
                   int stride, int pad, int dtype, void* stream);
int cn_maxpool_bwd(const void* dy, const unsigned char* argmax_tap, void* dx, int N, int H, int W, int C,
                   int k, int stride, int pad, int dtype, void* stream);
/* The stem's bn1 -> relu -> maxpool (models/resnet.py:228-230) without materialising the normalised map:
 * forward = cn_bn_fwd_train(_partials) with z = NULL (statistics + coefficients only) followed by
 * cn_maxpool_fwd_bnrelu on the pre-BN tensor; backward = cn_bn_bwd_maxpool, which folds the pool's gather
 * backward into both BatchNorm-backward passes. */
int cn_maxpool_fwd_bnrelu(const void* x_prebn, const float* scale, const float* shift, void* y, unsigned char* idx,
                          int N, int H, int W, int C, int k, int stride, int pad, int dtype, void* stream);
int cn_bn_bwd_maxpool(const void* dpool, const unsigned char* idx, const void* y_prebn, const float* gamma,
                      const float* stats, void* dy, float* dgamma, float* dbeta, float beta_acc, float gscale,
                      float* coef_scratch /*3C*/, int N, int H, int W, int C, int k, int stride, int pad, int dtype,
                      void* workspace, size_t ws_bytes, void* stream);
/* the same pair with the pre-BatchNorm value of every winning tap kept by the forward (xmax, shaped like the pooled
 * map): the backward sums sum g and sum g*xhat are then taken over the pooled map (dpool, xmax) instead of the 4x larger
 * input map with a 2x2-window gather per pixel; dy is computed as before */
int cn_maxpool_fwd_bnrelu_xmax(const void* x_prebn, const float* scale, const float* shift, void* y, unsigned char* idx,
                               void* xmax, int N, int H, int W, int C, int k, int stride, int pad, int dtype,
                               void* stream);
int cn_bn_bwd_maxpool_xmax(const void* dpool, const unsigned char* idx, const void* y_prebn, const void* xmax,
                           const float* gamma, const float* stats, void* dy, float* dgamma, float* dbeta,
                           float beta_acc, float gscale, float* coef_scratch, int N, int H, int W, int C, int k,
                           int stride, int pad, int dtype, void* workspace, size_t ws_bytes, void* stream);
int cn_avgpool_fwd(const void* x, void* y, int N, int HW, int C, int dtype, void* stream);
int cn_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype, void* stream);

/* ---- host->device boundary and autograd fan-in (trainer.py:116-117; models/resnet.py:115,162) */
int cn_nchw_to_nhwc(const float* x_nchw, void* y_nhwc, int N, int C, int H, int W, int Cpad, int dtype,
                    void* stream);
/* transforms.ToTensor() + Normalize(mean, std) (preprocess.py:23-25) on the device: uint8 NHWC crops -> fp32 NCHW batch,
 * y[n][c][h][w] = lut[c][x[n][h][w][c]]; the caller fills lut[C][256] = (u / 255 - mean[c]) / std[c] with the reference's fp32
 * operations, so the batch is bit-identical to the host pipeline's.  1 <= C <= 4. */
int cn_u8_nhwc_to_nchw_lut(const unsigned char* x_nhwc, float* y_nchw, int N, int H, int W, int C, const float* lut,
                           void* stream);
/* the Resize step of the input pipeline (preprocess.py:21-41,71-77: PIL's fixed-point two-pass BILINEAR resampler) on the
 * device, bit for bit: B uint8 HWC crops of any size -> out[B][S][S][C].  pixels = the crops back to back; meta[B][8] = {byte
 * offset, h, w, flip, horizontal table offset (int32 units), its taps, vertical table offset, its taps}; a table = S entries
 * {first input index, count, coefficients round(k * 2^22)} computed on the host by PIL's recipe (data.resample_table);
 * row_owner[total_rows] / row_off[B] index the crops' rows in tmp (total_rows * S * C bytes). */
int cn_resize_u8_crops(const unsigned char* pixels, const long long* meta, const int* tables, const int* row_owner,
                       const int* row_off, unsigned char* tmp, unsigned char* out, int B, int total_rows, int S, int C,
                       void* stream);
/* Stride-2 stem (models/resnet.py:226, 7x7/2 pad 3 on 3 channels) in "pixel pair" form: the fp32 NCHW batch
 * becomes a zero-padded bf16 image [N][H+2*pad_h][(W+2*pad_w)/2][8] whose 16-byte chunks hold two adjacent
 * pixels x 4 channels; with the filter packed the same way (cn_weight_prep_pairs: [K][R][ceil(S/2)][8]) the
 * stem is cn_conv2d_fwd(C=8, R, S=ceil(S/2), stride (2,1), pad 0) - 28 instead of 49 reduction chunks for
 * 7x7 and no bounds tests; cn_conv2d_wgrad on the same view + cn_wgrad_unpack_pairs gives the KRSC gradient. */
int cn_nchw_to_pairs(const float* x_nchw, void* y_pairs, int N, int C, int H, int W, int pad_h, int pad_w,
                     void* stream);
int cn_weight_prep_pairs(const float* master_krsc, void* out_pairs, int K, int R, int S, int C, void* stream);
int cn_wgrad_unpack_pairs(const float* packed, float* dw_krsc, int K, int R, int S, int C, float beta, void* stream);
int cn_nhwc_to_nchw(const void* x_nhwc, float* y_nchw, int N, int C, int H, int W, int Cpad, int dtype,
                    void* stream);
/* op 0: a += b;  1: a = relu(b);  2: a = b * (c > 0);  3: a = b * c;  4: a = relu(b + c).  n elements (multiple of
 * the chunk). */
int cn_eltwise(int op, void* a, const void* b, const void* c, long long n, int dtype, void* stream);

/* ---- criterion + accuracy + meters (main.py:231-235; trainer.py:143,153,224-229) ------------ */
/* logits fp32 [B][K], target int64 [B]; dlogits (optional, grad_dtype) = (softmax - smoothed
 * one-hot) * gscale; row_scratch 3*B floats; step_out[0..2] = mean loss, prec@1, prec@5 (%)
 * of this batch; meters[0..3] += {loss*B, prec1*B, prec5*B, B} (either may be NULL). */
int cn_softmax_ce(const float* logits, const long long* target, void* dlogits, int grad_dtype,
                  float* row_scratch, float* step_out, float* meters, int B, int K, float gscale,
                  const float* gscale_dev /*optional device scalar folded into gscale*/, float smooth_eps,
                  void* stream);

/* ---- optimizer.step / grad clipping / filter preparation (trainer.py:165-173) --------------- */
/* hyper_dev (optional, DEVICE, 2 floats {lr, momentum}): when given it overrides the lr / momentum arguments, so
 * a step captured in a HIP graph follows the learning-rate schedule without being re-captured */
int cn_sgd_momentum(float* p, const float* g, float* buf, long long n, float lr, float momentum,
                    float weight_decay, float gscale, const float* clip_coef, const float* hyper_dev,
                    void* stream);
size_t cn_grad_norm_workspace(void);
/* out2[0] = ||g||_2 * gscale, out2[1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0). */
int cn_grad_norm_clip(const float* g, long long n, float gscale, float max_norm, float* out2, float* meters2,
                      float meter_weight, float* workspace, void* stream);
int cn_weight_prep(const float* w_master_krsc, void* w_krsc, void* w_crsk /*optional*/, int Co, int taps,
                   int Creal, int Cpad, int dtype, void* stream);
/* all filters of a model in one launch; desc = int64[nd][8] on the device:
 * {src_off, start, krsc_off, crsk_off (<0: none), Co, taps, Creal, Cpad} */
int cn_weight_prep_multi(const float* master_arena, void* wbuf, const long long* desc, int nd, long long total,
                         int dtype, void* stream);
/* Same conversion for the descriptors with Cpad == Creal, one workgroup per 64x64 tile of a filter
 * matrix (coalesced KRSC and CRSK stores).  tiles: int[ntiles][4] = {descriptor row, co0, j0, 0},
 * j = tap*C + c. */
int cn_weight_prep_tiled(const float* master, void* wbuf, const long long* desc, const int* tiles, int ntiles,
                         int dtype, void* stream);
size_t cn_colsum_workspace(int C);
int cn_colsum(const void* x, float* out, int M, int C, int dtype, float beta, float scale, float* workspace,
              void* stream);
/* dense layers whose width is not a multiple of the chunk (10-way head of models/mnist.py:30), fp32 master
 * weights w[K][C]: mode 0 y=x.w^T+b (x T, out fp32); 1 dx=dy.w (x:=dy fp32, out T); 2 dw,db += (x T, out:=dy fp32) */
int cn_small_linear(int mode, const void* x, const float* w, const float* bias, void* out, float* dw, float* db,
                    int B, int C, int K, int dtype, void* stream);
int cn_cast_from_f32(const float* x, void* y, long long n, int dtype, void* stream);
int cn_fill_f32(float* x, long long n, float v, void* stream);

/* ---- simulated 8-bit training operators, BASELINE config 5 `{'quantize': True}` (models/modules/quantize.py,
 * switched in by models/resnet.py:387-391).  Tensors are snapped to a 2^bits-level grid and stay floating
 * point, as in the reference; the convolutions run on cn_conv2d_*.  A zero quantisation range is treated as 1
 * (identity on a constant tensor) where the reference divides by zero (quantize.py:64-66). --------------- */
size_t cn_minmax_workspace(int rows, long long row_len);
/* calculate_qparams' per-sample x.flatten(1).min(-1)/.max(-1) (quantize.py:19-27): minmax[r] = {min, max} of
 * row r of a contiguous [rows][row_len] tensor */
int cn_minmax_rows(const void* x, int rows, long long row_len, int dtype, float* minmax, float* workspace,
                   size_t workspace_bytes, void* stream);
/* qp[0] = zero_point, qp[1] = range from the per-row min/max: mode 0 = batch mean (activations,
 * quantize.py:28-30), 1 = extremes (gradients, :31-33); optional QuantMeasure running update
 * running = running*momentum + new*(1-momentum) (quantize.py:163-172) */
int cn_qparams(const float* minmax, int rows, int mode, float* qp, float* running_zero_point, float* running_range,
               float momentum, void* stream);
/* UniformQuantize.forward, unsigned + dequantised (quantize.py:41-76): y = round(clamp((x - zp)/scale + noise,
 * 0, 2^bits-1))*scale + zp, scale = range/(2^bits-1); zero_point / range are device scalars (cn_qparams' qp and
 * qp + 1, or QuantMeasure's running buffers in eval mode).  noise: optional fp32 U(-0.5,0.5)
 * per element (quantize.py:67-69); else the built-in counter-based generator keyed by (seed, index) when
 * stochastic != 0; else deterministic rounding. */
int cn_quantize(const void* x, void* y, long long n, int dtype, const float* zero_point, const float* range,
                int num_bits, const float* noise, int stochastic, unsigned long long seed, void* stream);
/* cn_quantize with its generator seed mixed with a device-resident step counter (one unsigned 64-bit word the caller
 * advances with cn_counter_inc once per training step): a launch replayed from a captured HIP graph - frozen kernel
 * arguments - still draws fresh rounding noise every step. */
int cn_quantize_s(const void* x, void* y, long long n, int dtype, const float* zero_point, const float* range,
                  int num_bits, const float* noise, int stochastic, unsigned long long seed,
                  const unsigned long long* step_counter, void* stream);
int cn_counter_inc(unsigned long long* counter, void* stream);
/* per-row (= per output channel) quantisation of an fp32 filter matrix with the row's own min / max
 * (QConv2d / QLinear weights, quantize.py:201-203,239-240) */
int cn_quantize_rows(const float* x, float* y, int rows, int row_len, int num_bits, void* stream);
/* the same for every filter of a model in one launch: x / y = the flat fp32 parameter arena and its quantised shadow,
 * rowtab[rows][3] = {element offset, row length, 2^bits - 1} per output channel (engine.ParamArena.prepare_weights) */
int cn_quantize_rows_multi(const float* x, float* y, const long long* rowtab, int rows, void* stream);
/* RangeBN (quantize.py:256-330) on an input-quantised x[M][C]: per channel mean and
 * scale = (mean of `chunks` chunk maxima - mean of chunk minima) * scale_fix over the M values in (n,h,w) order;
 * z = act(((x - mean)/(scale + eps))*weight + bias [+ residual]).  training: running statistics updated
 * (running = running*momentum + new*(1-momentum)), stats[2C] = {mean | scale+eps} and arg[C][2*chunks] (pixel index
 * of every chunk's first maximum / minimum) are kept for cn_rangebn_bwd.  training == 0: running statistics. */
size_t cn_rangebn_workspace(int M, int C, int chunks);
int cn_rangebn_fwd(const void* x, const void* residual, void* z, const float* weight, const float* bias,
                   float* running_mean, float* running_var, float momentum, float eps, int chunks, float scale_fix,
                   float* stats, int* arg, int M, int C, int relu, int training, int dtype, float* workspace,
                   size_t workspace_bytes, void* stream);
/* gradient of the above for an already quantised output gradient g: dx (mean path + routing of the scale
 * gradient to the chunk maxima / minima), dweight += sum g*(x-mean)/(scale+eps), dbias += sum g */
int cn_rangebn_bwd(const void* g, const void* x, const float* weight, const float* stats, const int* arg, void* dx,
                   float* dweight, float* dbias, int M, int C, int chunks, float scale_fix, int dtype,
                   float* workspace, size_t workspace_bytes, void* stream);
/* Round-4 producer-side fusions of the quantised chain (same bits as the separate passes; quantize.py:158-182, 101-112,
 * 288-326):
 *  cn_rangebn_fwd_q   training forward on the RAW convolution output: x_qparams = [zero_point, range] of RangeBN's
 *                     x_bits-bit input quantiser (cn_qparams); the statistics pass snaps every element on load and stores
 *                     the snapped tensor to qx_out (= cn_quantize(x); the backward pass reads it); z_minmax (optional):
 *                     [mm_rows][2] = cn_minmax_rows(z, mm_rows), mm_rows = batch samples, for the next activation quantiser
 *  cn_rangebn_bwd_mm  cn_rangebn_bwd with the routing of dL/dscale folded into the apply pass and
 *                     dx_minmax[mm_rows][2] = cn_minmax_rows(dx, mm_rows) for the gradient quantiser of the convolution
 *  cn_eltwise_mm      a = b * (c > 0) (op 2) / a = relu(b + c) (op 4) with minmax[rows][2] = cn_minmax_rows(a, rows) */
int cn_rangebn_fwd_q(const void* x, const float* x_qparams, int x_bits, void* qx_out, const void* residual, void* z,
                     const float* weight, const float* bias, float* running_mean, float* running_var, float momentum,
                     float eps, int chunks, float scale_fix, float* stats, int* arg, int M, int C, int relu, int dtype,
                     int mm_rows, float* z_minmax, float* ws, size_t ws_bytes, void* stream);
int cn_rangebn_bwd_mm(const void* g, const void* x, const float* weight, const float* stats, const int* arg, void* dx,
                      float* dweight, float* dbias, int M, int C, int chunks, float scale_fix, int dtype, int mm_rows,
                      float* dx_minmax, float* ws, size_t ws_bytes, void* stream);
/* ---- 8-bit LEVEL storage of the snapped tensors (round 6): one byte per element instead of a value in the compute dtype.
 * cn_quantize_levels = cn_quantize_s writing levels; cn_rangebn_fwd_q8 = cn_rangebn_fwd_q keeping the snapped input as levels
 * (qx8_out) for the backward pass; cn_rangebn_bwd_q8 = cn_rangebn_bwd_mm taking g and / or x as levels with their
 * [zero_point, range] (NULL qparams: that operand holds values).  value = T(level * range / (2^bits - 1) + zero_point):
 * every output has the bits of the value-storing entry points.  bits <= 8, element counts in whole 16-byte chunks. */
int cn_quantize_levels(const void* x, unsigned char* y8, long long n, int dtype, const float* zero_point, const float* range,
                       int num_bits, const float* noise, int stochastic, unsigned long long seed,
                       const unsigned long long* step_counter, void* stream);
int cn_rangebn_fwd_q8(const void* x, const float* x_qparams, int x_bits, unsigned char* qx8_out, void* z, const float* weight,
                      const float* bias, float* running_mean, float* running_var, float momentum, float eps, int chunks,
                      float scale_fix, float* stats, int* arg, int M, int C, int relu, int dtype, int mm_rows,
                      float* z_minmax, float* ws, size_t ws_bytes, void* stream);
int cn_rangebn_bwd_q8(const void* g, const float* g_qparams, int g_bits, const void* x, const float* x_qparams, int x_bits,
                      const float* weight, const float* stats, const int* arg, void* dx, float* dweight, float* dbias,
                      int M, int C, int chunks, float scale_fix, int dtype, int mm_rows, float* dx_minmax,
                      float* dx_qp_extreme /* optional, mm_rows <= 256: cn_qparams(dx_minmax, mm_rows, 1) */, float* ws,
                      size_t ws_bytes, void* stream);
size_t cn_eltwise_mm_workspace(long long n, int rows, int dtype);
int cn_eltwise_mm(int op, void* a, const void* b, const void* c, long long n, int dtype, int rows, float* minmax, float* ws,
                  size_t ws_bytes, void* stream);
/* ... additionally qp_extreme[2] = cn_qparams(minmax, rows, 1) when the consumer of `a` is a gradient quantiser (rows <= 256) */
int cn_eltwise_mm_qp(int op, void* a, const void* b, const void* c, long long n, int dtype, int rows, float* minmax,
                     float* qp_extreme, float* ws, size_t ws_bytes, void* stream);

/* ---- true int8 MFMA forward product of QConv2d (v_mfma_i32_32x32x32_i8; csrc/qconv_i8.hip).  Both operands of
 * the reference's simulated convolution (quantize.py:195-219) live on integer grids, so
 *   y[p,k] = sx*sw[k]*ACC[p,k] + sx*zw'[k]*A[p] + zx'*(sw[k]*B[cls(p)][k] + zw'[k]*n_valid(cls(p))),
 * ACC = the int8 GEMM of the levels - 128 (exact int32), A = window sums of the activation levels, B = filter
 * level sums over the taps valid for border class cls (zero padding contributes 0, not the zero point).
 * Forward only: dgrad reduces across the per-output-channel filter scales and wgrad uses the full-precision dy
 * (quantize.py:115-121); both stay on the float kernels. */
/* activation (NHWC, dtype) -> q = level - 128 (int8 NHWC), A[N*P*Q] window sums, cls[N*P*Q] border class =
 * rowcls[p] * ncolcls + colcls[q]; chansum = int32 scratch [N*H*W]; C % 16 == 0 */
int cn_i8_prepare_activation(const void* x, signed char* q, int* chansum, int* A, unsigned char* cls, int N, int H,
                             int W, int C, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                             const float* zero_point, const float* range, const unsigned char* rowcls,
                             const unsigned char* colcls, int ncolcls, void* stream);
/* fp32 master filter [K][taps][C] -> q = level - 128 with the row's own min / max (quantize.py:201-203),
 * wsum[K][taps] = sum_c q, wpar[K][2] = {scale, zero point + 128*scale} */
int cn_i8_prepare_weight(const float* w_master, signed char* q, int* wsum, float* wpar, int K, int taps, int C,
                         void* stream);
/* clsmask[ncls][taps] = 1 where the tap is inside the image for that class; tables = (2 + ncls)*K floats scratch */
int cn_conv2d_fwd_i8(const signed char* xq, const signed char* wq, void* y, const int* A, const unsigned char* cls,
                     const float* zero_point, const float* range, const float* wpar, const int* wsum,
                     const unsigned char* clsmask, int ncls, float* tables, int N, int H, int W, int C, int K, int R,
                     int S, int stride_h, int stride_w, int pad_h, int pad_w, int out_dtype, void* stream);

/* ---- data-parallel exchange step, directly on RCCL (trainer.py:79-82 DistributedDataParallel; main.py:190-191
 * SyncBatchNorm) ---------------------------------------------------------------------------------
 * One communicator handle per process (= per GPU).  The handle is the ONLY state the library keeps for the
 * caller: an ncclComm_t, one high-priority HIP stream for the gradient buckets and a ring of events.
 * Rendezvous is the caller's job: rank 0 calls cn_comm_unique_id and ships the 128 bytes to the other ranks
 * (torch.distributed's store, MPI, a file ...), then every rank calls cn_comm_init with its HIP device current.
 * RCCL is bound at run time (the process's librccl.so.1).
 * Multi-rank set-up is done in phases that the caller separates with its own agreement step, so that a rank which
 * cannot proceed is found out before any rank blocks in a collective: (1) every rank: cn_comm_load (resolves the
 * library, nothing else); (2) rank 0: cn_comm_unique_id, shipped to all; (3) every rank: cn_comm_init. */
int cn_comm_load(void);
int cn_comm_unique_id(char* id128 /* HOST, 128 bytes */);
int cn_comm_init(void** handle /* HOST */, const char* id128 /* HOST */, int rank, int world);
int cn_comm_info(void* handle, int* rank, int* world, int* rccl_version /* HOST, any may be NULL */);
/* in-place SUM all-reduce of one fp32 gradient bucket on the handle's own stream, ordered after everything
 * queued so far on the first n_after (0..2) producer streams after_a, after_b; producers are not stalled */
int cn_comm_allreduce_bucket(void* handle, float* buf, long long count, void* after_a, void* after_b,
                             int n_after);
/* `stream` waits for every bucket queued so far (call before the optimizer step) */
int cn_comm_join(void* handle, void* stream);
/* in-stream, in-place SUM all-reduce; dtype 0 = fp32, 2 = fp64 (SyncBatchNorm sums) */
int cn_comm_allreduce(void* handle, void* buf, long long count, int dtype, void* stream);
/* in-stream broadcast of nbytes bytes from rank `root` (parameter arena at construction, BN buffers) */
int cn_comm_broadcast(void* handle, void* buf, long long nbytes, int root, void* stream);
int cn_comm_destroy(void* handle);

/* ---- hardware lane-map probes (tests only) -------------------------------------------------- */
int cn_probe_mfma_bf16(const unsigned short* A /*32x16*/, const unsigned short* B /*16x32*/, float* D /*32x32*/,
                       void* stream);
int cn_probe_mfma_f16(const unsigned short* A /*32x16*/, const unsigned short* B /*16x32*/, float* D /*32x32*/,
                      void* stream);
int cn_probe_mfma_f32(const float* A /*32x2*/, const float* B /*2x32*/, float* D, void* stream);
int cn_probe_tr16(const unsigned short* src /*256*/, unsigned short* out /*64x4*/, void* stream);
int cn_probe_mfma_i8(const signed char* A /*32x32*/, const signed char* B /*32x32*/, int* D /*32x32*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONVNET_HIP_H */
