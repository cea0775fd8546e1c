// plan.hip -- launch plans: the device side of one training step recorded once and re-issued from ONE C call.
//
// Replaces the per-batch host work of the reference's step loop (/root/reference/trainer.py:106-177: ~470 ATen
// dispatches through Python per ResNet-50 step; this package's eager step: the same number of cn_* calls through
// ctypes + the autograd tape, 11-12 ms of host time per 17 ms step).  A HIP graph removes the host work but its
// executor collapses the two-stream schedule of the step (backward chain + weight-gradient side stream) onto one
// hardware queue (NOTES.md, round 4: 7 % slower at b = 256).  A plan keeps the schedule: it is the ordered list of
//   - kernel launches (function, grid, block, a private copy of the argument block, the stream they were issued on),
//   - cross-stream hand-offs (cn_stream_fork; cn_stream_arm marks = "this kernel signals event e on completion",
//     cn_stream_wait_mark = "that stream waits for e"),
//   - communicator calls (cn_comm_allreduce_bucket / _join / _allreduce: re-issued LIVE on RCCL, never captured),
//   - whatever else the step put on the streams that is not a launch of this library (a torch fill kernel, a
//     memset): imported from the HIP graph the recording ran under (cn_plan_import_graph), re-launched with the
//     graph node's own parameter block,
// and cn_plan_replay() walks that list with hipLaunchKernel / hipExtLaunchKernel / hipEventRecord /
// hipStreamWaitEvent on the recorded streams.  Nothing is compiled, instantiated or optimised: the device sees the
// same launches in the same per-stream order with the same event edges as in the eager step, the host spends a few
// microseconds per launch instead of a Python call.
//
// Recording is process-wide (the autograd engine runs the backward pass on its own thread) and normally happens
// while the caller's stream is being CAPTURED (torch.cuda.graph: nothing executes, torch's allocator gives the step
// a private pool, so every address in the plan stays valid for as long as the captured graph object lives).  Under
// capture a mark is an event recorded right behind its kernel (a captured hipExtLaunchKernel would drop the stop
// event); in a replay it is the kernel's own completion signal, as in the eager step.
#include "cn_api_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

int cn_plan_recording = 0;   // read by CN_LAUNCH and the stream / communicator entry points

namespace {

enum OpKind { OP_KERNEL = 0, OP_FORK, OP_WAIT_MARK, OP_COMM_BUCKET, OP_COMM_JOIN, OP_COMM_ALLREDUCE, OP_FOREIGN_KERNEL,
              OP_MEMSET, OP_MEMCPY, OP_CLOSURE };

struct PlanOp {
  int kind = OP_KERNEL;
  hipStream_t s0 = nullptr, s1 = nullptr;     // kernel / wait: s0; fork: s0 -> s1; comm bucket: producers s0, s1
  const void* func = nullptr;
  dim3 grid, block;
  unsigned shmem = 0;
  std::vector<void*> args;                    // kernel: pointers into `blob`; foreign kernel: the graph node's array
  void** fargs = nullptr;                     // foreign kernel: kernelParams of the node (owned by the graph)
  void** fextra = nullptr;                    // foreign kernel launched through the module API: its `extra` block
  std::vector<char> blob;                     // private copy of the argument values
  int event = -1;                             // kernel: event signalled on completion; fork / wait_mark: event used
  void* node = nullptr;                       // graph node this launch became (recording under capture)
  // communicator ops
  void* comm = nullptr;
  void* buf = nullptr;
  long long count = 0;
  int dtype = 0, n_after = 0;
  // memset / memcpy (imported)
  void* dst = nullptr;
  const void* src = nullptr;
  size_t bytes = 0;
  int value = 0;
#ifdef CN_EMULATE
  std::function<void()> body;
#endif
};

struct InputSite { size_t op; size_t arg; size_t byte; uintptr_t delta; };   // an 8-byte argument word holding base + delta

struct Plan {
  std::vector<InputSite> sites[4];   // rebindable inputs (cn_plan_bind_input)
  hipStream_t main = nullptr;
  std::vector<PlanOp> ops;
  int n_events = 0;
  std::vector<int> handle_event;     // mark ring handle -> plan event index of the kernel that last took it
  bool ended = false;
  int n_foreign = 0;
  long long replays = 0;
#ifndef CN_EMULATE
  std::vector<hipEvent_t> events;
#endif
  std::string text;
};

std::mutex g_mu;
Plan* g_rec = nullptr;

}  // namespace

// ---- recording hooks (called by CN_LAUNCH and by runtime.hip / comm.hip while cn_plan_recording != 0) ----------
#ifndef CN_EMULATE
static void* capture_tail(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &ndeps) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (st != hipStreamCaptureStatusActive || ndeps != 1) return nullptr;
  return (void*)deps[0];
}

// One launch of this library: called BEFORE the launch with copies of the kernel's parameters.
int cn_plan_rec_kernel(const void* func, dim3 grid, dim3 block, hipStream_t stream, int nargs, const void* const* ptrs,
                       const size_t* sizes, const size_t* aligns, int mark_handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  Plan* p = g_rec;
  if (p == nullptr) return -1;
  p->ops.emplace_back();
  PlanOp& op = p->ops.back();
  op.kind = OP_KERNEL;
  op.s0 = stream;
  op.func = func;
  op.grid = grid;
  op.block = block;
  size_t total = 0;
  std::vector<size_t> offs((size_t)nargs);
  for (int i = 0; i < nargs; ++i) {
    total = (total + aligns[i] - 1) / aligns[i] * aligns[i];
    offs[(size_t)i] = total;
    total += sizes[i];
  }
  op.blob.resize(total + 16);
  char* base = op.blob.data();
  base += (16 - ((uintptr_t)base & 15)) & 15;
  op.args.resize((size_t)nargs);
  for (int i = 0; i < nargs; ++i) {
    memcpy(base + offs[(size_t)i], ptrs[i], sizes[i]);
    op.args[(size_t)i] = base + offs[(size_t)i];
  }
  if (mark_handle >= 0) {
    op.event = p->n_events++;
    if ((int)p->handle_event.size() <= mark_handle) p->handle_event.resize((size_t)mark_handle + 1, -1);
    p->handle_event[(size_t)mark_handle] = op.event;
  }
  return (int)p->ops.size() - 1;
}
// ... and AFTER the launch: which graph node it became (NULL when the stream is not being captured).
void cn_plan_rec_kernel_node(int op_index, hipStream_t stream) {
  void* node = capture_tail(stream);
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rec != nullptr && op_index >= 0 && op_index < (int)g_rec->ops.size()) g_rec->ops[(size_t)op_index].node = node;
}
#else
int cn_plan_rec_closure(const std::function<void()>& launch) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rec == nullptr) return -1;
  g_rec->ops.emplace_back();
  PlanOp& op = g_rec->ops.back();
  op.kind = OP_CLOSURE;
  op.body = launch;
  return (int)g_rec->ops.size() - 1;
}
#endif

void cn_plan_rec_fork(void* from, void* to) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rec == nullptr) return;
  g_rec->ops.emplace_back();
  PlanOp& op = g_rec->ops.back();
  op.kind = OP_FORK;
  op.s0 = (hipStream_t)from;
  op.s1 = (hipStream_t)to;
  op.event = g_rec->n_events++;
}
int cn_plan_rec_wait_mark(int handle, void* to) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rec == nullptr) return CN_OK;
  if (handle < 0 || handle >= (int)g_rec->handle_event.size() || g_rec->handle_event[(size_t)handle] < 0) {
    cn_set_error("plan: cn_stream_wait_mark(%d) without a marked kernel in this recording", handle);
    return CN_EINVAL;
  }
  g_rec->ops.emplace_back();
  PlanOp& op = g_rec->ops.back();
  op.kind = OP_WAIT_MARK;
  op.s0 = (hipStream_t)to;
  op.event = g_rec->handle_event[(size_t)handle];
  return CN_OK;
}
void cn_plan_rec_comm(int kind, void* comm, void* buf, long long count, int dtype, void* s0, void* s1, int n_after) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rec == nullptr) return;
  g_rec->ops.emplace_back();
  PlanOp& op = g_rec->ops.back();
  op.kind = kind == 0 ? OP_COMM_BUCKET : kind == 1 ? OP_COMM_JOIN : OP_COMM_ALLREDUCE;
  op.comm = comm;
  op.buf = buf;
  op.count = count;
  op.dtype = dtype;
  op.s0 = (hipStream_t)s0;
  op.s1 = (hipStream_t)s1;
  op.n_after = n_after;
}

// ---- C ABI ---------------------------------------------------------------------------------------------------
extern "C" int cn_plan_begin(void** plan, void* main_stream) {
  if (plan == nullptr) { cn_set_error("cn_plan_begin: null handle pointer"); return CN_EINVAL; }
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rec != nullptr) { cn_set_error("cn_plan_begin: another plan is being recorded"); return CN_EINVAL; }
  Plan* p = new Plan();
  p->main = (hipStream_t)main_stream;
  g_rec = p;
  cn_plan_recording = 1;
  *plan = p;
  return CN_OK;
}

extern "C" int cn_plan_end(void* plan) {
  Plan* p = (Plan*)plan;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (p == nullptr || g_rec != p) { cn_set_error("cn_plan_end: this plan is not being recorded"); return CN_EINVAL; }
    g_rec = nullptr;
    cn_plan_recording = 0;
  }
#ifndef CN_EMULATE
  p->events.resize((size_t)p->n_events);
  for (int i = 0; i < p->n_events; ++i)
    if (hipEventCreateWithFlags(&p->events[(size_t)i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
      p->events.resize((size_t)i);
      cn_set_error("cn_plan_end: hipEventCreate failed");
      return CN_EHIP;
    }
#endif
  p->ended = true;
  return CN_OK;
}

extern "C" int cn_plan_destroy(void* plan) {
  Plan* p = (Plan*)plan;
  if (p == nullptr) return CN_OK;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rec == p) { g_rec = nullptr; cn_plan_recording = 0; }
  }
#ifndef CN_EMULATE
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
#endif
  delete p;
  return CN_OK;
}

// counts[0..7]: ops, kernel launches of this library, imported launches / memsets, events, hand-offs (fork + wait),
// communicator calls, distinct streams, replays so far
extern "C" int cn_plan_info(void* plan, long long* counts) {
  Plan* p = (Plan*)plan;
  if (p == nullptr || counts == nullptr) { cn_set_error("cn_plan_info: bad arguments"); return CN_EINVAL; }
  long long k = 0, f = 0, h = 0, c = 0;
  std::vector<hipStream_t> streams;
  for (const PlanOp& op : p->ops) {
    if (op.kind == OP_KERNEL || op.kind == OP_CLOSURE) ++k;
    else if (op.kind == OP_FOREIGN_KERNEL || op.kind == OP_MEMSET || op.kind == OP_MEMCPY) ++f;
    else if (op.kind == OP_FORK || op.kind == OP_WAIT_MARK) ++h;
    else ++c;
    if (op.kind == OP_KERNEL || op.kind == OP_FOREIGN_KERNEL) {
      bool seen = false;
      for (hipStream_t s : streams) seen = seen || s == op.s0;
      if (!seen) streams.push_back(op.s0);
    }
  }
  counts[0] = (long long)p->ops.size();
  counts[1] = k;
  counts[2] = f;
  counts[3] = p->n_events;
  counts[4] = h;
  counts[5] = c;
  counts[6] = (long long)streams.size();
  counts[7] = p->replays;
  return CN_OK;
}

#ifndef CN_EMULATE
static const char* kernel_name(const PlanOp& op) {
  const char* n = nullptr;
  if (op.kind == OP_KERNEL || (op.kind == OP_FOREIGN_KERNEL && op.fargs != nullptr)) n = hipKernelNameRefByPtr(op.func, op.s0);
  else if (op.kind == OP_FOREIGN_KERNEL) n = hipKernelNameRef((hipFunction_t)op.func);
  (void)hipGetLastError();
  return n != nullptr ? n : "?";
}
#endif

// Human-readable listing (debugging, tests): one line per op.
extern "C" const char* cn_plan_describe(void* plan) {
  Plan* p = (Plan*)plan;
  if (p == nullptr) return "";
  p->text.clear();
  char line[512];
  int i = 0;
  for (const PlanOp& op : p->ops) {
    switch (op.kind) {
#ifndef CN_EMULATE
      case OP_KERNEL:
      case OP_FOREIGN_KERNEL:
        snprintf(line, sizeof(line), "%d %s s=%p grid=%u,%u,%u block=%u ev=%d %.300s\n", i,
                 op.kind == OP_KERNEL ? "kernel" : "foreign", (void*)op.s0, op.grid.x, op.grid.y, op.grid.z, op.block.x,
                 op.event, kernel_name(op));
        break;
#endif
      case OP_CLOSURE: snprintf(line, sizeof(line), "%d kernel (emulated)\n", i); break;
      case OP_FORK: snprintf(line, sizeof(line), "%d fork %p -> %p ev=%d\n", i, (void*)op.s0, (void*)op.s1, op.event); break;
      case OP_WAIT_MARK: snprintf(line, sizeof(line), "%d wait s=%p ev=%d\n", i, (void*)op.s0, op.event); break;
      case OP_COMM_BUCKET: snprintf(line, sizeof(line), "%d comm_bucket n=%lld after=%d\n", i, op.count, op.n_after); break;
      case OP_COMM_JOIN: snprintf(line, sizeof(line), "%d comm_join s=%p\n", i, (void*)op.s0); break;
      case OP_COMM_ALLREDUCE: snprintf(line, sizeof(line), "%d comm_allreduce n=%lld dtype=%d\n", i, op.count, op.dtype); break;
      case OP_MEMSET: snprintf(line, sizeof(line), "%d memset %zu bytes\n", i, op.bytes); break;
      case OP_MEMCPY: snprintf(line, sizeof(line), "%d memcpy %zu bytes\n", i, op.bytes); break;
      default: snprintf(line, sizeof(line), "%d ?\n", i);
    }
    p->text += line;
    ++i;
  }
  return p->text.c_str();
}

// Cross-check against the HIP graph the recording ran under and import what this library did not launch.
// Every kernel op knows the graph node it became; a node of the graph nobody here launched is "foreign" (a torch
// kernel on the step's stream).  A foreign node goes onto the plan's main stream at the EARLIEST position that is
// behind all its dependencies: behind the op of every dependency, and - for a dependency on another stream - behind
// the first hand-off that makes the main stream wait for that stream after it.  Earliest is always safe for its
// dependents (a hand-off recorded later only waits for more).  Returns the number of imported nodes, or < 0 when
// the graph holds something a plan cannot re-issue (then the caller stays with eager launches).
extern "C" int cn_plan_import_graph(void* plan, void* graph_) {
#ifdef CN_EMULATE
  (void)plan; (void)graph_;
  cn_set_error("cn_plan_import_graph: no HIP graphs in the emulator build");
  return CN_EINVAL;
#else
  Plan* p = (Plan*)plan;
  hipGraph_t graph = (hipGraph_t)graph_;
  if (p == nullptr || graph == nullptr || !p->ended) { cn_set_error("cn_plan_import_graph: bad arguments / plan still recording"); return CN_EINVAL; }
  size_t n = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) { cn_set_error("cn_plan_import_graph: hipGraphGetNodes failed"); return CN_EHIP; }
  std::vector<hipGraphNode_t> nodes(n);
  if (n > 0 && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) { cn_set_error("cn_plan_import_graph: hipGraphGetNodes failed"); return CN_EHIP; }
  std::unordered_map<void*, int> pos;     // graph node -> index of the op it sits behind (own kernels: their op)
  long long own = 0, own_with_node = 0;
  for (size_t i = 0; i < p->ops.size(); ++i)
    if (p->ops[i].kind == OP_KERNEL) {
      ++own;
      if (p->ops[i].node != nullptr) { pos[p->ops[i].node] = (int)i; ++own_with_node; }
    }
  if (own_with_node != own) {
    cn_set_error("cn_plan_import_graph: %lld of %lld launches were recorded outside the capture", own - own_with_node, own);
    return CN_EINVAL;
  }
  struct Ins { int after; int seq; PlanOp op; };
  std::vector<Ins> ins;
  std::vector<hipGraphNode_t> todo;
  for (hipGraphNode_t nd : nodes)
    if (pos.find((void*)nd) == pos.end()) todo.push_back(nd);
  int seq = 0;
  size_t guard = todo.size() * todo.size() + 8;
  while (!todo.empty() && guard-- > 0) {
    hipGraphNode_t nd = todo.front();
    todo.erase(todo.begin());
    size_t nd_deps = 0;
    (void)hipGraphNodeGetDependencies(nd, nullptr, &nd_deps);
    std::vector<hipGraphNode_t> deps(nd_deps);
    if (nd_deps > 0 && hipGraphNodeGetDependencies(nd, deps.data(), &nd_deps) != hipSuccess) { cn_set_error("cn_plan_import_graph: dependencies unreadable"); return CN_EHIP; }
    bool ready = true;
    int after = -1;
    for (hipGraphNode_t d : deps) {
      auto it = pos.find((void*)d);
      if (it == pos.end()) { ready = false; break; }
      int a = it->second;
      if (a >= 0 && p->ops[(size_t)a].kind == OP_KERNEL && p->ops[(size_t)a].s0 != p->main) {
        // the dependency ran on another stream: the node must sit behind a hand-off main <- that stream
        const hipStream_t other = p->ops[(size_t)a].s0;
        int w = -1;
        for (size_t j = (size_t)a + 1; j < p->ops.size() && w < 0; ++j) {
          const PlanOp& o = p->ops[j];
          if (o.kind == OP_FORK && o.s0 == other && o.s1 == p->main) w = (int)j;
          if (o.kind == OP_WAIT_MARK && o.s0 == p->main && o.event == p->ops[(size_t)a].event && o.event >= 0) w = (int)j;
        }
        if (w < 0) { cn_set_error("cn_plan_import_graph: a foreign node depends on a side-stream kernel no hand-off covers"); return CN_EINVAL; }
        a = w;
      }
      after = a > after ? a : after;
    }
    if (!ready) { todo.push_back(nd); continue; }
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(nd, &ty) != hipSuccess) { cn_set_error("cn_plan_import_graph: node type unreadable"); return CN_EHIP; }
    Ins in;
    in.after = after;
    in.seq = seq++;
    in.op.s0 = p->main;
    if (ty == hipGraphNodeTypeKernel) {
      hipKernelNodeParams kp;
      memset(&kp, 0, sizeof(kp));
      if (hipGraphKernelNodeGetParams(nd, &kp) != hipSuccess) { cn_set_error("cn_plan_import_graph: kernel node parameters unreadable"); return CN_EHIP; }
      in.op.kind = OP_FOREIGN_KERNEL;
      in.op.func = kp.func;
      in.op.grid = kp.gridDim;
      in.op.block = kp.blockDim;
      in.op.shmem = kp.sharedMemBytes;
      in.op.fargs = kp.kernelParams;
      in.op.fextra = kp.extra;
      if (kp.kernelParams == nullptr && kp.extra == nullptr) { cn_set_error("cn_plan_import_graph: kernel node without parameters"); return CN_EINVAL; }
    } else if (ty == hipGraphNodeTypeMemset) {
      hipMemsetParams mp;
      memset(&mp, 0, sizeof(mp));
      if (hipGraphMemsetNodeGetParams(nd, &mp) != hipSuccess) { cn_set_error("cn_plan_import_graph: memset node parameters unreadable"); return CN_EHIP; }
      if (mp.height > 1 || (mp.elementSize != 1 && mp.elementSize != 4)) { cn_set_error("cn_plan_import_graph: 2-D / 16-bit memset node"); return CN_EINVAL; }
      in.op.kind = OP_MEMSET;
      in.op.dst = mp.dst;
      in.op.value = (int)mp.value;
      in.op.bytes = mp.width * mp.elementSize;
      in.op.count = mp.elementSize;
    } else if (ty == hipGraphNodeTypeEmpty) {
      pos[(void*)nd] = after;      // carries dependencies only
      continue;
    } else {
      cn_set_error("cn_plan_import_graph: the step holds a graph node of type %d (memcpy / host / child graph): not plannable", (int)ty);
      return CN_EINVAL;
    }
    pos[(void*)nd] = after;
    ins.push_back(in);
  }
  if (!todo.empty()) { cn_set_error("cn_plan_import_graph: dependency cycle among foreign nodes"); return CN_EINVAL; }
  if (!ins.empty()) {
    std::vector<PlanOp> merged;
    merged.reserve(p->ops.size() + ins.size());
    // (ins is in placement order per position already: seq increases; bucket by `after`)
    std::vector<std::vector<size_t>> at(p->ops.size() + 1);
    for (size_t i = 0; i < ins.size(); ++i) at[(size_t)(ins[i].after + 1)].push_back(i);
    for (size_t i : at[0]) merged.push_back(std::move(ins[i].op));
    for (size_t j = 0; j < p->ops.size(); ++j) {
      merged.push_back(std::move(p->ops[j]));     // moved, not copied: the argument pointers into `blob` stay valid
      for (size_t i : at[j + 1]) merged.push_back(std::move(ins[i].op));
    }
    p->ops.swap(merged);
  }
  p->n_foreign = (int)ins.size();
  return (int)ins.size();
#endif
}

// Rebindable inputs: the step reads its batch through pointers baked into argument blocks.  bind finds every 8-byte
// aligned argument word of this library's launches whose value lies inside [base, base + bytes) (a batch tensor of the
// recording) and remembers it; set_input points all of them at another buffer of the same layout - the next replay
// reads the caller's batch in place instead of a copy of it.  Returns the number of words found (0: nothing to
// rebind, keep copying).  Launches imported from the graph are not searched: the caller must know that only this
// library reads the tensor (Trainer: the model's first operator is the layout cast).
extern "C" int cn_plan_bind_input(void* plan, int slot, const void* base_, size_t bytes) {
  Plan* p = (Plan*)plan;
  if (p == nullptr || !p->ended || slot < 0 || slot >= 4 || base_ == nullptr || bytes == 0) { cn_set_error("cn_plan_bind_input: bad arguments"); return CN_EINVAL; }
  const uintptr_t base = (uintptr_t)base_;
  p->sites[slot].clear();
  for (size_t i = 0; i < p->ops.size(); ++i) {
    PlanOp& op = p->ops[i];
    if (op.kind != OP_KERNEL) continue;
    for (size_t a = 0; a < op.args.size(); ++a) {
      const char* lo = (const char*)op.args[a];
      const char* b0 = op.blob.data() + ((16 - ((uintptr_t)op.blob.data() & 15)) & 15);      // (as laid out by cn_plan_rec_kernel)
      const char* hi = a + 1 < op.args.size() ? (const char*)op.args[a + 1] : b0 + (op.blob.size() - 16);
      for (const char* q = lo; q + 8 <= hi; q += 8) {
        if (((uintptr_t)q & 7) != 0) break;
        uintptr_t v;
        memcpy(&v, q, 8);
        if (v >= base && v < base + bytes) p->sites[slot].push_back({i, a, (size_t)(q - lo), v - base});
      }
    }
  }
  return (int)p->sites[slot].size();
}
extern "C" int cn_plan_set_input(void* plan, int slot, const void* base_) {
  Plan* p = (Plan*)plan;
  if (p == nullptr || slot < 0 || slot >= 4 || base_ == nullptr) { cn_set_error("cn_plan_set_input: bad arguments"); return CN_EINVAL; }
  for (const InputSite& st : p->sites[slot]) {
    const uintptr_t v = (uintptr_t)base_ + st.delta;
    memcpy((char*)p->ops[st.op].args[st.arg] + st.byte, &v, 8);
  }
  return CN_OK;
}

// communicator entry points that really issue (comm.hip)
int cn_comm_allreduce_bucket_issue(void* handle, float* buf, long long count, void* after_a, void* after_b, int n_after);
int cn_comm_join_issue(void* handle, void* stream);
int cn_comm_allreduce_issue(void* handle, void* buf, long long count, int dtype, void* stream);

extern "C" int cn_plan_replay(void* plan) {
  Plan* p = (Plan*)plan;
  if (p == nullptr || !p->ended) { cn_set_error("cn_plan_replay: no finished plan"); return CN_EINVAL; }
  if (cn_plan_recording) { cn_set_error("cn_plan_replay: a plan is being recorded"); return CN_EINVAL; }
  int idx = 0;
  for (PlanOp& op : p->ops) {
#ifdef CN_EMULATE
    if (op.kind == OP_CLOSURE) op.body();
#else
    hipError_t e = hipSuccess;
    int rc = CN_OK;
    switch (op.kind) {
      case OP_KERNEL:
        if (op.event >= 0)
          e = hipExtLaunchKernel(op.func, op.grid, op.block, op.args.data(), 0, op.s0, nullptr, p->events[(size_t)op.event], 0);
        else
          e = hipLaunchKernel(op.func, op.grid, op.block, op.args.data(), 0, op.s0);
        break;
      case OP_FOREIGN_KERNEL:
        if (op.fargs != nullptr)
          e = hipLaunchKernel(op.func, op.grid, op.block, op.fargs, op.shmem, op.s0);
        else
          e = hipModuleLaunchKernel((hipFunction_t)op.func, op.grid.x, op.grid.y, op.grid.z, op.block.x, op.block.y,
                                    op.block.z, op.shmem, op.s0, nullptr, op.fextra);
        break;
      case OP_FORK:
        e = hipEventRecord(p->events[(size_t)op.event], op.s0);
        if (e == hipSuccess) e = hipStreamWaitEvent(op.s1, p->events[(size_t)op.event], 0);
        break;
      case OP_WAIT_MARK:
        e = hipStreamWaitEvent(op.s0, p->events[(size_t)op.event], 0);
        break;
      case OP_MEMSET:
        e = op.count == 4 ? hipMemsetD32Async((hipDeviceptr_t)op.dst, op.value, op.bytes / 4, op.s0)
                          : hipMemsetAsync(op.dst, op.value, op.bytes, op.s0);
        break;
      case OP_COMM_BUCKET:
        rc = cn_comm_allreduce_bucket_issue(op.comm, (float*)op.buf, op.count, (void*)op.s0, (void*)op.s1, op.n_after);
        break;
      case OP_COMM_JOIN:
        rc = cn_comm_join_issue(op.comm, (void*)op.s0);
        break;
      case OP_COMM_ALLREDUCE:
        rc = cn_comm_allreduce_issue(op.comm, op.buf, op.count, op.dtype, (void*)op.s0);
        break;
      default:
        break;
    }
    if (rc != CN_OK) return rc;
    if (e != hipSuccess) {
      cn_set_error("cn_plan_replay: op %d (kind %d) failed: %s", idx, op.kind, hipGetErrorString(e));
      (void)hipGetLastError();
      return CN_EHIP;
    }
#endif
    ++idx;
  }
  p->replays++;
  return CN_OK;
}
