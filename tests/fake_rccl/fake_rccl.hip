// fake_rccl.hip -- TEST INFRASTRUCTURE, not product code.  An in-process stand-in for the handful of RCCL entry points
// liblentil_hip.so's exchange uses (pota_amd/csrc/lentil_comm.h), so that the matching of sends and receives, the
// metadata all-gather and the merge order of lentil_hip_exchange_bands / lentil_hip_allreduce can be run with several
// "ranks" on the ONE GPU a test box has: every rank is a thread with its own lentil_hip_ctx, all on the same device
// (real RCCL refuses two ranks per device).  tests/test_native_exchange.py points LENTIL_RCCL_LIB at this library.
//
// Semantics kept: ncclGroupStart/End batches point-to-point calls, which complete together; sends and receives match
// per (source, destination) pair in issue order, counts must agree (a mismatch aborts the test instead of hanging);
// collectives are called by every rank in the same order.  Everything is synchronous: a call returns when the data has
// arrived (stronger than RCCL's stream semantics, never weaker for the caller).
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#define FAKE_API extern "C" __attribute__((visibility("default")))

namespace {
struct Msg { const void *ptr; size_t bytes; bool taken = false; };
struct World {
  int n = 0, joined = 0;
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Msg *>> box;    // (src, dst) -> posted sends
  // barrier + collective staging
  int arrived = 0; uint64_t generation = 0;
  std::vector<const void *> ptrs;
  void *tmp = nullptr; size_t tmp_bytes = 0;
};
struct Comm { World *w; int rank; };
struct Op { bool send; void *ptr; size_t bytes; int peer; hipStream_t st; };

std::mutex g_m;
std::map<uint64_t, World *> g_worlds;
uint64_t g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local Comm *t_comm = nullptr;

size_t dtype_bytes(int t) { return (t == 0 || t == 1) ? 1 : (t == 4 || t == 5 || t == 8) ? 8 : 4; }

void barrier(World *w) {
  std::unique_lock<std::mutex> l(w->m);
  const uint64_t g = w->generation;
  if (++w->arrived == w->n) { w->arrived = 0; ++w->generation; w->cv.notify_all(); }
  else w->cv.wait(l, [&] { return w->generation != g; });
}

[[noreturn]] void die(const char *what) { fprintf(stderr, "fake_rccl: %s\n", what); abort(); }

int run_ops(Comm *c, std::vector<Op> &ops) {
  World *w = c->w;
  std::vector<Msg *> mine;
  for (Op &o : ops)
    if (o.send) {
      if (hipStreamSynchronize(o.st) != hipSuccess) return 1;     // the data is complete before it is offered
      Msg *m = new Msg{o.ptr, o.bytes};
      mine.push_back(m);
      std::lock_guard<std::mutex> l(w->m);
      w->box[{c->rank, o.peer}].push_back(m);
      w->cv.notify_all();
    }
  for (Op &o : ops)
    if (!o.send) {
      Msg *m;
      {
        std::unique_lock<std::mutex> l(w->m);
        auto &q = w->box[{o.peer, c->rank}];
        if (!w->cv.wait_for(l, std::chrono::seconds(60), [&] { return !q.empty(); })) die("receive without a matching send (60 s)");
        m = q.front(); q.pop_front();
      }
      if (m->bytes != o.bytes) die("send and receive sizes differ");
      if (hipMemcpyAsync(o.ptr, m->ptr, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) return 1;
      if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
      std::lock_guard<std::mutex> l(w->m);
      m->taken = true;
      w->cv.notify_all();
    }
  for (Msg *m : mine) {
    std::unique_lock<std::mutex> l(w->m);
    if (!w->cv.wait_for(l, std::chrono::seconds(60), [&] { return m->taken; })) die("send without a matching receive (60 s)");
    l.unlock();
    delete m;
  }
  return 0;
}

template <typename T, bool kMin>
__global__ void reduce_kernel(const void *const *src, int n, T *dst, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  T v = ((const T *)src[0])[i];
  for (int r = 1; r < n; ++r) {
    const T u = ((const T *)src[r])[i];
    v = kMin ? (u < v ? u : v) : (T)(v + u);
  }
  dst[i] = v;
}
}  // namespace

FAKE_API const char *ncclGetErrorString(int) { return "fake_rccl error"; }

FAKE_API int ncclGetUniqueId(void *id) {
  std::lock_guard<std::mutex> l(g_m);
  memset(id, 0, 128);
  const uint64_t v = g_next_id++;
  memcpy(id, &v, 8);
  return 0;
}

struct FakeId { char b[128]; };
FAKE_API int ncclCommInitRank(Comm **out, int nranks, FakeId id, int rank) {
  uint64_t key;
  memcpy(&key, id.b, 8);
  World *w;
  {
    std::lock_guard<std::mutex> l(g_m);
    World *&slot = g_worlds[key];
    if (!slot) { slot = new World(); slot->n = nranks; slot->ptrs.resize((size_t)nranks); }
    w = slot;
  }
  if (w->n != nranks) die("ranks disagree about the world size");
  *out = new Comm{w, rank};
  barrier(w);
  return 0;
}

FAKE_API int ncclCommDestroy(Comm *c) { delete c; return 0; }     // the World is leaked: tests only

FAKE_API int ncclGroupStart() { ++t_depth; return 0; }
FAKE_API int ncclGroupEnd() {
  if (--t_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(t_ops);
  Comm *c = t_comm;
  t_comm = nullptr;
  return ops.empty() ? 0 : run_ops(c, ops);
}

FAKE_API int ncclSend(const void *p, size_t count, int dtype, int peer, Comm *c, hipStream_t st) {
  t_ops.push_back(Op{true, (void *)p, count * dtype_bytes(dtype), peer, st});
  t_comm = c;
  if (t_depth == 0) { std::vector<Op> ops; ops.swap(t_ops); return run_ops(c, ops); }
  return 0;
}
FAKE_API int ncclRecv(void *p, size_t count, int dtype, int peer, Comm *c, hipStream_t st) {
  t_ops.push_back(Op{false, p, count * dtype_bytes(dtype), peer, st});
  t_comm = c;
  if (t_depth == 0) { std::vector<Op> ops; ops.swap(t_ops); return run_ops(c, ops); }
  return 0;
}

FAKE_API int ncclAllGather(const void *s, void *r, size_t sendcount, int dtype, Comm *c, hipStream_t st) {
  World *w = c->w;
  const size_t bytes = sendcount * dtype_bytes(dtype);
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  w->ptrs[(size_t)c->rank] = s;
  barrier(w);
  for (int q = 0; q < w->n; ++q)
    if (hipMemcpyAsync((char *)r + (size_t)q * bytes, w->ptrs[(size_t)q], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  barrier(w);
  return 0;
}

FAKE_API int ncclAllReduce(const void *s, void *r, size_t count, int dtype, int op, Comm *c, hipStream_t st) {
  World *w = c->w;
  const size_t bytes = count * dtype_bytes(dtype);
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  w->ptrs[(size_t)c->rank] = s;
  barrier(w);
  // every rank reduces all inputs into a buffer of its own, then (after everybody has read) into its output
  void *tmp = nullptr;
  const void **d_src = nullptr;
  if (hipMalloc(&tmp, bytes ? bytes : 8) != hipSuccess) return 1;
  if (hipMalloc((void **)&d_src, sizeof(void *) * (size_t)w->n) != hipSuccess) return 1;
  if (hipMemcpy(d_src, w->ptrs.data(), sizeof(void *) * (size_t)w->n, hipMemcpyHostToDevice) != hipSuccess) return 1;
  const unsigned blocks = (unsigned)((count + 255) / 256);
  if (count) {
    if (dtype == 7 && op == 0) reduce_kernel<float, false><<<blocks, 256, 0, st>>>(d_src, w->n, (float *)tmp, count);
    else if (dtype == 5 && op == 3) reduce_kernel<unsigned long long, true><<<blocks, 256, 0, st>>>(d_src, w->n, (unsigned long long *)tmp, count);
    else die("reduction not implemented in the fake");
  }
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  barrier(w);
  if (hipMemcpyAsync(r, tmp, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  (void)hipFree(tmp); (void)hipFree(d_src);
  barrier(w);
  return 0;
}
