// lentil_comm.h -- the exchange between the GPUs of one node, native: RCCL over xGMI, called through the C-ABI
// (lentil_hip_comm_* / _allreduce / _exchange_bands in include/lentil_hip.h).  Included by lentil_hip.hip.
//
// The reference has nothing like it (one process, threads sharing one set of buffers, src/lentil.h:823-851);
// SURVEY.md section 8(e) defines the step.  Two partitions, as in pota_amd/distributed.py, whose Python form of the
// same steps the world_size-2/3 gloo tests keep exercising:
//   interleaved rows + all-reduce   every rank holds the whole frame: closest-AOV winner keys are min-reduced
//                                   (ncclUint64 / ncclMin), winners gathered, then one sum all-reduce of the
//                                   accumulator block
//   row bands (tiled output)        touched rows all-gathered (one small ncclAllGather), what a rank added to
//                                   another's band goes there point to point -- a list of pixel entries when the rows
//                                   are mostly empty, packed rows otherwise -- the owner merges and resolves its band
// RCCL is loaded at run time (dlopen): liblentil_hip.so does not depend on it unless a communicator is asked for, and
// a process that already has an RCCL loaded (torch's) gets that same instance by soname.
#pragma once
#include <dlfcn.h>

typedef struct ncclComm *lentil_ncclComm_t;
struct LentilNcclId { char b[128]; };     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value
struct LentilRccl {
  void *lib = nullptr;
  int (*GetUniqueId)(void *id) = nullptr;
  int (*CommInitRank)(lentil_ncclComm_t *comm, int nranks, LentilNcclId id, int rank) = nullptr;
  int (*CommDestroy)(lentil_ncclComm_t comm) = nullptr;
  int (*AllReduce)(const void *s, void *r, size_t count, int dtype, int op, lentil_ncclComm_t comm, hipStream_t st) = nullptr;
  int (*AllGather)(const void *s, void *r, size_t sendcount, int dtype, lentil_ncclComm_t comm, hipStream_t st) = nullptr;
  int (*Send)(const void *s, size_t count, int dtype, int peer, lentil_ncclComm_t comm, hipStream_t st) = nullptr;
  int (*Recv)(void *r, size_t count, int dtype, int peer, lentil_ncclComm_t comm, hipStream_t st) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
// rccl.h: ncclSum 0, ncclMin 3; ncclInt8 0, ncclUint8 1, ncclInt64 4, ncclUint64 5, ncclFloat32 7
enum { kNcclSum = 0, kNcclMin = 3, kNcclUint8 = 1, kNcclInt64 = 4, kNcclUint64 = 5, kNcclFloat32 = 7 };

static LentilRccl g_rccl;
static std::mutex g_rccl_mutex;

static const char *load_rccl() {
  std::lock_guard<std::mutex> g(g_rccl_mutex);
  if (g_rccl.lib) return nullptr;
  const char *names[] = {getenv("LENTIL_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) {
    if (!n || !n[0]) continue;
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return "librccl.so not found (LENTIL_RCCL_LIB names it)";
  LentilRccl r;
  r.lib = h;
#define LENTIL_RCCL_SYM(field, name)                                \
  *(void **)(&r.field) = dlsym(h, name);                            \
  if (!r.field) return "librccl.so lacks " name;
  LENTIL_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
  LENTIL_RCCL_SYM(CommInitRank, "ncclCommInitRank")
  LENTIL_RCCL_SYM(CommDestroy, "ncclCommDestroy")
  LENTIL_RCCL_SYM(AllReduce, "ncclAllReduce")
  LENTIL_RCCL_SYM(AllGather, "ncclAllGather")
  LENTIL_RCCL_SYM(Send, "ncclSend")
  LENTIL_RCCL_SYM(Recv, "ncclRecv")
  LENTIL_RCCL_SYM(GroupStart, "ncclGroupStart")
  LENTIL_RCCL_SYM(GroupEnd, "ncclGroupEnd")
  LENTIL_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef LENTIL_RCCL_SYM
  g_rccl = r;
  return nullptr;
}

struct LentilComm {
  lentil_ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int64_t *d_meta_mine = nullptr, *d_meta_all = nullptr, *h_meta_all = nullptr;   // [2 + world] / [world][2 + world]
  std::vector<void *> scratch;          // grow-only device buffers, one per use slot
  std::vector<size_t> scratch_bytes;
  uint64_t last_sent = 0, last_received = 0;     // payload bytes of the last exchange_bands / allreduce (lentil_hip_exchange_stats)
  // Fixed-capacity form of the tiled exchange (exchange_bands_fixed): entries the last message of each directed pair
  // carried, as both of its ends know it (-1: none yet) -- the capacity of the next one follows from it on both sides
  std::vector<int64_t> hist_out, hist_in;
  uint32_t *h_hdr = nullptr;                     // pinned: [world] headers to send, [world] sent (read back), [world] received
  uint64_t n_fixed = 0, n_fixed_overflow = 0;    // exchanges in that form / directed pairs whose entries did not fit
  // An exchange in that form that returned an error on this rank.  Its peers' histories have moved on (or will, once their
  // receives time out or complete) while this rank's have not, so the two ends of a pair would size their next message
  // differently: every later exchange on this communicator is refused at once instead of posting sends and receives of
  // unequal length (round-4 ADVICE).  Destroy and re-create the communicators on all ranks.
  bool poisoned = false;
};

// One message of the fixed-capacity form, ONE send per directed pair: 4 header words {entries found, capacity, first row,
// end row of the region the sender compacted}, `cap` pixel indices, `cap` records of 4 n_aovs + 1 floats, `cap` winner keys
// per key plane the frame has.
constexpr uint32_t kHdrWords = 4;
static uint32_t pair_capacity(int64_t hist, uint64_t band_pix) {
  // (LENTIL_EXCHANGE_CAP_FIRST: the capacity of a pair's first message -- tests make first messages overflow with it)
  static const uint64_t first_ = getenv("LENTIL_EXCHANGE_CAP_FIRST") ? (uint64_t)atoll(getenv("LENTIL_EXCHANGE_CAP_FIRST")) : 0ull;
  uint64_t c = hist < 0 ? (first_ ? first_ : band_pix / 16) : 2 * (uint64_t)hist + 256;      // twice what the pair carried last time
  if (c < 1024 && !(hist < 0 && first_)) c = 1024;
  if (c > band_pix) c = band_pix;                                          // (never more entries than the band has pixels)
  return (uint32_t)((c + 1ull) & ~1ull);                                   // (even: the keys behind the records stay 8-byte aligned)
}

#define RCCL_TRY(ctx, call)                                                                           \
  do {                                                                                                \
    const int r_ = (call);                                                                            \
    if (r_ != 0) return fail(ctx, LENTIL_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

static int comm_scratch(lentil_hip_ctx *ctx, LentilComm *cm, size_t slot, size_t bytes, void **out) {
  if (cm->scratch.size() <= slot) { cm->scratch.resize(slot + 1, nullptr); cm->scratch_bytes.resize(slot + 1, 0); }
  if (cm->scratch_bytes[slot] < bytes) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // nobody is still using the old buffer
    (void)hipFree(cm->scratch[slot]);
    cm->scratch[slot] = nullptr;
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(ctx, hipMalloc(&cm->scratch[slot], want));
    cm->scratch_bytes[slot] = want;
  }
  *out = cm->scratch[slot];
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_comm_unique_id(uint8_t id[128]) {
  if (!id) return fail(nullptr, LENTIL_ERR_INVALID, "id is null");
  if (const char *e = load_rccl()) return fail(nullptr, LENTIL_ERR_UNSUPPORTED, e);
  const int r = g_rccl.GetUniqueId(id);
  if (r != 0) return fail(nullptr, LENTIL_ERR_HIP, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_comm_destroy(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  LentilComm *cm = ctx->comm;
  if (!cm) return LENTIL_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (cm->comm) (void)g_rccl.CommDestroy(cm->comm);
  (void)hipFree(cm->d_meta_mine); (void)hipFree(cm->d_meta_all);
  if (cm->h_meta_all) (void)hipHostFree(cm->h_meta_all);
  if (cm->h_hdr) (void)hipHostFree(cm->h_hdr);
  for (void *p : cm->scratch) (void)hipFree(p);
  delete cm;
  ctx->comm = nullptr;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_comm_init(lentil_hip_ctx *ctx, const uint8_t id[128], int rank, int world) {
  CHECK_CTX(ctx);
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(ctx, LENTIL_ERR_INVALID, "bad communicator arguments");
  if (const char *e = load_rccl()) return fail(ctx, LENTIL_ERR_UNSUPPORTED, e);
  int rc = lentil_hip_comm_destroy(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  LentilComm *cm = new (std::nothrow) LentilComm();
  if (!cm) return fail(ctx, LENTIL_ERR_NOMEM, "out of host memory");
  cm->rank = rank; cm->world = world;
  LentilNcclId uid;
  memcpy(uid.b, id, 128);
  // (the context only gets a communicator that is complete: a half-built one would be handed to RCCL by the next
  // exchange, and would make the cryptomatte pass refuse the context for being "multi-GPU")
  auto undo = [&]() {
    if (cm->comm) (void)g_rccl.CommDestroy(cm->comm);
    (void)hipFree(cm->d_meta_mine); (void)hipFree(cm->d_meta_all);
    if (cm->h_meta_all) (void)hipHostFree(cm->h_meta_all);
    if (cm->h_hdr) (void)hipHostFree(cm->h_hdr);
    delete cm;
  };
  {
    const int r = g_rccl.CommInitRank(&cm->comm, world, uid, rank);
    if (r != 0) { cm->comm = nullptr; undo(); return fail(ctx, LENTIL_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
  }
  const size_t m = (size_t)(2 + world);
  hipError_t e = hipMalloc(&cm->d_meta_mine, m * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc(&cm->d_meta_all, m * (size_t)world * sizeof(int64_t));
  if (e == hipSuccess) e = hipHostMalloc((void **)&cm->h_meta_all, m * (size_t)world * sizeof(int64_t), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void **)&cm->h_hdr, (size_t)3 * world * kHdrWords * sizeof(uint32_t), hipHostMallocDefault);
  if (e != hipSuccess) { undo(); return fail(ctx, LENTIL_ERR_HIP, std::string("communicator buffers: ") + hipGetErrorString(e)); }
  cm->hist_out.assign((size_t)world, -1);
  cm->hist_in.assign((size_t)world, -1);
  ctx->comm = cm;
  return LENTIL_OK;
}

// payload bytes this rank sent / received in its last lentil_hip_exchange_bands (entries or packed rows, keys included) or
// lentil_hip_allreduce (ring traffic 2 (G - 1) / G of the reduced buffers, both ways)
LENTIL_API int lentil_hip_exchange_stats(lentil_hip_ctx *ctx, uint64_t *bytes_sent, uint64_t *bytes_received) {
  CHECK_CTX(ctx);
  if (!ctx->comm) return fail(ctx, LENTIL_ERR_INVALID, "no communicator (lentil_hip_comm_init)");
  if (bytes_sent) *bytes_sent = ctx->comm->last_sent;
  if (bytes_received) *bytes_received = ctx->comm->last_received;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_exchange_counts(lentil_hip_ctx *ctx, uint64_t *fixed_form, uint64_t *pairs_overflowed) {
  CHECK_CTX(ctx);
  if (!ctx->comm) return fail(ctx, LENTIL_ERR_INVALID, "no communicator (lentil_hip_comm_init)");
  if (fixed_form) *fixed_form = ctx->comm->n_fixed;
  if (pairs_overflowed) *pairs_overflowed = ctx->comm->n_fixed_overflow;
  return LENTIL_OK;
}

// interleaved partition: every rank ends up with the whole frame's accumulators (distributed.frame_step)
LENTIL_API int lentil_hip_allreduce(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  ctx->resolved_valid = false;
  LentilComm *cm = ctx->comm;
  if (!cm) return fail(ctx, LENTIL_ERR_INVALID, "no communicator (lentil_hip_comm_init)");
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if (ctx->crypto) return fail(ctx, LENTIL_ERR_UNSUPPORTED, "cryptomatte AOVs travel with the tiled exchange only (lentil_hip_exchange_bands)");
  if ((rc = fold_direct(ctx, 0, ctx->F.np, true))) return rc;        // what the scan kept apart joins the sum
  untrust_touched(ctx);
  if (ctx->F.zkey || ctx->F.zkey_dbg) {
    if (!ctx->closest_deferred)
      return fail(ctx, LENTIL_ERR_INVALID, "closest-filtered AOVs: lentil_hip_set_closest_exchange(ctx, 1, ...) before the pass");
    if (ctx->F.zkey)
      RCCL_TRY(ctx, g_rccl.AllReduce(ctx->F.zkey, ctx->F.zkey, ctx->F.np, kNcclUint64, kNcclMin, cm->comm, ctx->stream));
    if (ctx->F.zkey_dbg)       // lentil_debug: a key plane of its own (src/lentil.h:838-845)
      RCCL_TRY(ctx, g_rccl.AllReduce(ctx->F.zkey_dbg, ctx->F.zkey_dbg, ctx->F.np, kNcclUint64, kNcclMin, cm->comm, ctx->stream));
    if ((rc = lentil_hip_closest_gather(ctx))) return rc;
  }
  RCCL_TRY(ctx, g_rccl.AllReduce(ctx->F.acc, ctx->F.acc, ctx->F.np * ctx->F.stride, kNcclFloat32, kNcclSum, cm->comm, ctx->stream));
  {
    uint64_t bytes = ctx->F.np * ctx->F.stride * 4ull;
    if (ctx->F.zkey) bytes += ctx->F.np * 8ull;
    if (ctx->F.zkey_dbg) bytes += ctx->F.np * 8ull;
    cm->last_sent = cm->last_received = cm->world > 1 ? bytes * 2ull * (uint64_t)(cm->world - 1) / (uint64_t)cm->world : 0ull;
  }
  return LENTIL_OK;
}

// The tiled exchange without a host round trip in its middle.  The form below this one compacts
// per destination, reads every count back (a stream sync each), all-gathers the counts (another sync) and only then knows
// the sizes of its sends and receives.  Here the size of every message is known to both of its ends BEFORE the pass: a
// capacity that follows from what the same directed pair carried in the previous exchange (pair_capacity; both ends saw
// that number -- the sender counted it, the receiver read it in the header), so compaction, sends, receives, merges and
// the band's resolve are enqueued back to back and the merge kernels take the entry count from the header word that
// arrived.  The host looks at the headers once, at the end: they give the next capacities, and a pair whose entries did
// not fit (count > capacity: nothing of it was merged) sends the rows of its region whole, as the sized form does.
static int exchange_bands_fixed(lentil_hip_ctx *ctx, LentilComm *cm, const int32_t *bounds, int32_t visit_rows,
                                int32_t b_lo, int32_t b_hi, int32_t lo, int32_t hi) {
  const int world = cm->world, rank = cm->rank;
  const int32_t yres = (int32_t)ctx->P.yres;
  const uint32_t xres = ctx->P.xres;
  auto band_of = [&](int r, int32_t &l, int32_t &h) {
    if (bounds) { l = bounds[r]; h = bounds[r + 1]; }
    else { l = (int32_t)((int64_t)visit_rows * r / world); h = (int32_t)((int64_t)visit_rows * (r + 1) / world); }
    if (r == world - 1) h = yres;
  };
  const uint32_t used = 4u * ctx->F.n_aovs + 1u;
  const bool keys = ctx->F.zkey != nullptr, dkeys = ctx->F.zkey_dbg != nullptr;
  int rc;
  struct Msg { uint32_t cap = 0; int32_t lo = 0, hi = 0; size_t bytes = 0; uint32_t *idx = nullptr; float *vals = nullptr; unsigned long long *k = nullptr, *kd = nullptr; };
  std::vector<Msg> out((size_t)world), in((size_t)world);
  uint32_t *h_send = cm->h_hdr, *h_sent = cm->h_hdr + (size_t)world * kHdrWords, *h_got = cm->h_hdr + (size_t)2 * world * kHdrWords;
  const uint64_t my_pix = (uint64_t)(b_hi > b_lo ? b_hi - b_lo : 0) * xres;
  auto buffers = [&](Msg &m, int q, size_t slot) -> int {
    m.bytes = ((size_t)kHdrWords + (size_t)m.cap * (1u + used)) * 4 + (size_t)m.cap * 8 * ((keys ? 1u : 0u) + (dkeys ? 1u : 0u));
    void *p;
    const int r = comm_scratch(ctx, cm, (size_t)q * 12 + slot, m.bytes, &p);
    if (r) return r;
    m.idx = (uint32_t *)p;                                   // (the header first: idx[0 .. kHdrWords), entries from idx + kHdrWords)
    m.vals = reinterpret_cast<float *>(m.idx + kHdrWords + m.cap);
    unsigned long long *kp = reinterpret_cast<unsigned long long *>(m.vals + (size_t)m.cap * used);
    if (keys) { m.k = kp; kp += m.cap; }
    if (dkeys) m.kd = kp;
    return LENTIL_OK;
  };
  // ---- what this rank added to every other band: compacted into that pair's message, the count stays on the device
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    int32_t q_lo, q_hi;
    band_of(q, q_lo, q_hi);
    Msg &o = out[(size_t)q];
    o.cap = pair_capacity(cm->hist_out[(size_t)q], (uint64_t)(q_hi > q_lo ? q_hi - q_lo : 0) * xres);
    o.lo = lo > q_lo ? lo : q_lo;
    o.hi = hi < q_hi ? hi : q_hi;
    if (o.hi < o.lo) o.hi = o.lo;
    if ((rc = buffers(o, q, 0))) return rc;
    uint32_t *h = h_send + (size_t)q * kHdrWords;
    h[0] = 0u; h[1] = o.cap; h[2] = (uint32_t)o.lo; h[3] = (uint32_t)o.hi;
    HIP_TRY(ctx, hipMemcpyAsync(o.idx, h, kHdrWords * 4, hipMemcpyHostToDevice, ctx->stream));
    if (o.hi > o.lo && o.cap)
      if ((rc = compact_rows_impl(ctx, (uint32_t)o.lo, (uint32_t)(o.hi - o.lo), o.idx + kHdrWords, o.vals, o.k, o.kd, o.cap, nullptr,
                                  reinterpret_cast<unsigned int *>(o.idx)))) return rc;
    Msg &i = in[(size_t)q];
    i.cap = pair_capacity(cm->hist_in[(size_t)q], my_pix);
    if ((rc = buffers(i, q, 4))) return rc;
  }
  // ---- the exchange: every message has its size on both sides already
  RCCL_TRY(ctx, g_rccl.GroupStart());
  int g_err = 0;
  uint64_t sent = 0, received = 0;
  auto snd = [&](const void *p, size_t n, int type, size_t elem, int q) {
    if (!n) return;
    if (!g_err) g_err = g_rccl.Send(p, n, type, q, cm->comm, ctx->stream);
    sent += (uint64_t)n * elem;
  };
  auto rcv = [&](void *p, size_t n, int type, size_t elem, int q) {
    if (!n) return;
    if (!g_err) g_err = g_rccl.Recv(p, n, type, q, cm->comm, ctx->stream);
    received += (uint64_t)n * elem;
  };
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    snd(out[(size_t)q].idx, out[(size_t)q].bytes, kNcclUint8, 1, q);
    rcv(in[(size_t)q].idx, in[(size_t)q].bytes, kNcclUint8, 1, q);
  }
  {
    const int r_end = g_rccl.GroupEnd();
    if (g_err) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(g_err));
    if (r_end) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r_end));
  }
  // ---- cryptomatte maps: what this rank's draws added outside its band goes to the owners (lentil_crypto.h)
  if (ctx->crypto) {
    std::vector<int32_t> bands((size_t)(2 * world), 0);
    for (int q = 0; q < world; ++q) band_of(q, bands[(size_t)(2 * q)], bands[(size_t)(2 * q + 1)]);
    if ((rc = crypto_exchange_bands(ctx, bands.data(), lo, hi))) return rc;
  }
  // ---- merge what arrived, senders in rank order; the kernels read the count (the rows touched: anywhere in the band)
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    const Msg &i = in[(size_t)q];
    if (i.cap && b_hi > b_lo)
      if ((rc = merge_sparse_impl(ctx, (uint32_t)b_lo, (uint32_t)(b_hi - b_lo), i.cap, i.idx + kHdrWords, i.vals, i.k, i.kd, i.idx))) return rc;
  }
  if ((rc = lentil_hip_resolve_rows(ctx, (uint32_t)b_lo, (uint32_t)(b_hi - b_lo)))) return rc;
  // ---- the headers, once everything is on its way
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    HIP_TRY(ctx, hipMemcpyAsync(h_sent + (size_t)q * kHdrWords, out[(size_t)q].idx, kHdrWords * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h_got + (size_t)q * kHdrWords, in[(size_t)q].idx, kHdrWords * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ++cm->n_fixed;
  struct Redo { int q; bool send; int32_t lo, hi; float *packed = nullptr; unsigned long long *key_rows = nullptr, *dkey_rows = nullptr; };
  std::vector<Redo> redo;
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    const uint32_t *hs = h_sent + (size_t)q * kHdrWords, *hg = h_got + (size_t)q * kHdrWords;
    if (hg[1] != in[(size_t)q].cap)
      return fail(ctx, LENTIL_ERR_HIP, "tiled exchange: the ranks disagree about a message's capacity (an exchange failed on one of them: "
                                       "destroy and re-create the communicators)");
    cm->hist_out[(size_t)q] = hs[0];
    cm->hist_in[(size_t)q] = hg[0];
    if (hs[0] > out[(size_t)q].cap) redo.push_back({q, true, out[(size_t)q].lo, out[(size_t)q].hi});
    if (hg[0] > in[(size_t)q].cap) redo.push_back({q, false, (int32_t)hg[2], (int32_t)hg[3]});
  }
  if (!redo.empty()) {
    // entries that did not fit their message: those regions travel as whole rows (both ends of the pair know it)
    cm->n_fixed_overflow += redo.size();
    for (Redo &r : redo) {
      if (r.lo < 0 || r.hi > yres || r.hi <= r.lo) return fail(ctx, LENTIL_ERR_HIP, "tiled exchange: bad region in a message header");
      const uint64_t n_pix = (uint64_t)(r.hi - r.lo) * xres;
      void *p;
      if ((rc = comm_scratch(ctx, cm, (size_t)r.q * 12 + (r.send ? 3 : 5), (size_t)n_pix * used * 4, &p))) return rc;
      r.packed = (float *)p;
      if (r.send) {
        if ((rc = lentil_hip_pack_rows(ctx, (uint32_t)r.lo, (uint32_t)(r.hi - r.lo), r.packed))) return rc;
      } else {
        if (keys) { if ((rc = comm_scratch(ctx, cm, (size_t)r.q * 12 + 6, (size_t)n_pix * 8, &p))) return rc; r.key_rows = (unsigned long long *)p; }
        if (dkeys) { if ((rc = comm_scratch(ctx, cm, (size_t)r.q * 12 + 9, (size_t)n_pix * 8, &p))) return rc; r.dkey_rows = (unsigned long long *)p; }
      }
    }
    RCCL_TRY(ctx, g_rccl.GroupStart());
    for (const Redo &r : redo) {
      const uint64_t n_pix = (uint64_t)(r.hi - r.lo) * xres;
      if (r.send) {
        snd(r.packed, (size_t)n_pix * used, kNcclFloat32, 4, r.q);
        if (keys) snd(ctx->F.zkey + (uint64_t)r.lo * xres, (size_t)n_pix, kNcclUint64, 8, r.q);
        if (dkeys) snd(ctx->F.zkey_dbg + (uint64_t)r.lo * xres, (size_t)n_pix, kNcclUint64, 8, r.q);
      } else {
        rcv(r.packed, (size_t)n_pix * used, kNcclFloat32, 4, r.q);
        if (keys) rcv(r.key_rows, (size_t)n_pix, kNcclUint64, 8, r.q);
        if (dkeys) rcv(r.dkey_rows, (size_t)n_pix, kNcclUint64, 8, r.q);
      }
    }
    {
      const int r_end = g_rccl.GroupEnd();
      if (g_err) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(g_err));
      if (r_end) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r_end));
    }
    bool merged = false;
    for (const Redo &r : redo)
      if (!r.send) {
        if ((rc = merge_rows_impl(ctx, (uint32_t)r.lo, (uint32_t)(r.hi - r.lo), r.packed, r.key_rows, r.dkey_rows, true))) return rc;
        merged = true;
      }
    if (merged && (rc = lentil_hip_resolve_rows(ctx, (uint32_t)b_lo, (uint32_t)(b_hi - b_lo)))) return rc;
  }
  cm->last_sent = sent; cm->last_received = received;
  return LENTIL_OK;
}

// row bands (distributed.frame_step_bands): call after lentil_hip_redistribute; ends with the band resolved
LENTIL_API int lentil_hip_exchange_bands(lentil_hip_ctx *ctx, const int32_t *bounds, int32_t visit_rows, int32_t sparse,
                                         int32_t *band_lo, int32_t *band_hi) {
  CHECK_CTX(ctx);
  ctx->resolved_valid = false;
  LentilComm *cm = ctx->comm;
  if (!cm) return fail(ctx, LENTIL_ERR_INVALID, "no communicator (lentil_hip_comm_init)");
  if (!ctx->have_frame || visit_rows <= 0) return fail(ctx, LENTIL_ERR_INVALID, "bad exchange_bands arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int world = cm->world, rank = cm->rank;
  const int32_t yres = (int32_t)ctx->P.yres;
  const uint32_t xres = ctx->P.xres;
  auto band_of = [&](int r, int32_t &lo, int32_t &hi) {
    if (bounds) { lo = bounds[r]; hi = bounds[r + 1]; }
    else { lo = (int32_t)((int64_t)visit_rows * r / world); hi = (int32_t)((int64_t)visit_rows * (r + 1) / world); }
    if (r == world - 1) hi = yres;      // the last band also owns the rows beyond the visits (yres = H + 1)
  };
  int32_t b_lo, b_hi;
  band_of(rank, b_lo, b_hi);
  if (band_lo) *band_lo = b_lo;
  if (band_hi) *band_hi = b_hi;
  int rc;
  int32_t lo = 0, hi = 0;
  if ((rc = lentil_hip_touched_rows(ctx, &lo, &hi))) { if (sparse) cm->poisoned = true; return rc; }      // (the peers go on without this rank's message)
  const uint32_t used = 4u * ctx->F.n_aovs + 1u;
  const bool keys = ctx->F.zkey != nullptr, dkeys = ctx->F.zkey_dbg != nullptr;
  if ((keys || dkeys) && ctx->closest_deferred)
    return fail(ctx, LENTIL_ERR_INVALID, "tiled exchange: the pass must gather its own winners (set_closest_exchange(ctx, 0, ...))");

  // (LENTIL_EXCHANGE_FIXED=0: the sized form below, whose sends carry exactly the entries found -- and whose host waits for
  // every count; sparse == 0, whole rows always, is that form too)
  static const bool fixed_form = !(getenv("LENTIL_EXCHANGE_FIXED") && getenv("LENTIL_EXCHANGE_FIXED")[0] == '0');
  if (sparse && fixed_form) {
    if (cm->poisoned)
      return fail(ctx, LENTIL_ERR_INVALID, "tiled exchange: an earlier exchange on this communicator failed on this rank; its message "
                                           "sizes no longer agree with its peers' (lentil_hip_comm_destroy / _comm_init on every rank)");
    rc = exchange_bands_fixed(ctx, cm, bounds, visit_rows, b_lo, b_hi, lo, hi);
    if (rc) cm->poisoned = true;
    return rc;
  }

  // ---- what this rank added to every other band, and the form it will travel in
  struct Out { int form = 0; int32_t s_lo = 0, s_hi = 0; uint32_t *idx = nullptr; float *vals = nullptr; unsigned long long *k = nullptr, *kd = nullptr; float *packed = nullptr; };
  std::vector<Out> out((size_t)world);
  std::vector<int64_t> mine((size_t)(2 + world), 0);
  mine[0] = lo; mine[1] = hi;
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    int32_t q_lo, q_hi;
    band_of(q, q_lo, q_hi);
    Out &o = out[(size_t)q];
    o.s_lo = lo > q_lo ? lo : q_lo;
    o.s_hi = hi < q_hi ? hi : q_hi;
    if (o.s_hi <= o.s_lo) continue;
    o.form = -1;
    const uint64_t n_pix = (uint64_t)(o.s_hi - o.s_lo) * xres;
    if (sparse) {
      const uint32_t cap = (uint32_t)(n_pix / 4 > 1024 ? n_pix / 4 : 1024);
      void *p;
      if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 0, (size_t)cap * 4, &p))) return rc;
      o.idx = (uint32_t *)p;
      if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 1, (size_t)cap * used * 4, &p))) return rc;
      o.vals = (float *)p;
      if (keys) { if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 2, (size_t)cap * 8, &p))) return rc; o.k = (unsigned long long *)p; }
      if (dkeys) { if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 8, (size_t)cap * 8, &p))) return rc; o.kd = (unsigned long long *)p; }
      uint32_t n = 0;
      if ((rc = compact_rows_impl(ctx, (uint32_t)o.s_lo, (uint32_t)(o.s_hi - o.s_lo), o.idx, o.vals, o.k, o.kd, cap, &n, nullptr))) return rc;
      if (n <= cap) o.form = (int)n;
    }
    if (o.form < 0) {
      void *p;
      if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 3, (size_t)n_pix * used * 4, &p))) return rc;
      o.packed = (float *)p;
      if ((rc = lentil_hip_pack_rows(ctx, (uint32_t)o.s_lo, (uint32_t)(o.s_hi - o.s_lo), o.packed))) return rc;
    }
    mine[(size_t)(2 + q)] = o.form;
  }
  // ---- one small all-gather: [touched lo, hi, form per destination] of every rank
  const size_t m = (size_t)(2 + world);
  HIP_TRY(ctx, hipMemcpyAsync(cm->d_meta_mine, mine.data(), m * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  RCCL_TRY(ctx, g_rccl.AllGather(cm->d_meta_mine, cm->d_meta_all, m, kNcclInt64, cm->comm, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(cm->h_meta_all, cm->d_meta_all, m * (size_t)world * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // ---- receive buffers from what the others announced
  struct In { int form = 0; int32_t r_lo = 0, r_hi = 0; uint32_t *idx = nullptr; float *vals = nullptr; unsigned long long *k = nullptr, *kd = nullptr; float *packed = nullptr; unsigned long long *key_rows = nullptr, *dkey_rows = nullptr; };
  std::vector<In> in((size_t)world);
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    const int64_t *info = cm->h_meta_all + (size_t)q * m;
    In &i = in[(size_t)q];
    i.r_lo = (int32_t)(info[0] > b_lo ? info[0] : b_lo);
    i.r_hi = (int32_t)(info[1] < b_hi ? info[1] : b_hi);
    i.form = (int)info[2 + rank];
    if (i.r_hi <= i.r_lo || i.form == 0) { i.form = 0; continue; }
    void *p;
    if (i.form > 0) {
      if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 4, (size_t)i.form * 4, &p))) return rc;
      i.idx = (uint32_t *)p;
      if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 5, (size_t)i.form * used * 4, &p))) return rc;
      i.vals = (float *)p;
      if (keys) { if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 6, (size_t)i.form * 8, &p))) return rc; i.k = (unsigned long long *)p; }
      if (dkeys) { if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 9, (size_t)i.form * 8, &p))) return rc; i.kd = (unsigned long long *)p; }
    } else {
      const uint64_t n_pix = (uint64_t)(i.r_hi - i.r_lo) * xres;
      if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 5, (size_t)n_pix * used * 4, &p))) return rc;
      i.packed = (float *)p;
      if (keys) { if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 6, (size_t)n_pix * 8, &p))) return rc; i.key_rows = (unsigned long long *)p; }
      if (dkeys) { if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 9, (size_t)n_pix * 8, &p))) return rc; i.dkey_rows = (unsigned long long *)p; }
    }
  }
  // ---- the exchange itself: every send has its receive on the other side, in the same order per pair
  // (an error inside the group is remembered and returned after GroupEnd: an early return would leave the group open)
  RCCL_TRY(ctx, g_rccl.GroupStart());
  int g_err = 0;
  uint64_t sent = 0, received = 0;
  auto snd = [&](const void *p, size_t n, int type, size_t elem, int q) {
    if (!g_err) g_err = g_rccl.Send(p, n, type, q, cm->comm, ctx->stream);
    sent += (uint64_t)n * elem;
  };
  auto rcv = [&](void *p, size_t n, int type, size_t elem, int q) {
    if (!g_err) g_err = g_rccl.Recv(p, n, type, q, cm->comm, ctx->stream);
    received += (uint64_t)n * elem;
  };
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    const Out &o = out[(size_t)q];
    if (o.form > 0) {
      snd(o.idx, (size_t)o.form * 4, kNcclUint8, 1, q);
      snd(o.vals, (size_t)o.form * used, kNcclFloat32, 4, q);
      if (keys) snd(o.k, (size_t)o.form, kNcclUint64, 8, q);
      if (dkeys) snd(o.kd, (size_t)o.form, kNcclUint64, 8, q);
    } else if (o.form < 0) {
      const uint64_t n_pix = (uint64_t)(o.s_hi - o.s_lo) * xres;
      snd(o.packed, (size_t)n_pix * used, kNcclFloat32, 4, q);
      if (keys) snd(ctx->F.zkey + (uint64_t)o.s_lo * xres, (size_t)n_pix, kNcclUint64, 8, q);
      if (dkeys) snd(ctx->F.zkey_dbg + (uint64_t)o.s_lo * xres, (size_t)n_pix, kNcclUint64, 8, q);
    }
    const In &i = in[(size_t)q];
    if (i.form > 0) {
      rcv(i.idx, (size_t)i.form * 4, kNcclUint8, 1, q);
      rcv(i.vals, (size_t)i.form * used, kNcclFloat32, 4, q);
      if (keys) rcv(i.k, (size_t)i.form, kNcclUint64, 8, q);
      if (dkeys) rcv(i.kd, (size_t)i.form, kNcclUint64, 8, q);
    } else if (i.form < 0) {
      const uint64_t n_pix = (uint64_t)(i.r_hi - i.r_lo) * xres;
      rcv(i.packed, (size_t)n_pix * used, kNcclFloat32, 4, q);
      if (keys) rcv(i.key_rows, (size_t)n_pix, kNcclUint64, 8, q);
      if (dkeys) rcv(i.dkey_rows, (size_t)n_pix, kNcclUint64, 8, q);
    }
  }
  {
    const int r_end = g_rccl.GroupEnd();
    if (g_err) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(g_err));
    if (r_end) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r_end));
  }
  cm->last_sent = sent; cm->last_received = received;
  // ---- cryptomatte maps: what this rank's draws added outside its band goes to the owners (lentil_crypto.h)
  if (ctx->crypto) {
    std::vector<int32_t> bands((size_t)(2 * world), 0);
    for (int q = 0; q < world; ++q) band_of(q, bands[(size_t)(2 * q)], bands[(size_t)(2 * q + 1)]);
    if ((rc = crypto_exchange_bands(ctx, bands.data(), lo, hi))) return rc;
  }
  // ---- merge what arrived (senders in rank order: the merge of one sender's entries is not atomic against another's)
  for (int q = 0; q < world; ++q) {
    const In &i = in[(size_t)q];
    if (i.form > 0) {
      if ((rc = merge_sparse_impl(ctx, (uint32_t)i.r_lo, (uint32_t)(i.r_hi - i.r_lo), (uint32_t)i.form, i.idx, i.vals, i.k, i.kd))) return rc;
    } else if (i.form < 0) {
      if ((rc = merge_rows_impl(ctx, (uint32_t)i.r_lo, (uint32_t)(i.r_hi - i.r_lo), i.packed, i.key_rows, i.dkey_rows, true))) return rc;
    }
  }
  return lentil_hip_resolve_rows(ctx, (uint32_t)b_lo, (uint32_t)(b_hi - b_lo));
}
