// lentil_upload.h -- the visit stream handed over piece by piece while the renderer is still rendering
// (lentil_hip_visits_begin / _append / _wait / _end, lentil_hip_host_alloc; include/lentil_hip.h).  Included by
// lentil_hip.hip.
//
// The reference adds every visit to the frame buffers the moment filter_pixel sees it (src/lentil_filter.cpp:66-436);
// a capturing filter_pixel instead has, at the end of the render, 80 B per visit on the host -- 6 GB for a 4K frame,
// 0.11 s of PCIe time against a 3 ms pass if it is sent in one piece then (lentil_hip_upload_visits).  Sent as the
// buckets finish, from pinned blocks, the transfer hides behind the render and the frame end only waits for the last
// block.  Copies go out on a stream of their own; a block may be reused once its ticket has been waited for.
#pragma once

struct LentilUpload {
  bool open = false;
  lentil_visits layout{};            // geometry of the stream being assembled (pointer members unused)
  uint64_t capacity = 0, n = 0;      // visits allocated / appended so far
  void *col[5 + LENTIL_MAX_AOVS - 1] = {};   // device columns: rgba, pos_z, raydir_time, volume_ignore, transmission, extras
  uint32_t *pixel = nullptr;
  float *inv_density = nullptr;
  // cryptomatte caches riding with the stream (lentil_hip_visits_begin_crypto): per AOV `crypto_entries` ids and
  // weights per visit; the columns are kept from frame to frame like the others
  uint32_t crypto_n = 0, crypto_entries = 0;           // this frame's stream (0: none)
  uint32_t crypto_alloc_n = 0, crypto_alloc_entries = 0;   // what the columns below were allocated for
  uint64_t crypto_capacity = 0;
  float *chash[LENTIL_MAX_CRYPTO] = {}, *cweight[LENTIL_MAX_CRYPTO] = {};
  hipStream_t stream = nullptr;
  std::mutex m;                      // appends come from many render threads
  // tickets: one event per append, in a ring (a ticket older than the ring is known to be complete when the oldest
  // event of the ring is)
  static constexpr uint32_t kRing = 256;
  hipEvent_t ev[kRing] = {};
  uint64_t next_ticket = 1;          // ticket t uses ev[t % kRing]
};

LENTIL_API int lentil_hip_host_alloc(void **host_ptr, uint64_t bytes) {
  if (!host_ptr || !bytes) return fail(nullptr, LENTIL_ERR_INVALID, "bad host_alloc arguments");
  hipError_t e = hipHostMalloc(host_ptr, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { *host_ptr = nullptr; return fail(nullptr, LENTIL_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_host_free(void *host_ptr) {
  if (host_ptr && hipHostFree(host_ptr) != hipSuccess) return fail(nullptr, LENTIL_ERR_HIP, "hipHostFree failed");
  return LENTIL_OK;
}

static void upload_free_crypto(lentil_hip_ctx *ctx, LentilUpload *u) {
  crypto_columns_gone(ctx);          // the context's cryptomatte module may be looking at them
  for (uint32_t a = 0; a < LENTIL_MAX_CRYPTO; ++a) {
    (void)hipFree(u->chash[a]); u->chash[a] = nullptr;
    (void)hipFree(u->cweight[a]); u->cweight[a] = nullptr;
  }
  u->crypto_alloc_n = u->crypto_alloc_entries = 0;
  u->crypto_capacity = 0;
}

static void upload_release(lentil_hip_ctx *ctx, bool free_columns) {
  LentilUpload *u = ctx->upload;
  if (!u) return;
  if (u->stream) (void)hipStreamSynchronize(u->stream);
  if (free_columns) {
    for (void *&p : u->col) { (void)hipFree(p); p = nullptr; }
    (void)hipFree(u->pixel); u->pixel = nullptr;
    (void)hipFree(u->inv_density); u->inv_density = nullptr;
    upload_free_crypto(ctx, u);
  }
  u->open = false;
  u->capacity = u->n = 0;
}

static void upload_destroy(lentil_hip_ctx *ctx) {
  LentilUpload *u = ctx->upload;
  if (!u) return;
  upload_release(ctx, true);
  for (hipEvent_t &e : u->ev) if (e) (void)hipEventDestroy(e);
  if (u->stream) (void)hipStreamDestroy(u->stream);
  delete u;
  ctx->upload = nullptr;
}

// (re)allocates the columns for `capacity` visits, keeping the first `keep` of them
static int upload_reserve(lentil_hip_ctx *ctx, LentilUpload *u, uint64_t capacity, uint64_t keep) {
  const uint32_t n_col = 5 + u->layout.n_extra;
  const bool ragged = u->layout.visits_per_pixel == 0;
  auto regrow = [&](void **p, size_t elem) -> int {
    void *q = nullptr;
    HIP_TRY(ctx, hipMalloc(&q, (size_t)capacity * elem));
    if (keep && *p) HIP_TRY(ctx, hipMemcpyAsync(q, *p, (size_t)keep * elem, hipMemcpyDeviceToDevice, u->stream));
    if (*p) { HIP_TRY(ctx, hipStreamSynchronize(u->stream)); (void)hipFree(*p); }
    *p = q;
    return LENTIL_OK;
  };
  int rc;
  for (uint32_t c = 0; c < n_col; ++c) if ((rc = regrow(&u->col[c], 16))) return rc;
  if (ragged) {
    if ((rc = regrow((void **)&u->pixel, 4))) return rc;
    if ((rc = regrow((void **)&u->inv_density, 4))) return rc;
  }
  u->capacity = capacity;
  if (u->crypto_n) {
    crypto_columns_gone(ctx);
    for (uint32_t a = 0; a < u->crypto_n; ++a) {
      if ((rc = regrow((void **)&u->chash[a], (size_t)u->crypto_entries * 4))) return rc;
      if ((rc = regrow((void **)&u->cweight[a], (size_t)u->crypto_entries * 4))) return rc;
    }
    u->crypto_capacity = capacity;
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_visits_begin(lentil_hip_ctx *ctx, const lentil_visits *layout, uint64_t capacity_hint) {
  CHECK_CTX(ctx);
  if (!layout) return fail(ctx, LENTIL_ERR_INVALID, "layout is null");
  if (layout->n_extra > LENTIL_MAX_AOVS - 1) return fail(ctx, LENTIL_ERR_INVALID, "too many extra AOV columns");
  if (layout->visits_per_pixel && (layout->pixels_per_row == 0 || layout->pixel_row_stride == 0))
    return fail(ctx, LENTIL_ERR_INVALID, "pixels_per_row / pixel_row_stride must be non-zero");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // the previous frame's pass no longer reads its visits
  free_visits(ctx);
  if (!ctx->upload) {
    ctx->upload = new (std::nothrow) LentilUpload();
    if (!ctx->upload) return fail(ctx, LENTIL_ERR_NOMEM, "out of host memory");
    const hipError_t e = hipStreamCreateWithFlags(&ctx->upload->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {         // (no half-made state: later begins would copy on the legacy null stream)
      delete ctx->upload;
      ctx->upload = nullptr;
      return fail(ctx, LENTIL_ERR_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
    }
  }
  LentilUpload *u = ctx->upload;
  std::lock_guard<std::mutex> g(u->m);
  const uint64_t want = capacity_hint ? capacity_hint : (1u << 20);
  const bool reuse = u->capacity >= want && u->layout.n_extra == layout->n_extra &&
                     (u->layout.visits_per_pixel == 0) == (layout->visits_per_pixel == 0);
  if (!reuse) {                      // the previous frame's columns serve again when they fit
    upload_release(ctx, true);
    u->layout = *layout;
    u->crypto_n = 0; u->crypto_entries = 0;       // (the previous frame's: lentil_hip_visits_begin_crypto announces this frame's, and allocates them)
    int rc = upload_reserve(ctx, u, want, 0);
    if (rc) return rc;
  }
  u->layout = *layout;
  u->n = 0;
  u->crypto_n = u->crypto_entries = 0;      // a plain stream unless lentil_hip_visits_begin_crypto follows
  u->open = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_visits_begin_crypto(lentil_hip_ctx *ctx, uint32_t entries) {
  CHECK_CTX(ctx);
  LentilUpload *u = ctx->upload;
  if (!u || !u->open) return fail(ctx, LENTIL_ERR_INVALID, "visits_begin_crypto without visits_begin");
  const uint32_t n_crypto = crypto_count(ctx);
  if (!n_crypto) return fail(ctx, LENTIL_ERR_INVALID, "no cryptomatte AOVs allocated (lentil_hip_alloc_crypto)");
  if (entries == 0 || entries > 64) return fail(ctx, LENTIL_ERR_INVALID, "crypto entries per visit must be 1..64");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> g(u->m);
  if (u->n) return fail(ctx, LENTIL_ERR_INVALID, "visits_begin_crypto after the first append");
  if (u->crypto_alloc_n != n_crypto || u->crypto_alloc_entries != entries || u->crypto_capacity < u->capacity) {
    upload_free_crypto(ctx, u);
    for (uint32_t a = 0; a < n_crypto; ++a) {
      HIP_TRY(ctx, hipMalloc(&u->chash[a], (size_t)u->capacity * entries * 4));
      HIP_TRY(ctx, hipMalloc(&u->cweight[a], (size_t)u->capacity * entries * 4));
    }
    u->crypto_alloc_n = n_crypto; u->crypto_alloc_entries = entries; u->crypto_capacity = u->capacity;
  }
  u->crypto_n = n_crypto;
  u->crypto_entries = entries;
  return LENTIL_OK;
}

static int visits_append_impl(lentil_hip_ctx *ctx, const lentil_visits *part, const lentil_crypto_visits *caches, uint64_t *ticket);
LENTIL_API int lentil_hip_visits_append(lentil_hip_ctx *ctx, const lentil_visits *part, uint64_t *ticket) {
  return visits_append_impl(ctx, part, nullptr, ticket);
}
LENTIL_API int lentil_hip_visits_append_crypto(lentil_hip_ctx *ctx, const lentil_visits *part, const lentil_crypto_visits *caches,
                                               uint64_t *ticket) {
  CHECK_CTX(ctx);
  if (!caches) return fail(ctx, LENTIL_ERR_INVALID, "caches is null");
  return visits_append_impl(ctx, part, caches, ticket);
}

static int visits_append_impl(lentil_hip_ctx *ctx, const lentil_visits *part, const lentil_crypto_visits *caches, uint64_t *ticket) {
  CHECK_CTX(ctx);
  LentilUpload *u = ctx->upload;
  if (!u || !u->open) return fail(ctx, LENTIL_ERR_INVALID, "visits_append without visits_begin");
  if (!part) return fail(ctx, LENTIL_ERR_INVALID, "part is null");
  if (ticket) *ticket = 0;
  if ((u->crypto_n != 0) != (caches != nullptr))
    return fail(ctx, LENTIL_ERR_INVALID, caches ? "visits_append_crypto on a stream without cryptomatte caches (lentil_hip_visits_begin_crypto)"
                                                : "this stream carries cryptomatte caches: lentil_hip_visits_append_crypto");
  if (caches) {
    if (caches->n != part->n || caches->n_crypto != u->crypto_n || caches->entries != u->crypto_entries)
      return fail(ctx, LENTIL_ERR_INVALID, "the caches do not match the part / the stream");
    for (uint32_t a = 0; a < caches->n_crypto; ++a)
      if (part->n && (!caches->hash[a] || !caches->weight[a])) return fail(ctx, LENTIL_ERR_INVALID, "a crypto column of the part is null");
  }
  if (part->n == 0) return LENTIL_OK;
  if (part->n_extra != u->layout.n_extra) return fail(ctx, LENTIL_ERR_INVALID, "part has another number of AOV columns");
  const bool ragged = u->layout.visits_per_pixel == 0;
  if (!part->rgba || !part->pos_z || !part->raydir_time || !part->volume_ignore || !part->transmission || (ragged && !part->pixel))
    return fail(ctx, LENTIL_ERR_INVALID, "a visit column of the part is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> g(u->m);
  if (u->n + part->n > 0xFFFFFFFFull) return fail(ctx, LENTIL_ERR_UNSUPPORTED, "more than 2^32 visits per frame");
  if (u->n + part->n > u->capacity) {
    uint64_t cap = u->capacity * 2;
    if (cap < u->n + part->n) cap = u->n + part->n;
    int rc = upload_reserve(ctx, u, cap, u->n);
    if (rc) return rc;
  }
  const void *src[5 + LENTIL_MAX_AOVS - 1] = {part->rgba, part->pos_z, part->raydir_time, part->volume_ignore, part->transmission};
  for (uint32_t k = 0; k < part->n_extra; ++k) {
    if (!part->extra[k]) return fail(ctx, LENTIL_ERR_INVALID, "an extra AOV column of the part is null");
    src[5 + k] = part->extra[k];
  }
  for (uint32_t c = 0; c < 5 + part->n_extra; ++c)
    HIP_TRY(ctx, hipMemcpyAsync((char *)u->col[c] + (size_t)u->n * 16, src[c], (size_t)part->n * 16, hipMemcpyHostToDevice, u->stream));
  if (ragged) {
    HIP_TRY(ctx, hipMemcpyAsync(u->pixel + u->n, part->pixel, (size_t)part->n * 4, hipMemcpyHostToDevice, u->stream));
    if (part->inv_density)
      HIP_TRY(ctx, hipMemcpyAsync(u->inv_density + u->n, part->inv_density, (size_t)part->n * 4, hipMemcpyHostToDevice, u->stream));
    else if (u->layout.inv_density)      // layout.inv_density != NULL announces per-visit densities: every part must bring them
      return fail(ctx, LENTIL_ERR_INVALID, "part without inv_density in a stream that has it");
  }
  if (caches) {
    const size_t row = (size_t)u->crypto_entries * 4;
    for (uint32_t a = 0; a < u->crypto_n; ++a) {
      HIP_TRY(ctx, hipMemcpyAsync((char *)u->chash[a] + (size_t)u->n * row, caches->hash[a], (size_t)part->n * row, hipMemcpyHostToDevice, u->stream));
      HIP_TRY(ctx, hipMemcpyAsync((char *)u->cweight[a] + (size_t)u->n * row, caches->weight[a], (size_t)part->n * row, hipMemcpyHostToDevice, u->stream));
    }
  }
  u->n += part->n;
  const uint64_t t = u->next_ticket++;
  hipEvent_t &e = u->ev[t % LentilUpload::kRing];
  if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventRecord(e, u->stream));
  if (ticket) *ticket = t;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_visits_wait(lentil_hip_ctx *ctx, uint64_t ticket) {
  CHECK_CTX(ctx);
  LentilUpload *u = ctx->upload;
  if (!u) return fail(ctx, LENTIL_ERR_INVALID, "no upload in progress");
  if (ticket == 0) return LENTIL_OK;
  hipEvent_t e;
  {
    std::lock_guard<std::mutex> g(u->m);
    if (ticket >= u->next_ticket) return fail(ctx, LENTIL_ERR_INVALID, "unknown ticket");
    // the ring slot of an old ticket now belongs to a later append of the same in-order stream: waiting for that
    // one covers it
    e = u->ev[ticket % LentilUpload::kRing];
  }
  HIP_TRY(ctx, hipEventSynchronize(e));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_visits_end(lentil_hip_ctx *ctx, uint64_t *n_visits) {
  CHECK_CTX(ctx);
  LentilUpload *u = ctx->upload;
  if (!u || !u->open) return fail(ctx, LENTIL_ERR_INVALID, "visits_end without visits_begin");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::lock_guard<std::mutex> g(u->m);
  HIP_TRY(ctx, hipStreamSynchronize(u->stream));
  lentil_visits d = u->layout;
  d.n = u->n;
  d.rgba = (const float *)u->col[0]; d.pos_z = (const float *)u->col[1]; d.raydir_time = (const float *)u->col[2];
  d.volume_ignore = (const float *)u->col[3]; d.transmission = (const float *)u->col[4];
  for (uint32_t k = 0; k < d.n_extra; ++k) d.extra[k] = (const float *)u->col[5 + k];
  const bool ragged = d.visits_per_pixel == 0;
  d.pixel = ragged ? u->pixel : nullptr;
  d.inv_density = (ragged && u->layout.inv_density) ? u->inv_density : nullptr;
  u->open = false;
  if (n_visits) *n_visits = u->n;
  int rc = check_visits(ctx, &d);
  if (rc) return rc;
  to_dev(ctx->V, &d);
  apply_camera_motion(ctx);
  ctx->V.id_base = ctx->visit_id_base;
  rc = ensure_worklist(ctx, d.n);
  if (rc) return rc;
  ctx->have_visits = true;          // the columns stay with the upload object (reused by the next frame's begin)
  ++ctx->visits_gen;
  if (u->crypto_n) crypto_bind_uploaded(ctx, u->crypto_n, u->crypto_entries, u->n, u->chash, u->cweight);
  return LENTIL_OK;
}
