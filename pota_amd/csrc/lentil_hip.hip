// lentil_hip.hip -- kernels + C-ABI of liblentil_hip.so (gfx950 only).
//
// Kernels (SURVEY.md section 8a/8d names):
//   scan_kernel      K1+K2+K6: reads the visit columns (80+16K B/visit, HBM-bound), evaluates the
//                    redistribute predicate and draw count, wave-ballot/prefix-sum compacts the
//                    redistributed visits into a work list, and accumulates the non-redistributed
//                    visits of each source pixel in reference order through wave-private LDS.
//   draw_kernel      K3/K4/K5: one wave per redistributed visit, lanes = backward-trace attempts;
//                    polynomial tables and the bokeh row CDF staged in LDS; fp32 atomic splat.
//   resolve_kernel   K7: weight normalisation.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lentil_device.h"
#include "generated/lens_registry.h"

using namespace lentil;

#define LENTIL_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------
// device-side bookkeeping
// ---------------------------------------------------------------------------------------
struct DevCounters {
  unsigned long long work_count;     // entries pushed by the scan kernel
  unsigned long long queue_head;     // next work item to hand out
  unsigned long long visits;
  unsigned long long redistributed;
  unsigned long long attempted;
  unsigned long long accepted;
  unsigned long long overflow;
  unsigned long long log_count;
  unsigned long long newton_iters;   // lane-iterations of the Newton solver (draw kernel)
  unsigned long long tries;          // aperture draws / solves started
  unsigned long long lane_rounds;    // 64 x scheduler rounds (issue slots offered)
};

// Shared scheduling state of one work item (redistributed visit); any number of waves may serve the
// same item.  slots = accepted + in-flight attempts (never exceeds `samples`), next = next attempt
// index to hand out (attempts are started strictly in increasing order).
struct ItemState {
  uint32_t slots;
  uint32_t next;
  uint32_t accepted;
  uint32_t last_ok;    // highest accepted attempt index
};

struct VisitsDev {
  uint64_t n;
  uint32_t visits_per_pixel, pixels_per_row;
  int32_t pixel_x0, pixel_y0;
  uint32_t pixel_row_stride, n_extra;
  const float4 *rgba, *pos_z, *raydir_time, *volume_ignore, *transmission;
  const float4 *extra[LENTIL_MAX_AOVS - 1];
  const uint32_t *pixel;
  const float *inv_density;
};

struct FrameDev {
  float *acc;        // [n_aovs][np][4]
  float *weight;     // [np]
  uint32_t n_aovs;
  uint64_t np;       // xres*yres
};

struct ScanArgs {
  lentil_params P;
  double lens_length;
  VisitsDev V;
  FrameDev F;
  uint2 *work;
  ItemState *state;
  uint64_t work_cap;
  DevCounters *ctr;
  uint32_t ppt;      // pixels per wave tile (uniform mode)
  uint32_t tv_pad;   // staging entries per wave (>= ppt * visits_per_pixel)
};

struct DrawArgs {
  lentil_params P;
  const DevLens *lens;     // header, global memory
  const DevTerm *terms;    // global memory
  DevBokeh bokeh;
  VisitsDev V;
  FrameDev F;
  const uint2 *work;
  ItemState *state;
  uint64_t work_cap;
  DevCounters *ctr;
  lentil_draw_record *log;
  uint64_t log_cap;
};

LD_DEV void visit_pixel(const VisitsDev &V, uint64_t v, int &px, int &py) {
  if (V.visits_per_pixel) {
    const uint64_t p = v / V.visits_per_pixel;
    px = V.pixel_x0 + (int)(p % V.pixels_per_row);
    py = V.pixel_y0 + (int)(p / V.pixels_per_row) * (int)V.pixel_row_stride;
  } else {
    const uint32_t q = V.pixel[v];
    px = (int)(q & 0xFFFFu);
    py = (int)(q >> 16);
  }
}

LD_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave-ballot + prefix-sum compaction of flagged lanes into the work list (K2)
LD_DEV void push_work(bool flagged, uint32_t visit, uint32_t samples, uint2 *work, ItemState *state, uint64_t cap,
                      DevCounters *ctr) {
  const unsigned long long mask = __ballot(flagged);
  if (mask == 0ull) return;
  const uint32_t lane = lane_id();
  const uint32_t leader = (uint32_t)__builtin_ctzll(mask);
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(&ctr->work_count, (unsigned long long)__builtin_popcountll(mask));
  base = __shfl(base, (int)leader);
  if (flagged) {
    const unsigned long long idx = base + (unsigned long long)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (idx < cap) {
      work[idx] = make_uint2(visit, samples);
      reinterpret_cast<uint4 *>(state)[idx] = make_uint4(0u, 0u, 0u, 0u);
    } else {
      atomicAdd(&ctr->overflow, 1ull);
    }
  }
}

// ---------------------------------------------------------------------------------------
// K1+K2+K6, uniform footprints.  One wave owns a tile of `ppt` consecutive source pixels
// (= ppt*M consecutive visits, read as fully coalesced 1 KiB column loads), stages the weighted
// contributions in wave-private LDS, then lane p adds up pixel p's M entries in iterator order --
// the order the reference accumulates them (filter_and_add_to_buffer_new, src/lentil.h:938-955).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_uniform_kernel(ScanArgs a) {
  extern __shared__ float4 smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t waves_per_block = blockDim.x >> 6;
  float4 *sval = smem + (size_t)wave * a.tv_pad;
  float *sw = reinterpret_cast<float *>(smem + (size_t)waves_per_block * a.tv_pad) + (size_t)wave * a.tv_pad;

  const VisitsDev &V = a.V;
  const uint32_t M = V.visits_per_pixel;
  const uint32_t ppt = a.ppt;
  const uint32_t TV = ppt * M;
  const uint64_t n_pixels = (V.n + M - 1) / M;
  const uint64_t n_tiles = (n_pixels + ppt - 1) / ppt;
  const uint64_t wave_global = (uint64_t)blockIdx.x * waves_per_block + wave;
  const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;
  const uint32_t xres = a.P.xres;
  unsigned long long n_redis = 0;

  for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_stride) {
    const uint64_t pix0 = tile * ppt;
    const uint64_t v0 = pix0 * M;
    for (uint32_t eb = 0; eb < TV; eb += 64) {
      const uint32_t e = eb + lane;
      const uint64_t v = v0 + e;
      const bool valid = (e < TV) && (v < V.n);
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      float w = 0.f;
      bool flagged = false;
      int samples = 0;
      if (valid) {
        const float4 rgba = V.rgba[v];
        const float4 pz = V.pos_z[v];
        const float4 rt = V.raydir_time[v];
        const float4 vi = V.volume_ignore[v];
        const float4 tr = V.transmission[v];
        const float invd = V.inv_density ? V.inv_density[v] : a.P.inverse_sample_density;
        const VisitInfo I = visit_prologue(a.P, a.lens_length, rgba, pz, rt, vi, tr, invd);
        if (I.redistribute) {
          flagged = true;
          samples = I.samples;
        } else {
          w = 1.0f * invd;                              // filter_weight * inv_density, lentil.h:949-953
          val = make_float4((rgba.x + 0.0f) * w, (rgba.y + 0.0f) * w, (rgba.z + 0.0f) * w, (rgba.w + 0.0f) * w);
        }
      }
      push_work(flagged, (uint32_t)v, (uint32_t)samples, a.work, a.state, a.work_cap, a.ctr);
      n_redis += flagged ? 1ull : 0ull;
      if (e < TV) { sval[e] = val; sw[e] = w; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint64_t pix = pix0 + lane;
    const bool own = (lane < ppt) && (pix < n_pixels);
    uint64_t lin = 0;
    if (own) {
      const int px = V.pixel_x0 + (int)(pix % V.pixels_per_row);
      const int py = V.pixel_y0 + (int)(pix / V.pixels_per_row) * (int)V.pixel_row_stride;
      lin = (uint64_t)px + (uint64_t)py * xres;
      float4 s = reinterpret_cast<float4 *>(a.F.acc)[lin];
      float ws = a.F.weight[lin];
      for (uint32_t j = 0; j < M; ++j) {
        const float4 c = sval[lane * M + j];
        const float cw = sw[lane * M + j];
        if (cw != 0.0f) { s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; ws += cw; }
      }
      reinterpret_cast<float4 *>(a.F.acc)[lin] = s;
      a.F.weight[lin] = ws;
    }
    // extra AOVs: same weights, one column at a time through the same staging area
    for (uint32_t k = 0; k < V.n_extra; ++k) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (uint32_t eb = 0; eb < TV; eb += 64) {
        const uint32_t e = eb + lane;
        const uint64_t v = v0 + e;
        if (e < TV) {
          const float w = sw[e];
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (v < V.n && w != 0.0f) {
            const float4 c = V.extra[k][v];
            x = make_float4((c.x + 0.0f) * w, (c.y + 0.0f) * w, (c.z + 0.0f) * w, (c.w + 0.0f) * w);
          }
          sval[e] = x;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (own) {
        float4 *dst = reinterpret_cast<float4 *>(a.F.acc) + (size_t)(k + 1) * a.F.np + lin;
        float4 s = *dst;
        for (uint32_t j = 0; j < M; ++j) {
          if (sw[lane * M + j] != 0.0f) {
            const float4 c = sval[lane * M + j];
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
          }
        }
        *dst = s;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // counters: one atomic per wave
  for (int off = 32; off > 0; off >>= 1) n_redis += __shfl_down(n_redis, off);
  if (lane == 0 && n_redis) atomicAdd(&a.ctr->redistributed, n_redis);
}

// K1+K2+K6 for ragged footprints (explicit per-visit pixel): lane per visit, fp32 atomics for the
// direct accumulation.
__global__ __launch_bounds__(256) void scan_ragged_kernel(ScanArgs a) {
  const VisitsDev &V = a.V;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long n_redis = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t n_round = (V.n + 63ull) & ~63ull;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_round; v += stride) {
    bool flagged = false;
    int samples = 0;
    if (v < V.n) {
      const float4 rgba = V.rgba[v];
      const float invd = V.inv_density ? V.inv_density[v] : a.P.inverse_sample_density;
      const VisitInfo I = visit_prologue(a.P, a.lens_length, rgba, V.pos_z[v], V.raydir_time[v],
                                         V.volume_ignore[v], V.transmission[v], invd);
      if (I.redistribute) {
        flagged = true;
        samples = I.samples;
      } else {
        int px, py;
        visit_pixel(V, v, px, py);
        const uint64_t lin = (uint64_t)px + (uint64_t)py * a.P.xres;
        const float w = 1.0f * invd;
        float *d = a.F.acc + lin * 4;
        atomicAdd(d + 0, (rgba.x + 0.0f) * w);
        atomicAdd(d + 1, (rgba.y + 0.0f) * w);
        atomicAdd(d + 2, (rgba.z + 0.0f) * w);
        atomicAdd(d + 3, (rgba.w + 0.0f) * w);
        atomicAdd(a.F.weight + lin, w);
        for (uint32_t k = 0; k < V.n_extra; ++k) {
          const float4 c = V.extra[k][v];
          float *dk = a.F.acc + ((size_t)(k + 1) * a.F.np + lin) * 4;
          atomicAdd(dk + 0, (c.x + 0.0f) * w);
          atomicAdd(dk + 1, (c.y + 0.0f) * w);
          atomicAdd(dk + 2, (c.z + 0.0f) * w);
          atomicAdd(dk + 3, (c.w + 0.0f) * w);
        }
      }
    }
    push_work(flagged, (uint32_t)v, (uint32_t)samples, a.work, a.state, a.work_cap, a.ctr);
    n_redis += flagged ? 1ull : 0ull;
  }
  for (int off = 32; off > 0; off >>= 1) n_redis += __shfl_down(n_redis, off);
  if (lane == 0 && n_redis) atomicAdd(&a.ctr->redistributed, n_redis);
}

// ---------------------------------------------------------------------------------------
// K3/K4/K5: draws.  One wave per work item (redistributed visit).
//
// Acceptance rule of the reference (src/lentil_filter.cpp:248,272,285): the accepted draws are the
// first `samples` successes among attempts n = 0 .. 5*samples-1, in attempt order.  Both variants
// below start attempts strictly in increasing n and never have more attempts outstanding than
// accepted draws are still missing, so every success they see belongs to that set -- no ranking,
// no ordered commit.
//
// Polynomial optics (draw_po_kernel): a lane-level scheduler.  Every round each busy lane advances
// its own Newton solve by ONE iteration (the straight-line polynomial block, ~2k fp64 ops, runs
// with a full exec mask); lanes whose solve ended finish it (pupil tests, retry with the next
// aperture draw, or sensor->pixel + atomic splat) and are refilled with the next attempt.  Solves
// that need 11 or 40 iterations, or 1 or 16 vignetting retries, no longer hold 63 other lanes.
// ---------------------------------------------------------------------------------------
constexpr int kMaxBokehRows = 2048;

struct Splat {
  const DrawArgs &a;
  uint32_t visit;
  float4 rgba;
  float ae, w;
  LD_DEV void operator()(uint32_t pix, uint32_t attempt) const {
    // Camera::add_to_buffer, src/lentil.h:827-830
    float *d = a.F.acc + (size_t)pix * 4;
    atomicAdd(d + 0, (rgba.x + ae) * w);
    atomicAdd(d + 1, (rgba.y + ae) * w);
    atomicAdd(d + 2, (rgba.z + ae) * w);
    atomicAdd(d + 3, (rgba.w + ae) * w);
    atomicAdd(a.F.weight + pix, w);
    for (uint32_t k = 0; k < a.V.n_extra; ++k) {
      const float4 c = a.V.extra[k][visit];
      float *dk = a.F.acc + ((size_t)(k + 1) * a.F.np + pix) * 4;
      atomicAdd(dk + 0, (c.x + ae) * w);
      atomicAdd(dk + 1, (c.y + ae) * w);
      atomicAdd(dk + 2, (c.z + ae) * w);
      atomicAdd(dk + 3, (c.w + ae) * w);
    }
    if (a.log_cap) {
      const unsigned long long li = atomicAdd(&a.ctr->log_count, 1ull);
      if (li < a.log_cap) { a.log[li].visit = visit; a.log[li].attempt = attempt; a.log[li].pixel = pix; }
    }
  }
};

struct ItemHeader {
  uint32_t visit, samples;
  int px, py;
  VisitInfo I;
  float4 rgba;
  float w;
};

LD_DEV ItemHeader load_item(const DrawArgs &a, unsigned long long item, double lens_length) {
  ItemHeader h;
  const uint2 wi = a.work[item];
  h.visit = __builtin_amdgcn_readfirstlane(wi.x);
  h.samples = __builtin_amdgcn_readfirstlane(wi.y);
  const uint32_t v = h.visit;
  h.rgba = a.V.rgba[v];
  const float invd = a.V.inv_density ? a.V.inv_density[v] : a.P.inverse_sample_density;
  h.I = visit_prologue(a.P, lens_length, h.rgba, a.V.pos_z[v], a.V.raydir_time[v], a.V.volume_ignore[v],
                       a.V.transmission[v], invd);
  visit_pixel(a.V, v, h.px, h.py);
  const float inv_samples = (float)(1.0 / (double)(float)(int)h.samples);
  h.w = 1.0f * invd * inv_samples;              // src/lentil_filter.cpp:297
  return h;
}

// One wave serving (part of) one work item.  Any number of waves may serve the same item
// concurrently; they coordinate through the item's ItemState in global memory:
//   - a lane may only start an attempt after reserving a slot (slots <= samples), so the number of
//     successes can never exceed `samples`;
//   - attempt indices come from one fetch-add counter, so the started attempts always form a prefix
//     0..next-1 and every started attempt is run to its end;
//   => the successes are exactly "the first `samples` successes in attempt order" of the reference.
// A failed attempt gives its slot back; the wave that owned it re-acquires on its next round, so an
// item always has a wave working on it while it is incomplete.
template <class LensT>
LD_DEV void po_item(const DrawArgs &a, const LensT &L, const float *cdfRow, const ItemHeader &h, ItemState *st,
                    uint32_t &st_iters, uint32_t &st_tries, uint32_t &st_rounds) {
  const lentil_params &P = a.P;
  const DevLens &k = L.consts();
  const uint32_t lane = lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const double target[3] = {-(double)h.I.cs[0] * 10.0, -(double)h.I.cs[1] * 10.0, -(double)h.I.cs[2] * 10.0};
  const Splat splat{a, h.visit, h.rgba, h.I.add_energy, h.w};
  const uint32_t samples = h.samples, max_total = samples * 5u;
  const uint32_t seed_a = (uint32_t)(h.px * h.py + h.px);

  bool busy = false, need_init = false, exhausted = false;
  uint32_t n = 0, backoff = 0;
  int t = 0;
  double ap_x = 0.0, ap_y = 0.0;
  NewtonState s;
  newton_init(s);

  while (true) {
    // ---- acquire attempts for the idle lanes
    const unsigned long long busy_mask = __ballot(busy);
    const uint32_t inflight = (uint32_t)__builtin_popcountll(busy_mask);
    const uint32_t n_idle = 64u - inflight;
    uint32_t grant = 0, base = 0;
    if (n_idle > 0 && !exhausted && backoff == 0) {
      if (lane == 0) {
        const uint32_t old = atomicAdd(&st->slots, n_idle);
        uint32_t g = (old >= samples) ? 0u : (samples - old < n_idle ? samples - old : n_idle);
        if (g < n_idle) atomicSub(&st->slots, n_idle - g);
        if (g > 0) {
          const uint32_t b = atomicAdd(&st->next, g);
          const uint32_t g2 = (b >= max_total) ? 0u : (max_total - b < g ? max_total - b : g);
          if (g2 < g) atomicSub(&st->slots, g - g2);
          if (g2 == 0) grant = 0xFFFFFFFFu;      // no attempts left at all
          else { grant = g2; base = b; }
        }
      }
      grant = __builtin_amdgcn_readfirstlane(grant);
      base = __builtin_amdgcn_readfirstlane(base);
      if (grant == 0xFFFFFFFFu) { exhausted = true; grant = 0; }
      else if (grant == 0) backoff = 4;            // item full right now: do not hammer the counter
    } else if (backoff) {
      --backoff;
    }
    if (!busy) {
      const uint32_t my_rank = (uint32_t)__builtin_popcountll(~busy_mask & lt_mask);
      if (my_rank < grant) { busy = true; need_init = true; n = base + my_rank; t = 0; }
    }
    if (inflight + grant == 0u) break;             // nothing in flight here and nothing to get

    // ---- (re)start a solve: aperture draw for (attempt n, try t), src/lentil.h:596-609
    if (busy && need_init) {
      po_aperture_sample(P, a.bokeh, cdfRow, seed_a, n + (uint32_t)t, ap_x, ap_y);
      newton_init(s);
      need_init = false;
      ++st_tries;
    }
    // ---- one Newton iteration for every busy lane
    if (busy) { newton_iter(L, target, ap_x, ap_y, s); ++st_iters; }
    ++st_rounds;

    // ---- lanes whose solve ended
    bool succ = false, failed = false;
    if (busy && !newton_continue(s)) {
      double out4;
      const float transmittance = (float)newton_finish(L, s, out4);
      bool try_ok = !(transmittance <= 0);
      if (try_ok) {
        const double ipx = s.x + s.dx * k.back_focal_length;
        const double ipy = s.y + s.dy * k.back_focal_length;
        if (ipx * ipx + ipy * ipy > k.inner_pupil_radius * k.inner_pupil_radius) try_ok = false;
      }
      if (try_ok) {
        const double sx = s.x + s.dx * -P.sensor_shift;
        const double sy = s.y + s.dy * -P.sensor_shift;
        uint32_t pix;
        if (po_sensor_to_pixel(P, sx, sy, pix)) {
          succ = true;
          splat(pix, n);
        } else {
          failed = true;
        }
        busy = false;
      } else {
        ++t;
        if (t > P.vignetting_retries) { busy = false; failed = true; } else need_init = true;
      }
    }
    const unsigned long long succ_mask = __ballot(succ);
    const unsigned long long fail_mask = __ballot(failed);
    if (succ_mask | fail_mask) {
      // highest successful attempt index of this round (attempt indices grow with the lane rank only
      // within one grant, so take a real max)
      uint32_t mx = succ ? n : 0u;
      for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(mx, off);
        if (o > mx) mx = o;
      }
      if (lane == 0) {
        const uint32_t nf = (uint32_t)__builtin_popcountll(fail_mask);
        const uint32_t ns = (uint32_t)__builtin_popcountll(succ_mask);
        if (nf) atomicSub(&st->slots, nf);
        if (ns) { atomicAdd(&st->accepted, ns); atomicMax(&st->last_ok, mx); }
      }
      if (fail_mask) backoff = 0;                  // a slot just became free
    }
  }
}

template <class LensT, bool kTables>
__global__ __launch_bounds__(256) void draw_po_kernel(DrawArgs a) {
  __shared__ DevTerm s_terms[kTables ? kMaxTerms : 1];
  __shared__ DevLens s_k;
  __shared__ float s_cdfRow[kMaxBokehRows];
  if (kTables) {
    const uint32_t nt = a.lens->n_terms;
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) s_terms[i] = a.terms[i];
  }
  if (threadIdx.x == 0) s_k = *a.lens;
  const bool row_in_lds = a.P.bokeh_enable_image && a.bokeh.y <= kMaxBokehRows;
  if (row_in_lds)
    for (int i = threadIdx.x; i < a.bokeh.y; i += blockDim.x) s_cdfRow[i] = a.bokeh.cdfRow[i];
  __syncthreads();
  const float *cdfRow = row_in_lds ? s_cdfRow : a.bokeh.cdfRow;
  LensT L;
  if constexpr (kTables) { L.terms = s_terms; L.k = &s_k; } else { L.k = &s_k; }

  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long n_items = a.ctr->work_count;
  if (n_items > a.work_cap) n_items = a.work_cap;
  // Tickets: ticket q serves item q % n_items as its (q / n_items)-th helper wave, so every item gets a
  // first wave before any item gets a second one.  An item with `samples` draws can use at most
  // ceil(samples / 64) waves at once.
  const uint32_t max_samples = a.P.samples_override > 0 ? (uint32_t)a.P.samples_override : 2000u;
  const unsigned long long max_helpers = (max_samples + 63u) / 64u;
  const unsigned long long n_tickets = n_items * max_helpers;
  uint32_t st_iters = 0, st_tries = 0, st_rounds = 0;
  while (true) {
    unsigned long long q = 0;
    if (lane == 0) q = atomicAdd(&a.ctr->queue_head, 1ull);
    q = __shfl(q, 0);
    if (q >= n_tickets) break;
    const unsigned long long item = q % n_items;
    const uint32_t helper = (uint32_t)(q / n_items);
    const uint32_t samples = __builtin_amdgcn_readfirstlane(a.work[item].y);
    if (helper * 64u >= samples) continue;
    ItemState *st = a.state + item;
    if (helper > 0) {
      // cheap look before doing the full item set-up: complete or fully reserved items need no helper
      const uint32_t sl = __hip_atomic_load(&st->slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t nx = __hip_atomic_load(&st->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sl >= samples || nx >= samples * 5u) continue;
    }
    const ItemHeader h = load_item(a, item, s_k.length);
    po_item(a, L, cdfRow, h, st, st_iters, st_tries, st_rounds);
  }
  unsigned long long it64 = st_iters, tr64 = st_tries, rd64 = st_rounds;
  for (int off = 32; off > 0; off >>= 1) {
    it64 += __shfl_down(it64, off);
    tr64 += __shfl_down(tr64, off);
    rd64 += __shfl_down(rd64, off);
  }
  if (lane == 0) {
    if (it64) atomicAdd(&a.ctr->newton_iters, it64);
    if (tr64) atomicAdd(&a.ctr->tries, tr64);
    if (rd64) atomicAdd(&a.ctr->lane_rounds, rd64);
  }
}

// total_samples_taken / accepted statistics of the PO items (src/lentil_filter.cpp:248): one thread
// per item after the draw kernel.
__global__ __launch_bounds__(256) void po_item_stats_kernel(DrawArgs a) {
  unsigned long long n_items = a.ctr->work_count;
  if (n_items > a.work_cap) n_items = a.work_cap;
  unsigned long long att = 0, acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += stride) {
    const ItemState st = a.state[i];
    const uint32_t samples = a.work[i].y;
    acc += st.accepted;
    att += (st.accepted >= samples) ? (unsigned long long)st.last_ok + 1ull : (unsigned long long)samples * 5ull;
  }
  for (int off = 32; off > 0; off >>= 1) {
    att += __shfl_down(att, off);
    acc += __shfl_down(acc, off);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (att) atomicAdd(&a.ctr->attempted, att);
    if (acc) atomicAdd(&a.ctr->accepted, acc);
  }
}

// Thin lens (K4): closed form, no retries -- lanes are consecutive attempts, a chunk never holds
// more attempts than accepted draws are still missing.
__global__ __launch_bounds__(256) void draw_thinlens_kernel(DrawArgs a) {
  __shared__ float s_cdfRow[kMaxBokehRows];
  const bool row_in_lds = a.P.bokeh_enable_image && a.bokeh.y <= kMaxBokehRows;
  if (row_in_lds)
    for (int i = threadIdx.x; i < a.bokeh.y; i += blockDim.x) s_cdfRow[i] = a.bokeh.cdfRow[i];
  __syncthreads();
  const float *cdfRow = row_in_lds ? s_cdfRow : a.bokeh.cdfRow;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long n_items = a.ctr->work_count;
  if (n_items > a.work_cap) n_items = a.work_cap;
  unsigned long long tot_attempted = 0, tot_accepted = 0;
  while (true) {
    unsigned long long item = 0;
    if (lane == 0) item = atomicAdd(&a.ctr->queue_head, 1ull);
    item = __shfl(item, 0);
    if (item >= n_items) break;
    const ItemHeader h = load_item(a, item, 0.0);
    const Splat splat{a, h.visit, h.rgba, h.I.add_energy, h.w};
    uint32_t accepted = 0, n_base = 0;
    const uint32_t samples = h.samples, max_total = samples * 5u;
    while (accepted < samples && n_base < max_total) {
      uint32_t chunk = samples - accepted;
      if (chunk > 64u) chunk = 64u;
      if (chunk > max_total - n_base) chunk = max_total - n_base;
      const uint32_t n = n_base + lane;
      bool ok = false;
      uint32_t pix = 0;
      if (lane < chunk) ok = thinlens_draw(a.P, a.bokeh, cdfRow, h.I.cs, h.px, h.py, n, pix);
      if (ok) splat(pix, n);
      accepted += (uint32_t)__builtin_popcountll(__ballot(ok));
      n_base += chunk;
    }
    tot_attempted += n_base;
    tot_accepted += accepted;
  }
  if (lane == 0) {
    if (tot_attempted) atomicAdd(&a.ctr->attempted, tot_attempted);
    if (tot_accepted) atomicAdd(&a.ctr->accepted, tot_accepted);
  }
}

// K7 -- driver_process_bucket's normalisation, src/lentil_imager.cpp:169-186
__global__ __launch_bounds__(256) void resolve_kernel(FrameDev F, float *resolved) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t total = F.np * F.n_aovs;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint64_t p = i % F.np;
    float4 c = reinterpret_cast<const float4 *>(F.acc)[i];
    const float wt = F.weight[p];
    if (wt != 0.0f) {                      // AtRGBA /= float multiplies by 1.0f/f
      const float inv = 1.0f / wt;
      c.x *= inv; c.y *= inv; c.z *= inv; c.w *= inv;
    }
    reinterpret_cast<float4 *>(resolved)[i] = c;
  }
}

// ---------------------------------------------------------------------------------------
// single-function test kernels
// ---------------------------------------------------------------------------------------
struct TestArgs {
  lentil_params P;
  const DevLens *lens;
  const DevTerm *terms;
  DevBokeh bokeh;
  uint64_t n;
  const double *in0;   // scene / target
  const double *in1;   // ap
  const int32_t *i0, *i1, *i2;
  const uint32_t *u0, *u1;
  double lambda;
  double *o0, *o1, *o2;
  int32_t *oi;
};

template <int WHAT>
__global__ __launch_bounds__(256) void test_kernel(TestArgs t) {
  __shared__ DevTerm s_terms[kMaxTerms];
  __shared__ DevLens s_k;
  if (t.lens) {
    const uint32_t nt = t.lens->n_terms;
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) s_terms[i] = t.terms[i];
    if (threadIdx.x == 0) {
      s_k = *t.lens;
      if (WHAT == 0) {   // explicit lambda: recompute lens_ipow(lambda, e) exactly like the host does
        s_k.lambda_pow[0] = 1.0; s_k.lambda_pow[1] = t.lambda;
        for (uint32_t e = 2; e <= kMaxExp; ++e) s_k.lambda_pow[e] = ipow_u(t.lambda, e);
      }
    }
  }
  __syncthreads();
  const LdsLens L{s_terms, &s_k};
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // keep whole waves converged: table reads go through readfirstlane
  const uint64_t ii = i < t.n ? i : t.n - 1;
  if (WHAT == 0) {
    double sensor[4], out[5] = {0, 0, 0, 0, t.lambda};
    const double scene[3] = {t.in0[ii * 3], t.in0[ii * 3 + 1], t.in0[ii * 3 + 2]};
    int iters = 0;
    const double T = lt_sample_aperture(L, scene, t.in1[ii * 2], t.in1[ii * 2 + 1], sensor, out, &iters);
    if (i < t.n) {
      for (int c = 0; c < 4; ++c) t.o0[i * 5 + c] = sensor[c];
      t.o0[i * 5 + 4] = t.lambda;
      for (int c = 0; c < 5; ++c) t.o1[i * 5 + c] = out[c];
      t.o2[i] = T;
      if (t.oi) t.oi[i] = iters;
    }
  } else if (WHAT == 1) {
    const double target[3] = {t.in0[ii * 3], t.in0[ii * 3 + 1], t.in0[ii * 3 + 2]};
    double sx = 0, sy = 0;
    const bool ok = trace_ray_bw_po(t.P, L, t.bokeh, t.bokeh.cdfRow, target, t.i0[ii], t.i1[ii], t.i2[ii], sx, sy);
    if (i < t.n) { t.o0[i * 2] = sx; t.o0[i * 2 + 1] = sy; t.oi[i] = ok ? 1 : 0; }
  } else {
    double ax, ay;
    po_aperture_sample(t.P, t.bokeh, t.bokeh.cdfRow, t.u0[ii], t.u1[ii], ax, ay);
    if (i < t.n) { t.o0[i * 2] = ax; t.o0[i * 2 + 1] = ay; }
  }
}

// =======================================================================================
// host side
// =======================================================================================
struct lentil_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::string err;
  int num_cu = 256;

  lentil_params P{};
  bool have_params = false;

  DevLens hlens{};
  DevLens *d_lens = nullptr;
  DevTerm *d_terms = nullptr;
  bool have_lens = false;
  unsigned long long lens_hash = 0;   // FNV-1a of the base table, matches gen::Lens_*::kTableHash
  bool use_generated = true;          // LENTIL_FORCE_TABLES=1 forces the table interpreter

  DevBokeh bokeh{};
  bool have_bokeh = false;

  FrameDev F{};
  uint8_t kind[LENTIL_MAX_AOVS] = {0};
  float *d_resolved = nullptr;
  bool have_frame = false;

  VisitsDev V{};
  bool have_visits = false;
  std::vector<void *> owned_visit_mem;

  uint2 *d_work = nullptr;
  ItemState *d_state = nullptr;
  uint64_t work_cap = 0;
  DevCounters *d_ctr = nullptr;
  lentil_draw_record *d_log = nullptr;
  uint64_t log_cap = 0;
  bool timed_draw = false, timed_resolve = false;
};

static thread_local std::string g_err;

static int fail(lentil_hip_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg; else g_err = msg;
  return code;
}
#define HIP_TRY(ctx, call)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(ctx, LENTIL_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
  } while (0)
#define CHECK_CTX(ctx) \
  if (!(ctx)) return fail(nullptr, LENTIL_ERR_INVALID, "null context")

static double host_ipow(double x, int e) {   // lens_ipow, src/lens.h:226-233
  if (e == 0) return 1.0;
  if (e == 1) return x;
  if (e == 2) return x * x;
  const double p2 = host_ipow(x, e / 2);
  if (e & 1) return x * p2 * p2;
  return p2 * p2;
}

LENTIL_API int lentil_hip_abi_version(void) { return LENTIL_ABI_VERSION; }

LENTIL_API int lentil_hip_create(int device, lentil_hip_ctx **out_ctx) {
  if (!out_ctx) return fail(nullptr, LENTIL_ERR_INVALID, "out_ctx is null");
  *out_ctx = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, LENTIL_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(nullptr, LENTIL_ERR_INVALID, "device index out of range");
  lentil_hip_ctx *ctx = new (std::nothrow) lentil_hip_ctx();
  if (!ctx) return fail(nullptr, LENTIL_ERR_NOMEM, "out of host memory");
  ctx->device = device;
  HIP_TRY(ctx, hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (auto &ev : ctx->ev) HIP_TRY(ctx, hipEventCreate(&ev));
  HIP_TRY(ctx, hipMalloc(&ctx->d_ctr, sizeof(DevCounters)));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctr, 0, sizeof(DevCounters), ctx->stream));
  if (const char *ft = getenv("LENTIL_FORCE_TABLES")) ctx->use_generated = !(ft[0] == '1');
  *out_ctx = ctx;
  return LENTIL_OK;
}

static void free_visits(lentil_hip_ctx *ctx) {
  for (void *p : ctx->owned_visit_mem) (void)hipFree(p);
  ctx->owned_visit_mem.clear();
  ctx->have_visits = false;
}
static void free_bokeh(lentil_hip_ctx *ctx) {
  if (ctx->have_bokeh) {
    (void)hipFree((void *)ctx->bokeh.cdfRow);
    (void)hipFree((void *)ctx->bokeh.rowIndices);
    (void)hipFree((void *)ctx->bokeh.cdfColumn);
    (void)hipFree((void *)ctx->bokeh.columnIndices);
  }
  ctx->bokeh = DevBokeh{};
  ctx->have_bokeh = false;
}

LENTIL_API int lentil_hip_destroy(lentil_hip_ctx *ctx) {
  if (!ctx) return LENTIL_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_visits(ctx);
  free_bokeh(ctx);
  (void)hipFree(ctx->d_lens);
  (void)hipFree(ctx->d_terms);
  (void)hipFree(ctx->F.acc);
  (void)hipFree(ctx->d_resolved);
  (void)hipFree(ctx->d_work);
  (void)hipFree(ctx->d_state);
  (void)hipFree(ctx->d_ctr);
  (void)hipFree(ctx->d_log);
  for (auto &ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return LENTIL_OK;
}

LENTIL_API const char *lentil_hip_last_error(const lentil_hip_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_err.c_str();
}

LENTIL_API int lentil_hip_set_params(lentil_hip_ctx *ctx, const lentil_params *p) {
  CHECK_CTX(ctx);
  if (!p) return fail(ctx, LENTIL_ERR_INVALID, "params is null");
  if (p->cameraType != LENTIL_THINLENS && p->cameraType != LENTIL_POLYNOMIAL_OPTICS)
    return fail(ctx, LENTIL_ERR_INVALID, "cameraType must be ThinLens or PolynomialOptics");
  if (p->xres == 0 || p->yres == 0) return fail(ctx, LENTIL_ERR_INVALID, "xres/yres must be non-zero");
  if (p->abb_chromatic > 0.0f)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED,
                "abb_chromatic > 0 is not implemented on the GPU (per-channel traces / global xor128 state)");
  if (p->cameraType == LENTIL_THINLENS && p->abb_coma != 0.0f)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "thin-lens abb_coma != 0 is not implemented on the GPU");
  if (p->samples_override < 0 || p->samples_override > (1 << 24))
    return fail(ctx, LENTIL_ERR_INVALID, "samples_override out of range");
  if (ctx->have_frame && (p->xres != ctx->P.xres || p->yres != ctx->P.yres))
    return fail(ctx, LENTIL_ERR_INVALID, "xres/yres changed after alloc_frame");
  ctx->P = *p;
  ctx->have_params = true;
  // lambda powers depend on the params' wavelength
  if (ctx->have_lens) {
    const double lam = (double)p->lambda_bw;
    for (int e = 0; e <= kMaxExp; ++e) ctx->hlens.lambda_pow[e] = host_ipow(lam, e);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lens, &ctx->hlens, sizeof(DevLens), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return LENTIL_OK;
}

static bool pack_terms(const lentil_lens_table *t, const lentil_poly &p, int derive_var, std::vector<DevTerm> &out,
                       uint16_t &first, uint16_t &count) {
  first = (uint16_t)out.size();
  uint32_t c = 0;
  for (uint32_t i = 0; i < p.count; ++i) {
    const lentil_term &s = t->terms[p.first + i];
    uint8_t e[5] = {s.e[0], s.e[1], s.e[2], s.e[3], s.e[4]};
    double coef = s.c;
    if (derive_var >= 0) {
      if (e[derive_var] == 0) continue;
      coef = s.c * (double)e[derive_var];          // derivative: c*e, exponent-1
      e[derive_var] = (uint8_t)(e[derive_var] - 1);
    }
    for (int v = 0; v < 5; ++v) if (e[v] > kMaxExp) return false;
    DevTerm d;
    d.c = coef;
    d.e = (uint32_t)e[0] | ((uint32_t)e[1] << 4) | ((uint32_t)e[2] << 8) | ((uint32_t)e[3] << 12) | ((uint32_t)e[4] << 16);
    d.pad = 0;
    out.push_back(d);
    ++c;
  }
  count = (uint16_t)c;
  return true;
}

LENTIL_API int lentil_hip_set_lens(lentil_hip_ctx *ctx, const lentil_lens_table *t) {
  CHECK_CTX(ctx);
  if (!t || !t->terms) return fail(ctx, LENTIL_ERR_INVALID, "lens table is null");
  for (int i = 0; i < 5; ++i)
    if ((uint64_t)t->out[i].first + t->out[i].count > t->n_terms) return fail(ctx, LENTIL_ERR_INVALID, "lens poly out of range");
  for (int i = 0; i < 4; ++i)
    if ((uint64_t)t->ap[i].first + t->ap[i].count > t->n_terms) return fail(ctx, LENTIL_ERR_INVALID, "lens poly out of range");
  std::vector<DevTerm> terms;
  DevLens h{};
  bool ok = true;
  for (int i = 0; i < 5; ++i) ok &= pack_terms(t, t->out[i], -1, terms, h.first[P_OUT_X + i], h.count[P_OUT_X + i]);
  for (int i = 0; i < 4; ++i) ok &= pack_terms(t, t->ap[i], -1, terms, h.first[P_AP_X + i], h.count[P_AP_X + i]);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      ok &= pack_terms(t, t->ap[i], 2 + j, terms, h.first[P_DAP_00 + i * 2 + j], h.count[P_DAP_00 + i * 2 + j]);
      ok &= pack_terms(t, t->out[2 + i], j, terms, h.first[P_DOUT_00 + i * 2 + j], h.count[P_DOUT_00 + i * 2 + j]);
    }
  if (!ok) return fail(ctx, LENTIL_ERR_UNSUPPORTED, "lens table exponent > 15");
  if (terms.size() > (size_t)kMaxTerms)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "lens table has too many terms for the LDS staging area");
  h.outer_pupil_radius = t->lens_outer_pupil_radius;
  h.inner_pupil_radius = t->lens_inner_pupil_radius;
  h.length = t->lens_length;
  h.back_focal_length = t->lens_back_focal_length;
  h.outer_pupil_curvature_radius = t->lens_outer_pupil_curvature_radius;
  h.outer_pupil_geometry = t->lens_outer_pupil_geometry;
  h.n_terms = (uint32_t)terms.size();
  {  // FNV-1a over the base table (tools/gen_lens_code.py: table_hash) -> compiled-in specialisation
    unsigned long long hs = 0xCBF29CE484222325ull;
    auto mix = [&](const void *p, size_t n) {
      const unsigned char *b = (const unsigned char *)p;
      for (size_t i = 0; i < n; ++i) hs = (hs ^ b[i]) * 0x100000001B3ull;
    };
    auto poly = [&](const lentil_poly &pl) {
      for (uint32_t i = 0; i < pl.count; ++i) {
        const lentil_term &tt = t->terms[pl.first + i];
        mix(&tt.c, 8);
        mix(tt.e, 5);
      }
      const unsigned char sep = 0xFF;
      mix(&sep, 1);
    };
    for (int i = 0; i < 5; ++i) poly(t->out[i]);
    for (int i = 0; i < 4; ++i) poly(t->ap[i]);
    ctx->lens_hash = hs;
  }
  const double lam = ctx->have_params ? (double)ctx->P.lambda_bw : (double)0.55f;
  for (int e = 0; e <= kMaxExp; ++e) h.lambda_pow[e] = host_ipow(lam, e);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->d_lens) HIP_TRY(ctx, hipMalloc(&ctx->d_lens, sizeof(DevLens)));
  if (!ctx->d_terms) HIP_TRY(ctx, hipMalloc(&ctx->d_terms, sizeof(DevTerm) * kMaxTerms));
  ctx->hlens = h;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lens, &ctx->hlens, sizeof(DevLens), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_terms, terms.data(), sizeof(DevTerm) * terms.size(), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_lens = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_set_lens_mode(lentil_hip_ctx *ctx, int mode) {
  CHECK_CTX(ctx);
  if (mode != 0 && mode != 1) return fail(ctx, LENTIL_ERR_INVALID, "lens mode must be 0 (auto) or 1 (tables)");
  ctx->use_generated = (mode == 0);
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_lens_is_compiled(lentil_hip_ctx *ctx) {
  if (!ctx || !ctx->have_lens) return 0;
#define LENTIL_HASH_MATCH(NAME) if (ctx->lens_hash == gen::Lens_##NAME::kTableHash) return 1;
  LENTIL_GENERATED_LENSES(LENTIL_HASH_MATCH)
#undef LENTIL_HASH_MATCH
  return 0;
}

LENTIL_API int lentil_hip_set_bokeh(lentil_hip_ctx *ctx, const lentil_bokeh_table *b) {
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_bokeh(ctx);
  if (!b) return LENTIL_OK;
  if (b->x <= 0 || b->y <= 0 || b->x != b->y || !b->cdfRow || !b->rowIndices || !b->cdfColumn || !b->columnIndices)
    return fail(ctx, LENTIL_ERR_INVALID, "bokeh table must be square with all four arrays");
  const size_t ny = (size_t)b->y, nn = (size_t)b->x * b->y;
  float *cr = nullptr, *cc = nullptr;
  int32_t *ri = nullptr, *ci = nullptr;
  HIP_TRY(ctx, hipMalloc(&cr, ny * 4));
  HIP_TRY(ctx, hipMalloc(&ri, ny * 4));
  HIP_TRY(ctx, hipMalloc(&cc, nn * 4));
  HIP_TRY(ctx, hipMalloc(&ci, nn * 4));
  HIP_TRY(ctx, hipMemcpy(cr, b->cdfRow, ny * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ri, b->rowIndices, ny * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(cc, b->cdfColumn, nn * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ci, b->columnIndices, nn * 4, hipMemcpyHostToDevice));
  ctx->bokeh.x = b->x; ctx->bokeh.y = b->y;
  ctx->bokeh.cdfRow = cr; ctx->bokeh.rowIndices = ri; ctx->bokeh.cdfColumn = cc; ctx->bokeh.columnIndices = ci;
  ctx->have_bokeh = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_alloc_frame(lentil_hip_ctx *ctx, uint32_t n_aovs, const uint8_t *kind) {
  CHECK_CTX(ctx);
  if (!ctx->have_params) return fail(ctx, LENTIL_ERR_INVALID, "set_params must precede alloc_frame");
  if (n_aovs < 1 || n_aovs > LENTIL_MAX_AOVS) return fail(ctx, LENTIL_ERR_INVALID, "n_aovs out of range");
  for (uint32_t i = 0; i < n_aovs; ++i) {
    const uint8_t k = kind ? kind[i] : LENTIL_FILTER_GAUSSIAN;
    if (k != LENTIL_FILTER_GAUSSIAN)
      return fail(ctx, LENTIL_ERR_UNSUPPORTED, "only gaussian-original AOVs are redistributed on the GPU so far");
    ctx->kind[i] = k;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->F.acc);
  (void)hipFree(ctx->d_resolved);
  ctx->F = FrameDev{};
  ctx->d_resolved = nullptr;
  ctx->have_frame = false;
  const uint64_t np = (uint64_t)ctx->P.xres * ctx->P.yres;
  const uint64_t nfl = np * 4 * n_aovs + np;
  HIP_TRY(ctx, hipMalloc(&ctx->F.acc, nfl * sizeof(float)));
  HIP_TRY(ctx, hipMalloc(&ctx->d_resolved, np * 4 * n_aovs * sizeof(float)));
  ctx->F.weight = ctx->F.acc + np * 4 * n_aovs;
  ctx->F.n_aovs = n_aovs;
  ctx->F.np = np;
  ctx->have_frame = true;
  HIP_TRY(ctx, hipMemsetAsync(ctx->F.acc, 0, nfl * sizeof(float), ctx->stream));
  return LENTIL_OK;
}

static int check_visits(lentil_hip_ctx *ctx, const lentil_visits *v) {
  if (!v) return fail(ctx, LENTIL_ERR_INVALID, "visits is null");
  if (v->n > 0xFFFFFFFFull) return fail(ctx, LENTIL_ERR_UNSUPPORTED, "more than 2^32 visits per call");
  if (v->n_extra > LENTIL_MAX_AOVS - 1) return fail(ctx, LENTIL_ERR_INVALID, "too many extra AOV columns");
  if (v->n && (!v->rgba || !v->pos_z || !v->raydir_time || !v->volume_ignore || !v->transmission))
    return fail(ctx, LENTIL_ERR_INVALID, "a visit column is null");
  for (uint32_t k = 0; k < v->n_extra; ++k)
    if (v->n && !v->extra[k]) return fail(ctx, LENTIL_ERR_INVALID, "an extra AOV column is null");
  if (v->visits_per_pixel == 0 && v->n && !v->pixel)
    return fail(ctx, LENTIL_ERR_INVALID, "visits_per_pixel == 0 needs the per-visit pixel array");
  if (v->visits_per_pixel && (v->pixels_per_row == 0 || v->pixel_row_stride == 0))
    return fail(ctx, LENTIL_ERR_INVALID, "pixels_per_row / pixel_row_stride must be non-zero");
  return LENTIL_OK;
}

static void to_dev(VisitsDev &d, const lentil_visits *v) {
  d.n = v->n;
  d.visits_per_pixel = v->visits_per_pixel;
  d.pixels_per_row = v->pixels_per_row;
  d.pixel_x0 = v->pixel_x0;
  d.pixel_y0 = v->pixel_y0;
  d.pixel_row_stride = v->pixel_row_stride;
  d.n_extra = v->n_extra;
  d.rgba = (const float4 *)v->rgba;
  d.pos_z = (const float4 *)v->pos_z;
  d.raydir_time = (const float4 *)v->raydir_time;
  d.volume_ignore = (const float4 *)v->volume_ignore;
  d.transmission = (const float4 *)v->transmission;
  for (int k = 0; k < LENTIL_MAX_AOVS - 1; ++k) d.extra[k] = (const float4 *)v->extra[k];
  d.pixel = v->pixel;
  d.inv_density = v->inv_density;
}

static int ensure_worklist(lentil_hip_ctx *ctx, uint64_t n) {
  if (n <= ctx->work_cap && ctx->d_work) return LENTIL_OK;
  (void)hipFree(ctx->d_work);
  (void)hipFree(ctx->d_state);
  ctx->d_work = nullptr;
  ctx->d_state = nullptr;
  ctx->work_cap = 0;
  const uint64_t cap = n < 1024 ? 1024 : n;
  HIP_TRY(ctx, hipMalloc(&ctx->d_work, cap * sizeof(uint2)));
  HIP_TRY(ctx, hipMalloc(&ctx->d_state, cap * sizeof(ItemState)));
  ctx->work_cap = cap;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_bind_visits(lentil_hip_ctx *ctx, const lentil_visits *v) {
  CHECK_CTX(ctx);
  int rc = check_visits(ctx, v);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_visits(ctx);
  to_dev(ctx->V, v);
  rc = ensure_worklist(ctx, v->n);
  if (rc) return rc;
  ctx->have_visits = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_upload_visits(lentil_hip_ctx *ctx, const lentil_visits *v) {
  CHECK_CTX(ctx);
  int rc = check_visits(ctx, v);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_visits(ctx);
  lentil_visits d = *v;
  auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
    *dst = nullptr;
    if (!src || !bytes) return LENTIL_OK;
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    ctx->owned_visit_mem.push_back(p);
    HIP_TRY(ctx, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = p;
    return LENTIL_OK;
  };
  const size_t col = (size_t)v->n * 16;
  if ((rc = up(v->rgba, col, (const void **)&d.rgba))) return rc;
  if ((rc = up(v->pos_z, col, (const void **)&d.pos_z))) return rc;
  if ((rc = up(v->raydir_time, col, (const void **)&d.raydir_time))) return rc;
  if ((rc = up(v->volume_ignore, col, (const void **)&d.volume_ignore))) return rc;
  if ((rc = up(v->transmission, col, (const void **)&d.transmission))) return rc;
  for (uint32_t k = 0; k < v->n_extra; ++k)
    if ((rc = up(v->extra[k], col, (const void **)&d.extra[k]))) return rc;
  if ((rc = up(v->pixel, (size_t)v->n * 4, (const void **)&d.pixel))) return rc;
  if ((rc = up(v->inv_density, (size_t)v->n * 4, (const void **)&d.inv_density))) return rc;
  to_dev(ctx->V, &d);
  rc = ensure_worklist(ctx, v->n);
  if (rc) return rc;
  ctx->have_visits = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_clear_frame(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t nfl = ctx->F.np * 4 * ctx->F.n_aovs + ctx->F.np;
  HIP_TRY(ctx, hipMemsetAsync(ctx->F.acc, 0, nfl * sizeof(float), ctx->stream));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_redistribute(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  if (!ctx->have_params || !ctx->have_frame || !ctx->have_visits)
    return fail(ctx, LENTIL_ERR_INVALID, "redistribute needs set_params, alloc_frame and visits");
  const lentil_params &P = ctx->P;
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS && !ctx->have_lens)
    return fail(ctx, LENTIL_ERR_INVALID, "polynomial optics needs set_lens");
  if (P.bokeh_enable_image && !ctx->have_bokeh) return fail(ctx, LENTIL_ERR_INVALID, "bokeh_enable_image needs set_bokeh");
  if (ctx->V.n_extra + 1 != ctx->F.n_aovs)
    return fail(ctx, LENTIL_ERR_INVALID, "visit stream carries a different number of AOVs than the frame");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctr, 0, sizeof(DevCounters), ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  if (ctx->V.n) {
    ScanArgs sa{};
    sa.P = P;
    sa.lens_length = ctx->have_lens ? ctx->hlens.length : 0.0;
    sa.V = ctx->V;
    sa.F = ctx->F;
    sa.work = ctx->d_work;
    sa.state = ctx->d_state;
    sa.work_cap = ctx->work_cap;
    sa.ctr = ctx->d_ctr;
    if (ctx->V.visits_per_pixel) {
      const uint32_t M = ctx->V.visits_per_pixel;
      // staging: 20 B per visit per wave, 4 waves per block, keep a block under ~48 KiB
      uint32_t ppt = 64;
      while (ppt > 1 && (uint64_t)ppt * M * 20ull * 4ull > 48ull * 1024ull) ppt >>= 1;
      if ((uint64_t)ppt * M * 20ull * 4ull > 150ull * 1024ull)
        return fail(ctx, LENTIL_ERR_UNSUPPORTED, "visits_per_pixel too large for the LDS staging area");
      sa.ppt = ppt;
      sa.tv_pad = ppt * M;
      const size_t lds = (size_t)sa.tv_pad * 20 * 4;
      const uint64_t n_pixels = (ctx->V.n + M - 1) / M;
      const uint64_t n_tiles = (n_pixels + ppt - 1) / ppt;
      uint64_t blocks = (n_tiles + 3) / 4;
      const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
      if (blocks > max_blocks) blocks = max_blocks;
      if (blocks < 1) blocks = 1;
      hipLaunchKernelGGL(scan_uniform_kernel, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, sa);
    } else {
      uint64_t blocks = (ctx->V.n + 255) / 256;
      const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
      if (blocks > max_blocks) blocks = max_blocks;
      hipLaunchKernelGGL(scan_ragged_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, sa);
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  if (ctx->V.n) {
    DrawArgs da{};
    da.P = P;
    da.lens = ctx->d_lens;
    da.terms = ctx->d_terms;
    da.bokeh = ctx->bokeh;
    da.V = ctx->V;
    da.F = ctx->F;
    da.work = ctx->d_work;
    da.state = ctx->d_state;
    da.work_cap = ctx->work_cap;
    da.ctr = ctx->d_ctr;
    da.log = ctx->d_log;
    da.log_cap = ctx->log_cap;
    const unsigned blocks = (unsigned)ctx->num_cu * 4;
    if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS) {
      bool launched = false;
#define LENTIL_LAUNCH_GEN(NAME)                                                                        \
      if (!launched && ctx->use_generated && ctx->lens_hash == gen::Lens_##NAME::kTableHash) {          \
        hipLaunchKernelGGL((draw_po_kernel<GenLens<gen::Lens_##NAME>, false>), dim3(blocks), dim3(256), 0, \
                           ctx->stream, da);                                                           \
        launched = true;                                                                               \
      }
      LENTIL_GENERATED_LENSES(LENTIL_LAUNCH_GEN)
#undef LENTIL_LAUNCH_GEN
      if (!launched)
        hipLaunchKernelGGL((draw_po_kernel<LdsLens, true>), dim3(blocks), dim3(256), 0, ctx->stream, da);
      hipLaunchKernelGGL(po_item_stats_kernel, dim3((unsigned)ctx->num_cu), dim3(256), 0, ctx->stream, da);
    } else {
      hipLaunchKernelGGL(draw_thinlens_kernel, dim3(blocks), dim3(256), 0, ctx->stream, da);
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  ctx->timed_draw = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_resolve(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
  const uint64_t total = ctx->F.np * ctx->F.n_aovs;
  uint64_t blocks = (total + 255) / 256;
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->F, ctx->d_resolved);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  ctx->timed_resolve = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_sync(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_aov(lentil_hip_ctx *ctx, uint32_t aov, float *host_rgba) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame || aov >= ctx->F.n_aovs || !host_rgba) return fail(ctx, LENTIL_ERR_INVALID, "bad download_aov arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(host_rgba, ctx->d_resolved + (size_t)aov * ctx->F.np * 4, ctx->F.np * 16,
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_accum(lentil_hip_ctx *ctx, uint32_t aov, float *host_rgba, float *host_weight) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame || aov >= ctx->F.n_aovs) return fail(ctx, LENTIL_ERR_INVALID, "bad download_accum arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (host_rgba)
    HIP_TRY(ctx, hipMemcpyAsync(host_rgba, ctx->F.acc + (size_t)aov * ctx->F.np * 4, ctx->F.np * 16,
                                hipMemcpyDeviceToHost, ctx->stream));
  if (host_weight)
    HIP_TRY(ctx, hipMemcpyAsync(host_weight, ctx->F.weight, ctx->F.np * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_accum_buffer(lentil_hip_ctx *ctx, void **device_ptr, uint64_t *n_floats) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame || !device_ptr || !n_floats) return fail(ctx, LENTIL_ERR_INVALID, "bad accum_buffer arguments");
  *device_ptr = ctx->F.acc;
  *n_floats = ctx->F.np * 4 * ctx->F.n_aovs + ctx->F.np;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_stream(lentil_hip_ctx *ctx, void **hip_stream) {
  CHECK_CTX(ctx);
  if (!hip_stream) return fail(ctx, LENTIL_ERR_INVALID, "hip_stream is null");
  *hip_stream = (void *)ctx->stream;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_get_counters(lentil_hip_ctx *ctx, lentil_counters *out) {
  CHECK_CTX(ctx);
  if (!out) return fail(ctx, LENTIL_ERR_INVALID, "out is null");
  DevCounters c;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(&c, ctx->d_ctr, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  out->visits = ctx->have_visits ? ctx->V.n : 0;
  out->redistributed_visits = c.redistributed;
  out->attempted_draws = c.attempted;
  out->accepted_draws = c.accepted;
  out->worklist_overflow = c.overflow;
  out->newton_iterations = c.newton_iters;
  out->tries = c.tries;
  out->lane_rounds = c.lane_rounds;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_last_timing(lentil_hip_ctx *ctx, float ms[3]) {
  CHECK_CTX(ctx);
  if (!ms) return fail(ctx, LENTIL_ERR_INVALID, "ms is null");
  ms[0] = ms[1] = ms[2] = 0.f;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->timed_draw) {
    HIP_TRY(ctx, hipEventElapsedTime(&ms[0], ctx->ev[0], ctx->ev[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&ms[1], ctx->ev[1], ctx->ev[2]));
  }
  if (ctx->timed_resolve) HIP_TRY(ctx, hipEventElapsedTime(&ms[2], ctx->ev[3], ctx->ev[4]));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_set_draw_log(lentil_hip_ctx *ctx, uint64_t capacity) {
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->d_log);
  ctx->d_log = nullptr;
  ctx->log_cap = 0;
  if (capacity) {
    HIP_TRY(ctx, hipMalloc(&ctx->d_log, capacity * sizeof(lentil_draw_record)));
    ctx->log_cap = capacity;
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_draw_log(lentil_hip_ctx *ctx, lentil_draw_record *out, uint64_t capacity,
                                            uint64_t *n_records) {
  CHECK_CTX(ctx);
  if (!n_records) return fail(ctx, LENTIL_ERR_INVALID, "n_records is null");
  DevCounters c;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(&c, ctx->d_ctr, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *n_records = c.log_count;
  uint64_t n = c.log_count < ctx->log_cap ? c.log_count : ctx->log_cap;
  if (n > capacity) n = capacity;
  if (out && n) HIP_TRY(ctx, hipMemcpy(out, ctx->d_log, n * sizeof(lentil_draw_record), hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

// ----- single-function tests -----------------------------------------------------------
template <typename T>
static int dev_copy_in(lentil_hip_ctx *ctx, const T *src, size_t n, T **dst, std::vector<void *> &tmp) {
  *dst = nullptr;
  if (!src || !n) return LENTIL_OK;
  HIP_TRY(ctx, hipMalloc((void **)dst, n * sizeof(T)));
  tmp.push_back(*dst);
  HIP_TRY(ctx, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return LENTIL_OK;
}
template <typename T>
static int dev_alloc(lentil_hip_ctx *ctx, size_t n, T **dst, std::vector<void *> &tmp) {
  HIP_TRY(ctx, hipMalloc((void **)dst, n * sizeof(T)));
  tmp.push_back(*dst);
  return LENTIL_OK;
}
struct TmpFree {
  std::vector<void *> v;
  ~TmpFree() { for (void *p : v) (void)hipFree(p); }
};

LENTIL_API int lentil_hip_test_lt_sample_aperture(lentil_hip_ctx *ctx, uint64_t n, const double *scene,
                                                  const double *ap, double lambda, double *sensor,
                                                  double *out, double *transmittance) {
  CHECK_CTX(ctx);
  if (!ctx->have_lens) return fail(ctx, LENTIL_ERR_INVALID, "set_lens first");
  if (!n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpFree tf;
  TestArgs t{};
  t.P = ctx->P; t.lens = ctx->d_lens; t.terms = ctx->d_terms; t.bokeh = ctx->bokeh; t.n = n; t.lambda = lambda;
  double *d_scene, *d_ap, *o0, *o1, *o2;
  int rc;
  if ((rc = dev_copy_in(ctx, scene, n * 3, &d_scene, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, ap, n * 2, &d_ap, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 5, &o0, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 5, &o1, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n, &o2, tf.v))) return rc;
  t.in0 = d_scene; t.in1 = d_ap; t.o0 = o0; t.o1 = o1; t.o2 = o2;
  hipLaunchKernelGGL(test_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(sensor, o0, n * 5 * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(out, o1, n * 5 * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(transmittance, o2, n * 8, hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_test_trace_bw_po(lentil_hip_ctx *ctx, uint64_t n, const double *target,
                                           const int32_t *px, const int32_t *py, const int32_t *attempt,
                                           double *sensor_xy, int32_t *ok) {
  CHECK_CTX(ctx);
  if (!ctx->have_lens || !ctx->have_params) return fail(ctx, LENTIL_ERR_INVALID, "set_params and set_lens first");
  if (ctx->P.bokeh_enable_image && !ctx->have_bokeh) return fail(ctx, LENTIL_ERR_INVALID, "set_bokeh first");
  if (!n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpFree tf;
  TestArgs t{};
  t.P = ctx->P; t.lens = ctx->d_lens; t.terms = ctx->d_terms; t.bokeh = ctx->bokeh; t.n = n;
  double *d_t, *o0;
  int32_t *d_px, *d_py, *d_at, *oi;
  int rc;
  if ((rc = dev_copy_in(ctx, target, n * 3, &d_t, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, px, n, &d_px, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, py, n, &d_py, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, attempt, n, &d_at, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 2, &o0, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n, &oi, tf.v))) return rc;
  t.in0 = d_t; t.i0 = d_px; t.i1 = d_py; t.i2 = d_at; t.o0 = o0; t.oi = oi;
  hipLaunchKernelGGL(test_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(sensor_xy, o0, n * 2 * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(ok, oi, n * 4, hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_test_aperture_sample(lentil_hip_ctx *ctx, uint64_t n, const uint32_t *a,
                                               const uint32_t *b, double *xy) {
  CHECK_CTX(ctx);
  if (!ctx->have_params) return fail(ctx, LENTIL_ERR_INVALID, "set_params first");
  if (ctx->P.bokeh_enable_image && !ctx->have_bokeh) return fail(ctx, LENTIL_ERR_INVALID, "set_bokeh first");
  if (!n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpFree tf;
  TestArgs t{};
  t.P = ctx->P; t.lens = nullptr; t.terms = nullptr; t.bokeh = ctx->bokeh; t.n = n;
  uint32_t *d_a, *d_b;
  double *o0;
  int rc;
  if ((rc = dev_copy_in(ctx, a, n, &d_a, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, b, n, &d_b, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 2, &o0, tf.v))) return rc;
  t.u0 = d_a; t.u1 = d_b; t.o0 = o0;
  hipLaunchKernelGGL(test_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(xy, o0, n * 2 * 8, hipMemcpyDeviceToHost));
  return LENTIL_OK;
}
