// lentil_closest_replay.h -- closest-filtered AOVs where a candidate's depth is 0 or NaN (round 5).
//
// Reference, src/lentil.h:832-837 (one z-buffer for all closest AOVs; lentil_debug, :838-845, keeps its own with the same
// test):      if (std::abs(depth) <= zbuffer[px] || zbuffer[px] == 0.0) { buffer[px] = value; zbuffer[px] = std::abs(depth); }
// 0.0 doubles as "empty".  Over depths that are positive numbers the outcome is order-free -- smallest |Z| wins, equal
// depths go to the later sample -- and that is what the pass computes with atomicMin on packed keys (lentil_kernels.h,
// closest_key).  A candidate at |Z| == 0 wins and at once re-opens the pixel: the NEXT candidate replaces it whatever its
// depth.  A NaN is taken only by an open pixel and then never replaced (`x <= NaN` and `NaN == 0` are both false); elsewhere
// it is ignored.  So at a pixel that sees such a candidate the winner depends on the ORDER of the candidates, which the
// reference leaves to its bucket threads' timing and this repo defines, as everywhere, as the single-threaded one: visits
// in stream order (frame-wide visit id), a redistributed visit's draws where that visit stands.
//
// Until round 4 a pass that met such a candidate was refused.  Now it is replayed: the pass raises
// DevCounters::degenerate_depth; afterwards (lentil_hip.hip, closest_degenerate_replay)
//   1. the pixels a degenerate candidate reached are flagged -- its own pixel if the visit stayed there, the pixels of its
//      accepted draws (the pass's draw log) if it was redistributed;
//   2. every candidate at a flagged pixel -- own visits that were not redistributed, every logged draw that landed there --
//      goes into a per-pixel list (visit id, depth bits, which z-buffers it feeds);
//   3. one thread per flagged pixel derives the sequential outcome from its list WITHOUT sorting it.  With the candidates
//      c_1 .. c_n in visit order (equal ids are one candidate: all draws of a visit carry its depth and its values):
//        - the first NaN whose predecessor is a zero depth, or that has no predecessor, is taken by an open pixel and keeps
//          it for good: it wins;
//        - otherwise everything up to the last zero depth is overwritten by it; among the candidates behind it (all of them
//          if there is no zero) NaNs are ignored and the ordinary rule holds -- smallest |Z|, the later on a tie; if nothing
//          is left behind the last zero, that zero wins;
//      and writes the winner's key where the pass's atomicMin left its own, so that closest_gather_kernel /
//      debug_gather_kernel copy the right sample's values.
// Candidates at other pixels are untouched: their keys are already right.
// Not across GPUs: the exchange reduces keys with min, which has no order either; such a pass is still refused there.
#pragma once
#include "lentil_kernels.h"

struct ReplayNode { uint32_t gid, depth_bits, next, planes; };
constexpr uint32_t kReplayClosest = 1u, kReplayDebug = 2u;

struct ReplayArgs {
  lentil_params P;
  double lens_length;
  VisitsDev V;
  FrameDev F;
  uint8_t *flag;                   // [np]: a degenerate candidate reached the pixel
  const lentil_draw_record *log;
  uint64_t n_log;
  uint32_t *head;                  // [np]: first node of the pixel's list (0xFFFFFFFF: none)
  ReplayNode *nodes;               // null: count only
  uint32_t node_cap;
  unsigned int *n_nodes;
};

LD_DEV bool replay_degenerate(float depth) { return !(fabsf(depth) > 0.0f); }

LD_DEV uint64_t replay_pixel_of(const ReplayArgs &a, uint64_t v) {
  int px, py;
  visit_pixel(a.V, v, px, py);
  return (uint64_t)px + (uint64_t)py * a.P.xres;
}

// does visit v stay in its own pixel, and which z-buffers does it feed there?  (the scan kernels' conditions: not
// redistributed, a weight that is not zero; lentil_debug only through visit_feeds_debug_directly)
LD_DEV uint32_t replay_direct_planes(const ReplayArgs &a, uint64_t v) {
  const float invd = a.V.inv_density ? a.V.inv_density[v] : a.P.inverse_sample_density;
  if (invd == 0.0f) return 0u;
  const float4 pz = a.V.pos_z[v], vi = a.V.volume_ignore[v], tr = a.V.transmission[v];
  if (visit_redistributes(a.P, a.lens_length, pz, vi, tr, invd, [&]() { return a.V.raydir_time[v]; }, a.V.cam)) return 0u;
  uint32_t planes = a.F.zkey ? kReplayClosest : 0u;
  if (a.F.zkey_dbg && visit_feeds_debug_directly(a.P, pz, vi, tr, invd, [&]() { return a.V.raydir_time[v]; }, a.V.cam)) planes |= kReplayDebug;
  return planes;
}

__global__ __launch_bounds__(256) void replay_mark_visits_kernel(ReplayArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < a.V.n; v += stride) {
    if (!replay_degenerate(a.V.pos_z[v].w)) continue;
    if (replay_direct_planes(a, v)) a.flag[replay_pixel_of(a, v)] = 1u;
  }
}
__global__ __launch_bounds__(256) void replay_mark_log_kernel(ReplayArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_log; r += stride) {
    const lentil_draw_record d = a.log[r];
    if ((uint64_t)d.visit < a.V.n && (uint64_t)d.pixel < a.F.np && replay_degenerate(a.V.pos_z[d.visit].w)) a.flag[d.pixel] = 1u;
  }
}

LD_DEV void replay_push(const ReplayArgs &a, uint64_t p, uint32_t gid, float depth, uint32_t planes) {
  const uint32_t idx = atomicAdd(a.n_nodes, 1u);
  if (!a.nodes || idx >= a.node_cap) return;
  ReplayNode n;
  n.gid = gid; n.depth_bits = __float_as_uint(fabsf(depth)); n.planes = planes;
  n.next = atomicExch(a.head + p, idx);
  a.nodes[idx] = n;
}
__global__ __launch_bounds__(256) void replay_push_visits_kernel(ReplayArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < a.V.n; v += stride) {
    const uint64_t p = replay_pixel_of(a, v);
    if (p >= a.F.np || !a.flag[p]) continue;
    const uint32_t planes = replay_direct_planes(a, v);
    if (planes) replay_push(a, p, visit_gid(a.V, (uint32_t)v), a.V.pos_z[v].w, planes);
  }
}
__global__ __launch_bounds__(256) void replay_push_log_kernel(ReplayArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_log; r += stride) {
    const lentil_draw_record d = a.log[r];
    if ((uint64_t)d.visit >= a.V.n || (uint64_t)d.pixel >= a.F.np || !a.flag[d.pixel]) continue;
    // an accepted draw feeds both z-buffers (src/lentil.h:832-845; lentil_debug's value, the draw count, is not zero)
    replay_push(a, d.pixel, visit_gid(a.V, d.visit), a.V.pos_z[d.visit].w,
                (a.F.zkey ? kReplayClosest : 0u) | (a.F.zkey_dbg ? kReplayDebug : 0u));
  }
}

// the sequential outcome of one pixel's list for one z-buffer (see the head of this file); ~0: no candidate
LD_DEV unsigned long long replay_winner(const ReplayArgs &a, uint32_t first, uint32_t plane) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  const uint32_t cap = a.node_cap;
  auto is_nan = [](uint32_t bits) { return bits > 0x7F800000u; };      // (the bits of |depth|)
  // walk 1: is there anything, the last zero depth, are there NaNs
  bool any = false, any_nan = false, any_zero = false;
  uint32_t z_last = 0;
  for (uint32_t i = first; i != kNone && i < cap; i = a.nodes[i].next) {
    const ReplayNode n = a.nodes[i];
    if (!(n.planes & plane)) continue;
    any = true;
    if (is_nan(n.depth_bits)) any_nan = true;
    else if (n.depth_bits == 0u) { if (!any_zero || n.gid > z_last) z_last = n.gid; any_zero = true; }
  }
  if (!any) return ~0ull;
  // NaNs in visit order: the first one an open pixel takes keeps it
  if (any_nan) {
    bool have_bound = false;
    uint32_t bound = 0;
    while (true) {
      bool found = false;
      uint32_t g = 0;
      for (uint32_t i = first; i != kNone && i < cap; i = a.nodes[i].next) {
        const ReplayNode n = a.nodes[i];
        if (!(n.planes & plane) || !is_nan(n.depth_bits)) continue;
        if (have_bound && n.gid <= bound) continue;
        if (!found || n.gid < g) { g = n.gid; found = true; }
      }
      if (!found) break;
      bool have_pred = false;
      uint32_t pred = 0, pred_bits = 0;
      for (uint32_t i = first; i != kNone && i < cap; i = a.nodes[i].next) {
        const ReplayNode n = a.nodes[i];
        if (!(n.planes & plane) || n.gid >= g) continue;
        if (!have_pred || n.gid > pred) { pred = n.gid; pred_bits = n.depth_bits; have_pred = true; }
      }
      if (!have_pred || pred_bits == 0u) return ((unsigned long long)0x7FC00000u << 32) | (unsigned long long)(0xFFFFFFFFu - g);
      bound = g; have_bound = true;
    }
  }
  // the ordinary rule over what stands behind the last zero depth
  unsigned long long best = ~0ull;
  for (uint32_t i = first; i != kNone && i < cap; i = a.nodes[i].next) {
    const ReplayNode n = a.nodes[i];
    if (!(n.planes & plane) || is_nan(n.depth_bits)) continue;
    if (any_zero && n.gid <= z_last) continue;
    const unsigned long long key = ((unsigned long long)n.depth_bits << 32) | (unsigned long long)(0xFFFFFFFFu - n.gid);
    if (key < best) best = key;
  }
  if (best == ~0ull && any_zero) best = (unsigned long long)(0xFFFFFFFFu - z_last);      // (depth bits 0)
  return best;
}

__global__ __launch_bounds__(256) void replay_resolve_kernel(ReplayArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.F.np; p += stride) {
    if (!a.flag[p]) continue;
    const uint32_t first = a.head[p];
    if (a.F.zkey) {
      const unsigned long long k = replay_winner(a, first, kReplayClosest);
      if (k != ~0ull) a.F.zkey[p] = k;
    }
    if (a.F.zkey_dbg) {
      const unsigned long long k = replay_winner(a, first, kReplayDebug);
      if (k != ~0ull) a.F.zkey_dbg[p] = k;
    }
  }
}
