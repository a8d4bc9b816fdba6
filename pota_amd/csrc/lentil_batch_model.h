// lentil_batch_model.h -- how many backward traces an item's FIRST batch should hold (round 5).
//
// Reference semantics (src/lentil_filter.cpp:248-299, src/lentil.h:592-648): a redistributed sample ("item") keeps making
// attempts until `samples` of them have landed inside the frame, at most 5 x samples; an attempt is the first of up to
// vignetting_retries + 1 aperture draws that passes the lens.  The draw kernels compute every trace R(m) once
// (lentil_kernels.h, "solve once") in batches; an item whose first batch leaves it short needs a second round of solves, and a
// round is a chain of latencies (accept -> tasks -> Newton iterations -> stragglers -> accept, ~0.4 ms on the headline frame)
// whatever the number of items in it.  Which items fall short is a property of the lens and of the frame, known BEFORE any
// trace of the item has run: the traces that fail are the ones that land outside the frame (an item near the frame's edge,
// by as much as its circle of confusion reaches over it) or that the lens vignettes.  So:
//
//   calibration  (batch_model_kernel, once per lens / camera parameters / aperture tables): on a grid of camera-space targets
//                -- field x, field y (the tangents x / -z, y / -z over the frame's field of view), inverse depth 1 / -z --
//                the SAME kBmK aperture points (a stratified 16 x 16 set pushed through the aperture sampler the draws use:
//                disc, polygon or image) are traced through the lens: per node and point the continuous pixel position where
//                the ray meets the sensor and its SLACK -- how far inside the lens it stayed, the smallest of 1 - (r / R)^2 at
//                the outer and at the inner pupil and of its transmittance; negative: vignetted; -1: the Newton solve broke off.
//   estimate     (batch_estimate, a wave per item inside publish_kernel): the item's target lies in one cell of the grid; per
//                aperture point the landing position and the slack are the trilinear blend of the cell's eight nodes (both are
//                smooth in the target: the blend reproduces the paraxial part exactly and the distortion to a fraction of a
//                pixel, and a blended slack changes sign where the lens starts to clip the point -- blending pass / fail
//                flags instead put that edge half a cell off and overrated heavily vignetting lenses by a third).  The share
//                of the passing points that land inside the frame is the success rate of an attempt's decisive try.
//   batch        (batch_from_estimate): samples / rate, plus four standard deviations of the binomial count, plus a margin
//                for the model itself -- or the plain samples + retries + spare where every point of every node around lands
//                well inside the frame.
//
// Nothing of this can change a result: a surplus R(m) is never looked at, and an item that still falls short is served by a
// second round exactly as before (the pass then loses its bet on the lean tail, lentil_hip.hip, and widens the margin).
#pragma once
#include "lentil_device.h"

namespace lentil {

constexpr uint32_t kBmSide = 16, kBmK = kBmSide * kBmSide;      // aperture points per node
constexpr float kBmBorder = 3.0f;                                // "well inside": pixels between a landing point and the frame's edge

struct BatchModelDev {
  const float4 *land;       // [nz][ny][nx][kBmK]: pixel position (x, y; x = NaN: no position), slack (z); null: no model
  const float4 *box;        // [nz][ny][nx]: x min, y min, x max, y max of the node's points (min > max: all vignetted)
  const uint32_t *npass;    // [nz][ny][nx]: points that pass the lens (slack > 0)
  float fx0, fx_inv, fy0, fy_inv, u0, u_inv;      // node coordinate = (value - v0) * v_inv
  uint32_t nx, ny, nz;
  uint32_t margin16;        // sixteenths added to the model's own margin (grows when a pass loses its bet)
  float xres, yres;
};

// the aperture point of try (d1, d2) -- po_aperture_sample with the two uniforms handed in (src/lentil.h:596-609)
LD_DEV void po_aperture_from_uniforms(const lentil_params &P, const DevBokeh &B, const float *cdfRow, float d1, float d2,
                                      double &ax, double &ay) {
  if (!P.enable_dof) { ax = 0.0; ay = 0.0; return; }
  if (P.bokeh_aperture_blades <= 2) {
    double ux = 0.0, uy = 0.0;
    if (P.bokeh_enable_image) bokeh_sample(B, cdfRow, d2, d1, ux, uy);
    else concentric_disk_sample((double)d2, (double)d1, ux, uy);
    ax = ux * P.aperture_radius;
    ay = uy * P.aperture_radius;
  } else {
    triangular_aperture(ax, ay, (double)d2, (double)d1, P.aperture_radius, P.bokeh_aperture_blades, B.blade_sc, B.blade_count);
  }
}

struct BatchModelArgs {
  lentil_params P;
  const DevLens *lens;
  const DevTerm *terms;
  DevBokeh bokeh;
  float4 *land;
  float4 *box;
  uint32_t *npass;
  float fx0, fx_step, fy0, fy_step, u0, u_step;
  uint32_t nx, ny, nz;
};

// one block per node, one thread per aperture point; the table interpreter (any lens; the outcome of a trace does not
// depend on which evaluator ran it, and this kernel runs once per camera set-up)
__global__ __launch_bounds__(256) void batch_model_kernel(BatchModelArgs a) {
  __shared__ DevTerm s_terms[kMaxTerms];
  __shared__ DevLens s_k;
  __shared__ float s_box[4][4];
  __shared__ uint32_t s_n[4];
  {
    const uint32_t nt = a.lens->n_terms;
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) s_terms[i] = a.terms[i];
    if (threadIdx.x == 0) s_k = *a.lens;
  }
  __syncthreads();
  const LdsLens L{s_terms, &s_k};
  const uint32_t node = blockIdx.x;
  const uint32_t ix = node % a.nx, iy = (node / a.nx) % a.ny, iz = node / (a.nx * a.ny);
  const double fx = (double)(a.fx0 + a.fx_step * (float)ix), fy = (double)(a.fy0 + a.fy_step * (float)iy);
  const double d = 1.0 / (double)(a.u0 + a.u_step * (float)iz);
  // camera space (cm): (fx d, fy d, -d); the trace's target is -P_cs * 10 (src/lentil_filter.cpp:271)
  const double target[3] = {-fx * d * 10.0, -fy * d * 10.0, d * 10.0};
  const uint32_t k = threadIdx.x;
  const float d1 = ((float)(k / kBmSide) + 0.5f) / (float)kBmSide, d2 = ((float)(k % kBmSide) + 0.5f) / (float)kBmSide;
  double ax, ay;
  po_aperture_from_uniforms(a.P, a.bokeh, a.bokeh.cdfRow, d1, d2, ax, ay);
  // lt_sample_aperture (lentil_device.h) taken apart: the loop, then the tests of newton_finish and of trace_ray_bw_po
  // (src/lentil.h:633-645) as margins instead of verdicts
  NewtonState st;
  newton_init(st);
  while (newton_continue(st)) newton_iter(L, target, ax, ay, st);
  const DevLens &kk = s_k;
  float slack = -1.0f;
  bool have_pos = st.error == 0 && st.x == st.x && st.y == st.y && st.dx == st.dx && st.dy == st.dy;
  if (have_pos) {
    const double ro = (st.out[0] * st.out[0] + st.out[1] * st.out[1]) / (kk.outer_pupil_radius * kk.outer_pupil_radius);
    const double ipx = st.x + st.dx * kk.back_focal_length;
    const double ipy = st.y + st.dy * kk.back_focal_length;
    const double ri = (ipx * ipx + ipy * ipy) / (kk.inner_pupil_radius * kk.inner_pupil_radius);
    const double begin[4] = {st.x, st.y, st.dx, st.dy};
    const double T = L.transmittance(begin);
    double sl = 1.0 - ro;
    sl = fmin(sl, 1.0 - ri);
    sl = fmin(sl, T);
    slack = (sl == sl) ? (float)fmax(-1.0, fmin(1.0, sl)) : -1.0f;
  }
  // sensor -> continuous pixel position, the arithmetic of po_sensor_to_pixel without its frame test
  const double sx = st.x + st.dx * -a.P.sensor_shift;
  const double sy = st.y + st.dy * -a.P.sensor_shift;
  const double aspect = (double)a.P.xres_without_region / (double)a.P.yres_without_region;
  const double s0 = sx / (a.P.sensor_width * 0.5);
  const double s1 = sy / (a.P.sensor_width * 0.5) * aspect;
  float px = (float)((((s0 + 1.0) / 2.0) * a.P.xres_without_region) - a.P.region_min_x);
  float py = (float)((((-s1 + 1.0) / 2.0) * a.P.yres_without_region) - a.P.region_min_y);
  if (!(px == px) || !(py == py) || fabsf(px) > 1.0e7f || fabsf(py) > 1.0e7f) have_pos = false;
  if (!have_pos) slack = -1.0f;
  const bool ok = have_pos && slack > 0.0f;
  a.land[(size_t)node * kBmK + k] = have_pos ? make_float4(px, py, slack, 0.0f) : make_float4(__int_as_float(0x7FC00000), 0.0f, -1.0f, 0.0f);
  float x0 = ok ? px : 3.0e38f, y0 = ok ? py : 3.0e38f, x1 = ok ? px : -3.0e38f, y1 = ok ? py : -3.0e38f;
  uint32_t n = ok ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) {
    x0 = fminf(x0, __shfl_xor(x0, off)); y0 = fminf(y0, __shfl_xor(y0, off));
    x1 = fmaxf(x1, __shfl_xor(x1, off)); y1 = fmaxf(y1, __shfl_xor(y1, off));
    n += __shfl_xor(n, off);
  }
  const uint32_t w = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0u) { s_box[w][0] = x0; s_box[w][1] = y0; s_box[w][2] = x1; s_box[w][3] = y1; s_n[w] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t j = 1; j < 4; ++j) {
      s_box[0][0] = fminf(s_box[0][0], s_box[j][0]); s_box[0][1] = fminf(s_box[0][1], s_box[j][1]);
      s_box[0][2] = fmaxf(s_box[0][2], s_box[j][2]); s_box[0][3] = fmaxf(s_box[0][3], s_box[j][3]);
      s_n[0] += s_n[j];
    }
    a.box[node] = make_float4(s_box[0][0], s_box[0][1], s_box[0][2], s_box[0][3]);
    a.npass[node] = s_n[0];
  }
}

struct BatchEstimate {
  float q_strict;     // share of the passing weight that lands at least kBmBorder pixels inside the frame
  float q;            // ... that lands inside the frame
  float fail;         // share of the aperture points the lens vignettes
};

// All 64 lanes of a wave, one item: camera-space position cs (cm, z < 0 in front of the camera).  The result is wave-uniform.
LD_DEV BatchEstimate batch_estimate(const BatchModelDev &M, float cx, float cy, float cz, uint32_t lane) {
  BatchEstimate e;
  e.q_strict = 0.0f; e.q = 0.0f; e.fail = 1.0f;
  const float d = -cz;
  if (!(d > 0.0f)) return e;                   // behind the camera / NaN: nothing is promised, the item gets every attempt at once
  const float inv = 1.0f / d;
  // node coordinate -> cell and position inside it.  The field grid stops short of the frame's edge (the outermost nodes sit
  // at 97 % of the half field: at the very edge, and above all in the corners, the lens passes nothing and a node there knows
  // nothing), so a target in the rim beyond them is EXTRAPOLATED from the outermost cell, up to 0.35 of a cell: positions
  // move a pixel per pixel of field, holding them constant over the rim would put every landing point 30-60 pixels off.
  auto cell = [](float v, float v0, float v_inv, uint32_t n, float slack_cells, uint32_t &i, float &t) {
    float c = (v - v0) * v_inv;
    const float lo = -slack_cells, hi = (float)(n - 1u) + slack_cells;
    c = c > lo ? c : lo;                       // (NaN -> lo)
    c = c < hi ? c : hi;
    const float fl = floorf(c);
    int ii = (int)fl;
    if (ii < 0) ii = 0;
    if (ii > (int)n - 2) ii = (int)n - 2;
    i = (uint32_t)ii;
    t = c - (float)ii;
  };
  uint32_t ix, iy, iz;
  float tx, ty, tz;
  cell(cx * inv, M.fx0, M.fx_inv, M.nx, 0.35f, ix, tx);
  cell(cy * inv, M.fy0, M.fy_inv, M.ny, 0.35f, iy, ty);
  cell(inv, M.u0, M.u_inv, M.nz, 0.0f, iz, tz);
  const bool inner = tx >= 0.0f && tx <= 1.0f && ty >= 0.0f && ty <= 1.0f;       // (the hull argument below holds for blends only)
  uint32_t node[8];
  float wn[8];
  bool all_safe = true;
#pragma unroll
  for (uint32_t c = 0; c < 8; ++c) {
    const uint32_t dx = c & 1u, dy = (c >> 1) & 1u, dz = c >> 2;
    node[c] = ((iz + dz) * M.ny + (iy + dy)) * M.nx + (ix + dx);
    wn[c] = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) * (dz ? tz : 1.0f - tz);
    const float4 b = M.box[node[c]];
    all_safe = all_safe && inner && M.npass[node[c]] == kBmK && b.x >= kBmBorder && b.y >= kBmBorder &&
               b.z < M.xres - kBmBorder && b.w < M.yres - kBmBorder;
  }
  if (all_safe) {         // (a blend of positions lies inside the hull of the positions blended)
    e.q_strict = 1.0f; e.q = 1.0f; e.fail = 0.0f;
    return e;
  }
  // A point counts where the nodes that have a position for it carry most of the blend's weight; where they do not (beside
  // a region in which the Newton solves break off) nothing is known about it: it is taken to pass the lens and to miss the
  // frame -- the reading that asks for the most traces.
  float n_pass = 0.0f, n_in = 0.0f, n_strict = 0.0f;
  for (uint32_t p = lane; p < kBmK; p += 64u) {
    float w = 0.0f, wabs = 0.0f, x = 0.0f, y = 0.0f, sl = 0.0f;
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) {
      const float4 l = M.land[(size_t)node[c] * kBmK + p];
      if (l.x == l.x) { w += wn[c]; x += wn[c] * l.x; y += wn[c] * l.y; sl += wn[c] * l.z; }
      else wabs += fabsf(wn[c]);
    }
    if (wabs > 0.4f || !(w > 0.5f)) { n_pass += 1.0f; continue; }
    if (sl > 0.0f) {
      x /= w; y /= w;
      n_pass += 1.0f;
      if (x >= 0.0f && y >= 0.0f && x < M.xres && y < M.yres) n_in += 1.0f;
      if (x >= kBmBorder && y >= kBmBorder && x < M.xres - kBmBorder && y < M.yres - kBmBorder) n_strict += 1.0f;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_pass += __shfl_xor(n_pass, off); n_in += __shfl_xor(n_in, off); n_strict += __shfl_xor(n_strict, off);
  }
  e.fail = 1.0f - n_pass / (float)kBmK;
  if (n_pass > 0.0f) { e.q = n_in / n_pass; e.q_strict = n_strict / n_pass; }
  return e;
}

// R(m) the first batch of an item with `samples` draws should hold, m < samples * 5 + retries
LD_DEV uint32_t batch_from_estimate(const BatchEstimate &e, uint32_t samples, uint32_t retries, uint32_t plain, uint32_t margin16) {
  const uint32_t limit = samples * 5u + retries;
  const float S = (float)samples;
  // an attempt whose tries are all vignetted fails whatever the frame
  // (the share of vignetted tries is known to a few points in a hundred: taken 0.03 larger)
  const float f = e.fail > 0.0f ? (e.fail + 0.03f < 1.0f ? e.fail + 0.03f : 1.0f) : 0.0f;
  float allfail = 1.0f;
  for (uint32_t t = 0; t <= retries && t < 64u; ++t) allfail *= f;
  float qa;
  if (e.q_strict >= 0.9999f) {
    if (allfail < 1.0e-7f) return plain < limit ? plain : limit;
    qa = 1.0f - allfail;
  } else {
    // the model's own error: 8 % of the rate and 0.02, and what earlier passes added after falling short
    const float rel = 0.08f + (float)margin16 * (1.0f / 16.0f);
    qa = (e.q * (1.0f - (rel < 0.9f ? rel : 0.9f)) - 0.02f) * (1.0f - allfail);
  }
  if (!(qa > 0.02f)) return limit;
  const float n = S / qa + 4.0f * sqrtf(S * (1.0f - qa)) / qa + (float)retries + 8.0f;
  if (!(n < (float)limit)) return limit;
  const uint32_t c = (uint32_t)n + 1u;
  return c < plain ? plain : c;
}

}  // namespace lentil
