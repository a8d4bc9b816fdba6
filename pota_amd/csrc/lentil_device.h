// lentil_device.h -- gfx950 device functions of the redistribution path.
//
// Arithmetic mirrors the reference operation by operation (float where the reference is
// float, double where it is double; build with -ffp-contract=off) so that pixel indices
// come out bit-identical.  Reference citations are relative to the upstream tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lentil_hip.h"

#define LD_DEV __device__ __forceinline__

namespace lentil {

// Arnold SDK constants (ai_constants.h)
constexpr float kAiPi = 3.14159265358979323846f;
constexpr float kAiPiOver2 = 1.57079632679489661923f;
constexpr float kAiEpsilon = 1.0e-4f;
constexpr float kAiInfinite = 1.0e30f;
constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------
// Device-side lens table.  21 polynomials: out[5], ap[4], d ap_{x,y}/d{dx,dy}, d out_{dx,dy}/d{x,y}, d ap_{x,y}/d{x,y}.
// A term packs its five exponents in 4 bits each (x | y<<4 | dx<<8 | dy<<12 | lambda<<16).
// ---------------------------------------------------------------------------------------
enum PolyId {
  P_OUT_X = 0, P_OUT_Y, P_OUT_DX, P_OUT_DY, P_OUT_T,
  P_AP_X, P_AP_Y, P_AP_DX, P_AP_DY,
  P_DAP_00, P_DAP_01, P_DAP_10, P_DAP_11,      // d ap_x/d dx, d ap_x/d dy, d ap_y/d dx, d ap_y/d dy
  P_DOUT_00, P_DOUT_01, P_DOUT_10, P_DOUT_11,  // d out_dx/d x, d out_dx/d y, d out_dy/d x, d out_dy/d y
  P_DAPPOS_00, P_DAPPOS_01, P_DAPPOS_10, P_DAPPOS_11,   // d ap_x/d x, d ap_x/d y, d ap_y/d x, d ap_y/d y (lens_pt_sample_aperture only)
  P_COUNT
};

struct DevTerm {
  double c;
  uint32_t e;
  uint32_t pad;
};

constexpr int kMaxTerms = 1536;   // 24 KiB of LDS
constexpr int kMaxExp = 15;

struct DevLens {
  double outer_pupil_radius, inner_pupil_radius, length, back_focal_length;
  double outer_pupil_curvature_radius;
  int32_t outer_pupil_geometry;
  uint32_t n_terms;
  uint16_t first[P_COUNT];
  uint16_t count[P_COUNT];
  double lambda_pow[kMaxExp + 1];   // lens_ipow(lambda, e), computed on the host with the same recursion
};

struct DevBokeh {
  int32_t x, y;
  const float *cdfRow;
  const int32_t *rowIndices;
  const float *cdfColumn;
  const int32_t *columnIndices;
  // Polygonal apertures (bokeh_aperture_blades): sin and cos of the blades' corner angles, [2 k] = sin, [2 k + 1] = cos of
  // (double)(2.0f * AI_PI / blades * k), k = 0 .. blades, computed by the HOST's libm -- the reference calls std::sin /
  // std::cos on exactly these few arguments (src/lentil.h:964-982), and the device's own sin / cos differ from glibc's
  // in the last bit now and then.  Null: more blades than the table holds, device functions.
  const double *blade_sc;
  int32_t blade_count;
};

// ---------------------------------------------------------------------------------------
// a6 -- tea<8> / rng, src/global.h:32-57
// ---------------------------------------------------------------------------------------
LD_DEV uint32_t tea8(uint32_t v0, uint32_t v1) {
  uint32_t s0 = 0;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4u);
    v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761Eu);
  }
  return v0;
}
LD_DEV float lcg(uint32_t &s) {
  s = s * 1664525u + 1013904223u;
  return float(s & 0x00FFFFFFu) / float(0x01000000u);
}

// ---------------------------------------------------------------------------------------
// a9 -- fast trig + disk samplers, src/lens.h:17-37,309-333,477-514
// ---------------------------------------------------------------------------------------
LD_DEV float fast_sin(float x) {
  x = fmodf(x + kAiPi, kAiPi * 2) - kAiPi;
  const float B = 4.0f / kAiPi;
  const float C = -4.0f / (kAiPi * kAiPi);
  float y = B * x + C * x * fabsf(x);
  const float P = 0.225f;
  return P * (y * fabsf(y) - y) + y;
}
LD_DEV float fast_cos(float x) {
  x = (float)((double)x + (double)kAiPi * 0.5);
  x = fmodf(x + kAiPi, kAiPi * 2) - kAiPi;
  const float B = 4.0f / kAiPi;
  const float C = -4.0f / (kAiPi * kAiPi);
  float y = B * x + C * x * fabsf(x);
  const float P = 0.225f;
  return P * (y * fabsf(y) - y) + y;
}

LD_DEV void concentric_disk_sample(double ox, double oy, double &ux, double &uy) {
  double phi, r;
  const double a = 2.0 * ox - 1.0;
  const double b = 2.0 * oy - 1.0;
  if ((a * a) > (b * b)) {
    r = a;
    phi = (0.78539816339) * (b / a);
  } else {
    r = b;
    phi = (kPi / 2.0) - (0.78539816339) * (a / b);
  }
  ux = r * (double)fast_cos((float)phi);
  uy = r * (double)fast_sin((float)phi);
}

LD_DEV float ai_bias(float a, float b) {
  return (a > 0) ? ((b > 0) ? powf(a, logf(b) * -1.442695041f) : 0.0f) : 0.0f;
}
LD_DEV float lerpf(float perc, float a, float b) { return a + perc * (b - a); }

LD_DEV void concentricDiskSample_tl(float ox, float oy, double &lx, double &ly, float bias, float squarelerp) {
  if (ox == 0.0f && oy == 0.0f) { lx = 0.0; ly = 0.0; return; }
  float phi, r;
  const float a = (float)(2.0 * (double)ox - 1.0);
  const float b = (float)(2.0 * (double)oy - 1.0);
  if ((a * a) > (b * b)) {
    r = a;
    phi = (float)(0.78539816339 * (double)(b / a));
  } else {
    r = b;
    phi = (float)((double)kAiPiOver2 - ((0.78539816339) * (double)(a / b)));
  }
  if ((double)bias != 0.5) r = ai_bias(fabsf(r), bias) * (float)(r < 0 ? -1 : 1);
  const float cos_phi = fast_cos(phi);
  const float sin_phi = fast_sin(phi);
  lx = (double)(r * cos_phi);
  ly = (double)(r * sin_phi);
  if (squarelerp > 0.0f) {
    lx = (double)lerpf(squarelerp, (float)lx, a);
    ly = (double)lerpf(squarelerp, (float)ly, b);
  }
}

// a10 -- Camera::lens_sample_triangular_aperture, src/lentil.h:964-982
LD_DEV void triangular_aperture(double &x, double &y, double r1, double r2, double radius, int blades,
                                const double *blade_sc = nullptr, int blade_count = 0) {
  const int tri = (int)(r1 * blades);
  r1 = r1 * blades - tri;
  const double a = sqrt(r1);
  const double b = (1.0 - r2) * a;
  const double c = r2 * a;
  double s1, c1, s2, c2;
  if (blade_sc && blade_count == blades && tri >= 0 && tri < blades) {
    s1 = blade_sc[2 * (tri + 1)]; c1 = blade_sc[2 * (tri + 1) + 1];
    s2 = blade_sc[2 * tri]; c2 = blade_sc[2 * tri + 1];
  } else {
    const double ph1 = (double)(2.0f * kAiPi / (float)blades * (float)(tri + 1));
    const double ph2 = (double)(2.0f * kAiPi / (float)blades * (float)tri);
    s1 = sin(ph1); c1 = cos(ph1); s2 = sin(ph2); c2 = cos(ph2);
  }
  x = radius * (b * c1 + c * c2);
  y = radius * (b * s1 + c * s2);
}

// a7 -- imageData::bokehSample, src/imagebokeh.h:341-412.  upper_bound = first element > value.
template <typename T>
LD_DEV int upper_bound_f(const T *a, int n, float v) {
  int lo = 0, len = n;
  while (len > 0) {
    const int half = len >> 1;
    if (!(v < a[lo + half])) { lo += half + 1; len -= half + 1; } else { len = half; }
  }
  return lo;
}
LD_DEV void bokeh_sample(const DevBokeh &B, const float *cdfRow, float uRow, float uCol, double &lx, double &ly) {
  const int x = B.x, y = B.y;
  int r = upper_bound_f(cdfRow, y, uRow);
  if (r >= y) r = y - 1;
  const int actualPixelRow = B.rowIndices[r];
  const int recalulatedPixelRow = actualPixelRow - ((x - 1) / 2);
  const int startPixel = actualPixelRow * x;
  int c = upper_bound_f(B.cdfColumn + startPixel, x, uCol);
  c = (c >= x) ? startPixel + x - 1 : startPixel + c;
  const int actualPixelColumn = B.columnIndices[c];
  const int relativePixelColumn = actualPixelColumn - startPixel;
  const int recalulatedPixelColumn = relativePixelColumn - ((y - 1) / 2);
  const float flippedRow = (float)recalulatedPixelColumn;
  const float flippedColumn = (float)recalulatedPixelRow * -1.0f;
  lx = (double)(flippedRow / (float)x) * 2.0;
  ly = (double)(flippedColumn / (float)y) * 2.0;
}

// ---------------------------------------------------------------------------------------
// a14 -- pupil transforms, src/lens.h:99-221
// ---------------------------------------------------------------------------------------
LD_DEV void normalise3(double &x, double &y, double &z) {
  const double ilen = 1.0 / sqrt(x * x + y * y + z * z);
  x *= ilen; y *= ilen; z *= ilen;
}

LD_DEV void sphereToCs(double ipx, double ipy, double idx, double idy, double pos[3], double dir[3],
                       double center, double R) {
  const double n0 = ipx / R, n1 = ipy / R;
  const double n2 = sqrt(fmax(0.0, R * R - ipx * ipx - ipy * ipy)) / fabs(R);
  const double t2 = sqrt(fmax(0.0, 1.0 - idx * idx - idy * idy));
  double ex0 = n2, ex1 = 0.0, ex2 = -n0;
  normalise3(ex0, ex1, ex2);
  const double ey0 = n1 * ex2 - n2 * ex1;
  const double ey1 = n2 * ex0 - n0 * ex2;
  const double ey2 = n0 * ex1 - n1 * ex0;
  dir[0] = idx * ex0 + idy * ey0 + t2 * n0;
  dir[1] = idx * ex1 + idy * ey1 + t2 * n1;
  dir[2] = idx * ex2 + idy * ey2 + t2 * n2;
  pos[0] = ipx;
  pos[1] = ipy;
  pos[2] = n2 * R + center;
}

LD_DEV void csToSphere(const double pos[3], double vx, double vy, double vz, double &odx, double &ody,
                       double center, double R) {
  const double n0 = pos[0] / R, n1 = pos[1] / R, n2 = fabs((pos[2] - center) / R);
  normalise3(vx, vy, vz);
  double ex0 = n2, ex1 = 0.0, ex2 = -n0;
  normalise3(ex0, ex1, ex2);
  const double ey0 = n1 * ex2 - n2 * ex1;
  const double ey1 = n2 * ex0 - n0 * ex2;
  const double ey2 = n0 * ex1 - n1 * ex0;
  odx = vx * ex0 + vy * ex1 + vz * ex2;
  ody = vx * ey0 + vy * ey1 + vz * ey2;
}

LD_DEV void cylinderToCs(double ipx, double ipy, double idx, double idy, double pos[3], double dir[3],
                         double center, double R, bool cyl_y) {
  double n0 = 0.0, n1 = 0.0, n2;
  if (cyl_y) { n0 = ipx / R; n2 = sqrt(fmax(0.0, R * R - ipx * ipx)) / fabs(R); }
  else { n1 = ipy / R; n2 = sqrt(fmax(0.0, R * R - ipy * ipy)) / fabs(R); }
  const double t2 = sqrt(fmax(0.0, 1.0 - idx * idx - idy * idy));
  double ex0 = n2, ex1 = 0.0, ex2 = -n0;
  normalise3(ex0, ex1, ex2);
  double ey0 = n1 * ex2 - n2 * ex1;
  double ey1 = n2 * ex0 - n0 * ex2;
  double ey2 = n0 * ex1 - n1 * ex0;
  normalise3(ey0, ey1, ey2);
  dir[0] = idx * ex0 + idy * ey0 + t2 * n0;
  dir[1] = idx * ex1 + idy * ey1 + t2 * n1;
  dir[2] = idx * ex2 + idy * ey2 + t2 * n2;
  pos[0] = ipx;
  pos[1] = ipy;
  pos[2] = n2 * R + center;
}

LD_DEV void csToCylinder(const double pos[3], double vx, double vy, double vz, double &odx, double &ody,
                         double center, double R, bool cyl_y) {
  double n0 = 0.0, n1 = 0.0;
  if (cyl_y) n0 = pos[0] / R; else n1 = pos[1] / R;
  const double n2 = fabs((pos[2] - center) / R);
  normalise3(vx, vy, vz);
  const double ex0 = n2, ex1 = 0.0, ex2 = -n0;     // not normalised, src/lens.h:171
  double ey0 = n1 * ex2 - n2 * ex1;
  double ey1 = n2 * ex0 - n0 * ex2;
  double ey2 = n0 * ex1 - n1 * ex0;
  normalise3(ey0, ey1, ey2);
  odx = vx * ex0 + vy * ex1 + vz * ex2;
  ody = vx * ey0 + vy * ey1 + vz * ey2;
}

// lens_ipow, src/lens.h:226-233, unrolled: e is wave-uniform (it comes from the lens table)
LD_DEV double ipow_u(double x, uint32_t e) {
  uint32_t odd = 0, n = 0;
  while (e > 2) { odd = (odd << 1) | (e & 1u); ++n; e >>= 1; }
  double p = (e == 2) ? x * x : x;
  for (uint32_t i = 0; i < n; ++i) {
    p = (odd & 1u) ? (x * p) * p : p * p;
    odd >>= 1;
  }
  return p;
}

// ---------------------------------------------------------------------------------------
// Table-driven polynomial evaluation: terms in LDS, exponents read through readfirstlane so
// that all control flow is scalar.  sum = t0 + t1 + ... left to right, term = c*f1*f2*...
// ---------------------------------------------------------------------------------------
struct LdsLens {
  const DevTerm *terms;   // LDS
  const DevLens *k;       // LDS copy of the header

  LD_DEV double eval(int pid, const double v[4]) const {
    const uint32_t first = k->first[pid], count = k->count[pid];
    double sum = 0.0;
    for (uint32_t i = 0; i < count; ++i) {
      const DevTerm t = terms[first + i];
      const uint32_t e = __builtin_amdgcn_readfirstlane(t.e);
      double term = t.c;
#pragma unroll
      for (int var = 0; var < 4; ++var) {
        const uint32_t ev = (e >> (4 * var)) & 15u;
        if (ev == 1) term = term * v[var];
        else if (ev > 1) term = term * ipow_u(v[var], ev);
      }
      const uint32_t el = (e >> 16) & 15u;
      if (el) term = term * k->lambda_pow[el];
      sum = (i == 0) ? term : sum + term;
    }
    return sum;
  }

  // everything one Newton iteration of lt_sample_aperture needs, all at the same point
  LD_DEV void eval_bw(const double v[4], double pred_ap[2], double Jap[4], double out[4], double Jout[4]) const {
    pred_ap[0] = eval(P_AP_X, v);
    pred_ap[1] = eval(P_AP_Y, v);
    Jap[0] = eval(P_DAP_00, v); Jap[1] = eval(P_DAP_01, v);
    Jap[2] = eval(P_DAP_10, v); Jap[3] = eval(P_DAP_11, v);
    out[0] = eval(P_OUT_X, v); out[1] = eval(P_OUT_Y, v);
    out[2] = eval(P_OUT_DX, v); out[3] = eval(P_OUT_DY, v);
    Jout[0] = eval(P_DOUT_00, v); Jout[1] = eval(P_DOUT_01, v);
    Jout[2] = eval(P_DOUT_10, v); Jout[3] = eval(P_DOUT_11, v);
  }
  LD_DEV double transmittance(const double v[4]) const { return eval(P_OUT_T, v); }
  LD_DEV const DevLens &consts() const { return *k; }
};

// ---------------------------------------------------------------------------------------
// a12 -- lens_lt_sample_aperture (src/lentil.h:1296-1313) with the generated Newton body as
// evidenced by tests/aperture_sampling_debug/writout.txt: all polynomials of an iteration
// at the iteration's begin state; aperture step undamped; outer-pupil step damped 0.72;
// error bits reset while k < 10; tolerance 1e-8 on both squared errors.
// Split into init / continue? / one iteration / finish so that the draw kernel can advance
// many independent solves in lock step (one iteration per lane per round).
// ---------------------------------------------------------------------------------------
struct NewtonState {
  double x, y, dx, dy;
  double sqr_err, sqr_ap_err;
  double out[4];
  int k, error;
};

LD_DEV void newton_init(NewtonState &s) {
  s.x = 0; s.y = 0; s.dx = 0; s.dy = 0;
  s.sqr_err = 1e30; s.sqr_ap_err = 1e30;
  s.out[0] = s.out[1] = s.out[2] = s.out[3] = 0.0;
  s.k = 0; s.error = 0;
}

LD_DEV bool newton_continue(const NewtonState &s) {
  const double eps = 1e-8;
  return s.k < 100 && (s.sqr_err > eps || s.sqr_ap_err > eps) && s.error == 0;
}

template <class Lens>
LD_DEV void newton_iter(const Lens &L, const double scene[3], double ap_x, double ap_y, NewtonState &s) {
  const DevLens &k = L.consts();
  const double R = k.outer_pupil_curvature_radius;
  const double prev_sqr_err = s.sqr_err, prev_sqr_ap_err = s.sqr_ap_err;
  const double begin[4] = {s.x, s.y, s.dx, s.dy};
  double pred_ap[2], Jap[4], Jout[4];
  L.eval_bw(begin, pred_ap, Jap, s.out, Jout);
  const double d0 = ap_x - pred_ap[0], d1 = ap_y - pred_ap[1];
  s.sqr_ap_err = d0 * d0 + d1 * d1;
  {
    const double invdet = 1.0 / (Jap[0] * Jap[3] - Jap[1] * Jap[2]);
    const double i00 = Jap[3] * invdet, i11 = Jap[0] * invdet;
    const double i01 = -Jap[1] * invdet, i10 = -Jap[2] * invdet;
    s.dx += i00 * d0; s.dy += i10 * d0;
    s.dx += i01 * d1; s.dy += i11 * d1;
  }
  double pos[3], dir[3];
  double on_dx, on_dy;
  // the view vector is normalised once here and once more inside csTo*, like the generated code
  if (k.outer_pupil_geometry == LENTIL_GEOM_SPHERICAL) {
    sphereToCs(s.out[0], s.out[1], s.out[2], s.out[3], pos, dir, -R, R);
    double vx = scene[0] - pos[0], vy = scene[1] - pos[1], vz = scene[2] - pos[2];
    normalise3(vx, vy, vz);
    csToSphere(pos, vx, vy, vz, on_dx, on_dy, -R, R);
  } else {
    const bool cy = k.outer_pupil_geometry == LENTIL_GEOM_CYL_Y;
    cylinderToCs(s.out[0], s.out[1], s.out[2], s.out[3], pos, dir, -R, R, cy);
    double vx = scene[0] - pos[0], vy = scene[1] - pos[1], vz = scene[2] - pos[2];
    normalise3(vx, vy, vz);
    csToCylinder(pos, vx, vy, vz, on_dx, on_dy, -R, R, cy);
  }
  const double e0 = on_dx - s.out[2], e1 = on_dy - s.out[3];
  s.sqr_err = e0 * e0 + e1 * e1;
  {
    const double invdet = 1.0 / (Jout[0] * Jout[3] - Jout[1] * Jout[2]);
    const double i00 = Jout[3] * invdet, i11 = Jout[0] * invdet;
    const double i01 = -Jout[1] * invdet, i10 = -Jout[2] * invdet;
    s.x += 0.72 * i00 * e0; s.y += 0.72 * i10 * e0;
    s.x += 0.72 * i01 * e1; s.y += 0.72 * i11 * e1;
  }
  int error = s.error;
  if (s.sqr_err > prev_sqr_err) error |= 1;
  if (s.sqr_ap_err > prev_sqr_ap_err) error |= 2;
  if (s.out[0] != s.out[0]) error |= 4;
  if (s.out[2] * s.out[2] + s.out[3] * s.out[3] > 1.0) error |= 8;
  if (s.k < 10) error = 0;
  s.error = error;
  s.k += 1;
}

// ---------------------------------------------------------------------------------------
// a13 -- Camera::lens_evaluate (src/lentil.h:1257-1266) and Camera::lens_pt_sample_aperture (:1272-1291; body as the
// polynomial-optics generator emits it: at most 5 Newton steps on the direction at the sensor, tolerance 1e-4, the
// Jacobian with the dist * d/dpos chain term of the shifted start point), and on top of them
// Camera::camera_get_y0_intersection_distance (:1361-1386) -- what the focus search evaluates per candidate.
// Table interpreter only (a camera update runs them 20 001 times, not per draw).
// ---------------------------------------------------------------------------------------
LD_DEV double lens_evaluate(const LdsLens &L, const double in[4], double out[4]) {
  out[0] = L.eval(P_OUT_X, in); out[1] = L.eval(P_OUT_Y, in);
  out[2] = L.eval(P_OUT_DX, in); out[3] = L.eval(P_OUT_DY, in);
  return fmax(0.0, L.eval(P_OUT_T, in));
}

// in: x, y, dx, dy at the sensor (dx, dy solved for); ap_x, ap_y: the aperture point to hit; out_dx/out_dy: the
// direction predicted at the aperture
LD_DEV void lens_pt_sample_aperture(const LdsLens &L, double in[4], double ap_x, double ap_y, double dist,
                                    double &out_dx, double &out_dy) {
  double dx = in[2], dy = in[3];
  double pred_dx = 0.0, pred_dy = 0.0;
  double sqr_err = 3.4028234663852886e38;
  for (int k = 0; k < 5 && sqr_err > 1e-4; ++k) {
    const double begin[4] = {in[0] + dist * dx, in[1] + dist * dy, dx, dy};
    const double pred_x = L.eval(P_AP_X, begin), pred_y = L.eval(P_AP_Y, begin);
    pred_dx = L.eval(P_AP_DX, begin);
    pred_dy = L.eval(P_AP_DY, begin);
    const double j00 = L.eval(P_DAP_00, begin) + dist * L.eval(P_DAPPOS_00, begin);
    const double j01 = L.eval(P_DAP_01, begin) + dist * L.eval(P_DAPPOS_01, begin);
    const double j10 = L.eval(P_DAP_10, begin) + dist * L.eval(P_DAPPOS_10, begin);
    const double j11 = L.eval(P_DAP_11, begin) + dist * L.eval(P_DAPPOS_11, begin);
    const double invdet = 1.0 / (j00 * j11 - j01 * j10);
    const double i00 = j11 * invdet, i11 = j00 * invdet, i01 = -j01 * invdet, i10 = -j10 * invdet;
    const double r0 = ap_x - pred_x, r1 = ap_y - pred_y;
    dx += i00 * r0; dy += i10 * r0;
    dx += i01 * r1; dy += i11 * r1;
    sqr_err = r0 * r0 + r1 * r1;
  }
  out_dx = pred_dx; out_dy = pred_dy;
  in[2] = dx; in[3] = dy;
}

// line_plane_intersection with the y = 0 plane (src/lens.h:412-419), z component: the direction is normalised
// (Eigen: v / norm), scaled by (0 - origin.y), then divided by its y component
LD_DEV double y0_plane_z(const double pos[3], const double dir[3]) {
  const double n = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  const double dy = dir[1] / n, dz = dir[2] / n;
  return pos[2] + (dz * (0.0 - pos[1])) / dy;
}

LD_DEV double camera_get_y0_intersection_distance(const LdsLens &L, double sensor_shift, double aperture_housing_radius,
                                                  double sensor_out[4], double out[4], double &transmittance) {
  const DevLens &k = L.consts();
  double sensor[4] = {0.0, 0.0, 0.0, 0.0};
  double adx, ady;
  lens_pt_sample_aperture(L, sensor, 0.0, aperture_housing_radius * 0.25, sensor_shift, adx, ady);
  sensor[0] += sensor[2] * sensor_shift;
  sensor[1] += sensor[3] * sensor_shift;
  transmittance = lens_evaluate(L, sensor, out);
  for (int c = 0; c < 4; ++c) sensor_out[c] = sensor[c];
  const double R = k.outer_pupil_curvature_radius;
  double pos[3], dir[3];
  if (k.outer_pupil_geometry == LENTIL_GEOM_SPHERICAL) sphereToCs(out[0], out[1], out[2], out[3], pos, dir, -R, R);
  else cylinderToCs(out[0], out[1], out[2], out[3], pos, dir, -R, R, k.outer_pupil_geometry == LENTIL_GEOM_CYL_Y);
  return y0_plane_z(pos, dir);
}

// after the loop: outer-pupil radius test + transmittance (0 on error); returns max(0, T)
template <class Lens>
LD_DEV double newton_finish(const Lens &L, const NewtonState &s, double &out4) {
  const DevLens &k = L.consts();
  int error = s.error;
  if (s.out[0] * s.out[0] + s.out[1] * s.out[1] > k.outer_pupil_radius * k.outer_pupil_radius) error |= 16;
  if (error == 0) {
    const double begin[4] = {s.x, s.y, s.dx, s.dy};
    out4 = L.transmittance(begin);
  } else {
    out4 = 0.0;
  }
  return fmax(0.0, out4);
}

// Returns the transmittance (>= 0).  sensor = (x, y, dx, dy).
template <class Lens>
LD_DEV double lt_sample_aperture(const Lens &L, const double scene[3], double ap_x, double ap_y,
                                 double sensor[4], double out[5], int *iters = nullptr) {
  NewtonState s;
  newton_init(s);
  while (newton_continue(s)) newton_iter(L, scene, ap_x, ap_y, s);
  const double T = newton_finish(L, s, out[4]);
  out[0] = s.out[0]; out[1] = s.out[1]; out[2] = s.out[2]; out[3] = s.out[3];
  if (iters) *iters = s.k;
  sensor[0] = s.x; sensor[1] = s.y; sensor[2] = s.dx; sensor[3] = s.dy;
  return T;
}

// Compiled-in lens (tools/gen_lens_code.py): coefficients are literals, nothing read from LDS.
template <class Gen>
struct GenLens {
  const DevLens *k;       // LDS copy of the header (constants + lambda powers)
  LD_DEV void eval_bw(const double v[4], double pred_ap[2], double Jap[4], double out[4], double Jout[4]) const {
    Gen::eval_bw(v, k->lambda_pow, pred_ap, Jap, out, Jout);
  }
  LD_DEV double transmittance(const double v[4]) const { return Gen::transmittance(v, k->lambda_pow); }
  LD_DEV const DevLens &consts() const { return *k; }
};

// ---------------------------------------------------------------------------------------
// Aperture draw of trace_ray_bw_po, src/lentil.h:596-609 (GCC argument order: see oracle)
// ---------------------------------------------------------------------------------------
LD_DEV void po_aperture_sample(const lentil_params &P, const DevBokeh &B, const float *cdfRow, uint32_t a,
                               uint32_t b, double &ax, double &ay) {
  if (!P.enable_dof) { ax = 0.0; ay = 0.0; return; }
  uint32_t seed = tea8(a, b);
  if (P.bokeh_aperture_blades <= 2) {
    double ux = 0.0, uy = 0.0;
    if (P.bokeh_enable_image) {
      lcg(seed); lcg(seed);
      const float d3 = lcg(seed), d4 = lcg(seed);
      bokeh_sample(B, cdfRow, d4, d3, ux, uy);
    } else {
      const float d1 = lcg(seed), d2 = lcg(seed);
      concentric_disk_sample((double)d2, (double)d1, ux, uy);
    }
    ax = ux * P.aperture_radius;
    ay = uy * P.aperture_radius;
  } else {
    const float d1 = lcg(seed), d2 = lcg(seed);
    triangular_aperture(ax, ay, (double)d2, (double)d1, P.aperture_radius, P.bokeh_aperture_blades, B.blade_sc, B.blade_count);
  }
}

// a11 -- Camera::trace_ray_bw_po, src/lentil.h:573-661 (AiTraceProbe == false)
template <class Lens>
LD_DEV bool trace_ray_bw_po(const lentil_params &P, const Lens &L, const DevBokeh &B, const float *cdfRow,
                            const double target[3], int px, int py, int total_samples_taken,
                            double &sx, double &sy) {
  const DevLens &k = L.consts();
  int tries = 0;
  bool ray_succes = false;
  double sensor[4] = {0, 0, 0, 0};
  double out[5] = {0, 0, 0, 0, (double)P.lambda_bw};
  double ax = 0.0, ay = 0.0;
  while (ray_succes == false && tries <= P.vignetting_retries) {
    po_aperture_sample(P, B, cdfRow, (uint32_t)(px * py + px), (uint32_t)(total_samples_taken + tries), ax, ay);
    const float transmittance = (float)lt_sample_aperture(L, target, ax, ay, sensor, out);
    if (transmittance <= 0) { ++tries; continue; }
    const double ipx = sensor[0] + sensor[2] * k.back_focal_length;
    const double ipy = sensor[1] + sensor[3] * k.back_focal_length;
    if (ipx * ipx + ipy * ipy > k.inner_pupil_radius * k.inner_pupil_radius) { ++tries; continue; }
    ray_succes = true;
  }
  if (!ray_succes) return false;
  sx = sensor[0] + sensor[2] * -P.sensor_shift;
  sy = sensor[1] + sensor[3] * -P.sensor_shift;
  return true;
}

// a15 -- sensor -> pixel, src/lentil_filter.cpp:276-290 (fp64, explicit NaN test)
LD_DEV bool po_sensor_to_pixel(const lentil_params &P, double sx, double sy, uint32_t &pixelnumber) {
  const double xres = (double)P.xres, yres = (double)P.yres;
  const double aspect = (double)P.xres_without_region / (double)P.yres_without_region;
  const double s0 = sx / (P.sensor_width * 0.5);
  const double s1 = sy / (P.sensor_width * 0.5) * aspect;
  const double pixel0 = (((s0 + 1.0) / 2.0) * P.xres_without_region) - P.region_min_x;
  const double pixel1 = (((-s1 + 1.0) / 2.0) * P.yres_without_region) - P.region_min_y;
  if ((pixel0 >= xres) || (pixel0 < 0) || (pixel1 >= yres) || (pixel1 < 0) || (pixel0 != pixel0) ||
      (pixel1 != pixel1))
    return false;
  const int ix = (int)floor(pixel0), iy = (int)floor(pixel1);
  pixelnumber = (uint32_t)(ix + (iy * (int)P.xres));
  return true;
}

// ---------------------------------------------------------------------------------------
// a3/a4/a5 -- scalar helpers of the visit prologue
// ---------------------------------------------------------------------------------------
LD_DEV float get_coc_thinlens(const lentil_params &P, float z_cs) {   // src/lentil.h:674-692
  float _focus_distance = (float)P.focus_distance;
  float _aperture_radius = (float)P.aperture_radius;
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS) _focus_distance = (float)((double)_focus_distance / 10.0);
  else _aperture_radius = (float)((double)_aperture_radius * 10.0);
  const float f = P.focal_length;
  const float image_dist_samplepos = (-f * z_cs) / (-f + z_cs);
  const float image_dist_focusdist = (-f * -_focus_distance) / (-f + -_focus_distance);
  return fabsf((_aperture_radius * (image_dist_samplepos - image_dist_focusdist)) / image_dist_samplepos);
}

LD_DEV float additional_luminance_soft_trans(const lentil_params &P, float lum) {   // src/lentil.h:1128-1138
  const double lo = P.bidir_add_energy_minimum_luminance;
  const float tr = P.bidir_add_energy_transition;
  if ((double)lum > lo && (double)lum < lo + (double)tr) {
    const float perc = (float)(((double)lum - lo) / (double)tr);
    return P.bidir_add_energy * perc;
  } else if ((double)lum > lo + (double)tr) {
    return P.bidir_add_energy;
  }
  return 0.0f;
}

// draw count, src/lentil_filter.cpp:177-202.  pow(x, 0.5) on a float-valued double is taken as the
// correctly rounded sqrt, and pow(x, 2) of a float-valued double is exact.
LD_DEV int draw_count(const lentil_params &P, float lum, float coc, float inv_density) {
  const float luminance_mult = (float)fmax(0.0, sqrt((double)fminf(lum, 20.0f)) * (double)P.bidir_sample_mult);
  const float cy = coc * (float)P.yres;
  const float coc_squared_pixels =
      (float)((((double)cy * (double)cy) * ((double)luminance_mult * (double)luminance_mult)) * 0.00001);
  // `int samples = std::ceil(float)`: x86 cvttss2si yields INT_MIN for out-of-range / NaN inputs
  const float cf = ceilf(coc_squared_pixels * inv_density);
  const int si = (cf >= 2147483648.0f || cf < -2147483648.0f || cf != cf) ? (int)0x80000000 : (int)cf;
  float s = (float)si;
  if (s < 4.0f) s = 4.0f;
  if (s > 2000.0f) s = 2000.0f;
  int samples = (int)s;
  if (P.samples_override > 0) samples = P.samples_override;
  return samples;
}

LD_DEV float v3len(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }
LD_DEV void v3norm(float &x, float &y, float &z) {    // AiV3Normalize
  float t = v3len(x, y, z);
  if (t != 0) t = 1 / t;
  x *= t; y *= t; z *= t;
}

// ---------------------------------------------------------------------------------------
// Visit prologue shared by the scan kernel and the draw kernels:
// src/lentil_filter.cpp:105-165,173-202,240.
// ---------------------------------------------------------------------------------------
// A moving camera (lentil_hip_set_camera_motion): n >= 2 world-to-camera matrices, row-vector convention like
// lentil_params::world_to_camera, at equidistant shutter-relative times 0 ... 1; a visit's matrix is the component-wise
// interpolation ((b - a) * f) + a of the two keys around its lentil_time (src/lentil_filter.cpp:141-144).  n < 2: the
// static matrix of the parameters.
// The keys lie at t0 + k / (n - 1) * (t1 - t0) of the camera's shutter [t0, t1] (lentil_hip_set_camera_shutter; 0 ... 1 unless
// set): a visit's lentil_time is Arnold's absolute sample time, (time - t0) * inv_dt its place between the first and the last key.
struct CamMotion {
  const float *keys;
  uint32_t n;
  float t0, inv_dt;
};
// column `c` of the visit's matrix (what one camera-space coordinate needs), rows 0..3
LD_DEV void cam_column(const lentil_params &P, const CamMotion &cm, float time, int c, float col[4]) {
  if (cm.n < 2u) { for (int r = 0; r < 4; ++r) col[r] = P.world_to_camera[r][c]; return; }
  float t = (time - cm.t0) * cm.inv_dt;
  t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
  const float sc = t * (float)(cm.n - 1u);
  uint32_t i0 = (uint32_t)sc;
  if (i0 > cm.n - 2u) i0 = cm.n - 2u;
  const float f = sc - (float)i0;
  const float *ka = cm.keys + (size_t)i0 * 16u, *kb = ka + 16;
  for (int r = 0; r < 4; ++r) col[r] = ((kb[r * 4 + c] - ka[r * 4 + c]) * f) + ka[r * 4 + c];
}

struct VisitInfo {
  bool redistribute;
  int samples;
  float cs[3];          // camera-space position after unit scaling
  float add_energy;     // fitted_bidir_add_energy
  float depth;
};

// The redistribute decision alone, for the scan kernel (all but a few visits in 10^5 stay in their own pixel
// and need nothing else): the same tests on the same fp32/fp64 values as visit_prologue below, minus
// everything that only feeds the draw count.  `raydir` is only read for visits at infinite depth.
template <class RaydirLoad>
LD_DEV bool visit_redistributes(const lentil_params &P, double lens_length, float4 pos_z, float4 volume_ignore,
                                float4 transmission, float inv_density, RaydirLoad load_raydir, const CamMotion &cm = CamMotion{nullptr, 0u, 0.0f, 1.0f}) {
  bool redistribute = true;
  if (P.adaptive_sampling) { if (inv_density > 0.2f) redistribute = false; }
  float wx = pos_z.x, wy = pos_z.y, wz = pos_z.z;
  const float depth = pos_z.w;
  const bool small = fabsf(wx) < kAiEpsilon && fabsf(wy) < kAiEpsilon && fabsf(wz) < kAiEpsilon;
  const bool far = ((double)depth == (double)kAiInfinite) || small;
  if (far && P.enable_skydome) {
    const float4 raydir_time = load_raydir();
    if (raydir_time.x == 0.0f && raydir_time.y == 0.0f && raydir_time.z == 0.0f) redistribute = false;
    else { wx = raydir_time.x * 100000000.0f; wy = raydir_time.y * 100000000.0f; wz = raydir_time.z * 100000000.0f; }
  }
  if (far && !P.enable_skydome) redistribute = false;
  if (fmaxf(fmaxf(volume_ignore.x, volume_ignore.y), volume_ignore.z) > 0.0f) redistribute = false;
  float mz[4];
  cam_column(P, cm, cm.n >= 2u ? load_raydir().w : 0.0f, 2, mz);      // (a moving camera reads the visit's lentil_time)
  float cz = wx * mz[0] + wy * mz[1] + wz * mz[2] + mz[3];
  float scale = 1.0f;
  if (P.unitModel == LENTIL_UNIT_MM) scale = 0.1f;
  else if (P.unitModel == LENTIL_UNIT_DM) scale = 10.0f;
  else if (P.unitModel == LENTIL_UNIT_M) scale = 100.0f;
  cz *= scale;
  if (!P.enable_bidir_transmission && fmaxf(fmaxf(transmission.x, transmission.y), transmission.z) > 0.0f) redistribute = false;
  if (volume_ignore.w > 0.0f) redistribute = false;
  if (get_coc_thinlens(P, cz) < 0.4f) redistribute = false;
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS)
    if ((double)fabsf(cz) < (lens_length * 0.1)) redistribute = false;
  return redistribute;
}

// camera-space position of a visit (the part of visit_prologue below that the item header needs; same arithmetic)
template <class RaydirLoad>
LD_DEV void visit_camera_space(const lentil_params &P, float4 pos_z, RaydirLoad load_raydir, float cs[3],
                               const CamMotion &cm = CamMotion{nullptr, 0u, 0.0f, 1.0f}) {
  float wx = pos_z.x, wy = pos_z.y, wz = pos_z.z;
  const float depth = pos_z.w;
  const bool small = fabsf(wx) < kAiEpsilon && fabsf(wy) < kAiEpsilon && fabsf(wz) < kAiEpsilon;
  const bool far = ((double)depth == (double)kAiInfinite) || small;
  if (far && P.enable_skydome) {
    const float4 raydir_time = load_raydir();
    if (!(raydir_time.x == 0.0f && raydir_time.y == 0.0f && raydir_time.z == 0.0f)) {
      wx = raydir_time.x * 100000000.0f; wy = raydir_time.y * 100000000.0f; wz = raydir_time.z * 100000000.0f;
    }
  }
  float m0[4], m1[4], m2[4];
  const float time = cm.n >= 2u ? load_raydir().w : 0.0f;
  cam_column(P, cm, time, 0, m0); cam_column(P, cm, time, 1, m1); cam_column(P, cm, time, 2, m2);
  float cx = wx * m0[0] + wy * m0[1] + wz * m0[2] + m0[3];
  float cy = wx * m1[0] + wy * m1[1] + wz * m1[2] + m1[3];
  float cz = wx * m2[0] + wy * m2[1] + wz * m2[2] + m2[3];
  float scale = 1.0f;
  if (P.unitModel == LENTIL_UNIT_MM) scale = 0.1f;
  else if (P.unitModel == LENTIL_UNIT_DM) scale = 10.0f;
  else if (P.unitModel == LENTIL_UNIT_M) scale = 100.0f;
  cs[0] = cx * scale; cs[1] = cy * scale; cs[2] = cz * scale;
}

LD_DEV VisitInfo visit_prologue(const lentil_params &P, double lens_length, float4 rgba, float4 pos_z,
                                float4 raydir_time, float4 volume_ignore, float4 transmission,
                                float inv_density, const CamMotion &cm = CamMotion{nullptr, 0u, 0.0f, 1.0f}) {
  VisitInfo I;
  bool redistribute = true;
  if (P.adaptive_sampling) { if (inv_density > 0.2f) redistribute = false; }
  float wx = pos_z.x, wy = pos_z.y, wz = pos_z.z;
  const float depth = pos_z.w;
  const bool small = fabsf(wx) < kAiEpsilon && fabsf(wy) < kAiEpsilon && fabsf(wz) < kAiEpsilon;
  const bool far = ((double)depth == (double)kAiInfinite) || small;
  if (far && P.enable_skydome) {
    if (raydir_time.x == 0.0f && raydir_time.y == 0.0f && raydir_time.z == 0.0f) redistribute = false;
    else { wx = raydir_time.x * 100000000.0f; wy = raydir_time.y * 100000000.0f; wz = raydir_time.z * 100000000.0f; }
  }
  if (far && !P.enable_skydome) redistribute = false;
  if (fmaxf(fmaxf(volume_ignore.x, volume_ignore.y), volume_ignore.z) > 0.0f) redistribute = false;

  float m0[4], m1[4], m2[4];
  cam_column(P, cm, raydir_time.w, 0, m0); cam_column(P, cm, raydir_time.w, 1, m1); cam_column(P, cm, raydir_time.w, 2, m2);
  float cx = wx * m0[0] + wy * m0[1] + wz * m0[2] + m0[3];
  float cy = wx * m1[0] + wy * m1[1] + wz * m1[2] + m1[3];
  float cz = wx * m2[0] + wy * m2[1] + wz * m2[2] + m2[3];
  float scale = 1.0f;
  if (P.unitModel == LENTIL_UNIT_MM) scale = 0.1f;
  else if (P.unitModel == LENTIL_UNIT_DM) scale = 10.0f;
  else if (P.unitModel == LENTIL_UNIT_M) scale = 100.0f;
  cx *= scale; cy *= scale; cz *= scale;

  float r = rgba.x, g = rgba.y, b = rgba.z;
  const bool transmitted = P.enable_bidir_transmission
                               ? false
                               : (fmaxf(fmaxf(transmission.x, transmission.y), transmission.z) > 0.0f);
  if (transmitted) { r -= transmission.x; g -= transmission.y; b -= transmission.z; redistribute = false; }
  const float lum = (float)((double)(r + g + b) / 3.0);
  if (volume_ignore.w > 0.0f) redistribute = false;

  float add_energy = 0.0f;
  if (P.bidir_add_energy > 0.0f) add_energy = additional_luminance_soft_trans(P, lum);
  const float coc = get_coc_thinlens(P, cz);
  if (coc < 0.4f) redistribute = false;
  I.samples = draw_count(P, lum, coc, inv_density);
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS)
    if ((double)fabsf(cz) < (lens_length * 0.1)) redistribute = false;
  I.redistribute = redistribute;
  I.cs[0] = cx; I.cs[1] = cy; I.cs[2] = cz;
  I.add_energy = add_energy;
  I.depth = depth;
  return I;
}

// Thin-lens coma, src/lens.h:563-582.  The rotation is Eigen's AngleAxisd -> Matrix3d, Matrix3d::inverse()
// and Matrix3d * Vector3d, restated from Eigen's published sources (3.3/3.4; Eigen is not in the reference
// tree, it is included from a sibling checkout, src/lens.h:5-6): Geometry/AngleAxis.h toRotationMatrix,
// LU/InverseImpl.h compute_inverse<.,.,3>, coefficient-based product with the reduction a0 + (a1 + a2).
LD_DEV float abb_coma_multipliers(float sensor_width, float focal_length, float dcx, float dcy, float dcz,
                                  double ux, double uy) {
  float mx = (float)(1.0 * ((double)sensor_width * 0.5)), my = mx, mz = -focal_length;
  v3norm(mx, my, mz);
  const float maximal_projection = mx * 0.0f + my * 0.0f + mz * -1.0f;
  const float current_projection = dcx * 0.0f + dcy * 0.0f + dcz * -1.0f;
  const float projection_perc =
      (float)((((double)(current_projection - maximal_projection) / (1.0 - (double)maximal_projection)) - 0.5) * 2.0);
  const float dist_from_sensor_center = (float)(1.0 - (double)projection_perc);
  const float dist_from_aperture = (float)sqrt(ux * ux + uy * uy);
  return dist_from_sensor_center * dist_from_aperture;
}

LD_DEV double cofactor3(const double m[3][3], int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
}

// rotates ray_to_perturb about normalize(cross(dir_from_lens, -z)) by -/+ abb_coma * 2.3456 degrees
LD_DEV void abb_coma_perturb(float lx, float ly, float lz, float rx, float ry, float rz, float abb_coma,
                             bool reverse, float &ox, float &oy, float &oz) {
  float ax = ly * -1.0f - lz * 0.0f, ay = lz * 0.0f - lx * -1.0f, az = lx * 0.0f - ly * 0.0f;   // AiV3Cross(l, (0,0,-1))
  v3norm(ax, ay, az);
  const double axis[3] = {(double)ax, (double)ay, (double)az};
  const double angle = ((double)abb_coma * 2.3456 * (double)kAiPi) / 180.0;
  const double sn = sin(angle), c = cos(angle);
  const double sin_axis[3] = {sn * axis[0], sn * axis[1], sn * axis[2]};
  const double cos1_axis[3] = {(1.0 - c) * axis[0], (1.0 - c) * axis[1], (1.0 - c) * axis[2]};
  double rot[3][3], inv[3][3];
  double tmp = cos1_axis[0] * axis[1];
  rot[0][1] = tmp - sin_axis[2]; rot[1][0] = tmp + sin_axis[2];
  tmp = cos1_axis[0] * axis[2];
  rot[0][2] = tmp + sin_axis[1]; rot[2][0] = tmp - sin_axis[1];
  tmp = cos1_axis[1] * axis[2];
  rot[1][2] = tmp - sin_axis[0]; rot[2][1] = tmp + sin_axis[0];
#pragma unroll
  for (int i = 0; i < 3; ++i) rot[i][i] = cos1_axis[i] * axis[i] + c;
  const double(*m)[3] = rot;
  if (reverse) {
    const double c0[3] = {cofactor3(rot, 0, 0), cofactor3(rot, 1, 0), cofactor3(rot, 2, 0)};
    const double det = c0[0] * rot[0][0] + (c0[1] * rot[1][0] + c0[2] * rot[2][0]);
    const double invdet = 1.0 / det;
    inv[1][0] = cofactor3(rot, 0, 1) * invdet;
    inv[1][1] = cofactor3(rot, 1, 1) * invdet;
    inv[2][0] = cofactor3(rot, 0, 2) * invdet;
    inv[1][2] = cofactor3(rot, 2, 1) * invdet;
    inv[2][1] = cofactor3(rot, 1, 2) * invdet;
    inv[2][2] = cofactor3(rot, 2, 2) * invdet;
    inv[0][0] = c0[0] * invdet; inv[0][1] = c0[1] * invdet; inv[0][2] = c0[2] * invdet;
    m = inv;
  }
  const double r[3] = {(double)rx, (double)ry, (double)rz};
  ox = (float)(m[0][0] * r[0] + (m[0][1] * r[1] + m[0][2] * r[2]));
  oy = (float)(m[1][0] * r[0] + (m[1][1] * r[1] + m[1][2] * r[2]));
  oz = (float)(m[2][0] * r[0] + (m[2][1] * r[1] + m[2][2] * r[2]));
}

// a16 -- one thin-lens draw, src/lentil_filter.cpp:311-434 (abb_chromatic == 0)
// The draw in two parts: the ray from the lens point towards the sample's image (everything up to the optical
// vignetting test, src/lentil_filter.cpp:304-386), and its projection onto the sensor for a given image distance of
// the focus plane (:389-434) -- chromatic aberration (:393-406) moves only that distance, per colour channel.
struct TlRay { float lx, ly, dlx, dly, dlz; };

LD_DEV bool thinlens_ray(const lentil_params &P, const DevBokeh &B, const float *cdfRow, const float cs[3],
                         int px, int py, uint32_t total_samples_taken, TlRay &ray) {
  uint32_t seed = tea8((uint32_t)(px * py + px), total_samples_taken);
  const float f = P.focal_length;
  const float image_dist_samplepos = (-f * cs[2]) / (-f + cs[2]);
  double ux = 0.0, uy = 0.0;
  if (P.bokeh_enable_image) {
    lcg(seed); lcg(seed);
    const float d3 = lcg(seed), d4 = lcg(seed);
    bokeh_sample(B, cdfRow, d4, d3, ux, uy);
  } else if (P.bokeh_aperture_blades < 2) {
    const float d1 = lcg(seed), d2 = lcg(seed);
    concentricDiskSample_tl(d2, d1, ux, uy, P.abb_spherical, P.circle_to_square);
  } else {
    const float d1 = lcg(seed), d2 = lcg(seed);
    triangular_aperture(ux, uy, (double)d2, (double)d1, 1.0, P.bokeh_aperture_blades, B.blade_sc, B.blade_count);
  }
  ux *= (double)P.bokeh_anamorphic;
  const float lx = (float)(ux * P.aperture_radius), ly = (float)(uy * P.aperture_radius), lz = 0.0f;
  float dcx = cs[0], dcy = cs[1], dcz = cs[2];
  v3norm(dcx, dcy, dcz);
  float ptx = dcx, pty = dcy, ptz = dcz;
  if (P.abb_coma != 0.0f) {
    // coma: the centre ray rotated about the axis orthogonal to the lens ray, :328-334 (zero coma rotates
    // by exactly the identity, so the branch only saves work)
    float qx = cs[0] - lx, qy = cs[1] - ly, qz = cs[2] - lz;
    v3norm(qx, qy, qz);
    const float mult = P.abb_coma * abb_coma_multipliers(P.sensor_width, P.focal_length, dcx, dcy, dcz, ux, uy);
    abb_coma_perturb(qx, qy, qz, dcx, dcy, dcz, mult, true, ptx, pty, ptz);
  }
  const float len = v3len(cs[0], cs[1], cs[2]);
  const float ppx = len * ptx, ppy = len * pty, ppz = len * ptz;
  dcx = ppx; dcy = ppy; dcz = ppz;
  v3norm(dcx, dcy, dcz);
  const float sii = fabsf(image_dist_samplepos / dcz);
  const float ipx = dcx * sii, ipy = dcy * sii, ipz = dcz * sii;
  float dlx = ipx - lx, dly = ipy - ly, dlz = ipz - lz;
  v3norm(dlx, dly, dlz);
  if (P.optical_vignetting_distance > 0.0f) {
    float qx = ppx - lx, qy = ppy - ly, qz = ppz - lz;
    v3norm(qx, qy, qz);
    const float squarebias = (float)(1.0 + log(1.0 + (double)P.circle_to_square) * exp((double)P.circle_to_square * 3.0));
    const float inter = fabsf(P.optical_vignetting_distance / qz);
    const float ovx = qx * inter - lx, ovy = qy * inter - ly;
    const float power = (float)(1.0 + (double)squarebias);
    const float radius = (float)P.aperture_radius * P.optical_vignetting_radius;
    const float dist = powf(fabsf(ovx), power) + powf(fabsf(ovy), power);
    if (dist > powf(radius, power)) return false;
  }
  ray.lx = lx; ray.ly = ly; ray.dlx = dlx; ray.dly = dly; ray.dlz = dlz;
  return true;
}

// Camera::get_image_dist_focusdist_thinlens / _abberated (src/lentil.h:665-671): double arithmetic narrowed to float
LD_DEV float thinlens_image_dist_focus(const lentil_params &P, float shift) {
  const float f = P.focal_length;
  return (float)(((double)-f * -(P.focus_distance + (double)shift)) / ((double)-f + -(P.focus_distance + (double)shift)));
}

LD_DEV bool thinlens_project(const lentil_params &P, const TlRay &ray, float image_dist_focusdist, uint32_t &pixelnumber) {
  const float f = P.focal_length;
  const float lx = ray.lx, ly = ray.ly, lz = 0.0f, dlx = ray.dlx, dly = ray.dly, dlz = ray.dlz;
  const float fi = fabsf(image_dist_focusdist / dlz);
  const float fx = lx + dlx * fi, fy = ly + dly * fi, fz = lz + dlz * fi;
  float spx = fx / fz, spy = fy / fz;
  const float div = (float)((P.sensor_width * 0.5) / (double)-f);
  { const float inv = 1.0f / div; spx *= inv; spy *= inv; }
  if (P.abb_distortion > 0.0f) {
    const double b = (double)P.abb_distortion;
    const float l = sqrtf(spx * spx + spy * spy);
    const double bl = (double)l;
    const float x0 = (float)pow(9. * b * b * bl + sqrt(3.) * sqrt(27. * b * b * b * b * bl * bl + 4. * b * b * b), 1. / 3.);
    const float xx = (float)((double)x0 / (pow(2., 1. / 3.) * pow(3., 2. / 3.) * b) - pow(2. / 3., 1. / 3.) / (double)x0);
    spx = spx * (xx / l); spy = spy * (xx / l);
  }
  const double aspect = (double)P.xres_without_region / (double)P.yres_without_region;
  const double s0 = (double)spx, s1 = (double)spy * aspect;
  const float pixel_x = (float)((((s0 + 1.0) / 2.0) * P.xres_without_region) - P.region_min_x);
  const float pixel_y = (float)((((-s1 + 1.0) / 2.0) * P.yres_without_region) - P.region_min_y);
  const double xres = (double)P.xres, yres = (double)P.yres;
  if (((double)pixel_x >= xres) || (pixel_x < 0) || ((double)pixel_y >= yres) || (pixel_y < 0)) return false;
  const int ix = (int)floorf(pixel_x), iy = (int)floorf(pixel_y);
  pixelnumber = (uint32_t)(ix + (iy * (int)P.xres));
  return true;
}

LD_DEV bool thinlens_draw(const lentil_params &P, const DevBokeh &B, const float *cdfRow, const float cs[3],
                          int px, int py, uint32_t total_samples_taken, uint32_t &pixelnumber) {
  TlRay ray;
  if (!thinlens_ray(P, B, cdfRow, cs, px, py, total_samples_taken, ray)) return false;
  const float image_dist_focusdist =
      (float)(((double)-P.focal_length * -P.focus_distance) / ((double)-P.focal_length + -P.focus_distance));
  return thinlens_project(P, ray, image_dist_focusdist, pixelnumber);
}

// abb_chromatic > 0 (src/lentil_filter.cpp:348-353, 393-406): the focus plane's image distance of colour channel
// `channel` (-1 red, 0 green, 1 blue) -- shifted by channel (or |channel|: green/magenta) * abb_chromatic * 5 * the
// distance of the unaberrated sensor point from the centre.  All factors float, multiplied left to right.
LD_DEV float thinlens_chroma_image_dist(const lentil_params &P, const TlRay &ray, int channel) {
  const float idf = thinlens_image_dist_focus(P, 0.0f);       // shift 0: x + 0.0 changes no bit of the double sum
  const float fi = fabsf(idf / ray.dlz);
  const float ux = ray.lx + ray.dlx * fi, uy = ray.ly + ray.dly * fi, uz = 0.0f + ray.dlz * fi;
  const float sx = ux / uz, sy = uy / uz;
  const float ddx = 0.0f - sx, ddy = 0.0f - sy;
  const float dist = sqrtf(ddx * ddx + ddy * ddy);            // AiV2Dist((0, 0), sensor_position_unperturbed)
  const float abb_chromatic_lateral = 5.0f;
  const float direction_shift = P.abb_chromatic_type == 0 ? (float)(channel < 0 ? -channel : channel) : (float)channel;
  return thinlens_image_dist_focus(P, direction_shift * P.abb_chromatic * abb_chromatic_lateral * dist);
}

}  // namespace lentil
