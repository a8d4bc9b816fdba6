// lentil_host.cpp -- CPU-side setup code of the plugin mirror (liblentil_host.so).
#include "../../../include/lentil_host.h"

#include <algorithm>
#include <vector>

#define HOST_API extern "C" __attribute__((visibility("default")))

namespace {
// descending-by-value index comparator, src/imagebokeh.h:21-27
struct ByValueDesc {
  const float *v;
  bool operator()(int a, int b) const { return v[a] > v[b]; }
};
}  // namespace

HOST_API int lentil_host_bokeh_probability(const float *pix, int32_t x, int32_t y, int32_t nch, float *cdfRow,
                                           int32_t *rowIndices, float *cdfColumn, int32_t *columnIndices) {
  if (!pix || x <= 0 || y <= 0 || nch < 3 || x != y) return -1;
  const int n = x * y;
  // luminance, fp32, sequential total (src/imagebokeh.h:164-168)
  std::vector<float> lum(n), prob(n), rowSum(y), rowProb(n);
  float total = 0.0f;
  for (int i = 0, j = 0; i < n; ++i, j += nch) {
    lum[i] = pix[j] * 0.3f + pix[j + 1] * 0.59f + pix[j + 2] * 0.11f;
    total += lum[i];
  }
  const float invTotal = 1.0f / total;                       // :180
  for (int i = 0; i < n; ++i) prob[i] = lum[i] * invTotal;   // :183-186
  for (int r = 0, k = 0; r < y; ++r) {                       // :204-212
    float s = 0.0f;
    for (int c = 0; c < x; ++c, ++k) s += prob[k];
    rowSum[r] = s;
  }
  for (int r = 0; r < y; ++r) rowIndices[r] = r;
  std::sort(rowIndices, rowIndices + y, ByValueDesc{rowSum.data()});   // :238
  float run = 0.0f;
  for (int r = 0; r < y; ++r) {                              // :254-258
    cdfRow[r] = run + rowSum[rowIndices[r]];
    run = cdfRow[r];
  }
  for (int r = 0, i = 0; r < y; ++r)                         // :273-285
    for (int c = 0; c < x; ++c, ++i)
      rowProb[i] = (prob[i] != 0 && rowSum[r] != 0) ? prob[i] / rowSum[r] : 0.0f;
  for (int i = 0; i < n; ++i) columnIndices[i] = i;
  for (int i = 0; i < n; i += x) std::sort(columnIndices + i, columnIndices + i + x, ByValueDesc{rowProb.data()});  // :301-303
  for (int r = 0, i = 0; r < y; ++r) {                       // :319-328
    run = 0.0f;
    for (int c = 0; c < x; ++c, ++i) {
      cdfColumn[i] = run + rowProb[columnIndices[i]];
      run = cdfColumn[i];
    }
  }
  return 0;
}

// =======================================================================================
// Polynomial-optics lens on the host (camera setup + forward rays).
// An independent implementation of the arithmetic the kernels use (checked against the oracle in
// tests/test_host_setup.py); evaluation order as in lentil_device.h / tools/gen_lens_code.py.
// =======================================================================================
#include <cmath>
#include <cstring>

namespace {

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kPiOver2F = 1.57079632679489661923f;

struct Monomial {
  double c;
  uint8_t e[5];
};
using Polynomial = std::vector<Monomial>;

double ipow(double x, int e) {           // lens_ipow, src/lens.h:226-233
  if (e == 0) return 1.0;
  if (e == 1) return x;
  if (e == 2) return x * x;
  const double h = ipow(x, e / 2);
  return (e & 1) ? x * h * h : h * h;
}

double eval(const Polynomial &p, const double v[5]) {
  double acc = 0.0;
  for (size_t i = 0; i < p.size(); ++i) {
    double t = p[i].c;
    for (int k = 0; k < 5; ++k) {
      const int e = p[i].e[k];
      if (e == 1) t = t * v[k];
      else if (e > 1) t = t * ipow(v[k], e);
    }
    acc = i == 0 ? t : acc + t;
  }
  return acc;
}

Polynomial differentiate(const Polynomial &p, int var) {
  Polynomial d;
  for (const Monomial &m : p) {
    if (!m.e[var]) continue;
    Monomial n = m;
    n.c = m.c * (double)m.e[var];
    n.e[var] = (uint8_t)(m.e[var] - 1);
    d.push_back(n);
  }
  return d;
}

struct Vec3 {
  double x, y, z;
};
Vec3 unit(Vec3 v) {                      // raytrace_normalise: multiply by 1/len
  const double il = 1.0 / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  return {v.x * il, v.y * il, v.z * il};
}
Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Ray3 {
  Vec3 pos, dir;
};

// sphereToCs / cylinderToCs, src/lens.h:99-125,188-221
Ray3 pupil_to_camera(const double o[4], double center, double R, int geom) {
  Vec3 n{0, 0, 0};
  if (geom == LENTIL_GEOM_CYL_Y) { n.x = o[0] / R; n.z = std::sqrt(std::max(0.0, R * R - o[0] * o[0])) / std::abs(R); }
  else if (geom == LENTIL_GEOM_CYL_X) { n.y = o[1] / R; n.z = std::sqrt(std::max(0.0, R * R - o[1] * o[1])) / std::abs(R); }
  else { n.x = o[0] / R; n.y = o[1] / R; n.z = std::sqrt(std::max(0.0, R * R - o[0] * o[0] - o[1] * o[1])) / std::abs(R); }
  const double tz = std::sqrt(std::max(0.0, 1.0 - o[2] * o[2] - o[3] * o[3]));
  const Vec3 ex = unit({n.z, 0.0, -n.x});
  Vec3 ey = cross(n, ex);
  if (geom != LENTIL_GEOM_SPHERICAL) ey = unit(ey);
  Ray3 r;
  r.dir = {o[2] * ex.x + o[3] * ey.x + tz * n.x, o[2] * ex.y + o[3] * ey.y + tz * n.y, o[2] * ex.z + o[3] * ey.z + tz * n.z};
  r.pos = {o[0], o[1], n.z * R + center};
  return r;
}

// csToSphere / csToCylinder, src/lens.h:127-185: direction of `dir` in the pupil's tangent frame at pos
void camera_to_pupil_dir(Vec3 pos, Vec3 dir, double center, double R, int geom, double &odx, double &ody) {
  Vec3 n{0, 0, std::abs((pos.z - center) / R)};
  if (geom == LENTIL_GEOM_CYL_Y) n.x = pos.x / R;
  else if (geom == LENTIL_GEOM_CYL_X) n.y = pos.y / R;
  else { n.x = pos.x / R; n.y = pos.y / R; }
  const Vec3 d = unit(dir);
  Vec3 ex{n.z, 0.0, -n.x};
  if (geom == LENTIL_GEOM_SPHERICAL) ex = unit(ex);
  Vec3 ey = cross(n, ex);
  if (geom != LENTIL_GEOM_SPHERICAL) ey = unit(ey);
  odx = d.x * ex.x + d.y * ex.y + d.z * ex.z;
  ody = d.x * ey.x + d.y * ey.y + d.z * ey.z;
}

void invert2(const double J[4], double inv[4]) {   // row-major [00 01 10 11]
  const double id = 1.0 / (J[0] * J[3] - J[1] * J[2]);
  inv[0] = J[3] * id; inv[3] = J[0] * id; inv[1] = -J[1] * id; inv[2] = -J[2] * id;
}

}  // namespace

struct lentil_host_lens {
  lentil_lens_table k;
  Polynomial out[5], ap[4];
  Polynomial dap_ddir[2][2];   // d ap_{x,y} / d {dx,dy}
  Polynomial dap_dpos[2][2];   // d ap_{x,y} / d {x,y}
  Polynomial dout_dpos[2][2];  // d out_{dx,dy} / d {x,y}
};

HOST_API lentil_host_lens *lentil_host_lens_create(const lentil_lens_table *t) {
  if (!t || !t->terms) return nullptr;
  lentil_host_lens *L = new lentil_host_lens();
  L->k = *t;
  L->k.terms = nullptr;
  auto take = [&](const lentil_poly &p) {
    Polynomial q;
    for (uint32_t i = 0; i < p.count; ++i) {
      const lentil_term &s = t->terms[p.first + i];
      Monomial m;
      m.c = s.c;
      memcpy(m.e, s.e, 5);
      q.push_back(m);
    }
    return q;
  };
  for (int i = 0; i < 5; ++i) L->out[i] = take(t->out[i]);
  for (int i = 0; i < 4; ++i) L->ap[i] = take(t->ap[i]);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      L->dap_ddir[i][j] = differentiate(L->ap[i], 2 + j);
      L->dap_dpos[i][j] = differentiate(L->ap[i], j);
      L->dout_dpos[i][j] = differentiate(L->out[2 + i], j);
    }
  return L;
}
HOST_API void lentil_host_lens_destroy(lentil_host_lens *L) { delete L; }

HOST_API double lentil_host_lens_evaluate(const lentil_host_lens *L, const double in[5], double out[5]) {
  for (int i = 0; i < 4; ++i) out[i] = eval(L->out[i], in);
  return std::max(0.0, eval(L->out[4], in));
}

HOST_API void lentil_host_lens_pt_sample_aperture(const lentil_host_lens *L, double in[5], double out[5], double dist) {
  double dx = in[2], dy = in[3];
  double pdx = 0, pdy = 0;
  double err2 = 3.4028234663852886e38;
  for (int k = 0; k < 5 && err2 > 1e-4; ++k) {
    const double v[5] = {in[0] + dist * dx, in[1] + dist * dy, dx, dy, in[4]};
    const double px = eval(L->ap[0], v), py = eval(L->ap[1], v);
    pdx = eval(L->ap[2], v);
    pdy = eval(L->ap[3], v);
    double J[4], inv[4];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) J[i * 2 + j] = eval(L->dap_ddir[i][j], v) + dist * eval(L->dap_dpos[i][j], v);
    invert2(J, inv);
    const double r0 = out[0] - px, r1 = out[1] - py;
    dx += inv[0] * r0; dy += inv[2] * r0;
    dx += inv[1] * r1; dy += inv[3] * r1;
    err2 = r0 * r0 + r1 * r1;
  }
  out[2] = pdx; out[3] = pdy;
  in[2] = dx; in[3] = dy;
}

HOST_API double lentil_host_lens_lt_sample_aperture(const lentil_host_lens *L, const double scene[3], const double ap[2],
                                                    double sensor[5], double out[5], double lambda) {
  const lentil_lens_table &k = L->k;
  const double R = k.lens_outer_pupil_curvature_radius;
  const int geom = k.lens_outer_pupil_geometry;
  double x = 0, y = 0, dx = 0, dy = 0;
  double e2 = 1e30, a2 = 1e30;
  int error = 0;
  for (int it = 0; it < 100 && (e2 > 1e-8 || a2 > 1e-8) && error == 0; ++it) {
    const double pe2 = e2, pa2 = a2;
    const double v[5] = {x, y, dx, dy, lambda};
    const double d0 = ap[0] - eval(L->ap[0], v), d1 = ap[1] - eval(L->ap[1], v);
    a2 = d0 * d0 + d1 * d1;
    double J[4], inv[4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) J[i * 2 + j] = eval(L->dap_ddir[i][j], v);
    invert2(J, inv);
    dx += inv[0] * d0; dy += inv[2] * d0;
    dx += inv[1] * d1; dy += inv[3] * d1;
    for (int i = 0; i < 4; ++i) out[i] = eval(L->out[i], v);
    const Ray3 r = pupil_to_camera(out, -R, R, geom);
    const Vec3 view = unit({scene[0] - r.pos.x, scene[1] - r.pos.y, scene[2] - r.pos.z});
    double ndx, ndy;
    camera_to_pupil_dir(r.pos, view, -R, R, geom, ndx, ndy);
    const double g0 = ndx - out[2], g1 = ndy - out[3];
    e2 = g0 * g0 + g1 * g1;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) J[i * 2 + j] = eval(L->dout_dpos[i][j], v);
    invert2(J, inv);
    x += 0.72 * inv[0] * g0; y += 0.72 * inv[2] * g0;
    x += 0.72 * inv[1] * g1; y += 0.72 * inv[3] * g1;
    if (e2 > pe2) error |= 1;
    if (a2 > pa2) error |= 2;
    if (out[0] != out[0]) error |= 4;
    if (out[2] * out[2] + out[3] * out[3] > 1.0) error |= 8;
    if (it < 10) error = 0;
  }
  if (out[0] * out[0] + out[1] * out[1] > k.lens_outer_pupil_radius * k.lens_outer_pupil_radius) error |= 16;
  if (error == 0) {
    const double v[5] = {x, y, dx, dy, lambda};
    out[4] = eval(L->out[4], v);
  } else {
    out[4] = 0.0;
  }
  sensor[0] = x; sensor[1] = y; sensor[2] = dx; sensor[3] = dy; sensor[4] = lambda;
  return std::max(0.0, out[4]);
}

// line_plane_intersection (src/lens.h:412-419) with the y = 0 plane, z component
static double y0_plane_z(const Ray3 &r) {
  const double n = std::sqrt(r.dir.x * r.dir.x + r.dir.y * r.dir.y + r.dir.z * r.dir.z);
  const double dy = r.dir.y / n, dz = r.dir.z / n;
  return r.pos.z + (dz * (0.0 - r.pos.y)) / dy;
}

HOST_API double lentil_host_camera_get_y0_intersection_distance(const lentil_host_lens *L, double shift, double lambda) {
  double s[5] = {0, 0, 0, 0, lambda};
  double a[5] = {0, L->k.lens_aperture_housing_radius * 0.25, 0, 0, 0};
  double o[5] = {0, 0, 0, 0, 0};
  lentil_host_lens_pt_sample_aperture(L, s, a, shift);
  s[0] += s[2] * shift;
  s[1] += s[3] * shift;
  lentil_host_lens_evaluate(L, s, o);
  return y0_plane_z(pupil_to_camera(o, -L->k.lens_outer_pupil_curvature_radius, L->k.lens_outer_pupil_curvature_radius,
                                    L->k.lens_outer_pupil_geometry));
}

HOST_API double lentil_host_logarithmic_focus_search(const lentil_host_lens *L, double focal_distance, double lambda) {
  double best = 0.0, closest = 999999999.0;
  for (double i = -1.0; i <= 1.0; i += 0.0001) {            // logarithmic_values, src/lens.h:395-407
    const double shift = (i < 0 ? -1 : 1) * std::pow(i, 2.0) * 45.0 + 0.0;
    const double miss = focal_distance - lentil_host_camera_get_y0_intersection_distance(L, shift, lambda);
    if (miss < closest && miss > 0.0) { closest = miss; best = shift; }
  }
  return best;
}

HOST_API void lentil_host_trace_backwards_for_fstop(const lentil_host_lens *L, double fstop_target, double lambda,
                                                    double *calculated_fstop, double *calculated_aperture_radius) {
  const lentil_lens_table &k = L->k;
  double best_f = 0.0, best_r = 0.0;
  for (int i = 1; i < 1000; ++i) {
    const double h = ((double)i / 1000.0) * k.lens_outer_pupil_radius;
    const double target[3] = {0, h, (double)1.0e12f};        // AI_BIG
    const double ap[2] = {0.01, h};
    double s[5] = {0, 0, 0, 0, lambda}, o[5] = {0, 0, 0, 0, 0};
    if (lentil_host_lens_lt_sample_aperture(L, target, ap, s, o, lambda) <= 0.0) continue;
    const double px = s[0] + (s[2] * k.lens_back_focal_length), py = s[1] + (s[3] * k.lens_back_focal_length);
    if (px * px + py * py > k.lens_inner_pupil_radius * k.lens_inner_pupil_radius) continue;
    const double Ri = k.lens_inner_pupil_curvature_radius;
    const Ray3 r = pupil_to_camera(o, -Ri + k.lens_back_focal_length, Ri, k.lens_inner_pupil_geometry);
    const double f = 1.0 / (std::sin(std::atan(r.pos.y / r.pos.z)) * 2.0);
    if (f < fstop_target) break;
    best_f = f;
    best_r = h;
  }
  *calculated_fstop = best_f;
  *calculated_aperture_radius = best_r;
}

HOST_API int lentil_host_trace_ray_focus_check(const lentil_host_lens *L, double shift, double lambda, double *dist) {
  const lentil_lens_table &k = L->k;
  double s[5] = {0, 0, 0, 0, lambda};
  double a[5] = {0, k.lens_aperture_housing_radius * 0.25, 0, 0, 0};
  double o[5] = {0, 0, 0, 0, 0};
  lentil_host_lens_pt_sample_aperture(L, s, a, shift);
  s[0] += s[2] * shift;
  s[1] += s[3] * shift;
  if (lentil_host_lens_evaluate(L, s, o) <= 0.0) return 0;
  if (o[0] * o[0] + o[1] * o[1] > k.lens_outer_pupil_radius * k.lens_outer_pupil_radius) return 0;
  const double px = s[0] + s[2] * k.lens_back_focal_length, py = s[1] + s[3] * k.lens_back_focal_length;
  if (px * px + py * py > k.lens_inner_pupil_radius * k.lens_inner_pupil_radius) return 0;
  *dist = y0_plane_z(pupil_to_camera(o, -k.lens_outer_pupil_curvature_radius, k.lens_outer_pupil_curvature_radius,
                                     k.lens_outer_pupil_geometry));
  return 1;
}

HOST_API int lentil_host_camera_model_specific_setup(lentil_params *p, const lentil_host_lens *L, double input_fstop,
                                                     double wavelength_nm, double extra_sensor_shift, double *tan_fov) {
  return lentil_host_camera_model_specific_setup_with(p, L, input_fstop, wavelength_nm, extra_sensor_shift, tan_fov, nullptr, nullptr);
}

HOST_API int lentil_host_camera_model_specific_setup_with(lentil_params *p, const lentil_host_lens *L, double input_fstop,
                                                          double wavelength_nm, double extra_sensor_shift, double *tan_fov,
                                                          lentil_focus_search_fn focus_search, void *user) {
  if (!p) return -1;
  if (p->cameraType == LENTIL_POLYNOMIAL_OPTICS) {
    if (!L) return -1;
    const lentil_lens_table &k = L->k;
    p->focus_distance *= 10.0;                                           // :1573
    const double lambda = (double)(float)wavelength_nm * 0.001;          // :1213
    if (input_fstop == 0.0) {
      p->aperture_radius = k.lens_aperture_radius_at_fstop;              // :1604-1605
    } else {
      double f = 0.0, r = 0.0;
      lentil_host_trace_backwards_for_fstop(L, input_fstop, lambda, &f, &r);
      p->aperture_radius = std::min(k.lens_aperture_radius_at_fstop, r); // :1614
    }
    double best_shift = 0.0;
    // the 20 001-candidate search: the caller's accelerated version (lentil_hip_focus_search), or the loop here
    if (!focus_search || focus_search(user, p->focus_distance, lambda, &best_shift) != 0)
      best_shift = lentil_host_logarithmic_focus_search(L, p->focus_distance, lambda);
    p->sensor_shift = best_shift + (double)(float)extra_sensor_shift;
    if (tan_fov) *tan_fov = std::tan(k.lens_field_of_view / 2.0);        // :1658
  } else {
    const float fov = 2.0 * std::atan(p->sensor_width / (2.0 * p->focal_length));   // :1665
    if (tan_fov) *tan_fov = std::tan(fov / 2.0);
    p->aperture_radius = (p->focal_length / (2.0 * input_fstop)) / 10.0;             // :1667
  }
  return 0;
}

// ---- aperture samplers (host copies of what the kernels do) -------------------------------------
namespace {
float fsin(float x) {                   // fast_sin, src/lens.h:17-24
  x = fmodf(x + kPiF, kPiF * 2) - kPiF;
  const float B = 4.0f / kPiF, C = -4.0f / (kPiF * kPiF);
  const float y = B * x + C * x * std::fabs(x);
  return 0.225f * (y * std::fabs(y) - y) + y;
}
float fcos(float x) {                   // fast_cos, src/lens.h:27-37
  x = (float)((double)x + (double)kPiF * 0.5);
  x = fmodf(x + kPiF, kPiF * 2) - kPiF;
  const float B = 4.0f / kPiF, C = -4.0f / (kPiF * kPiF);
  const float y = B * x + C * x * std::fabs(x);
  return 0.225f * (y * std::fabs(y) - y) + y;
}
void disk_po(double ox, double oy, double &ux, double &uy) {     // concentric_disk_sample(.., true)
  const double a = 2.0 * ox - 1.0, b = 2.0 * oy - 1.0;
  double r, phi;
  if (a * a > b * b) { r = a; phi = 0.78539816339 * (b / a); }
  else { r = b; phi = (3.14159265358979323846 / 2.0) - 0.78539816339 * (a / b); }
  ux = r * fcos((float)phi);
  uy = r * fsin((float)phi);
}
float lerp1(float t, float a, float b) { return a + t * (b - a); }
void disk_thinlens(float ox, float oy, double &lx, double &ly, float bias, float squarelerp) {   // concentricDiskSample
  if (ox == 0.0f && oy == 0.0f) { lx = ly = 0.0; return; }
  const float a = (float)(2.0 * ox - 1.0), b = (float)(2.0 * oy - 1.0);
  float r, phi;
  if (a * a > b * b) { r = a; phi = (float)(0.78539816339 * (b / a)); }
  else { r = b; phi = (float)(kPiOver2F - 0.78539816339 * (a / b)); }
  if ((double)bias != 0.5) {            // AiBias
    const float m = std::fabs(r);
    const float v = m > 0 ? (bias > 0 ? powf(m, logf(bias) * -1.442695041f) : 0.0f) : 0.0f;
    r = v * (r < 0 ? -1 : 1);
  }
  lx = r * fcos(phi);
  ly = r * fsin(phi);
  if (squarelerp > 0.0f) { lx = lerp1(squarelerp, (float)lx, a); ly = lerp1(squarelerp, (float)ly, b); }
}
void triangle(double &x, double &y, double r1, double r2, double radius, int blades) {   // src/lentil.h:964-982
  const int tri = (int)(r1 * blades);
  r1 = r1 * blades - tri;
  const double a = std::sqrt(r1), b = (1.0 - r2) * a, c = r2 * a;
  const double p1 = (double)(2.0f * kPiF / blades * (tri + 1)), p2 = (double)(2.0f * kPiF / blades * tri);
  x = radius * (b * std::cos(p1) + c * std::cos(p2));
  y = radius * (b * std::sin(p1) + c * std::sin(p2));
}
void bokeh_pick(const lentil_bokeh_table *B, float uRow, float uCol, double &lx, double &ly) {   // bokehSample
  const int x = B->x, y = B->y;
  int r = (int)(std::upper_bound(B->cdfRow, B->cdfRow + y, uRow) - B->cdfRow);
  if (r >= y) r = y - 1;
  const int row = B->rowIndices[r];
  const int start = row * x;
  int c = (int)(std::upper_bound(B->cdfColumn + start, B->cdfColumn + start + x, uCol) - B->cdfColumn);
  if (c >= start + x) c = start + x - 1;
  const int col = B->columnIndices[c] - start;
  const float fr = (float)(col - (y - 1) / 2), fc = (float)(row - (x - 1) / 2) * -1.0f;
  lx = (double)(fr / (float)x) * 2.0;
  ly = (double)(fc / (float)y) * 2.0;
}
uint32_t xs128(uint32_t s[4]) {          // xor128, src/global.h:22-27
  const uint32_t t = s[0] ^ (s[0] << 11);
  s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
  return s[3] = (s[3] ^ (s[3] >> 19) ^ t ^ (t >> 8));
}
void normalize3f(float v[3]) {           // AiV3Normalize
  float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (l != 0) l = 1 / l;
  v[0] *= l; v[1] *= l; v[2] *= l;
}
}  // namespace

HOST_API void lentil_host_xor128_init(uint32_t s[4]) { s[0] = 123456789u; s[1] = 362436069u; s[2] = 521288629u; s[3] = 88675123u; }

HOST_API void lentil_host_trace_ray_fw_po(const lentil_params *P, const lentil_host_lens *L, const lentil_bokeh_table *B,
                                          uint32_t rng[4], double lambda, double sx, double sy, double *r1, double *r2,
                                          int deriv_ray, float origin[3], float direction[3], float weight[3], int *tries_out) {
  const lentil_lens_table &k = L->k;
  int tries = 0;
  bool ok = false;
  double s[5] = {0, 0, 0, 0, 0}, a[5] = {0, 0, 0, 0, 0}, o[5] = {0, 0, 0, 0, 0};
  while (!ok && tries <= P->vignetting_retries) {
    s[0] = sx * (P->sensor_width * 0.5); s[1] = sy * (P->sensor_width * 0.5); s[2] = s[3] = 0.0; s[4] = lambda;
    std::fill(a, a + 5, 0.0);
    std::fill(o, o + 5, 0.0);
    double ux = 0.0, uy = 0.0;
    if (P->enable_dof) {
      if (!deriv_ray && tries > 0) { *r1 = xs128(rng) / 4294967296.0; *r2 = xs128(rng) / 4294967296.0; }
      if (P->bokeh_enable_image) { xs128(rng); xs128(rng); bokeh_pick(B, (float)*r1, (float)*r2, ux, uy); }
      else if (P->bokeh_aperture_blades < 2) disk_po(*r1, *r2, ux, uy);
      else triangle(ux, uy, *r1, *r2, 1.0, P->bokeh_aperture_blades);
    }
    a[0] = ux * P->aperture_radius;
    a[1] = uy * P->aperture_radius;
    if (P->enable_dof) lentil_host_lens_pt_sample_aperture(L, s, a, P->sensor_shift);
    s[0] += s[2] * P->sensor_shift;
    s[1] += s[3] * P->sensor_shift;
    if (lentil_host_lens_evaluate(L, s, o) <= 0.0) { ++tries; continue; }
    if (o[0] * o[0] + o[1] * o[1] > k.lens_outer_pupil_radius * k.lens_outer_pupil_radius) { ++tries; continue; }
    const double px = s[0] + s[2] * k.lens_back_focal_length, py = s[1] + s[3] * k.lens_back_focal_length;
    if (px * px + py * py > k.lens_inner_pupil_radius * k.lens_inner_pupil_radius) { ++tries; continue; }
    ok = true;
  }
  if (!ok) weight[0] = weight[1] = weight[2] = 0.0f;
  const Ray3 r = pupil_to_camera(o, -k.lens_outer_pupil_curvature_radius, k.lens_outer_pupil_curvature_radius,
                                 k.lens_outer_pupil_geometry);
  float og[3] = {(float)r.pos.x, (float)r.pos.y, (float)r.pos.z};
  float dg[3] = {(float)r.dir.x, (float)r.dir.y, (float)r.dir.z};
  const float sc = P->unitModel == LENTIL_UNIT_MM ? -1.0f : P->unitModel == LENTIL_UNIT_CM ? -0.1f
                 : P->unitModel == LENTIL_UNIT_DM ? -0.01f : -0.001f;
  for (int i = 0; i < 3; ++i) { og[i] *= sc; dg[i] *= sc; }
  normalize3f(dg);
  bool nan = false;
  for (int i = 0; i < 3; ++i) { origin[i] = og[i]; direction[i] = dg[i]; nan |= (og[i] != og[i]) || (dg[i] != dg[i]); }
  if (nan) weight[0] = weight[1] = weight[2] = 0.0f;
  if (tries_out) *tries_out = tries;
}

// Coma of the thin-lens model, src/lens.h:563-582: how far the centre ray is from the sensor's corner ray
// and how far the lens sample is from the aperture centre scale a rotation of the ray about the axis
// orthogonal to it and -z.  Eigen's AngleAxisd::toRotationMatrix / Matrix3d::inverse / Matrix3d * Vector3d
// are written out (operation order of Eigen 3.3/3.4).
static float coma_amount(const lentil_params *P, const float dc[3], double ux, double uy) {
  float corner[3] = {(float)(1.0 * (P->sensor_width * 0.5)), (float)(1.0 * (P->sensor_width * 0.5)), -P->focal_length};
  normalize3f(corner);
  const float max_proj = corner[0] * 0.0f + corner[1] * 0.0f + corner[2] * -1.0f;
  const float cur_proj = dc[0] * 0.0f + dc[1] * 0.0f + dc[2] * -1.0f;
  const float perc = (float)((((double)(cur_proj - max_proj) / (1.0 - (double)max_proj)) - 0.5) * 2.0);
  const float from_center = (float)(1.0 - (double)perc);
  const float from_aperture = (float)std::sqrt(ux * ux + uy * uy);
  return from_center * from_aperture;
}

static void coma_rotate(const float about[3], const float ray[3], float amount, bool inverse, float out[3]) {
  float ax[3] = {about[1] * -1.0f - about[2] * 0.0f, about[2] * 0.0f - about[0] * -1.0f, about[0] * 0.0f - about[1] * 0.0f};
  normalize3f(ax);
  const double a[3] = {ax[0], ax[1], ax[2]};
  const double angle = ((double)amount * 2.3456 * (double)3.14159265358979323846f) / 180.0;
  const double sn = std::sin(angle), cs = std::cos(angle);
  double R[3][3], I[3][3];
  const double sa[3] = {sn * a[0], sn * a[1], sn * a[2]};
  const double ca[3] = {(1.0 - cs) * a[0], (1.0 - cs) * a[1], (1.0 - cs) * a[2]};
  double t = ca[0] * a[1];  R[0][1] = t - sa[2];  R[1][0] = t + sa[2];
  t = ca[0] * a[2];         R[0][2] = t + sa[1];  R[2][0] = t - sa[1];
  t = ca[1] * a[2];         R[1][2] = t - sa[0];  R[2][1] = t + sa[0];
  for (int i = 0; i < 3; ++i) R[i][i] = ca[i] * a[i] + cs;
  const double (*M)[3] = R;
  if (inverse) {
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return R[i1][j1] * R[i2][j2] - R[i1][j2] * R[i2][j1];
    };
    const double c0[3] = {cof(0, 0), cof(1, 0), cof(2, 0)};
    const double invdet = 1.0 / (c0[0] * R[0][0] + (c0[1] * R[1][0] + c0[2] * R[2][0]));
    I[1][0] = cof(0, 1) * invdet; I[1][1] = cof(1, 1) * invdet; I[2][0] = cof(0, 2) * invdet;
    I[1][2] = cof(2, 1) * invdet; I[2][1] = cof(1, 2) * invdet; I[2][2] = cof(2, 2) * invdet;
    I[0][0] = c0[0] * invdet; I[0][1] = c0[1] * invdet; I[0][2] = c0[2] * invdet;
    M = I;
  }
  const double r[3] = {ray[0], ray[1], ray[2]};
  for (int i = 0; i < 3; ++i) out[i] = (float)(M[i][0] * r[0] + (M[i][1] * r[1] + M[i][2] * r[2]));
}

HOST_API void lentil_host_trace_ray_fw_thinlens(const lentil_params *P, const lentil_bokeh_table *B, uint32_t rng[4],
                                                double sx, double sy, double *r1, double *r2, int deriv_ray,
                                                float origin[3], float direction[3], float weight[3], int *tries_out) {
  int tries = 0;
  bool ok = false;
  float og[3] = {0, 0, 0}, dg[3] = {0, 0, 0};
  while (!ok && tries <= P->vignetting_retries) {
    float ssx = (float)sx, ssy = (float)sy;
    if (P->abb_distortion > 0.0f) {                          // barrelDistortion, src/lens.h:545-548
      const float f = (float)(1. + (double)((ssx * ssx + ssy * ssy) * P->abb_distortion));   // AiV2Dot * distortion in float
      ssx *= f; ssy *= f;
    }
    float dc[3] = {(float)(ssx * (P->sensor_width * 0.5)), (float)(ssy * (P->sensor_width * 0.5)), -P->focal_length};
    normalize3f(dc);
    double ux = 0.0, uy = 0.0;
    if (P->enable_dof) {
      if (!deriv_ray && tries > 0) { *r1 = xs128(rng) / 4294967296.0; *r2 = xs128(rng) / 4294967296.0; }
      if (P->bokeh_enable_image) { xs128(rng); xs128(rng); bokeh_pick(B, (float)*r1, (float)*r2, ux, uy); }
      else if (P->bokeh_aperture_blades < 2) disk_thinlens((float)*r1, (float)*r2, ux, uy, P->abb_spherical, P->circle_to_square);
      else triangle(ux, uy, *r1, *r2, 1.0, P->bokeh_aperture_blades);
    }
    ux *= (double)P->bokeh_anamorphic;
    const float lens[3] = {(float)(ux * P->aperture_radius), (float)(uy * P->aperture_radius), 0.0f};
    const float hit = (float)std::fabs(P->focus_distance / (double)lerp1(0.0f, dc[2], 1.0f));
    float dl[3] = {dc[0] * hit - lens[0], dc[1] * hit - lens[1], dc[2] * hit - lens[2]};
    normalize3f(dl);
    {                                                                   // src/lentil.h:490-491
      float rot[3];
      coma_rotate(dl, dl, P->abb_coma * coma_amount(P, dc, ux, uy), false, rot);
      dl[0] = rot[0]; dl[1] = rot[1]; dl[2] = rot[2];
    }
    if (P->optical_vignetting_distance > 0.0f && !deriv_ray) {          // src/lens.h:529-543
      const float squarebias = (float)(1.0 + std::log(1.0 + (double)P->circle_to_square) * std::exp((double)P->circle_to_square * 3.0));
      const float inter = std::fabs(P->optical_vignetting_distance / dl[2]);
      const float vx = dl[0] * inter - lens[0], vy = dl[1] * inter - lens[1];
      const float power = (float)(1.0 + (double)squarebias);
      const float radius = (float)P->aperture_radius * P->optical_vignetting_radius;
      if (powf(std::fabs(vx), power) + powf(std::fabs(vy), power) > powf(radius, power)) { ++tries; continue; }
    }
    const float sc = P->unitModel == LENTIL_UNIT_MM ? 10.0f : P->unitModel == LENTIL_UNIT_CM ? 1.0f
                   : P->unitModel == LENTIL_UNIT_DM ? 0.1f : 0.01f;
    for (int i = 0; i < 3; ++i) { og[i] = lens[i] * sc; dg[i] = dl[i] * sc; }
    ok = true;
  }
  normalize3f(dg);
  for (int i = 0; i < 3; ++i) { origin[i] = og[i]; direction[i] = dg[i]; }
  if (!ok) weight[0] = weight[1] = weight[2] = 0.0f;
  if (tries_out) *tries_out = tries;
}

HOST_API void lentil_host_camera_create_ray(const lentil_params *P, const lentil_host_lens *L, const lentil_bokeh_table *B,
                                            uint32_t rng[4], double lambda, float exposure, const float in[6],
                                            lentil_host_camera_ray *out) {
  // src/lentil_camera.cpp:78-125
  int tries = 0;
  double r1 = in[4], r2 = in[5];
  const float step = 0.001f;
  float w[3] = {1, 1, 1}, wdx[3] = {1, 1, 1}, wdy[3] = {1, 1, 1};
  float ox[3], dx_[3], oy[3], dy_[3];
  auto trace = [&](double sx, double sy, float *o, float *d, float *ww, int deriv) {
    if (P->cameraType == LENTIL_THINLENS) lentil_host_trace_ray_fw_thinlens(P, B, rng, sx, sy, &r1, &r2, deriv, o, d, ww, &tries);
    else lentil_host_trace_ray_fw_po(P, L, B, rng, lambda, sx, sy, &r1, &r2, deriv, o, d, ww, &tries);
  };
  trace(in[0], in[1], out->origin, out->dir, w, 0);
  const float sxd = in[0] + (in[2] * step), syd = in[1] + (in[3] * step);
  trace(sxd, in[1], ox, dx_, wdx, 1);
  trace(in[0], syd, oy, dy_, wdy, 1);
  for (int i = 0; i < 3; ++i) {
    const float inv = 1.0f / step;                           // AtVector / float multiplies by 1/f
    out->dOdx[i] = (ox[i] - out->origin[i]) * inv;
    out->dOdy[i] = (oy[i] - out->origin[i]) * inv;
    out->dDdx[i] = (dx_[i] - out->dir[i]) * inv;
    out->dDdy[i] = (dy_[i] - out->dir[i]) * inv;
    out->weight[i] = w[i] * exposure;
  }
}

HOST_API void lentil_host_camera_reverse_ray(double tan_fov, const float Po[3], float Ps[2]) {
  const double coeff = 1.0 / std::max(std::abs((double)Po[2] * tan_fov), 1e-3);   // src/lentil_camera.cpp:164-172
  Ps[0] = (float)(Po[0] * coeff);
  Ps[1] = (float)(Po[1] * coeff);
}
