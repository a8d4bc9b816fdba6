/*
 * lentil_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's (zpelgrims/pota, "lentil") bidirectional
 * redistribution hot path, used as the parity oracle for the HIP implementation and
 * as the "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (pota_amd/, liblentil_hip.so) never does.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Conventions pinned: GCC right-to-left
 * evaluation of the rng() call arguments, no FP contraction (-ffp-contract=off, the
 * reference is built -O3 for baseline x86-64 = no FMA), sequential pixel-major visit
 * order, AiTraceProbe == false, explicit buffer stride xres.
 *
 * PARITY STATUS
 *  - pinned against the reference's own data: tea<8>/rng/xor128 known answers
 *    (produced from src/global.h), and the per-iteration
 *    Newton-solver trace tests/aperture_sampling_debug/writout.txt (2x2 inverses, update
 *    steps incl. the 0.72 damping, sphereToCs, csToSphere, normalise, error-flag rules).
 *  - PARITY UNPINNED for polynomial values: the generated lens code
 *    (zpelgrims/polynomial-optics, database/lenses/<lens>/<fl>/code/<file>.h; no pinned version,
 *    not a submodule) is absent from the reference tree, so the polynomial tables are the
 *    build's own (tools/fit_lens.py) and the solver body restates the published
 *    polynomial-optics generator output as evidenced by that trace.
 *  - PARITY UNPINNED for Arnold SDK inline helpers (AiV3Normalize, AiM4PointByMatrixMult,
 *    AiBias, AI_* constants, AtRGBA * AtRGB = AtRGBA(rgb, 1) in the chromatic splat, an int assigned
 *    to an AtRGBA setting all four components for lentil_debug): the SDK is absent; they are
 *    restated from its public headers as recalled.
 *  - PARITY UNPINNED for the three Eigen operations of the thin-lens coma rotation (AngleAxisd ->
 *    Matrix3d, Matrix3d::inverse, Matrix3d * Vector3d): Eigen is included from a sibling checkout
 *    (src/lens.h:5-6, no version pinned) and absent here; restated from Eigen 3.3/3.4's sources.
 *  - PARITY UNPINNED for the cryptomatte path (cache construction, per-pixel maps, ranking): the reference
 *    holds scenes with cryptomatte AOVs (tests/tl_redistribution_bug) but no captured samples or images for
 *    them; AiColorToGrey is restated as recalled ((r + g + b) / 3).  Pinned only by hand-worked depth lists
 *    (tests/test_crypto.py) and by std::map / std::sort being libstdc++'s own here as there.
 *  The reference itself is unbuildable here (needs <ai.h>, Eigen, CryptomatteArnold and
 *  the generated lens code), so there is no oracle/_ref.
 *
 * Written as C-style C++ (g++) on purpose: std::pow / std::abs / std::ceil / std::min
 * overload resolution and std::sort / std::upper_bound tie behaviour are then the very
 * ones the reference gets from libstdc++.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include "../include/lentil_hip.h"

#define ORC_API extern "C" __attribute__((visibility("default")))

/* Arnold SDK constants (ai_constants.h; SDK absent, recalled -- SURVEY appendix C.16) */
static const float AI_PI_F = 3.14159265358979323846f;
static const float AI_PIOVER2_F = 1.57079632679489661923f;
static const float AI_EPSILON_F = 1.0e-4f;
static const float AI_INFINITE_F = 1.0e30f;
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* =====================================================================================
 * a6 -- src/global.h:22-57
 * ===================================================================================== */
ORC_API uint32_t orc_tea8(uint32_t val0, uint32_t val1) {      /* tea<8>, src/global.h:32-46 */
  uint32_t v0 = val0, v1 = val1, s0 = 0;
  for (int n = 0; n < 8; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xC8013EA4u);
    v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7E95761Eu);
  }
  return v0;
}

ORC_API float orc_rng(uint32_t *state) {                        /* rng, src/global.h:51-57 */
  *state = *state * 1664525u + 1013904223u;
  return float(*state & 0x00FFFFFFu) / float(0x01000000u);
}

ORC_API uint32_t orc_xor128(uint32_t st[4]) {                   /* xor128, src/global.h:22-27 */
  uint32_t t = st[0] ^ (st[0] << 11);
  st[0] = st[1]; st[1] = st[2]; st[2] = st[3];
  return st[3] = (st[3] ^ (st[3] >> 19) ^ t ^ (t >> 8));
}
ORC_API void orc_xor128_init(uint32_t st[4]) {
  st[0] = 123456789u; st[1] = 362436069u; st[2] = 521288629u; st[3] = 88675123u;
}

static inline float lerpf(float perc, float a, float b) { return a + perc * (b - a); } /* global.h:3-5 */
static inline float clampf(float in, float lo, float hi) {                              /* global.h:8-12 */
  if (in < lo) in = lo;
  if (in > hi) in = hi;
  return in;
}

/* =====================================================================================
 * a9 -- src/lens.h:17-37 (fast trig), :309-333 (PO disk), :477-514 (thin-lens disk)
 * ===================================================================================== */
ORC_API float orc_fast_sin(float x) {
  x = fmod(x + AI_PI_F, AI_PI_F * 2) - AI_PI_F;
  const float B = 4.0f / AI_PI_F;
  const float C = -4.0f / (AI_PI_F * AI_PI_F);
  float y = B * x + C * x * std::abs(x);
  const float P = 0.225f;
  return P * (y * std::abs(y) - y) + y;
}
ORC_API float orc_fast_cos(float x) {
  x += AI_PI_F * 0.5;
  x = fmod(x + AI_PI_F, AI_PI_F * 2) - AI_PI_F;
  const float B = 4.0f / AI_PI_F;
  const float C = -4.0f / (AI_PI_F * AI_PI_F);
  float y = B * x + C * x * std::abs(x);
  const float P = 0.225f;
  return P * (y * std::abs(y) - y) + y;
}

/* concentric_disk_sample(ox, oy, unit_disk, fast_trigo=true), src/lens.h:309-333 */
ORC_API void orc_concentric_disk_sample(double ox, double oy, double disk[2]) {
  double phi, r;
  double a = 2.0 * ox - 1.0;
  double b = 2.0 * oy - 1.0;
  if ((a * a) > (b * b)) {
    r = a;
    phi = (0.78539816339) * (b / a);
  } else {
    r = b;
    phi = (M_PI / 2.0) - (0.78539816339) * (a / b);
  }
  disk[0] = r * orc_fast_cos(phi);   /* phi narrows to float at the call, appendix C.11 */
  disk[1] = r * orc_fast_sin(phi);
}

/* AiBias (ai_math / SDK, recalled -- appendix C.16) */
static inline float ai_bias(float a, float b) {
  return (a > 0) ? ((b > 0) ? powf(a, logf(b) * -1.442695041f) : 0.0f) : 0.0f;
}

/* concentricDiskSample(ox, oy, lens, bias, squarelerp, squeeze_x), src/lens.h:477-514 */
ORC_API void orc_concentricDiskSample(float ox, float oy, double lens[2], float bias,
                                      float squarelerp) {
  if (ox == 0.0 && oy == 0.0) { lens[0] = 0.0; lens[1] = 0.0; return; }
  float phi, r;
  const float a = 2.0 * ox - 1.0;
  const float b = 2.0 * oy - 1.0;
  if ((a * a) > (b * b)) {
    r = a;
    phi = 0.78539816339 * (b / a);
  } else {
    r = b;
    phi = (AI_PIOVER2_F) - ((0.78539816339) * (a / b));
  }
  if (bias != 0.5) r = ai_bias(std::abs(r), bias) * (r < 0 ? -1 : 1);
  const float cos_phi = orc_fast_cos(phi);
  const float sin_phi = orc_fast_sin(phi);
  lens[0] = r * cos_phi;
  lens[1] = r * sin_phi;
  if (squarelerp > 0.0) {
    lens[0] = lerpf(squarelerp, lens[0], a);   /* double -> float args, float result */
    lens[1] = lerpf(squarelerp, lens[1], b);
  }
}

/* a10 -- Camera::lens_sample_triangular_aperture, src/lentil.h:964-982 */
ORC_API void orc_triangular_aperture(double *x, double *y, double r1, double r2, double radius,
                                     int blades) {
  const int tri = (int)(r1 * blades);
  r1 = r1 * blades - tri;
  double a = std::sqrt(r1);
  double b = (1.0f - r2) * a;
  double c = r2 * a;
  double p1[2], p2[2];
  double ph1 = 2.0f * AI_PI_F / blades * (tri + 1);   /* float expression, widened at the call */
  double ph2 = 2.0f * AI_PI_F / blades * tri;
  p1[0] = std::sin(ph1); p1[1] = std::cos(ph1);       /* common_sincosf, src/lens.h:40-43 */
  p2[0] = std::sin(ph2); p2[1] = std::cos(ph2);
  *x = radius * (b * p1[1] + c * p2[1]);
  *y = radius * (b * p1[0] + c * p2[0]);
}

/* =====================================================================================
 * a14 -- src/lens.h:47-60 (helpers), :99-221 (pupil transforms), :226-233 (lens_ipow)
 * ===================================================================================== */
static inline double dot3(const double u[3], const double v[3]) {
  return u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
}
static inline void cross3(double r[3], const double u[3], const double v[3]) {
  r[0] = u[1] * v[2] - u[2] * v[1];
  r[1] = u[2] * v[0] - u[0] * v[2];
  r[2] = u[0] * v[1] - u[1] * v[0];
}
ORC_API void orc_normalise(double v[3]) {                       /* raytrace_normalise */
  const double ilen = 1.0f / std::sqrt(dot3(v, v));
  for (int k = 0; k < 3; k++) v[k] *= ilen;
}

ORC_API void orc_sphereToCs(const double inpos[2], const double indir[2], double outpos[3],
                            double outdir[3], double center, double R) {
  const double normal[3] = {
      inpos[0] / R, inpos[1] / R,
      std::sqrt(std::max(0.0, R * R - inpos[0] * inpos[0] - inpos[1] * inpos[1])) / std::abs(R)};
  const double tempDir[3] = {
      indir[0], indir[1],
      std::sqrt(std::max(0.0, 1.0 - indir[0] * indir[0] - indir[1] * indir[1]))};
  double ex[3] = {normal[2], 0, -normal[0]};
  orc_normalise(ex);
  double ey[3];
  cross3(ey, normal, ex);
  outdir[0] = tempDir[0] * ex[0] + tempDir[1] * ey[0] + tempDir[2] * normal[0];
  outdir[1] = tempDir[0] * ex[1] + tempDir[1] * ey[1] + tempDir[2] * normal[1];
  outdir[2] = tempDir[0] * ex[2] + tempDir[1] * ey[2] + tempDir[2] * normal[2];
  outpos[0] = inpos[0];
  outpos[1] = inpos[1];
  outpos[2] = normal[2] * R + center;
}

ORC_API void orc_csToSphere(const double inpos[3], const double indir[3], double outpos[2],
                            double outdir[2], double center, double R) {
  const double normal[3] = {inpos[0] / R, inpos[1] / R, std::abs((inpos[2] - center) / R)};
  double tempDir[3] = {indir[0], indir[1], indir[2]};
  orc_normalise(tempDir);
  double ex[3] = {normal[2], 0, -normal[0]};
  orc_normalise(ex);
  double ey[3];
  cross3(ey, normal, ex);
  outdir[0] = dot3(tempDir, ex);
  outdir[1] = dot3(tempDir, ey);
  outpos[0] = inpos[0];
  outpos[1] = inpos[1];
}

ORC_API void orc_cylinderToCs(const double inpos[2], const double indir[2], double outpos[3],
                              double outdir[3], double center, double R, int cyl_y) {
  double normal[3] = {0, 0, 0};
  if (cyl_y) {
    normal[0] = inpos[0] / R;
    normal[2] = std::sqrt(std::max(0.0, R * R - inpos[0] * inpos[0])) / std::abs(R);
  } else {
    normal[1] = inpos[1] / R;
    normal[2] = std::sqrt(std::max(0.0, R * R - inpos[1] * inpos[1])) / std::abs(R);
  }
  const double tempDir[3] = {
      indir[0], indir[1],
      std::sqrt(std::max(0.0, 1.0 - indir[0] * indir[0] - indir[1] * indir[1]))};
  double ex[3] = {normal[2], 0, -normal[0]};
  orc_normalise(ex);
  double ey[3];
  cross3(ey, normal, ex);
  orc_normalise(ey);
  outdir[0] = tempDir[0] * ex[0] + tempDir[1] * ey[0] + tempDir[2] * normal[0];
  outdir[1] = tempDir[0] * ex[1] + tempDir[1] * ey[1] + tempDir[2] * normal[1];
  outdir[2] = tempDir[0] * ex[2] + tempDir[1] * ey[2] + tempDir[2] * normal[2];
  outpos[0] = inpos[0];
  outpos[1] = inpos[1];
  outpos[2] = normal[2] * R + center;
}

ORC_API void orc_csToCylinder(const double inpos[3], const double indir[3], double outpos[2],
                              double outdir[2], double center, double R, int cyl_y) {
  double normal[3] = {0, 0, 0};
  if (cyl_y) {
    normal[0] = inpos[0] / R;
    normal[2] = std::abs((inpos[2] - center) / R);
  } else {
    normal[1] = inpos[1] / R;
    normal[2] = std::abs((inpos[2] - center) / R);
  }
  double tempDir[3] = {indir[0], indir[1], indir[2]};
  orc_normalise(tempDir);
  double ex[3] = {normal[2], 0, -normal[0]};     /* not normalised here, src/lens.h:171 */
  double ey[3];
  cross3(ey, normal, ex);
  orc_normalise(ey);
  outdir[0] = dot3(tempDir, ex);
  outdir[1] = dot3(tempDir, ey);
  outpos[0] = inpos[0];
  outpos[1] = inpos[1];
}

ORC_API double orc_lens_ipow(double x, int exp) {               /* src/lens.h:226-233 */
  if (exp == 0) return 1.0f;
  if (exp == 1) return x;
  if (exp == 2) return x * x;
  const double p2 = orc_lens_ipow(x, exp / 2);
  if (exp & 1) return x * p2 * p2;
  return p2 * p2;
}

/* =====================================================================================
 * Polynomial tables (stand-in for the generated code spliced at src/lentil.h:1262,1278,1308).
 * Evaluation mirrors how polynomial-optics prints a polynomial:
 *     out = + c0*<factors> + c1*<factors> ...   (left to right)
 * a factor is the bare variable for exponent 1, lens_ipow(v, e) otherwise, in variable
 * order x, y, dx, dy, lambda.   PARITY UNPINNED (generator output absent).
 * ===================================================================================== */
struct OrcPoly { std::vector<lentil_term> t; };

struct OrcLens {
  lentil_lens_table k;          /* constants (terms pointer unused) */
  OrcPoly out[5], ap[4];
  OrcPoly dap[2][2];            /* d ap_{x,y} / d {dx,dy}   ("dx1_domega0") */
  OrcPoly dout[2][2];           /* d out_{dx,dy} / d {x,y}  ("domega2_dx0") */
};

static OrcPoly derive(const OrcPoly &p, int var) {
  /* derivative term: coefficient * exponent, exponent - 1; zero-exponent terms vanish */
  OrcPoly d;
  for (const lentil_term &t : p.t) {
    if (t.e[var] == 0) continue;
    lentil_term n = t;
    n.c = t.c * (double)t.e[var];
    n.e[var] = (uint8_t)(t.e[var] - 1);
    d.t.push_back(n);
  }
  return d;
}

static inline double eval_poly(const OrcPoly &p, const double v[5]) {
  double sum = 0.0;
  bool first = true;
  for (const lentil_term &t : p.t) {
    double term = t.c;
    for (int i = 0; i < 5; i++) {
      if (t.e[i] == 0) continue;
      term = term * (t.e[i] == 1 ? v[i] : orc_lens_ipow(v[i], t.e[i]));
    }
    if (first) { sum = term; first = false; } else { sum = sum + term; }
  }
  return sum;
}

ORC_API OrcLens *orc_lens_create(const lentil_lens_table *tab) {
  OrcLens *L = new OrcLens();
  L->k = *tab;
  L->k.terms = nullptr;
  for (int i = 0; i < 5; i++)
    L->out[i].t.assign(tab->terms + tab->out[i].first, tab->terms + tab->out[i].first + tab->out[i].count);
  for (int i = 0; i < 4; i++)
    L->ap[i].t.assign(tab->terms + tab->ap[i].first, tab->terms + tab->ap[i].first + tab->ap[i].count);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) {
      L->dap[i][j] = derive(L->ap[i], 2 + j);
      L->dout[i][j] = derive(L->out[2 + i], j);
    }
  return L;
}
ORC_API void orc_lens_destroy(OrcLens *L) { delete L; }

ORC_API double orc_poly_eval(const OrcLens *L, int which, const double v[5]) {
  /* which: 0-4 out, 5-8 ap, 9-12 dap[i][j], 13-16 dout[i][j] */
  if (which < 5) return eval_poly(L->out[which], v);
  if (which < 9) return eval_poly(L->ap[which - 5], v);
  if (which < 13) return eval_poly(L->dap[(which - 9) / 2][(which - 9) % 2], v);
  return eval_poly(L->dout[(which - 13) / 2][(which - 13) % 2], v);
}

/* 2x2 inverse exactly as the generated solvers write it (trace order [0][0],[1][1],[0][1],[1][0]) */
ORC_API void orc_inv2x2(const double J[2][2], double inv[2][2], double *invdet_out) {
  const double invdet = 1.0f / (J[0][0] * J[1][1] - J[0][1] * J[1][0]);
  inv[0][0] = J[1][1] * invdet;
  inv[1][1] = J[0][0] * invdet;
  inv[0][1] = -J[0][1] * invdet;
  inv[1][0] = -J[1][0] * invdet;
  if (invdet_out) *invdet_out = invdet;
}

/* the two Newton update steps, separately callable for the writout.txt known answers */
ORC_API void orc_newton_step(const double inv[2][2], const double delta[2], double damping,
                             double *a, double *b) {
  for (int i = 0; i < 2; i++) {
    *a += damping * inv[0][i] * delta[i];
    *b += damping * inv[1][i] * delta[i];
  }
}

ORC_API int orc_newton_error_bits(double sqr_err, double prev_sqr_err, double sqr_ap_err,
                                  double prev_sqr_ap_err, double out0, double out2, double out3) {
  int error = 0;
  if (sqr_err > prev_sqr_err) error |= 1;
  if (sqr_ap_err > prev_sqr_ap_err) error |= 2;
  if (out0 != out0) error |= 4;
  if (out2 * out2 + out3 * out3 > 1.0) error |= 8;
  return error;
}

static inline void pupil_to_cs(const lentil_lens_table &k, const double inpos[2], const double indir[2],
                               double pos[3], double dir[3]) {
  const double R = k.lens_outer_pupil_curvature_radius;
  if (k.lens_outer_pupil_geometry == LENTIL_GEOM_CYL_Y) orc_cylinderToCs(inpos, indir, pos, dir, -R, R, 1);
  else if (k.lens_outer_pupil_geometry == LENTIL_GEOM_CYL_X) orc_cylinderToCs(inpos, indir, pos, dir, -R, R, 0);
  else orc_sphereToCs(inpos, indir, pos, dir, -R, R);
}
static inline void cs_to_pupil(const lentil_lens_table &k, const double inpos[3], const double indir[3],
                               double pos[2], double dir[2]) {
  const double R = k.lens_outer_pupil_curvature_radius;
  if (k.lens_outer_pupil_geometry == LENTIL_GEOM_CYL_Y) orc_csToCylinder(inpos, indir, pos, dir, -R, R, 1);
  else if (k.lens_outer_pupil_geometry == LENTIL_GEOM_CYL_X) orc_csToCylinder(inpos, indir, pos, dir, -R, R, 0);
  else orc_csToSphere(inpos, indir, pos, dir, -R, R);
}

/* -------------------------------------------------------------------------------------
 * a12 -- Camera::lens_lt_sample_aperture, src/lentil.h:1296-1313, with the generated body
 * restated from tests/aperture_sampling_debug/writout.txt (all polynomials of one iteration
 * are evaluated at the state the iteration *began* with -- "begin_x ... begin_lambda",
 * writout.txt:13-17; aperture step undamped :19-24; outer-pupil step damped by 0.72 :33-39
 * and newton-w4.py:45; error bits reset while k<10 :40; stop when both squared errors
 * <= 1e-8, as in the converged trace ending at k=10 :302-333).
 * `out` is in/out like the reference's Eigen::VectorXd &out (appendix C.7).
 * ------------------------------------------------------------------------------------- */
ORC_API double orc_lt_sample_aperture(const OrcLens *L, const double scene[3], const double ap[2],
                                      double sensor[5], double out[5], double lambda,
                                      int *iterations) {
  const lentil_lens_table &k = L->k;
  double x = 0, y = 0, dx = 0, dy = 0;
  int error = 0;
  const double eps = 1e-8;
  double sqr_err = 1e30, sqr_ap_err = 1e30;
  double prev_sqr_err = 1e32, prev_sqr_ap_err = 1e32;
  int it = 0;
  for (int kk = 0; kk < 100 && (sqr_err > eps || sqr_ap_err > eps) && error == 0; kk++) {
    prev_sqr_err = sqr_err;
    prev_sqr_ap_err = sqr_ap_err;
    const double begin[5] = {x, y, dx, dy, lambda};
    const double pred_ap[2] = {eval_poly(L->ap[0], begin), eval_poly(L->ap[1], begin)};
    const double delta_ap[2] = {ap[0] - pred_ap[0], ap[1] - pred_ap[1]};
    sqr_ap_err = delta_ap[0] * delta_ap[0] + delta_ap[1] * delta_ap[1];
    double J[2][2], inv[2][2];
    J[0][0] = eval_poly(L->dap[0][0], begin);
    J[0][1] = eval_poly(L->dap[0][1], begin);
    J[1][0] = eval_poly(L->dap[1][0], begin);
    J[1][1] = eval_poly(L->dap[1][1], begin);
    orc_inv2x2(J, inv, nullptr);
    for (int i = 0; i < 2; i++) {
      dx += inv[0][i] * delta_ap[i];
      dy += inv[1][i] * delta_ap[i];
    }
    out[0] = eval_poly(L->out[0], begin);
    out[1] = eval_poly(L->out[1], begin);
    out[2] = eval_poly(L->out[2], begin);
    out[3] = eval_poly(L->out[3], begin);
    double pos[3], dir[3];
    pupil_to_cs(k, out, out + 2, pos, dir);
    double view[3] = {scene[0] - pos[0], scene[1] - pos[1], scene[2] - pos[2]};
    orc_normalise(view);
    double out_new[4];
    cs_to_pupil(k, pos, view, out_new, out_new + 2);
    const double delta_out[2] = {out_new[2] - out[2], out_new[3] - out[3]};
    sqr_err = delta_out[0] * delta_out[0] + delta_out[1] * delta_out[1];
    J[0][0] = eval_poly(L->dout[0][0], begin);
    J[0][1] = eval_poly(L->dout[0][1], begin);
    J[1][0] = eval_poly(L->dout[1][0], begin);
    J[1][1] = eval_poly(L->dout[1][1], begin);
    orc_inv2x2(J, inv, nullptr);
    for (int i = 0; i < 2; i++) {
      x += 0.72 * inv[0][i] * delta_out[i];
      y += 0.72 * inv[1][i] * delta_out[i];
    }
    error |= orc_newton_error_bits(sqr_err, prev_sqr_err, sqr_ap_err, prev_sqr_ap_err, out[0], out[2], out[3]);
    if (kk < 10) error = 0;
    it = kk + 1;
  }
  if (out[0] * out[0] + out[1] * out[1] > k.lens_outer_pupil_radius * k.lens_outer_pupil_radius) error |= 16;
  if (error == 0) {
    const double begin[5] = {x, y, dx, dy, lambda};
    out[4] = eval_poly(L->out[4], begin);
  } else {
    out[4] = 0.0f;
  }
  if (iterations) *iterations = it;
  sensor[0] = x; sensor[1] = y; sensor[2] = dx; sensor[3] = dy; sensor[4] = lambda;  /* lentil.h:1311 */
  return std::max(0.0, out[4]);                                                     /* lentil.h:1312 */
}

/* a13 -- Camera::lens_evaluate, src/lentil.h:1257-1266 */
ORC_API double orc_lens_evaluate(const OrcLens *L, const double in[5], double out[5]) {
  for (int i = 0; i < 4; i++) out[i] = eval_poly(L->out[i], in);
  const double t = eval_poly(L->out[4], in);
  return std::max(0.0, t);
}

/* a13 -- Camera::lens_pt_sample_aperture, src/lentil.h:1272-1291; body restated from the
 * polynomial-optics generator (<=5 Newton steps on the aperture position, tolerance 1e-4).
 * PARITY UNPINNED.  in/out are [x,y,dx,dy,lambda]; solves in[2..3]. */
ORC_API void orc_pt_sample_aperture(const OrcLens *L, double in[5], double out[5], double dist) {
  double out_x = out[0], out_y = out[1], out_dx = out[2], out_dy = out[3];
  double x = in[0], y = in[1], dx = in[2], dy = in[3], lambda = in[4];
  double pred_x, pred_y, pred_dx = 0, pred_dy = 0;
  double sqr_err = 3.4028234663852886e38;
  for (int k = 0; k < 5 && sqr_err > 1e-4; k++) {
    const double begin[5] = {x + dist * dx, y + dist * dy, dx, dy, lambda};
    pred_x = eval_poly(L->ap[0], begin);
    pred_y = eval_poly(L->ap[1], begin);
    pred_dx = eval_poly(L->ap[2], begin);
    pred_dy = eval_poly(L->ap[3], begin);
    /* the Jacobian wrt (dx,dy) includes the dist*d/dx chain term of the shifted start point */
    double J[2][2], inv[2][2];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) {
        OrcPoly dpos = derive(L->ap[i], j);
        J[i][j] = eval_poly(L->dap[i][j], begin) + dist * eval_poly(dpos, begin);
      }
    orc_inv2x2(J, inv, nullptr);
    const double dx1[2] = {out_x - pred_x, out_y - pred_y};
    for (int i = 0; i < 2; i++) {
      dx += inv[0][i] * dx1[i];
      dy += inv[1][i] * dx1[i];
    }
    sqr_err = dx1[0] * dx1[0] + dx1[1] * dx1[1];
  }
  out_dx = pred_dx;
  out_dy = pred_dy;
  out[0] = out_x; out[1] = out_y; out[2] = out_dx; out[3] = out_dy;
  in[0] = x; in[1] = y; in[2] = dx; in[3] = dy;
}

/* =====================================================================================
 * a7/a8 -- imageData, src/imagebokeh.h:143-412
 * ===================================================================================== */
struct OrcBokeh {
  int x = 0, y = 0;
  std::vector<float> cdfRow, cdfColumn;
  std::vector<int> rowIndices, columnIndices;
};

namespace {
struct arrayCompare {                                           /* src/imagebokeh.h:21-27 */
  const float *values;
  explicit arrayCompare(const float *v) : values(v) {}
  bool operator()(int l, int r) const { return values[l] > values[r]; }
};
}

/* bokehProbability, src/imagebokeh.h:143-338; pixelData = x*y*nchannels floats */
ORC_API OrcBokeh *orc_bokeh_create(const float *pixelData, int x, int y, int nchannels) {
  if (!(x * y * nchannels > 0 && nchannels >= 3) || x != y) return nullptr;
  OrcBokeh *B = new OrcBokeh();
  B->x = x; B->y = y;
  const int npixels = x * y;
  const int o1 = (nchannels >= 2 ? 1 : 0);
  const int o2 = (nchannels >= 3 ? 2 : o1);
  std::vector<float> pixelValues(npixels), normalized(npixels), summedRow(y), perRow(npixels);
  float totalValue = 0.0f;
  for (int i = 0, j = 0; i < npixels; ++i, j += nchannels) {
    pixelValues[i] = pixelData[j] * 0.3f + pixelData[j + o1] * 0.59f + pixelData[j + o2] * 0.11f;
    totalValue += pixelValues[i];
  }
  float invTotalValue = 1.0f / totalValue;
  for (int i = 0; i < npixels; ++i) normalized[i] = pixelValues[i] * invTotalValue;
  for (int i = 0, k = 0; i < y; ++i) {
    summedRow[i] = 0.0f;
    for (int j = 0; j < x; ++j, ++k) summedRow[i] += normalized[k];
  }
  B->rowIndices.resize(y);
  for (int i = 0; i < y; ++i) B->rowIndices[i] = i;
  std::sort(B->rowIndices.begin(), B->rowIndices.end(), arrayCompare(summedRow.data()));
  B->cdfRow.resize(y);
  float prevVal = 0.0f;
  for (int i = 0; i < y; ++i) {
    B->cdfRow[i] = prevVal + summedRow[B->rowIndices[i]];
    prevVal = B->cdfRow[i];
  }
  for (int r = 0, i = 0; r < y; ++r)
    for (int c = 0; c < x; ++c, ++i) {
      if ((normalized[i] != 0) && (summedRow[r] != 0)) perRow[i] = normalized[i] / summedRow[r];
      else perRow[i] = 0;
    }
  B->columnIndices.resize(npixels);
  for (int i = 0; i < npixels; i++) B->columnIndices[i] = i;
  for (int i = 0; i < npixels; i += x)
    std::sort(B->columnIndices.begin() + i, B->columnIndices.begin() + i + x, arrayCompare(perRow.data()));
  B->cdfColumn.resize(npixels);
  for (int r = 0, i = 0; r < y; ++r) {
    prevVal = 0.0f;
    for (int c = 0; c < x; ++c, ++i) {
      B->cdfColumn[i] = prevVal + perRow[B->columnIndices[i]];
      prevVal = B->cdfColumn[i];
    }
  }
  return B;
}
ORC_API OrcBokeh *orc_bokeh_from_tables(const lentil_bokeh_table *t) {
  OrcBokeh *B = new OrcBokeh();
  B->x = t->x; B->y = t->y;
  B->cdfRow.assign(t->cdfRow, t->cdfRow + t->y);
  B->rowIndices.assign(t->rowIndices, t->rowIndices + t->y);
  B->cdfColumn.assign(t->cdfColumn, t->cdfColumn + (size_t)t->x * t->y);
  B->columnIndices.assign(t->columnIndices, t->columnIndices + (size_t)t->x * t->y);
  return B;
}
ORC_API void orc_bokeh_destroy(OrcBokeh *B) { delete B; }
ORC_API void orc_bokeh_tables(const OrcBokeh *B, float *cdfRow, int *rowIndices, float *cdfColumn,
                              int *columnIndices) {
  memcpy(cdfRow, B->cdfRow.data(), sizeof(float) * B->y);
  memcpy(rowIndices, B->rowIndices.data(), sizeof(int) * B->y);
  memcpy(cdfColumn, B->cdfColumn.data(), sizeof(float) * B->x * B->y);
  memcpy(columnIndices, B->columnIndices.data(), sizeof(int) * B->x * B->y);
}

/* bokehSample, src/imagebokeh.h:341-412 (stratification inputs unused, :407-411) */
ORC_API void orc_bokeh_sample(const OrcBokeh *B, float randomNumberRow, float randomNumberColumn,
                              double lens[2]) {
  const int x = B->x, y = B->y;
  const float *cdfRow = B->cdfRow.data();
  const float *cdfColumn = B->cdfColumn.data();
  const float *pUpperBound = std::upper_bound(cdfRow, cdfRow + y, randomNumberRow);
  int r = 0;
  pUpperBound >= (cdfRow + y) ? r = y - 1 : r = static_cast<int>(pUpperBound - cdfRow);
  int actualPixelRow = B->rowIndices[r];
  int recalulatedPixelRow = actualPixelRow - ((x - 1) / 2);
  int startPixel = actualPixelRow * x;
  const float *pUpperBoundColumn =
      std::upper_bound(cdfColumn + startPixel, cdfColumn + startPixel + x, randomNumberColumn);
  int c = 0;
  pUpperBoundColumn >= cdfColumn + startPixel + x ? c = startPixel + x - 1
                                                  : c = static_cast<int>(pUpperBoundColumn - cdfColumn);
  int actualPixelColumn = B->columnIndices[c];
  int relativePixelColumn = actualPixelColumn - startPixel;
  int recalulatedPixelColumn = relativePixelColumn - ((y - 1) / 2);
  float flippedRow = static_cast<float>(recalulatedPixelColumn);
  float flippedColumn = recalulatedPixelRow * -1.0f;
  lens[0] = static_cast<float>(flippedRow) / static_cast<float>(x) * 2.0;
  lens[1] = static_cast<float>(flippedColumn) / static_cast<float>(y) * 2.0;
}

/* =====================================================================================
 * a11 -- aperture draw + Camera::trace_ray_bw_po, src/lentil.h:573-661
 * GCC evaluates the rng(seed) arguments right to left:
 *   bokehSample(rng#4, rng#3, disk, rng#2, rng#1);  concentric_disk_sample(rng#2, rng#1, ...)
 *   lens_sample_triangular_aperture(.., r1 = rng#2, r2 = rng#1, ...)
 * ===================================================================================== */
ORC_API void orc_po_aperture_sample(const lentil_params *P, const OrcBokeh *B, uint32_t seed_a,
                                    uint32_t seed_b, double aperture[2]) {
  if (!P->enable_dof) { aperture[0] = aperture[1] = 0.0; return; }
  uint32_t seed = orc_tea8(seed_a, seed_b);
  if (P->bokeh_aperture_blades <= 2) {
    double unit_disk[2] = {0.0, 0.0};
    if (P->bokeh_enable_image) {
      const float d1 = orc_rng(&seed), d2 = orc_rng(&seed), d3 = orc_rng(&seed), d4 = orc_rng(&seed);
      (void)d1; (void)d2;
      orc_bokeh_sample(B, d4, d3, unit_disk);
    } else {
      const float d1 = orc_rng(&seed), d2 = orc_rng(&seed);
      orc_concentric_disk_sample(d2, d1, unit_disk);
    }
    aperture[0] = unit_disk[0] * P->aperture_radius;
    aperture[1] = unit_disk[1] * P->aperture_radius;
  } else {
    const float d1 = orc_rng(&seed), d2 = orc_rng(&seed);
    orc_triangular_aperture(&aperture[0], &aperture[1], d2, d1, P->aperture_radius,
                            P->bokeh_aperture_blades);
  }
}

/* the inverse of a 4x4 matrix by cofactors, in fp64, rounded to float at the end (what stands in for AiCameraToWorldMatrix where the
 * caller gives none: the SDK's own inverse is not in the reference tree; the HIP library computes the same, lentil_hip.hip) */
static void invert4x4(const float m_[16], float out[16]) {
  double m[16], inv[16];
  for (int i = 0; i < 16; i++) m[i] = m_[i];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  det = det != 0.0 ? 1.0 / det : 0.0;
  for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] * det);
}

/* one try's probe, src/lentil.h:613-629 (polynomial optics: lens_point = (-aperture * 0.1, 0)) and src/lentil_filter.cpp:356-375 (thin
 * lens: lens_point = lens): the point in world space -- AtVector /= float multiplies by 1.0f / f (SDK, recalled) -- and the
 * renderer's answer */
struct OrcProbeCtx {
  lentil_probe_fn fn;
  void *user;
  const float (*cam_to_world)[4];
  const float *sample_pos_ws;
  bool sample_is_from_skydome;
  int unit_model;
};
static inline void probe_target(const float c2w[4][4], int unit_model, const float lens_point[3], float out[3]) {
  float v[3] = {lens_point[0], lens_point[1], lens_point[2]};
  float div = 1.0f;
  switch (unit_model) {
    case LENTIL_UNIT_MM: div = 0.1f; break;
    case LENTIL_UNIT_CM: div = 1.0f; break;
    case LENTIL_UNIT_DM: div = 10.0f; break;
    case LENTIL_UNIT_M: div = 100.0f; break;
  }
  const float c = 1.0f / div;
  v[0] *= c; v[1] *= c; v[2] *= c;
  out[0] = v[0] * c2w[0][0] + v[1] * c2w[1][0] + v[2] * c2w[2][0] + c2w[3][0];      /* AiM4PointByMatrixMult, as m4_point below */
  out[1] = v[0] * c2w[0][1] + v[1] * c2w[1][1] + v[2] * c2w[2][1] + c2w[3][1];
  out[2] = v[0] * c2w[0][2] + v[1] * c2w[1][2] + v[2] * c2w[2][2] + c2w[3][2];
}
static inline bool probe_occluded(const OrcProbeCtx *pc, const float lens_point[3]) {
  if (!pc || !pc->fn || pc->sample_is_from_skydome) return false;
  lentil_probe_segment s;
  for (int i = 0; i < 3; i++) s.origin[i] = pc->sample_pos_ws[i];
  probe_target(pc->cam_to_world, pc->unit_model, lens_point, s.target);
  uint8_t occ = 0;
  pc->fn(pc->user, 1, &s, &occ);
  return occ != 0;
}

static int trace_ray_bw_po_probed(const lentil_params *P, const OrcLens *L, const OrcBokeh *B,
                                  const double target[3], double sensor_position[2], int px, int py,
                                  int total_samples_taken, float lambda_in, int *tries_out, const OrcProbeCtx *pc);
ORC_API int orc_trace_ray_bw_po(const lentil_params *P, const OrcLens *L, const OrcBokeh *B,
                                const double target[3], double sensor_position[2], int px, int py,
                                int total_samples_taken, float lambda_in, int *tries_out) {
  return trace_ray_bw_po_probed(P, L, B, target, sensor_position, px, py, total_samples_taken, lambda_in, tries_out, nullptr);
}
static int trace_ray_bw_po_probed(const lentil_params *P, const OrcLens *L, const OrcBokeh *B,
                                  const double target[3], double sensor_position[2], int px, int py,
                                  int total_samples_taken, float lambda_in, int *tries_out, const OrcProbeCtx *pc) {
  int tries = 0;
  bool ray_succes = false;
  double sensor[5] = {0, 0, 0, 0, lambda_in};
  double out[5] = {0, 0, 0, 0, lambda_in};
  double aperture[2] = {0, 0};
  while (ray_succes == false && tries <= P->vignetting_retries) {
    if (!P->enable_dof) aperture[0] = aperture[1] = 0.0;
    else orc_po_aperture_sample(P, B, (uint32_t)(px * py + px), (uint32_t)(total_samples_taken + tries), aperture);
    /* raytrace for scene/geometrical occlusions along the ray, src/lentil.h:613-629 (no probe set: AiTraceProbe == false) */
    if (pc) {
      const float lens_point[3] = {(float)(-aperture[0] * 0.1), (float)(-aperture[1] * 0.1), 0.0f};
      if (probe_occluded(pc, lens_point)) { ++tries; continue; }
    }
    sensor[0] = sensor[1] = 0.0;
    float transmittance = orc_lt_sample_aperture(L, target, aperture, sensor, out, lambda_in, nullptr);
    if (transmittance <= 0) { ++tries; continue; }
    const double ipx = sensor[0] + sensor[2] * L->k.lens_back_focal_length;
    const double ipy = sensor[1] + sensor[3] * L->k.lens_back_focal_length;
    if (ipx * ipx + ipy * ipy > L->k.lens_inner_pupil_radius * L->k.lens_inner_pupil_radius) { ++tries; continue; }
    ray_succes = true;
  }
  if (tries_out) *tries_out = tries;
  if (!ray_succes) return 0;
  sensor[0] += sensor[2] * -P->sensor_shift;
  sensor[1] += sensor[3] * -P->sensor_shift;
  sensor_position[0] = sensor[0];
  sensor_position[1] = sensor[1];
  return 1;
}

/* =====================================================================================
 * a3/a4/a5 -- scalar helpers of the visit prologue
 * ===================================================================================== */
ORC_API float orc_get_coc_thinlens(const lentil_params *P, float z_cs) {   /* src/lentil.h:674-692 */
  float _focus_distance = P->focus_distance;
  float _aperture_radius = P->aperture_radius;
  if (P->cameraType == LENTIL_POLYNOMIAL_OPTICS) _focus_distance /= 10.0;
  else _aperture_radius *= 10.0;
  const float focal_length = P->focal_length;
  const float image_dist_samplepos = (-focal_length * z_cs) / (-focal_length + z_cs);
  const float image_dist_focusdist = (-focal_length * -_focus_distance) / (-focal_length + -_focus_distance);
  return std::abs((_aperture_radius * (image_dist_samplepos - image_dist_focusdist)) / image_dist_samplepos);
}

ORC_API float orc_additional_luminance_soft_trans(const lentil_params *P, float sample_luminance) { /* lentil.h:1128-1138 */
  const double lo = P->bidir_add_energy_minimum_luminance;
  const float tr = P->bidir_add_energy_transition;
  if (sample_luminance > lo && sample_luminance < lo + tr) {
    float perc = (sample_luminance - lo) / tr;
    return P->bidir_add_energy * perc;
  } else if (sample_luminance > lo + tr) {
    return P->bidir_add_energy;
  }
  return 0.0;
}

/* draw count, src/lentil_filter.cpp:177-202 */
ORC_API int orc_draw_count(const lentil_params *P, float sample_luminance, float circle_of_confusion,
                           float inverse_sample_density) {
  float luminance_mult = std::max(0.0, std::pow(std::min(sample_luminance, 20.0f), 0.5) * P->bidir_sample_mult);
  const float coc_squared_pixels = std::pow(circle_of_confusion * P->yres, 2) * std::pow(luminance_mult, 2) * 0.00001;
  int samples = std::ceil(coc_squared_pixels * inverse_sample_density);
  samples = clampf(samples, 4, 2000);
  if (P->samples_override > 0) samples = P->samples_override;     /* bench extension, not in the reference */
  return samples;
}

/* AiM4PointByMatrixMult (SDK, recalled): row-vector convention */
static inline void m4_point(const float m[4][4], const float p[3], float o[3]) {
  o[0] = p[0] * m[0][0] + p[1] * m[1][0] + p[2] * m[2][0] + m[3][0];
  o[1] = p[0] * m[0][1] + p[1] * m[1][1] + p[2] * m[2][1] + m[3][1];
  o[2] = p[0] * m[0][2] + p[1] * m[1][2] + p[2] * m[2][2] + m[3][2];
}
/* AiV3Normalize / AiV3Length (SDK, recalled) */
static inline float v3len(const float v[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static inline void v3norm(const float v[3], float o[3]) {
  float t = v3len(v);
  if (t != 0) t = 1 / t;
  o[0] = v[0] * t; o[1] = v[1] * t; o[2] = v[2] * t;
}

/* -------------------------------------------------------------------------------------
 * Thin-lens coma, src/lens.h:563-582.  The rotation goes through Eigen (AngleAxisd ->
 * Matrix3d, Matrix3d::inverse(), Matrix3d * Vector3d).  Eigen is not part of the reference
 * tree (it is included by relative path from a sibling checkout, src/lens.h:5-6, no version
 * pinned), so the three Eigen operations are restated from Eigen's published sources
 * (3.3/3.4, Geometry/AngleAxis.h toRotationMatrix, LU/InverseImpl.h compute_inverse<.,.,3>,
 * the coefficient-based product with its unrolled 3-term reduction a0 + (a1 + a2)):
 * parity unpinned for this branch beyond that restatement.
 * ------------------------------------------------------------------------------------- */
static inline float v3dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

ORC_API float orc_abb_coma_multipliers(float sensor_width, float focal_length, const float dir_from_center[3],
                                       const double unit_disk[2]) {                 /* src/lens.h:563-571 */
  const float maximal_perturbed_ray[3] = {(float)(1.0 * (sensor_width * 0.5)), (float)(1.0 * (sensor_width * 0.5)),
                                          -focal_length};
  const float minus_z[3] = {0.0f, 0.0f, -1.0f};
  float n[3];
  v3norm(maximal_perturbed_ray, n);
  float maximal_projection = v3dot(n, minus_z);
  float current_projection = v3dot(dir_from_center, minus_z);
  float projection_perc = ((current_projection - maximal_projection) / (1.0 - maximal_projection) - 0.5) * 2.0;
  float dist_from_sensor_center = 1.0 - projection_perc;
  float dist_from_aperture = std::sqrt(unit_disk[0] * unit_disk[0] + unit_disk[1] * unit_disk[1]);  /* Vector2d::norm */
  return dist_from_sensor_center * dist_from_aperture;
}

/* Eigen::AngleAxisd(angle, axis).toRotationMatrix() */
static inline void angle_axis_matrix(double angle, const double axis[3], double res[3][3]) {
  const double s = std::sin(angle), c = std::cos(angle);
  const double sin_axis[3] = {s * axis[0], s * axis[1], s * axis[2]};
  const double cos1_axis[3] = {(1.0 - c) * axis[0], (1.0 - c) * axis[1], (1.0 - c) * axis[2]};
  double tmp;
  tmp = cos1_axis[0] * axis[1];
  res[0][1] = tmp - sin_axis[2];
  res[1][0] = tmp + sin_axis[2];
  tmp = cos1_axis[0] * axis[2];
  res[0][2] = tmp + sin_axis[1];
  res[2][0] = tmp - sin_axis[1];
  tmp = cos1_axis[1] * axis[2];
  res[1][2] = tmp - sin_axis[0];
  res[2][1] = tmp + sin_axis[0];
  for (int i = 0; i < 3; i++) res[i][i] = cos1_axis[i] * axis[i] + c;
}
/* Eigen::Matrix3d::inverse(): cofactors, det from column 0, everything scaled by 1/det */
static inline double cofactor3(const double m[3][3], int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
}
static inline void inverse3(const double m[3][3], double r[3][3]) {
  const double c0[3] = {cofactor3(m, 0, 0), cofactor3(m, 1, 0), cofactor3(m, 2, 0)};
  const double det = c0[0] * m[0][0] + (c0[1] * m[1][0] + c0[2] * m[2][0]);
  const double invdet = 1.0 / det;
  r[1][0] = cofactor3(m, 0, 1) * invdet;
  r[1][1] = cofactor3(m, 1, 1) * invdet;
  r[2][0] = cofactor3(m, 0, 2) * invdet;
  r[1][2] = cofactor3(m, 2, 1) * invdet;
  r[2][1] = cofactor3(m, 1, 2) * invdet;
  r[2][2] = cofactor3(m, 2, 2) * invdet;
  r[0][0] = c0[0] * invdet; r[0][1] = c0[1] * invdet; r[0][2] = c0[2] * invdet;
}

ORC_API void orc_abb_coma_perturb(const float dir_from_lens[3], const float ray_to_perturb[3], float abb_coma,
                                  int reverse, float out[3]) {                      /* src/lens.h:575-582 */
  const float minus_z[3] = {0.0f, 0.0f, -1.0f};
  const float cr[3] = {dir_from_lens[1] * minus_z[2] - dir_from_lens[2] * minus_z[1],
                       dir_from_lens[2] * minus_z[0] - dir_from_lens[0] * minus_z[2],
                       dir_from_lens[0] * minus_z[1] - dir_from_lens[1] * minus_z[0]};     /* AiV3Cross */
  float axis_tmp[3];
  v3norm(cr, axis_tmp);
  const double axis[3] = {axis_tmp[0], axis_tmp[1], axis_tmp[2]};
  const double angle = (abb_coma * 2.3456 * AI_PI_F) / 180.0;
  double rot[3][3], inv[3][3];
  angle_axis_matrix(angle, axis, rot);
  const double (*m)[3] = rot;
  if (reverse) { inverse3(rot, inv); m = inv; }
  const double raydir[3] = {ray_to_perturb[0], ray_to_perturb[1], ray_to_perturb[2]};
  for (int i = 0; i < 3; i++) out[i] = (float)(m[i][0] * raydir[0] + (m[i][1] * raydir[1] + m[i][2] * raydir[2]));
}

/* =====================================================================================
 * The frame: buffers of Camera::setup_filter (src/lentil.h:1096-1121) and the whole
 * filter_pixel visit loop (src/lentil_filter.cpp:91-451) over a visit stream.
 * ===================================================================================== */
/* orc_redistribute_threads (below): what a worker thread keeps for itself while the frame's buffers are shared -- its counters,
 * its part of the draw log, and the accepted draws themselves, which are NOT added by the worker (another thread's rows) but
 * kept as records and added, thread after thread = in visit order, when all workers are through. */
struct OrcDeferredDraw {
  uint32_t pixel, visit;
  float add_energy, weight;
  uint32_t chan;                  /* rgb_weight: 0 white; 1, 2, 3: three-fold into r, g, b alone (abb_chromatic > 0) */
};
struct OrcWorker {
  lentil_counters ctr;
  std::vector<lentil_draw_record> log;
  std::vector<OrcDeferredDraw> draws;
  uint32_t cur_visit = 0;
};
static thread_local OrcWorker *tl_worker = nullptr;

struct OrcFrame {
  uint32_t xres, yres, n_aovs;
  uint8_t kind[LENTIL_MAX_AOVS];
  std::vector<float> buffer[LENTIL_MAX_AOVS];   /* AOVData::buffer, RGBA */
  std::vector<float> weight;                    /* filter_weight_buffer */
  std::vector<float> zbuffer;
  std::vector<float> zbuffer_debug;             /* lentil_debug's own z-buffer, src/lentil.h:101 */
  std::vector<uint32_t> zvisit;                 /* visit that wrote zbuffer last (multi-rank merge tests) */
  uint32_t cur_visit = 0;
  std::vector<double> buffer64[LENTIL_MAX_AOVS];/* fp64 shadow accumulation (tolerance studies) */
  std::vector<double> weight64;
  lentil_counters ctr;
  std::vector<lentil_draw_record> log;
  bool keep_log = false;
  bool shadow = true;             /* keep the fp64 shadow accumulators */
  /* xor128's state (src/global.h:22-27 keeps it in function statics, one per process): thin-lens abb_chromatic > 0
   * draws every attempt's colour channel from it (src/lentil_filter.cpp:397).  Explicit here, one per frame object,
   * advanced in the order this oracle walks the visits -- the single-threaded order. */
  uint32_t xor_state[4] = {123456789u, 362436069u, 521288629u, 88675123u};
  /* cryptomatte AOVs: AOVData::crypto_hash_map / crypto_total_weight, src/aov_data.h:127-128, and the visits' caches
   * (what cryptomatte_construct_cache leaves per visit) as `entries` (id, weight) pairs per visit */
  struct Crypto {
    std::vector<std::map<float, float>> hash_map;
    std::vector<float> total_weight;
    const float *hash = nullptr, *weight = nullptr;
  };
  std::vector<Crypto> crypto;
  uint32_t crypto_entries = 0;
  /* A moving camera: AiWorldToCameraMatrix(camera, time) (src/lentil_filter.cpp:141-144) as the linear interpolation of
   * n_cam_keys matrices at equidistant shutter-relative times 0 ... 1 (orc_frame_set_camera_motion).  How Arnold itself
   * interpolates a camera's matrix keys is the SDK's business and not in the reference tree: parity unpinned; the
   * component-wise AiLerp form ((b - a) * t) + a is what this oracle and the HIP path share. */
  std::vector<float> cam_keys;
  uint32_t n_cam_keys = 0;
  /* the keys span the camera's shutter [cam_t0, t1] (orc_frame_set_camera_shutter; 0 ... 1 unless set): lentil_time is Arnold's
   * absolute sample time (src/lentil_filter.cpp:141-143 hands it to AiWorldToCameraMatrix as it is) */
  float cam_t0 = 0.0f, cam_inv_dt = 1.0f;
  /* AiTraceProbe (src/lentil.h:613-629, src/lentil_filter.cpp:356-375): the renderer's answer to "is anything between the sample
   * and this point of the aperture", as a callback (orc_frame_set_probe; include/lentil_hip.h: lentil_probe_fn) called one
   * segment at a time where the reference calls AiTraceProbe; null: nothing ever is (every fixture before round 6).
   * cam_to_world: AiCameraToWorldMatrix -- given, or the inverse of world_to_camera (of each motion key) in fp64. */
  lentil_probe_fn probe = nullptr;
  void *probe_user = nullptr;
  bool have_c2w = false;
  float cam_to_world[4][4];
  std::vector<float> c2w_keys;
};

/* the counters / draw log / current visit of whoever runs do_visit: the frame's own, or the worker thread's */
static inline lentil_counters &ctr_of(OrcFrame *F) { return tl_worker ? tl_worker->ctr : F->ctr; }
static inline std::vector<lentil_draw_record> &log_of(OrcFrame *F) { return tl_worker ? tl_worker->log : F->log; }
static inline uint32_t &cur_visit_of(OrcFrame *F) { return tl_worker ? tl_worker->cur_visit : F->cur_visit; }

ORC_API OrcFrame *orc_frame_create(uint32_t xres, uint32_t yres, uint32_t n_aovs, const uint8_t *kind,
                                   int keep_log) {
  OrcFrame *F = new OrcFrame();
  F->xres = xres; F->yres = yres; F->n_aovs = n_aovs;
  const size_t np = (size_t)xres * yres;
  F->keep_log = (keep_log & 1) != 0;
  F->shadow = (keep_log & 2) == 0;      /* flag bit 1: timing runs skip the fp64 shadows */
  for (uint32_t a = 0; a < n_aovs; a++) {
    F->kind[a] = kind ? kind[a] : LENTIL_FILTER_GAUSSIAN;
    F->buffer[a].assign(np * 4, 0.0f);
    if (F->shadow) F->buffer64[a].assign(np * 4, 0.0);
  }
  F->weight.assign(np, 0.0f);
  if (F->shadow) F->weight64.assign(np, 0.0);
  F->zbuffer.assign(np, 0.0f);
  F->zbuffer_debug.assign(np, 0.0f);
  F->zvisit.assign(np, 0xFFFFFFFFu);
  memset(&F->ctr, 0, sizeof(F->ctr));
  return F;
}
ORC_API void orc_frame_destroy(OrcFrame *F) { delete F; }
ORC_API const float *orc_frame_buffer(const OrcFrame *F, uint32_t aov) { return F->buffer[aov].data(); }
ORC_API const float *orc_frame_weight(const OrcFrame *F) { return F->weight.data(); }
ORC_API const float *orc_frame_zbuffer(const OrcFrame *F) { return F->zbuffer.data(); }
ORC_API const uint32_t *orc_frame_zvisit(const OrcFrame *F) { return F->zvisit.data(); }
ORC_API const double *orc_frame_buffer64(const OrcFrame *F, uint32_t aov) { return F->buffer64[aov].data(); }
ORC_API const double *orc_frame_weight64(const OrcFrame *F) { return F->weight64.data(); }
ORC_API void orc_frame_counters(const OrcFrame *F, lentil_counters *c) { *c = F->ctr; }
ORC_API void orc_frame_set_xor128(OrcFrame *F, const uint32_t st[4]) { memcpy(F->xor_state, st, sizeof F->xor_state); }
ORC_API void orc_frame_get_xor128(const OrcFrame *F, uint32_t st[4]) { memcpy(st, F->xor_state, sizeof F->xor_state); }
ORC_API uint64_t orc_frame_log(const OrcFrame *F, lentil_draw_record *out, uint64_t cap) {
  uint64_t n = std::min<uint64_t>(cap, F->log.size());
  if (out && n) memcpy(out, F->log.data(), n * sizeof(lentil_draw_record));
  return F->log.size();
}
ORC_API void orc_frame_set_camera_motion(OrcFrame *F, uint32_t n_keys, const float *world_to_camera) {
  F->n_cam_keys = n_keys >= 2 ? n_keys : 0;
  F->cam_keys.assign(world_to_camera, world_to_camera + (size_t)F->n_cam_keys * 16);
}

/* the occlusion probe (OrcFrame::probe): fn null switches it off; camera_to_world 16 floats (n_keys > 1: one matrix per motion key,
 * as many as orc_frame_set_camera_motion got) or null for the inverses of the world-to-camera matrices */
ORC_API void orc_frame_set_probe(OrcFrame *F, lentil_probe_fn fn, void *user, const float *camera_to_world, uint32_t n_keys) {
  F->probe = fn;
  F->probe_user = user;
  F->have_c2w = false;
  F->c2w_keys.clear();
  if (camera_to_world && n_keys <= 1) { memcpy(F->cam_to_world, camera_to_world, sizeof F->cam_to_world); F->have_c2w = true; }
  else if (camera_to_world) F->c2w_keys.assign(camera_to_world, camera_to_world + (size_t)n_keys * 16);
}

/* An analytic occluder for the tests (a lentil_probe_fn): user -> {cx, cy, cz, r} floats; a segment is occluded when it passes
 * through the sphere.  Plain fp32-in, fp64 arithmetic: the same answers whoever calls it, any number of threads. */
ORC_API void orc_sphere_occluder(void *user, uint64_t n, const lentil_probe_segment *seg, uint8_t *occluded) {
  const float *sp = static_cast<const float *>(user);
  for (uint64_t i = 0; i < n; i++) {
    const double o[3] = {seg[i].origin[0], seg[i].origin[1], seg[i].origin[2]};
    const double d[3] = {seg[i].target[0] - o[0], seg[i].target[1] - o[1], seg[i].target[2] - o[2]};
    const double c[3] = {sp[0] - o[0], sp[1] - o[1], sp[2] - o[2]};
    const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    double t = dd > 0.0 ? (c[0] * d[0] + c[1] * d[1] + c[2] * d[2]) / dd : 0.0;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    const double q[3] = {c[0] - t * d[0], c[1] - t * d[1], c[2] - t * d[2]};
    occluded[i] = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) < (double)sp[3] * (double)sp[3] ? 1 : 0;
  }
}

ORC_API void orc_frame_set_camera_shutter(OrcFrame *F, float shutter_start, float shutter_end) {
  F->cam_t0 = shutter_start;
  F->cam_inv_dt = 1.0f / (shutter_end - shutter_start);
}

/* `src` holds the visits that FOLLOW dst's in the stream's order (ThreadedOracle merges its threads' frames in that order).
 * Gaussian AOVs add up.  A closest AOV (add_to_buffer's z-test, src/lentil.h:832-837, one zbuffer for all of them) keeps,
 * over nonzero depths, the LAST candidate of the smallest |Z|: src's survivor replaces dst's when its |Z| <= dst's or dst
 * has none -- what the sequential walk leaves.  (A candidate of |Z| == 0 re-opens the pixel for whatever comes next; a
 * stream with one cannot be split over threads this way, and the callers' generators have none.)  lentil_debug's own
 * z-buffer is not merged: single-threaded frames only. */
ORC_API void orc_frame_merge(OrcFrame *dst, const OrcFrame *src) {
  bool any_closest = false;
  for (uint32_t a = 0; a < dst->n_aovs; a++) {
    if (dst->kind[a] == LENTIL_FILTER_CLOSEST) { any_closest = true; continue; }
    for (size_t i = 0; i < dst->buffer[a].size(); i++) dst->buffer[a][i] += src->buffer[a][i];
  }
  for (size_t i = 0; i < dst->weight.size(); i++) dst->weight[i] += src->weight[i];
  if (dst->shadow && src->shadow) {   /* the exact sums add up exactly where the partial sums do (gaussian AOVs) */
    for (uint32_t a = 0; a < dst->n_aovs; a++) {
      if (dst->kind[a] == LENTIL_FILTER_CLOSEST) continue;
      for (size_t i = 0; i < dst->buffer64[a].size(); i++) dst->buffer64[a][i] += src->buffer64[a][i];
    }
    for (size_t i = 0; i < dst->weight64.size(); i++) dst->weight64[i] += src->weight64[i];
  }
  if (any_closest) {
    const size_t np = dst->zbuffer.size();
    for (size_t px = 0; px < np; px++) {
      if (src->zvisit[px] == 0xFFFFFFFFu) continue;                                   /* src never wrote here */
      if (!(dst->zvisit[px] == 0xFFFFFFFFu || src->zbuffer[px] <= dst->zbuffer[px])) continue;
      for (uint32_t a = 0; a < dst->n_aovs; a++) {
        if (dst->kind[a] != LENTIL_FILTER_CLOSEST) continue;
        for (int c = 0; c < 4; c++) {
          dst->buffer[a][px * 4 + c] = src->buffer[a][px * 4 + c];
          if (dst->shadow && src->shadow) dst->buffer64[a][px * 4 + c] = src->buffer64[a][px * 4 + c];
        }
      }
      dst->zbuffer[px] = src->zbuffer[px];
      dst->zvisit[px] = src->zvisit[px];
    }
  }
  dst->ctr.visits += src->ctr.visits;
  dst->ctr.redistributed_visits += src->ctr.redistributed_visits;
  dst->ctr.attempted_draws += src->ctr.attempted_draws;
  dst->ctr.accepted_draws += src->ctr.accepted_draws;
}

/* ---- a21: the display pass-through of filter_pixel ----------------------------------------
 * Camera::filter_closest_complete, src/lentil.h:696-735, and Camera::filter_gaussian_complete, :738-775, over one
 * pixel's AOV samples handed in as arrays (what the iterator yields, in its order).  aov_type: the SDK's AI_TYPE_*
 * codes (FLOAT 4, RGB 5, RGBA 6, VECTOR 7); value: 4 floats per sample as the typed getter would return them.
 * fast_exp: AiFastExp -- an exported SDK function whose bits are not in the reference tree; the caller supplies the one
 * of the SDK at hand (NULL: expf).  AtRGBA arithmetic as everywhere in this file: `float * AtRGBA` on all four
 * channels, `AtRGBA /= float` as a multiplication by 1.0f / f (SDK semantics recalled; parity unpinned). */
ORC_API void orc_filter_closest_complete(int n, const float *depth, const float *value, int aov_type, float out[4]) {
  float pixel_energy[4] = {0.0f, 0.0f, 0.0f, 0.0f};       /* AI_RGBA_ZERO */
  float z = 0.0;
  for (int i = 0; i < n; i++) {                          /* while (AiAOVSampleIteratorGetNext(iterator)) */
    const float d = depth[i];
    if ((std::abs(d) <= z) || z == 0.0) {
      z = std::abs(d);
      switch (aov_type) {
        case 7: {                                        /* AI_TYPE_VECTOR */
          pixel_energy[0] = value[4 * i]; pixel_energy[1] = value[4 * i + 1]; pixel_energy[2] = value[4 * i + 2]; pixel_energy[3] = 1.0f;
          break;
        }
        case 4: {                                        /* AI_TYPE_FLOAT */
          const float sample_energy = value[4 * i];
          pixel_energy[0] = sample_energy; pixel_energy[1] = sample_energy; pixel_energy[2] = sample_energy; pixel_energy[3] = 1.0f;
          break;
        }
        /* (no other case: an RGB / RGBA AOV leaves pixel_energy alone, :706-728) */
      }
    }
  }
  for (int c = 0; c < 4; c++) out[c] = pixel_energy[c];
}

ORC_API void orc_filter_gaussian_complete(int n, const float *offset_xy, const float *value, const float *sample_inv_density,
                                          int aov_type, float inverse_sample_density, int adaptive_sampling, float filter_width,
                                          float (*fast_exp)(float), float out[4]) {
  float aweight = 0.0f;
  float avalue[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float inv_density = inverse_sample_density;
  for (int i = 0; i < n; i++) {
    if (adaptive_sampling) inv_density = sample_inv_density[i];
    if (inv_density <= 0.f) continue;
    const float ox = offset_xy[2 * i], oy = offset_xy[2 * i + 1];
    const float q = 2 / filter_width;                    /* AiSqr(2 / filter_width): int / float */
    const float r = (q * q) * ((ox * ox) + (oy * oy));
    if (r > 1.0f) continue;
    const float e = fast_exp ? fast_exp(2 * -r) : std::exp(2 * -r);
    const float weight = e * inv_density;
    float sample_energy[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    switch (aov_type) {
      case 6: { for (int c = 0; c < 4; c++) sample_energy[c] = value[4 * i + c]; } break;          /* AI_TYPE_RGBA */
      case 5: { sample_energy[0] = value[4 * i]; sample_energy[1] = value[4 * i + 1]; sample_energy[2] = value[4 * i + 2];
                sample_energy[3] = 1.0f; } break;        /* AI_TYPE_RGB: AtRGB -> AtRGBA, alpha 1 */
    }
    for (int c = 0; c < 4; c++) avalue[c] += weight * sample_energy[c];
    aweight += weight;
  }
  if (aweight != 0.0f) { const float inv = 1.0f / aweight; for (int c = 0; c < 4; c++) avalue[c] *= inv; }
  for (int c = 0; c < 4; c++) out[c] = avalue[c];
}

/* ---- cryptomatte ---------------------------------------------------------------------- */
ORC_API void orc_frame_set_crypto(OrcFrame *F, uint32_t n_crypto, uint32_t entries, const float *const *hash,
                                  const float *const *weight) {
  F->crypto.assign(n_crypto, OrcFrame::Crypto());
  F->crypto_entries = entries;
  const size_t np = (size_t)F->xres * F->yres;
  for (uint32_t c = 0; c < n_crypto; c++) {
    F->crypto[c].hash_map.assign(np, std::map<float, float>());      /* allocate_cryptomatte_buffers, aov_data.h:145-150 */
    F->crypto[c].total_weight.assign(np, 0.0f);
    F->crypto[c].hash = hash[c];
    F->crypto[c].weight = weight[c];
  }
}

/* Camera::cryptomatte_construct_cache, src/lentil.h:781-811, for one cryptomatte AOV: the depth samples of one AOV
 * sample (opacity RGB and the AOV's float per depth) folded into id -> weight.  AiColorToGrey is the SDK's
 * (r + g + b) / 3 (recalled).  Returns the number of pairs (ids ascending, the map's order); -1 when cap is short. */
ORC_API int orc_crypto_construct_cache(int n_depth, const float *opacity_rgb, const float *value, float *ids,
                                       float *weights, int cap) {
  std::map<float, float> cache;
  float iterative_transparency_weight = 1.0f;
  float quota = 1.0;
  float sample_value = 0.0f;
  for (int d = 0; d < n_depth; d++) {                                 /* while (AiAOVSampleIteratorGetNextDepth) */
    const float sub_sample_opacity = (opacity_rgb[d * 3] + opacity_rgb[d * 3 + 1] + opacity_rgb[d * 3 + 2]) / 3;
    sample_value = value[d];
    const float sub_sample_weight = sub_sample_opacity * iterative_transparency_weight;
    iterative_transparency_weight *= (1.0f - sub_sample_opacity);
    quota -= sub_sample_weight;
    cache[sample_value] += sub_sample_weight;
  }
  if (quota > 0.0) cache[sample_value] += quota;                      /* :804 */
  if ((int)cache.size() > cap) return -1;
  int n = 0;
  for (auto const &e : cache) { ids[n] = e.first; weights[n] = e.second; n++; }
  return n;
}

/* the cache of visit v as the filter holds it while it adds the visit (src/lentil_filter.cpp:168-169) */
static inline void crypto_cache_of(const OrcFrame *F, uint32_t c, uint64_t v, std::map<float, float> &cache) {
  const OrcFrame::Crypto &K = F->crypto[c];
  for (uint32_t e = 0; e < F->crypto_entries; e++) {
    const float w = K.weight[v * F->crypto_entries + e];
    uint32_t bits; memcpy(&bits, &w, 4);
    if (bits == 0xFFFFFFFFu) continue;
    cache[K.hash[v * F->crypto_entries + e]] += w;
  }
}

/* Camera::add_to_buffer_cryptomatte, src/lentil.h:814-819 */
static inline void add_to_buffer_cryptomatte(OrcFrame::Crypto &aov, uint32_t px, const std::map<float, float> &cryptomatte_cache,
                                             const float sample_weight) {
  aov.total_weight[px] += sample_weight;
  for (auto const &sample : cryptomatte_cache) aov.hash_map[px][sample.first] += sample.second * sample_weight;
}

struct compareTail {                                                   /* src/lentil_imager.cpp:11-16 */
  bool operator()(const std::pair<float, float> x, const std::pair<float, float> y) { return x.second > y.second; }
};

/* the cryptomatte branch of driver_process_bucket, src/lentil_imager.cpp:121-161, for every pixel: out np*4,
 * has np (0 where the reference leaves the bucket row, :132-134; out is not written there) */
ORC_API void orc_crypto_rank(const OrcFrame *F, uint32_t c, int rank, float *out, uint8_t *has) {
  const size_t np = (size_t)F->xres * F->yres;
  const OrcFrame::Crypto &K = F->crypto[c];
  for (size_t p = 0; p < np; p++) {
    has[p] = 0;
    if ((int)K.hash_map[p].size() <= rank) continue;
    has[p] = 1;
    std::vector<std::pair<float, float>> all_vals;
    all_vals.reserve(K.hash_map[p].size());
    for (auto it = K.hash_map[p].begin(); it != K.hash_map[p].end(); ++it) all_vals.push_back(*it);
    std::sort(all_vals.begin(), all_vals.end(), compareTail());
    float o[4] = {0, 0, 0, 0};
    int iter = 0;
    for (auto it = all_vals.begin(); it != all_vals.end(); ++it) {
      if (iter == rank) { o[0] = it->first; o[1] = (it->second / K.total_weight[p]); }
      else if (iter == rank + 1) { o[2] = it->first; o[3] = (it->second / K.total_weight[p]); }
      iter++;
    }
    memcpy(out + p * 4, o, sizeof o);
  }
}

/* one pixel's map (ids ascending) and total weight; returns the number of entries */
ORC_API int orc_crypto_pixel(const OrcFrame *F, uint32_t c, uint64_t p, float *ids, float *weights, int cap, float *total) {
  const OrcFrame::Crypto &K = F->crypto[c];
  int n = 0;
  for (auto const &e : K.hash_map[p]) { if (n < cap) { ids[n] = e.first; weights[n] = e.second; } n++; }
  if (total) *total = K.total_weight[p];
  return n;
}

/* Camera::add_to_buffer, src/lentil.h:823-851 (rgb_weight is white on this path) */
static const float kWhite[3] = {1.0f, 1.0f, 1.0f};
/* rgb_weight: AtRGBA * AtRGB goes through AtRGBA(const AtRGB&, float a = 1) (SDK, recalled): alpha x 1 */
static inline void add_to_buffer(OrcFrame *F, uint32_t aov, uint32_t px, const float value[4],
                                 float add_energy, float depth, float filter_weight, const float rgb_weight[3] = kWhite) {
  if (F->kind[aov] == LENTIL_FILTER_GAUSSIAN) {
    if (aov == 0) { F->weight[px] += filter_weight; if (F->shadow) F->weight64[px] += (double)filter_weight; }
    for (int c = 0; c < 4; c++) {
      const float add = (value[c] + add_energy) * filter_weight * (c < 3 ? rgb_weight[c] : 1.0f);
      F->buffer[aov][(size_t)px * 4 + c] += add;
      if (F->shadow) F->buffer64[aov][(size_t)px * 4 + c] += (double)add;
    }
  } else if (F->kind[aov] == LENTIL_FILTER_CLOSEST) {
    if ((std::abs(depth) <= F->zbuffer[px]) || F->zbuffer[px] == 0.0) {
      for (int c = 0; c < 4; c++) {
        F->buffer[aov][(size_t)px * 4 + c] = value[c];
        if (F->shadow) F->buffer64[aov][(size_t)px * 4 + c] = value[c];
      }
      F->zbuffer[px] = std::abs(depth);
      F->zvisit[px] = cur_visit_of(F);
    }
  } else if (F->kind[aov] == LENTIL_FILTER_CLOSEST_DEBUG) {       /* aov.name == lentil_debug, src/lentil.h:838-845 */
    if ((std::abs(depth) <= F->zbuffer_debug[px]) || F->zbuffer_debug[px] == 0.0) {
      if (value[0] != 0.0) {
        for (int c = 0; c < 4; c++) {
          F->buffer[aov][(size_t)px * 4 + c] = value[c];
          if (F->shadow) F->buffer64[aov][(size_t)px * 4 + c] = value[c];
        }
        F->zbuffer_debug[px] = std::abs(depth);
      }
    }
  }
}

static inline float maxrgb(const float *c) { return std::max(std::max(c[0], c[1]), c[2]); }  /* AiColorMaxRGB */

/* One visit of filter_pixel's loop body, src/lentil_filter.cpp:105-448. */
static void do_visit(const lentil_params *P, const OrcLens *L, const OrcBokeh *B, OrcFrame *F,
                     const lentil_visits *V, uint64_t v, int px, int py, float inverse_sample_density) {
  const double xres = (double)P->xres, yres = (double)P->yres;
  const double frame_aspect_ratio_without_region = (double)P->xres_without_region / (double)P->yres_without_region;
  bool redistribute = true;
  cur_visit_of(F) = (uint32_t)v;
  if (P->adaptive_sampling) { if (inverse_sample_density > 0.2) redistribute = false; }   /* :108-113 */

  float sample[4] = {V->rgba[v * 4], V->rgba[v * 4 + 1], V->rgba[v * 4 + 2], V->rgba[v * 4 + 3]};
  float sample_pos_ws[3] = {V->pos_z[v * 4], V->pos_z[v * 4 + 1], V->pos_z[v * 4 + 2]};
  double depth = V->pos_z[v * 4 + 3];
  const float *raydir = &V->raydir_time[v * 4];
  const bool small = std::abs(sample_pos_ws[0]) < AI_EPSILON_F && std::abs(sample_pos_ws[1]) < AI_EPSILON_F &&
                     std::abs(sample_pos_ws[2]) < AI_EPSILON_F;      /* AiV3IsSmall */
  bool sample_is_from_skydome = false;                                 /* :119: exempt from the occlusion probe */
  if ((depth == AI_INFINITE_F || small) && P->enable_skydome) {       /* :122-129 */
    if (raydir[0] == 0 && raydir[1] == 0 && raydir[2] == 0) redistribute = false;
    else for (int i = 0; i < 3; i++) sample_pos_ws[i] = raydir[i] * (float)99999999.0;  /* AtVector * float */
    sample_is_from_skydome = true;
  }
  if ((depth == AI_INFINITE_F || small) && !P->enable_skydome) { redistribute = false; sample_is_from_skydome = true; }   /* :130-133 */
  if (maxrgb(&V->volume_ignore[v * 4]) > 0.0) redistribute = false;                 /* :135-137 */

  float cs[3];
  float c2w[4][4];                                                                  /* :142, AiCameraToWorldMatrix at the sample's time */
  if (F->n_cam_keys >= 2) {                                                         /* :141-143, per-sample camera time */
    float t = (V->raydir_time[v * 4 + 3] - F->cam_t0) * F->cam_inv_dt;
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    const float sc = t * (float)(F->n_cam_keys - 1);
    uint32_t i0 = (uint32_t)sc;
    if (i0 > F->n_cam_keys - 2) i0 = F->n_cam_keys - 2;
    const float f = sc - (float)i0;
    const float *ka = &F->cam_keys[(size_t)i0 * 16], *kb = ka + 16;
    float m[4][4];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) m[r][c] = ((kb[r * 4 + c] - ka[r * 4 + c]) * f) + ka[r * 4 + c];
    m4_point(m, sample_pos_ws, cs);
    if (F->probe && F->c2w_keys.size() == (size_t)F->n_cam_keys * 16) {
      const float *ia = &F->c2w_keys[(size_t)i0 * 16], *ib = ia + 16;
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) c2w[r][c] = ((ib[r * 4 + c] - ia[r * 4 + c]) * f) + ia[r * 4 + c];
    } else if (F->probe) {
      /* no camera-to-world keys given: the inverse of each world-to-camera key, blended like the keys themselves (what the HIP
       * library does with the keys it inverts on the host) */
      float ia[16], ib[16];
      invert4x4(ka, ia); invert4x4(kb, ib);
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) c2w[r][c] = ((ib[r * 4 + c] - ia[r * 4 + c]) * f) + ia[r * 4 + c];
    }
  } else {
    m4_point(P->world_to_camera, sample_pos_ws, cs);                                /* :144 */
    if (F->probe) {
      if (F->have_c2w) memcpy(c2w, F->cam_to_world, sizeof c2w);
      else invert4x4(&P->world_to_camera[0][0], &c2w[0][0]);
    }
  }
  OrcProbeCtx probe_ctx{F->probe, F->probe_user, c2w, sample_pos_ws, sample_is_from_skydome, P->unitModel};
  const OrcProbeCtx *pc = F->probe ? &probe_ctx : nullptr;
  switch (P->unitModel) {                                                           /* :145-150 */
    /* AtVector::operator*=(float): the double literals narrow to float at the call */
    case LENTIL_UNIT_MM: for (int i = 0; i < 3; i++) cs[i] *= 0.1f; break;
    case LENTIL_UNIT_CM: for (int i = 0; i < 3; i++) cs[i] *= 1.0f; break;
    case LENTIL_UNIT_DM: for (int i = 0; i < 3; i++) cs[i] *= 10.0f; break;
    case LENTIL_UNIT_M: for (int i = 0; i < 3; i++) cs[i] *= 100.0f; break;
  }
  const float *tr = &V->transmission[v * 4];
  bool transmitted = P->enable_bidir_transmission ? false : (maxrgb(tr) > 0.0);     /* :152-159 */
  if (transmitted) { sample[0] -= tr[0]; sample[1] -= tr[1]; sample[2] -= tr[2]; redistribute = false; }
  const float sample_luminance = (sample[0] + sample[1] + sample[2]) / 3.0;         /* :161 */
  if (V->volume_ignore[v * 4 + 3] > 0.0) redistribute = false;                      /* :162-164 */

  std::vector<std::map<float, float>> crypto_cache(F->crypto.size());               /* :167-169 */
  for (uint32_t c = 0; c < F->crypto.size(); c++) crypto_cache_of(F, c, v, crypto_cache[c]);

  float fitted_bidir_add_energy = 0.0;                                              /* :173-174 */
  if (P->bidir_add_energy > 0.0) fitted_bidir_add_energy = orc_additional_luminance_soft_trans(P, sample_luminance);

  float circle_of_confusion = orc_get_coc_thinlens(P, cs[2]);                       /* :178 */
  if (circle_of_confusion < 0.4f) redistribute = false;                             /* :183-187 */
  int samples = orc_draw_count(P, sample_luminance, circle_of_confusion, inverse_sample_density);
  float inv_samples = 1.0 / static_cast<float>(samples);                            /* :199 */
  unsigned int total_samples_taken = 0;
  unsigned int max_total_samples = samples * 5;

  /* aov_values, :206-234 -- AOV 0 is RGBA (the possibly transmission-reduced `sample` is NOT what is
   * stored: the reference re-reads the RGBA AOV through AiAOVSampleIteratorGetAOVRGBA, :216) */
  float aov_values[LENTIL_MAX_AOVS][4];
  for (int c = 0; c < 4; c++) aov_values[0][c] = V->rgba[v * 4 + c];
  for (uint32_t a = 1; a < F->n_aovs; a++) {
    if (F->kind[a] == LENTIL_FILTER_CLOSEST_DEBUG) {      /* :209-211: an int assigned to an AtRGBA sets all four */
      const float dbg = (float)(samples * (redistribute ? 1 : 0));
      for (int c = 0; c < 4; c++) aov_values[a][c] = dbg;
      continue;
    }
    for (int c = 0; c < 4; c++) aov_values[a][c] = V->extra[a - 1][v * 4 + c];
  }

  if (P->cameraType == LENTIL_POLYNOMIAL_OPTICS)
    if (std::abs(cs[2]) < (L->k.lens_length * 0.1)) redistribute = false;           /* :240 */

  ctr_of(F).visits++;
  if (!redistribute) {                                                              /* :243-246 / :306-309 */
    const uint32_t pixelnumber = P->xres * py + px;                                 /* lentil.h:945 */
    for (uint32_t a = 0; a < F->n_aovs; a++)
      add_to_buffer(F, a, pixelnumber, aov_values[a], 0.0, depth, 1.0f * inverse_sample_density);
    for (uint32_t c = 0; c < F->crypto.size(); c++)                                 /* lentil.h:952 */
      add_to_buffer_cryptomatte(F->crypto[c], pixelnumber, crypto_cache[c], inverse_sample_density);
    return;
  }
  ctr_of(F).redistributed_visits++;

  if (P->cameraType == LENTIL_POLYNOMIAL_OPTICS) {
    for (int count = 0; count < samples && total_samples_taken < max_total_samples; ++count, ++total_samples_taken) {
      double sensor_position[2] = {0, 0};
      const double target[3] = {-(double)cs[0] * 10.0, -(double)cs[1] * 10.0, -(double)cs[2] * 10.0};  /* :271 */
      ctr_of(F).attempted_draws++;
      float lambda_per_sample = P->lambda_bw;                                                           /* :254, 0.55 */
      for (int channel = -1; channel <= 1; channel++) {                                                 /* :255-268 */
        float rgb_weight[3] = {1.0f, 1.0f, 1.0f};
        if (P->abb_chromatic > 0.0) {
          if (channel == -1) {
            rgb_weight[0] = 3; rgb_weight[1] = 0; rgb_weight[2] = 0;
            lambda_per_sample = lerpf(1.0 - P->abb_chromatic, 0.35, 0.55);
          } else if (channel == 0) {
            rgb_weight[0] = 0; rgb_weight[1] = 3; rgb_weight[2] = 0;
            lambda_per_sample = 0.55;
          } else {
            rgb_weight[0] = 0; rgb_weight[1] = 0; rgb_weight[2] = 3;
            lambda_per_sample = lerpf(P->abb_chromatic, 0.55, 0.85);
          }
        } else if (P->abb_chromatic == 0.0 && channel > -1) continue;
        if (!trace_ray_bw_po_probed(P, L, B, target, sensor_position, px, py, (int)total_samples_taken,
                                    lambda_per_sample, nullptr, pc)) { --count; continue; }
        const double s0 = sensor_position[0] / (P->sensor_width * 0.5);                                 /* :276 */
        const double s1 = sensor_position[1] / (P->sensor_width * 0.5) * frame_aspect_ratio_without_region;
        const double pixel0 = (((s0 + 1.0) / 2.0) * P->xres_without_region) - P->region_min_x;          /* :277-278 */
        const double pixel1 = (((-s1 + 1.0) / 2.0) * P->yres_without_region) - P->region_min_y;
        if ((pixel0 >= xres) || (pixel0 < 0) || (pixel1 >= yres) || (pixel1 < 0) || (pixel0 != pixel0) ||
            (pixel1 != pixel1)) { --count; continue; }                                                  /* :282-287 */
        const int ix = floor(pixel0), iy = floor(pixel1);
        const unsigned pixelnumber = ix + (iy * P->xres);                                               /* :290, lentil.h:958-960 */
        if (tl_worker) {        /* a worker of orc_redistribute_threads: the draw is kept, and added when its turn comes */
          tl_worker->draws.push_back({pixelnumber, (uint32_t)v, fitted_bidir_add_energy, 1.0f * inverse_sample_density * inv_samples,
                                      (uint32_t)(P->abb_chromatic > 0.0 ? channel + 2 : 0)});
        } else
        for (uint32_t a = 0; a < F->n_aovs; a++)                                                        /* :295-298 */
          add_to_buffer(F, a, pixelnumber, aov_values[a], fitted_bidir_add_energy, depth,
                        1.0f * inverse_sample_density * inv_samples, rgb_weight);
        for (uint32_t c = 0; c < F->crypto.size(); c++)                                                 /* :296 */
          add_to_buffer_cryptomatte(F->crypto[c], pixelnumber, crypto_cache[c], inverse_sample_density * inv_samples);
        ctr_of(F).accepted_draws++;
        /* chromatic mode: the channel (0..2) rides in the attempt's top two bits */
        if (F->keep_log) log_of(F).push_back({(uint32_t)v, total_samples_taken | ((uint32_t)(P->abb_chromatic != 0.0 ? channel + 1 : 0) << 30),
                                           pixelnumber});
      }
    }
    return;
  }

  /* ---- ThinLens, src/lentil_filter.cpp:303-447.  abb_chromatic > 0 draws every attempt's channel from xor128
   * (:397): OrcFrame::xor_state, advanced in this function's visit order ---- */
  for (int count = 0; count < samples && total_samples_taken < max_total_samples; ++count, ++total_samples_taken) {
    ctr_of(F).attempted_draws++;
    unsigned int seed = orc_tea8((uint32_t)(px * py + px), total_samples_taken);
    const float focal_length = P->focal_length;
    float image_dist_samplepos = (-focal_length * cs[2]) / (-focal_length + cs[2]);
    double unit_disk[2] = {0, 0};
    if (P->bokeh_enable_image) {
      const float d1 = orc_rng(&seed), d2 = orc_rng(&seed), d3 = orc_rng(&seed), d4 = orc_rng(&seed);
      (void)d1; (void)d2;
      orc_bokeh_sample(B, d4, d3, unit_disk);
    } else if (P->bokeh_aperture_blades < 2) {
      const float d1 = orc_rng(&seed), d2 = orc_rng(&seed);
      orc_concentricDiskSample(d2, d1, unit_disk, P->abb_spherical, P->circle_to_square);
    } else {
      const float d1 = orc_rng(&seed), d2 = orc_rng(&seed);
      orc_triangular_aperture(&unit_disk[0], &unit_disk[1], d2, d1, 1.0, P->bokeh_aperture_blades);
    }
    unit_disk[0] *= P->bokeh_anamorphic;                                                    /* :322 */
    const float lens[3] = {(float)(unit_disk[0] * P->aperture_radius), (float)(unit_disk[1] * P->aperture_radius), 0.0f};
    float dir_from_center[3];
    v3norm(cs, dir_from_center);                                                            /* :327 */
    float dir_lens_to_P[3];
    { const float d[3] = {cs[0] - lens[0], cs[1] - lens[1], cs[2] - lens[2]}; v3norm(d, dir_lens_to_P); }  /* :328 */
    /* coma: the centre ray rotated about the axis orthogonal to the lens ray (:333-334) */
    {
      const float abb_coma_multiplied = P->abb_coma * orc_abb_coma_multipliers(P->sensor_width, P->focal_length,
                                                                               dir_from_center, unit_disk);
      float rotated[3];
      orc_abb_coma_perturb(dir_lens_to_P, dir_from_center, abb_coma_multiplied, 1, rotated);
      dir_lens_to_P[0] = rotated[0]; dir_lens_to_P[1] = rotated[1]; dir_lens_to_P[2] = rotated[2];
    }
    const float len = v3len(cs);
    float perturbed[3] = {len * dir_lens_to_P[0], len * dir_lens_to_P[1], len * dir_lens_to_P[2]};   /* :336 */
    v3norm(perturbed, dir_from_center);                                                     /* :337 */
    float samplepos_image_intersection = std::abs(image_dist_samplepos / dir_from_center[2]);
    float samplepos_image_point[3] = {dir_from_center[0] * samplepos_image_intersection,
                                      dir_from_center[1] * samplepos_image_intersection,
                                      dir_from_center[2] * samplepos_image_intersection};
    float tmp[3] = {samplepos_image_point[0] - lens[0], samplepos_image_point[1] - lens[1], samplepos_image_point[2] - lens[2]};
    float dir_from_lens_to_image_sample[3];
    v3norm(tmp, dir_from_lens_to_image_sample);                                             /* :344 */
    /* raytrace for scene/geometrical occlusions along the ray, :356-375 (no probe set: AiTraceProbe == false) */
    if (pc && probe_occluded(pc, lens)) { --count; continue; }
    if (P->optical_vignetting_distance > 0.0) {                                             /* :379-386, lens.h:529-543 */
      float t2[3] = {perturbed[0] - lens[0], perturbed[1] - lens[1], perturbed[2] - lens[2]};
      v3norm(t2, dir_lens_to_P);
      const float squarebias = 1.0 + std::log(1.0 + P->circle_to_square) * std::exp(P->circle_to_square * 3.0);
      float intersection = std::abs(P->optical_vignetting_distance / dir_lens_to_P[2]);
      float ovx = dir_lens_to_P[0] * intersection - lens[0];
      float ovy = dir_lens_to_P[1] * intersection - lens[1];
      float power = 1.0 + squarebias;
      float radius = (float)P->aperture_radius * P->optical_vignetting_radius;
      float dist = std::pow(std::abs(ovx), power) + std::pow(std::abs(ovy), power);
      if (dist > std::pow(radius, power)) { --count; continue; }
    }
    const float image_dist_focusdist = (-focal_length * -P->focus_distance) / (-focal_length + -P->focus_distance); /* lentil.h:665-667 */
    float focusdist_intersection = std::abs(image_dist_focusdist / dir_from_lens_to_image_sample[2]);     /* :389 */
    float rgb_weight[3] = {1.0f, 1.0f, 1.0f};                                                             /* :392 */
    int channel = 0;
    if (P->abb_chromatic > 0.0) {                                                                         /* :393-406 */
      /* sensor point of the unaberrated ray, for scaling the aberration (less in the centre), :348-353 */
      const float fi_u = std::abs(image_dist_focusdist / dir_from_lens_to_image_sample[2]);
      const float fip_u[3] = {lens[0] + dir_from_lens_to_image_sample[0] * fi_u,
                              lens[1] + dir_from_lens_to_image_sample[1] * fi_u,
                              lens[2] + dir_from_lens_to_image_sample[2] * fi_u};
      const float spu[2] = {fip_u[0] / fip_u[2], fip_u[1] / fip_u[2]};
      const float ddx = 0.0f - spu[0], ddy = 0.0f - spu[1];
      const float distance_to_center_unperturbed = sqrtf(ddx * ddx + ddy * ddy);      /* AiV2Dist (SDK, recalled) */
      const float abb_chromatic_lateral = 5.0;
      channel = static_cast<int>(std::floor((orc_xor128(F->xor_state) / 4294967296.0) * 3.0)) - 1;
      if (channel == -1) { rgb_weight[0] = 3; rgb_weight[1] = 0; rgb_weight[2] = 0; }
      else if (channel == 0) { rgb_weight[0] = 0; rgb_weight[1] = 3; rgb_weight[2] = 0; }
      else if (channel == 1) { rgb_weight[0] = 0; rgb_weight[1] = 0; rgb_weight[2] = 3; }
      float direction_shift = P->abb_chromatic_type == 0 /* green_magenta, src/lentil.h:82 */ ? std::abs(channel) : channel;
      const float shift = direction_shift * P->abb_chromatic * abb_chromatic_lateral * distance_to_center_unperturbed;
      /* Camera::get_image_dist_focusdist_thinlens_abberated, src/lentil.h:669-671 */
      const float aberrated = (-focal_length * -(P->focus_distance + shift)) / (-focal_length + -(P->focus_distance + shift));
      focusdist_intersection = std::abs(aberrated / dir_from_lens_to_image_sample[2]);
    }
    float fip[3] = {lens[0] + dir_from_lens_to_image_sample[0] * focusdist_intersection,
                    lens[1] + dir_from_lens_to_image_sample[1] * focusdist_intersection,
                    lens[2] + dir_from_lens_to_image_sample[2] * focusdist_intersection};               /* :409 */
    float sp[2] = {fip[0] / fip[2], fip[1] / fip[2]};                                                     /* :413-414 */
    /* AtVector2::operator/=(float) multiplies by 1.0f/f (SDK, recalled) */
    const float div = (P->sensor_width * 0.5) / -focal_length;                                            /* :416 */
    { const float inv = 1.0f / div; sp[0] *= inv; sp[1] *= inv; }
    if (P->abb_distortion > 0.0) {                                                                        /* :420, lens.h:550-559 */
      float b = P->abb_distortion;
      float l = sqrtf(sp[0] * sp[0] + sp[1] * sp[1]);
      float x0 = std::pow(9. * b * b * l + std::sqrt(3.) * std::sqrt(27. * b * b * b * b * l * l + 4. * b * b * b), 1. / 3.);
      float xx = x0 / (std::pow(2., 1. / 3.) * std::pow(3., 2. / 3.) * b) - std::pow(2. / 3., 1. / 3.) / x0;
      sp[0] = sp[0] * (xx / l); sp[1] = sp[1] * (xx / l);
    }
    const double s0 = sp[0], s1 = sp[1] * frame_aspect_ratio_without_region;                              /* :424 */
    const float pixel_x = (((s0 + 1.0) / 2.0) * P->xres_without_region) - P->region_min_x;               /* :425-426 */
    const float pixel_y = (((-s1 + 1.0) / 2.0) * P->yres_without_region) - P->region_min_y;
    if ((pixel_x >= xres) || (pixel_x < 0) || (pixel_y >= yres) || (pixel_y < 0)) { --count; continue; }  /* :429-432 */
    const int ix = floor(pixel_x), iy = floor(pixel_y);
    const unsigned pixelnumber = ix + (iy * P->xres);                                                     /* :434 */
    if (tl_worker) {          /* (thin lens without abb_chromatic > 0: rgb_weight is white) */
      tl_worker->draws.push_back({pixelnumber, (uint32_t)v, fitted_bidir_add_energy, 1.0f * inverse_sample_density * inv_samples, 0u});
    } else
    for (uint32_t a = 0; a < F->n_aovs; a++)                                                              /* :442-445 */
      add_to_buffer(F, a, pixelnumber, aov_values[a], fitted_bidir_add_energy, depth,
                    1.0f * inverse_sample_density * inv_samples, rgb_weight);
    for (uint32_t c = 0; c < F->crypto.size(); c++)                                                       /* :443 */
      add_to_buffer_cryptomatte(F->crypto[c], pixelnumber, crypto_cache[c], inverse_sample_density * inv_samples);
    ctr_of(F).accepted_draws++;
    /* chromatic mode: the channel (0..2) rides in the attempt's top two bits, as on the polynomial-optics path */
    if (F->keep_log) log_of(F).push_back({(uint32_t)v, total_samples_taken | ((uint32_t)(P->abb_chromatic > 0.0 ? channel + 1 : 0) << 30),
                                       pixelnumber});
  }
}

/* visit -> source pixel (see lentil_visits in include/lentil_hip.h) */
static inline void visit_pixel(const lentil_visits *V, uint64_t v, int *px, int *py) {
  if (V->visits_per_pixel) {
    const uint64_t p = v / V->visits_per_pixel;
    *px = V->pixel_x0 + (int)(p % V->pixels_per_row);
    *py = V->pixel_y0 + (int)(p / V->pixels_per_row) * (int)V->pixel_row_stride;
  } else {
    *px = (int)(V->pixel[v] & 0xFFFFu);
    *py = (int)(V->pixel[v] >> 16);
  }
}

/* Runs visits [v_begin, v_end) in order.  Returns 0, or LENTIL_ERR_UNSUPPORTED. */
ORC_API int orc_redistribute(const lentil_params *P, const OrcLens *L, const OrcBokeh *B, OrcFrame *F,
                             const lentil_visits *V, uint64_t v_begin, uint64_t v_end) {
  if (P->cameraType == LENTIL_POLYNOMIAL_OPTICS && !L) return LENTIL_ERR_UNSUPPORTED;
  if (P->bokeh_enable_image && !B) return LENTIL_ERR_INVALID;
  for (uint64_t v = v_begin; v < v_end; v++) {
    int px, py;
    visit_pixel(V, v, &px, &py);
    const float inv_density = V->inv_density ? V->inv_density[v] : P->inverse_sample_density;
    do_visit(P, L, B, F, V, v, px, py, inv_density);
  }
  return 0;
}

/* The same visits, [v_begin, v_end), over n_threads threads and ONE frame (test infrastructure and bench.py's cpu_baseline: no
 * counterpart in the reference, whose render threads share the buffers and race).  The range is cut where pixel
 * rows end (row_visits visits per row: a uniform stream), so a visit that stays in its pixel (src/lentil.h:938-955) is
 * added by the one thread that owns the pixel, in iterator order -- bit for bit what a single thread leaves there.  A draw
 * lands anywhere: the worker keeps it as a record, and the records are added thread after thread, which is visit order,
 * once all workers are through.  A pixel's fp32 sum then takes its own visits first and the draws after them -- another
 * order than the single-threaded walk's, inside the 1e-5 bar like every other (the fp64 shadows are exact either way);
 * counters and the draw log are the single-threaded ones.  Closest-filtered AOVs: the last candidate of the smallest |Z|
 * wins whatever the order the candidates are met in (zvisit says who holds the pixel); a stream with a candidate of |Z| == 0
 * cannot be split this way (orc_frame_merge's note).  Not for lentil_debug, cryptomatte, or thin lens with abb_chromatic > 0
 * (one xor128 stream in visit order): LENTIL_ERR_UNSUPPORTED.
 * Memory: one frame, whatever n_threads -- the per-thread frames of the Python-level merge were 60 B per pixel and AOV each. */
ORC_API int orc_redistribute_threads(const lentil_params *P, const OrcLens *L, const OrcBokeh *B, OrcFrame *F,
                                     const lentil_visits *V, uint64_t v_begin, uint64_t v_end, uint32_t n_threads,
                                     uint64_t row_visits) {
  if (P->cameraType == LENTIL_POLYNOMIAL_OPTICS && !L) return LENTIL_ERR_UNSUPPORTED;
  if (P->bokeh_enable_image && !B) return LENTIL_ERR_INVALID;
  if (!F->crypto.empty() || (P->cameraType == LENTIL_THINLENS && P->abb_chromatic > 0.0f) || !row_visits) return LENTIL_ERR_UNSUPPORTED;
  for (uint32_t a = 0; a < F->n_aovs; a++) if (F->kind[a] == LENTIL_FILTER_CLOSEST_DEBUG) return LENTIL_ERR_UNSUPPORTED;
  const uint64_t n = v_end > v_begin ? v_end - v_begin : 0;
  const uint64_t rows = (n + row_visits - 1) / row_visits;
  if (n_threads < 1) n_threads = 1;
  if ((uint64_t)n_threads > rows) n_threads = (uint32_t)(rows ? rows : 1);
  std::vector<OrcWorker> workers(n_threads);
  std::vector<uint64_t> bound(n_threads + 1);
  for (uint32_t i = 0; i <= n_threads; i++) {
    const uint64_t r = (uint64_t)((double)i * (double)rows / (double)n_threads + 0.5);
    bound[i] = std::min(v_end, v_begin + r * row_visits);
  }
  bound[0] = v_begin; bound[n_threads] = v_end;
  auto work = [&](uint32_t i) {
    OrcWorker &w = workers[i];
    memset(&w.ctr, 0, sizeof w.ctr);
    tl_worker = &w;
    for (uint64_t v = bound[i]; v < bound[i + 1]; v++) {
      int px, py;
      visit_pixel(V, v, &px, &py);
      const float inv_density = V->inv_density ? V->inv_density[v] : P->inverse_sample_density;
      do_visit(P, L, B, F, V, v, px, py, inv_density);
    }
    tl_worker = nullptr;
  };
  {
    std::vector<std::thread> ts;
    for (uint32_t i = 1; i < n_threads; i++) ts.emplace_back(work, i);
    work(0);
    for (std::thread &t : ts) t.join();
  }
  /* the draws, in visit order */
  for (uint32_t i = 0; i < n_threads; i++) {
    OrcWorker &w = workers[i];
    F->ctr.visits += w.ctr.visits; F->ctr.redistributed_visits += w.ctr.redistributed_visits;
    F->ctr.attempted_draws += w.ctr.attempted_draws; F->ctr.accepted_draws += w.ctr.accepted_draws;
    if (F->keep_log) F->log.insert(F->log.end(), w.log.begin(), w.log.end());
    for (const OrcDeferredDraw &d : w.draws) {
      const uint64_t v = d.visit;
      const float depth = V->pos_z[v * 4 + 3];
      float rgbw[3] = {1.0f, 1.0f, 1.0f};
      if (d.chan) { rgbw[0] = rgbw[1] = rgbw[2] = 0.0f; rgbw[d.chan - 1] = 3.0f; }
      for (uint32_t a = 0; a < F->n_aovs; a++) {
        const float *val = a == 0 ? &V->rgba[v * 4] : &V->extra[a - 1][v * 4];
        if (F->kind[a] == LENTIL_FILTER_CLOSEST) {
          /* Camera::add_to_buffer's z-test (src/lentil.h:832-837) met out of order: a pixel's own candidates of LATER visits are
           * in already -- the survivor is the candidate of the smallest |Z|, the latest of equals */
          const uint32_t px = d.pixel;
          const float z = std::abs(depth);
          const bool none = F->zvisit[px] == 0xFFFFFFFFu;
          if (none || z < F->zbuffer[px] || (z == F->zbuffer[px] && d.visit > F->zvisit[px])) {
            for (uint32_t b = 0; b < F->n_aovs; b++) {
              if (F->kind[b] != LENTIL_FILTER_CLOSEST) continue;
              const float *vb = b == 0 ? &V->rgba[v * 4] : &V->extra[b - 1][v * 4];
              for (int c = 0; c < 4; c++) {
                F->buffer[b][(size_t)px * 4 + c] = vb[c];
                if (F->shadow) F->buffer64[b][(size_t)px * 4 + c] = vb[c];
              }
            }
            F->zbuffer[px] = z;
            F->zvisit[px] = d.visit;
          }
        } else {
          add_to_buffer(F, a, d.pixel, val, d.add_energy, depth, d.weight, rgbw);
        }
      }
    }
    std::vector<OrcDeferredDraw>().swap(w.draws);
  }
  return 0;
}

/* a20 -- driver_process_bucket resolve, src/lentil_imager.cpp:112-118,169-186; out = xres*yres*4 */
ORC_API void orc_resolve(const OrcFrame *F, uint32_t aov, float *out) {
  const size_t np = (size_t)F->xres * F->yres;
  for (size_t p = 0; p < np; p++) {
    const float *b = &F->buffer[aov][p * 4];
    if (F->kind[aov] == LENTIL_FILTER_GAUSSIAN) {
      float r = b[0], g = b[1], bl = b[2], a = b[3];
      /* AtRGBA::operator/=(float) multiplies by 1.0f/f (SDK, recalled) */
      if (F->weight[p] != 0.0) { const float c = 1.0f / F->weight[p]; r *= c; g *= c; bl *= c; a *= c; }
      out[p * 4] = r; out[p * 4 + 1] = g; out[p * 4 + 2] = bl; out[p * 4 + 3] = a;
    } else {
      out[p * 4] = b[0]; out[p * 4 + 1] = b[1]; out[p * 4 + 2] = b[2]; out[p * 4 + 3] = 1.0f;
    }
  }
}

/* a1 -- visit prologue, src/lentil_filter.cpp:79-88: AA and inverse density from the footprint count */
ORC_API float orc_inverse_sample_density(int samples_counter, float filter_width, int aa_samples_set_by_user,
                                         int *redistribution_ok) {
  float AA_samples = std::sqrt(samples_counter) / filter_width;
  float inverse_sample_density = 1.0 / (AA_samples * AA_samples);
  if (redistribution_ok)
    *redistribution_ok = !(static_cast<int>(std::round(AA_samples)) != aa_samples_set_by_user || (aa_samples_set_by_user < 3));
  return inverse_sample_density;
}

/* =====================================================================================
 * Host-side camera setup -- src/lentil.h:1316-1460,1568-1670
 * ===================================================================================== */
/* line_plane_intersection with the fixed y = 0 plane, src/lens.h:412-419; returns the z component */
static double line_plane_y0_z(const double o[3], const double d_in[3]) {
  double d[3] = {d_in[0], d_in[1], d_in[2]};
  const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);   /* Eigen normalize(): v / norm */
  d[0] /= n; d[1] /= n; d[2] /= n;
  /* coord = normalized (100,0,100); coord.dot(planeNormal) = 0 exactly; planeNormal = (0,1,0) */
  /* rayOrigin + (rayDirection * (A - B)) / C : the vector is scaled first, then divided */
  return o[2] + (d[2] * (0.0 - o[1])) / d[1];
}

/* Camera::camera_get_y0_intersection_distance, src/lentil.h:1361-1386 */
ORC_API double orc_camera_get_y0_intersection_distance(const OrcLens *L, double sensor_shift, double lambda) {
  double sensor[5] = {0, 0, 0, 0, lambda};
  double aperture[5] = {0, L->k.lens_aperture_housing_radius * 0.25, 0, 0, 0};
  double out[5] = {0, 0, 0, 0, 0};
  orc_pt_sample_aperture(L, sensor, aperture, sensor_shift);
  sensor[0] += sensor[2] * sensor_shift;
  sensor[1] += sensor[3] * sensor_shift;
  orc_lens_evaluate(L, sensor, out);
  double pos[3], dir[3];
  pupil_to_cs(L->k, out, out + 2, pos, dir);
  return line_plane_y0_z(pos, dir);
}

/* logarithmic_values, src/lens.h:395-407 + Camera::logarithmic_focus_search, src/lentil.h:1445-1460 */
ORC_API double orc_logarithmic_focus_search(const OrcLens *L, double focal_distance, double lambda) {
  double closest_distance = 999999999.0;
  double best_sensor_shift = 0.0;
  for (double i = -1.0; i <= 1.0; i += 0.0001) {
    const double sensorshift = (i < 0 ? -1 : 1) * std::pow(i, 2.0) * (45.0 - 0.0) + 0.0;
    const double intersection_distance = orc_camera_get_y0_intersection_distance(L, sensorshift, lambda);
    const double new_distance = focal_distance - intersection_distance;
    if (new_distance < closest_distance && new_distance > 0.0) {
      closest_distance = new_distance;
      best_sensor_shift = sensorshift;
    }
  }
  return best_sensor_shift;
}

/* Camera::trace_backwards_for_fstop, src/lentil.h:1390-1441 (AI_BIG = 1.0e12f) */
ORC_API void orc_trace_backwards_for_fstop(const OrcLens *L, double fstop_target, double lambda,
                                           double *calculated_fstop, double *calculated_aperture_radius) {
  const int maxrays = 1000;
  double best_valid_fstop = 0.0, best_valid_aperture_radius = 0.0;
  const lentil_lens_table &k = L->k;
  for (int i = 1; i < maxrays; i++) {
    const double parallel_ray_height = (static_cast<double>(i) / static_cast<double>(maxrays)) * k.lens_outer_pupil_radius;
    const double target[3] = {0, parallel_ray_height, (double)1.0e12f};
    double sensor[5] = {0, 0, 0, 0, lambda};
    double out[5] = {0, 0, 0, 0, 0};
    const double aperture[2] = {0.01, parallel_ray_height};
    if (orc_lt_sample_aperture(L, target, aperture, sensor, out, lambda, nullptr) <= 0.0) continue;
    const double px = sensor[0] + (sensor[2] * k.lens_back_focal_length);
    const double py = sensor[1] + (sensor[3] * k.lens_back_focal_length);
    if (px * px + py * py > k.lens_inner_pupil_radius * k.lens_inner_pupil_radius) continue;
    double pos[3], dir[3];
    const double Ri = k.lens_inner_pupil_curvature_radius;
    if (k.lens_inner_pupil_geometry == LENTIL_GEOM_CYL_Y) orc_cylinderToCs(out, out + 2, pos, dir, -Ri + k.lens_back_focal_length, Ri, 1);
    else if (k.lens_inner_pupil_geometry == LENTIL_GEOM_CYL_X) orc_cylinderToCs(out, out + 2, pos, dir, -Ri + k.lens_back_focal_length, Ri, 0);
    else orc_sphereToCs(out, out + 2, pos, dir, -Ri + k.lens_back_focal_length, Ri);
    const double theta = std::atan(pos[1] / pos[2]);
    const double fstop = 1.0 / (std::sin(theta) * 2.0);
    if (fstop < fstop_target) {
      *calculated_fstop = best_valid_fstop;
      *calculated_aperture_radius = best_valid_aperture_radius;
      return;
    } else {
      best_valid_fstop = fstop;
      best_valid_aperture_radius = parallel_ray_height;
    }
  }
  *calculated_fstop = best_valid_fstop;
  *calculated_aperture_radius = best_valid_aperture_radius;
}

/* Camera::trace_ray_focus_check, src/lentil.h:1316-1357 */
ORC_API int orc_trace_ray_focus_check(const OrcLens *L, double sensor_shift, double lambda, double *test_focus_distance) {
  const lentil_lens_table &k = L->k;
  double sensor[5] = {0, 0, 0, 0, lambda};
  double aperture[5] = {0, k.lens_aperture_housing_radius * 0.25, 0, 0, 0};
  double out[5] = {0, 0, 0, 0, 0};
  orc_pt_sample_aperture(L, sensor, aperture, sensor_shift);
  sensor[0] += sensor[2] * sensor_shift;
  sensor[1] += sensor[3] * sensor_shift;
  const double transmittance = orc_lens_evaluate(L, sensor, out);
  if (transmittance <= 0.0) return 0;
  if (out[0] * out[0] + out[1] * out[1] > k.lens_outer_pupil_radius * k.lens_outer_pupil_radius) return 0;
  const double px = sensor[0] + sensor[2] * k.lens_back_focal_length;
  const double py = sensor[1] + sensor[3] * k.lens_back_focal_length;
  if (px * px + py * py > k.lens_inner_pupil_radius * k.lens_inner_pupil_radius) return 0;
  double pos[3], dir[3];
  pupil_to_cs(k, out, out + 2, pos, dir);
  *test_focus_distance = line_plane_y0_z(pos, dir);
  return 1;
}

/* =====================================================================================
 * Forward camera rays -- src/lentil.h:283-569.
 * xor128 state is explicit (the reference keeps it in function-local statics, src/global.h:22-27).
 * ===================================================================================== */
ORC_API void orc_trace_ray_fw_po(const lentil_params *P, const OrcLens *L, const OrcBokeh *B, uint32_t rng[4],
                                 double lambda, double sx, double sy, double *r1, double *r2, int deriv_ray,
                                 float origin[3], float direction[3], float weight[3], int *tries_out) {
  const lentil_lens_table &k = L->k;
  int tries = 0;
  bool ray_succes = false;
  double sensor[5] = {0, 0, 0, 0, 0}, aperture[5] = {0, 0, 0, 0, 0}, out[5] = {0, 0, 0, 0, 0};
  while (!ray_succes && tries <= P->vignetting_retries) {
    sensor[0] = sx * (P->sensor_width * 0.5);
    sensor[1] = sy * (P->sensor_width * 0.5);
    sensor[2] = sensor[3] = 0.0;
    sensor[4] = lambda;
    for (int i = 0; i < 5; i++) { aperture[i] = 0; out[i] = 0; }
    double unit_disk[2] = {0.0, 0.0};
    if (P->enable_dof) {
      if (!deriv_ray && tries > 0) {
        *r1 = orc_xor128(rng) / 4294967296.0;
        *r2 = orc_xor128(rng) / 4294967296.0;
      }
      if (P->bokeh_enable_image) {
        orc_xor128(rng); orc_xor128(rng);                 /* the two unused stratification draws */
        orc_bokeh_sample(B, *r1, *r2, unit_disk);
      } else if (P->bokeh_aperture_blades < 2) {
        orc_concentric_disk_sample(*r1, *r2, unit_disk);
      } else {
        orc_triangular_aperture(&unit_disk[0], &unit_disk[1], *r1, *r2, 1.0, P->bokeh_aperture_blades);
      }
    }
    aperture[0] = unit_disk[0] * P->aperture_radius;
    aperture[1] = unit_disk[1] * P->aperture_radius;
    if (P->enable_dof) orc_pt_sample_aperture(L, sensor, aperture, P->sensor_shift);
    sensor[0] += sensor[2] * P->sensor_shift;
    sensor[1] += sensor[3] * P->sensor_shift;
    const double transmittance = orc_lens_evaluate(L, sensor, out);
    if (transmittance <= 0.0) { ++tries; continue; }
    if (out[0] * out[0] + out[1] * out[1] > k.lens_outer_pupil_radius * k.lens_outer_pupil_radius) { ++tries; continue; }
    const double px = sensor[0] + sensor[2] * k.lens_back_focal_length;
    const double py = sensor[1] + sensor[3] * k.lens_back_focal_length;
    if (px * px + py * py > k.lens_inner_pupil_radius * k.lens_inner_pupil_radius) { ++tries; continue; }
    ray_succes = true;
  }
  if (!ray_succes) weight[0] = weight[1] = weight[2] = 0.0f;
  double pos[3], dir[3];
  pupil_to_cs(k, out, out + 2, pos, dir);
  float o[3] = {(float)pos[0], (float)pos[1], (float)pos[2]};
  float d[3] = {(float)dir[0], (float)dir[1], (float)dir[2]};
  float s = -1.0f;                                           /* src/lentil.h:395-416 */
  if (P->unitModel == LENTIL_UNIT_CM) s = -0.1f;
  else if (P->unitModel == LENTIL_UNIT_DM) s = -0.01f;
  else if (P->unitModel == LENTIL_UNIT_M) s = -0.001f;
  for (int i = 0; i < 3; i++) { o[i] *= s; d[i] *= s; }
  v3norm(d, d);
  for (int i = 0; i < 3; i++) { origin[i] = o[i]; direction[i] = d[i]; }
  if (o[0] != o[0] || o[1] != o[1] || o[2] != o[2] || d[0] != d[0] || d[1] != d[1] || d[2] != d[2])
    weight[0] = weight[1] = weight[2] = 0.0f;
  if (tries_out) *tries_out = tries;
}

ORC_API void orc_trace_ray_fw_thinlens(const lentil_params *P, const OrcBokeh *B, uint32_t rng[4], double sx, double sy,
                                       double *r1, double *r2, int deriv_ray, float origin[3], float dir[3],
                                       float weight[3], int *tries_out) {
  int tries = 0;
  bool ray_succes = false;
  float o[3] = {0, 0, 0}, dd[3] = {0, 0, 0};
  while (!ray_succes && tries <= P->vignetting_retries) {
    float s[3] = {(float)sx, (float)sy, 0.0f};
    if (P->abb_distortion > 0.0) {                           /* barrelDistortion, src/lens.h:545-548 */
      float ux = sx, uy = sy;
      const float f = 1. + (ux * ux + uy * uy) * P->abb_distortion;
      s[0] = ux * f; s[1] = uy * f;
    }
    const float p[3] = {(float)(s[0] * (P->sensor_width * 0.5)), (float)(s[1] * (P->sensor_width * 0.5)), -P->focal_length};
    float dir_from_center[3];
    v3norm(p, dir_from_center);
    double unit_disk[2] = {0, 0};
    if (P->enable_dof) {
      if (!deriv_ray && tries > 0) {
        *r1 = orc_xor128(rng) / 4294967296.0;
        *r2 = orc_xor128(rng) / 4294967296.0;
      }
      if (P->bokeh_enable_image) {
        orc_xor128(rng); orc_xor128(rng);
        orc_bokeh_sample(B, *r1, *r2, unit_disk);
      } else if (P->bokeh_aperture_blades < 2) {
        orc_concentricDiskSample(*r1, *r2, unit_disk, P->abb_spherical, P->circle_to_square);
      } else {
        orc_triangular_aperture(&unit_disk[0], &unit_disk[1], *r1, *r2, 1.0, P->bokeh_aperture_blades);
      }
    }
    unit_disk[0] *= P->bokeh_anamorphic;
    const float lens[3] = {(float)(unit_disk[0] * P->aperture_radius), (float)(unit_disk[1] * P->aperture_radius), 0.0f};
    const float intersection = std::abs(P->focus_distance / lerpf(0.0f, dir_from_center[2], 1.0));
    const float focusPoint[3] = {dir_from_center[0] * intersection, dir_from_center[1] * intersection, dir_from_center[2] * intersection};
    const float t[3] = {focusPoint[0] - lens[0], focusPoint[1] - lens[1], focusPoint[2] - lens[2]};
    float dir_from_lens[3];
    v3norm(t, dir_from_lens);
    {                                                            /* coma, src/lentil.h:490-491 */
      const float abb_coma_multiplied = P->abb_coma * orc_abb_coma_multipliers(P->sensor_width, P->focal_length,
                                                                               dir_from_center, unit_disk);
      float rotated[3];
      orc_abb_coma_perturb(dir_from_lens, dir_from_lens, abb_coma_multiplied, 0, rotated);
      dir_from_lens[0] = rotated[0]; dir_from_lens[1] = rotated[1]; dir_from_lens[2] = rotated[2];
    }
    if (P->optical_vignetting_distance > 0.0 && !deriv_ray) {     /* src/lens.h:529-543 */
      const float squarebias = 1.0 + std::log(1.0 + P->circle_to_square) * std::exp(P->circle_to_square * 3.0);
      float inter = std::abs(P->optical_vignetting_distance / dir_from_lens[2]);
      float ovx = dir_from_lens[0] * inter - lens[0];
      float ovy = dir_from_lens[1] * inter - lens[1];
      float power = 1.0 + squarebias;
      float radius = (float)P->aperture_radius * P->optical_vignetting_radius;
      float dist = std::pow(std::abs(ovx), power) + std::pow(std::abs(ovy), power);
      if (dist > std::pow(radius, power)) { ++tries; continue; }
    }
    float sc = 1.0f;                                          /* src/lentil.h:540-561 */
    if (P->unitModel == LENTIL_UNIT_MM) sc = 10.0f;
    else if (P->unitModel == LENTIL_UNIT_DM) sc = 0.1f;
    else if (P->unitModel == LENTIL_UNIT_M) sc = 0.01f;
    for (int i = 0; i < 3; i++) { o[i] = lens[i] * sc; dd[i] = dir_from_lens[i] * sc; }
    ray_succes = true;
  }
  v3norm(dd, dd);
  for (int i = 0; i < 3; i++) { origin[i] = o[i]; dir[i] = dd[i]; }
  if (!ray_succes) weight[0] = weight[1] = weight[2] = 0.0f;
  if (tries_out) *tries_out = tries;
}
