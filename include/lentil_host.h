/*
 * lentil_host.h -- host-side (CPU, once-per-render) helpers of the plugin mirror:
 * liblentil_host.so.  These produce the tables / parameters that liblentil_hip.so consumes.
 * C-ABI, plain pointers and sizes.
 */
#ifndef LENTIL_HOST_H
#define LENTIL_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* imageData::bokehProbability (src/imagebokeh.h:143-338): builds the row CDF and the per-row
 * column CDFs, both sorted by descending probability with std::sort (tie order matters and is
 * libstdc++'s), from x*y*nchannels float texels as AiTextureLoad delivers them
 * (src/imagebokeh.h:16-18,107).  Output arrays: cdfRow[y], rowIndices[y], cdfColumn[x*y],
 * columnIndices[x*y].  Returns 0, or -1 for an invalid image (non-square, < 3 channels:
 * src/imagebokeh.h:50-52,97-101). */
int lentil_host_bokeh_probability(const float *pixelData, int32_t x, int32_t y, int32_t nchannels,
                                  float *cdfRow, int32_t *rowIndices, float *cdfColumn,
                                  int32_t *columnIndices);

/* ------------------------------------------------------------------------------------------
 * Polynomial-optics lens on the host: what the reference evaluates on CPU threads around the
 * redistribution path -- camera setup (once per render) and forward camera rays (per camera sample).
 * A lentil_host_lens owns a copy of the table and the derivative tables derived from it.
 * ---------------------------------------------------------------------------------------- */
#include "lentil_hip.h"

typedef struct lentil_host_lens lentil_host_lens;
lentil_host_lens *lentil_host_lens_create(const lentil_lens_table *table);
void lentil_host_lens_destroy(lentil_host_lens *lens);

/* Camera::lens_evaluate (src/lentil.h:1257-1266): in/out = [x, y, dx, dy, lambda]; returns max(0, T) */
double lentil_host_lens_evaluate(const lentil_host_lens *lens, const double in[5], double out[5]);
/* Camera::lens_pt_sample_aperture (src/lentil.h:1272-1291): solves in[2..3] so that the ray through the
 * sensor point in[0..1] (shifted by dist) hits the aperture point out[0..1]; out[2..3] = direction there */
void lentil_host_lens_pt_sample_aperture(const lentil_host_lens *lens, double in[5], double out[5], double dist);
/* Camera::lens_lt_sample_aperture (src/lentil.h:1296-1313) */
double lentil_host_lens_lt_sample_aperture(const lentil_host_lens *lens, const double scene[3], const double ap[2],
                                           double sensor[5], double out[5], double lambda);
/* Camera::camera_get_y0_intersection_distance / logarithmic_focus_search / trace_backwards_for_fstop /
 * trace_ray_focus_check (src/lentil.h:1361-1386, 1445-1460, 1390-1441, 1316-1357) */
double lentil_host_camera_get_y0_intersection_distance(const lentil_host_lens *lens, double sensor_shift, double lambda);
double lentil_host_logarithmic_focus_search(const lentil_host_lens *lens, double focal_distance, double lambda);
void lentil_host_trace_backwards_for_fstop(const lentil_host_lens *lens, double fstop_target, double lambda,
                                           double *calculated_fstop, double *calculated_aperture_radius);
int lentil_host_trace_ray_focus_check(const lentil_host_lens *lens, double sensor_shift, double lambda,
                                      double *test_focus_distance);

/* Camera::camera_model_specific_setup (src/lentil.h:1568-1670).  Inputs are the node parameters after
 * get_lentil_camera_params (src/lentil.h:1189-1243): params->cameraType, sensor_width, focal_length,
 * focus_distance (still in cm as the user gave it); input_fstop (0 = wide open), wavelength_nm,
 * extra_sensor_shift.  Fills focus_distance (x10 for PO), aperture_radius, sensor_shift, and returns
 * tan_fov; lens may be NULL for the thin lens.  Returns 0, or -1 on invalid arguments. */
int lentil_host_camera_model_specific_setup(lentil_params *params, const lentil_host_lens *lens, double input_fstop,
                                            double wavelength_nm, double extra_sensor_shift, double *tan_fov);
/* The same with the focus search (src/lentil.h:1632: 20 001 candidate sensor shifts, two polynomial solves each)
 * delegated: focus_search(user, focal_distance_mm, lambda_um, &shift) returns 0 and the shift of
 * Camera::logarithmic_focus_search -- lentil_hip_focus_search does, one GPU lane per candidate; NULL or a
 * non-zero return runs the sequential loop of lentil_host_logarithmic_focus_search. */
typedef int (*lentil_focus_search_fn)(void *user, double focal_distance, double lambda, double *best_sensor_shift);
int lentil_host_camera_model_specific_setup_with(lentil_params *params, const lentil_host_lens *lens, double input_fstop,
                                                 double wavelength_nm, double extra_sensor_shift, double *tan_fov,
                                                 lentil_focus_search_fn focus_search, void *user);

/* Forward camera rays: Camera::trace_ray_fw_po / trace_ray_fw_thinlens (src/lentil.h:283-569).
 * rng: the xor128 state (src/global.h:22-27; the reference keeps it in function statics -- pass
 * lentil_host_xor128_init'ed storage, one per thread).  r1, r2: Arnold's lens samples, replaced by
 * xor128 draws on retries.  weight is set to 0 when every try is vignetted. */
void lentil_host_xor128_init(uint32_t state[4]);
void lentil_host_trace_ray_fw_po(const lentil_params *params, const lentil_host_lens *lens,
                                 const lentil_bokeh_table *bokeh, uint32_t rng[4], double lambda, double sx, double sy,
                                 double *r1, double *r2, int deriv_ray, float origin[3], float direction[3],
                                 float weight[3], int *tries);
void lentil_host_trace_ray_fw_thinlens(const lentil_params *params, const lentil_bokeh_table *bokeh, uint32_t rng[4],
                                       double sx, double sy, double *r1, double *r2, int deriv_ray, float origin[3],
                                       float direction[3], float weight[3], int *tries);
/* camera_create_ray (src/lentil_camera.cpp:78-125): the ray plus finite-difference differentials
 * (step 0.001).  in = {sx, sy, dsx, dsy, lensx, lensy}; exposure multiplies the weight. */
typedef struct lentil_host_camera_ray {
  float origin[3], dir[3], weight[3];
  float dOdx[3], dOdy[3], dDdx[3], dDdy[3];
} lentil_host_camera_ray;
void lentil_host_camera_create_ray(const lentil_params *params, const lentil_host_lens *lens,
                                   const lentil_bokeh_table *bokeh, uint32_t rng[4], double lambda, float exposure,
                                   const float in[6], lentil_host_camera_ray *out);
/* camera_reverse_ray (src/lentil_camera.cpp:164-172): pinhole approximation */
void lentil_host_camera_reverse_ray(double tan_fov, const float Po[3], float Ps[2]);

#ifdef __cplusplus
}
#endif
#endif
