/*
 * lentil_hip.h -- C-ABI of liblentil_hip.so: the MI355X (gfx950) implementation of
 * lentil's bidirectional bokeh redistribution + polynomial-optics ray transfer.
 *
 * This is the drop-in boundary.  A `Camera` object of the
 * reference plugin (src/lentil.h:92-196) keeps owning all render state; instead of
 * accumulating into std::vector buffers from `filter_pixel` it owns one
 * `lentil_hip_ctx` and drives it with the calls below.  Every entry point names the
 * reference code it replaces.  Plain pointers and sizes only: no torch, no HIP and no
 * Arnold types appear in any signature.
 *
 * Conventions
 *   - All functions return LENTIL_OK (0) or a negative LENTIL_ERR_* code; the text of
 *     the last error is available from lentil_hip_last_error().  Nothing aborts or
 *     exits (the reference's failures on this path are per-draw rejects, and
 *     AiRenderAbort on setup errors, src/lentil.h:225-228).
 *   - A context is bound to one GPU and one HIP stream.  Calls on one context must be
 *     serialised by the caller (the reference takes l_critsec only around
 *     setup_camera, src/lentil.h:212,235); different contexts are independent
 *     (multi-GPU = one context per device, one process per GPU).
 *   - "visit" = one (pixel, AOV-sample) pair yielded by the filter's sample iterator
 *     (src/lentil_filter.cpp:105); "draw" = one backward trace attempt for a visit
 *     (src/lentil_filter.cpp:248,311).
 */
#ifndef LENTIL_HIP_H
#define LENTIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LENTIL_ABI_VERSION 1

#define LENTIL_OK 0
#define LENTIL_ERR_INVALID (-1)     /* bad argument / call order */
#define LENTIL_ERR_HIP (-2)         /* a HIP runtime call failed */
#define LENTIL_ERR_UNSUPPORTED (-3) /* parameter combination not implemented on the GPU */
#define LENTIL_ERR_NOMEM (-4)

/* enum CameraType, src/lentil.h:75-78 */
#define LENTIL_THINLENS 0
#define LENTIL_POLYNOMIAL_OPTICS 1
/* enum UnitModel, src/lentil.h:67-72 */
#define LENTIL_UNIT_MM 0
#define LENTIL_UNIT_CM 1
#define LENTIL_UNIT_DM 2
#define LENTIL_UNIT_M 3
/* AOVData::original_filter, src/aov_data.h:120, src/lentil.h:189-191 */
#define LENTIL_FILTER_GAUSSIAN 0
#define LENTIL_FILTER_CLOSEST 1
#define LENTIL_FILTER_VARIANCE 2
/* the lentil_debug AOV (src/lentil_operator.cpp:99-111): closest-filtered through its own z-buffer, fed by
 * redistributed draws only, value = the visit's draw count; takes no visit column (extra[k-1] may be NULL) */
#define LENTIL_FILTER_CLOSEST_DEBUG 3
/* pupil geometry strings "cyl-y" / "cyl-x" / anything else, src/lentil.h:387-389 */
#define LENTIL_GEOM_SPHERICAL 0
#define LENTIL_GEOM_CYL_Y 1
#define LENTIL_GEOM_CYL_X 2

#define LENTIL_MAX_AOVS 16 /* RGBA + up to 15 more redistributed AOVs */
#define LENTIL_MAX_CRYPTO 9 /* cryptomatte AOVs: crypto_{material,object,asset}{00,01,02} */

/* ------------------------------------------------------------------------------------
 * Camera state the kernels need: the fields of `struct Camera` (src/lentil.h:92-196,
 * SURVEY.md appendix B) that filter_pixel / trace_ray_bw_po / add_to_buffer read, with
 * the reference's own names and float/double widths (appendix C.10).  Values are the
 * ones *after* get_lentil_camera_params() clamping (src/lentil.h:1189-1243) and
 * camera_model_specific_setup() (src/lentil.h:1568-1670): e.g. PO focus_distance is
 * already x10 (mm), bokeh_anamorphic is already 1 - param.
 * ---------------------------------------------------------------------------------- */
typedef struct lentil_params {
  int32_t cameraType;      /* LENTIL_THINLENS | LENTIL_POLYNOMIAL_OPTICS */
  int32_t unitModel;       /* LENTIL_UNIT_* */
  int32_t enable_dof;
  int32_t vignetting_retries;      /* default 15, src/lentil_camera.cpp:44 */
  int32_t bokeh_aperture_blades;
  int32_t bokeh_enable_image;
  int32_t bidir_sample_mult;
  int32_t enable_bidir_transmission;
  int32_t enable_skydome;
  int32_t abb_chromatic_type;
  int32_t adaptive_sampling;       /* options.enable_adaptive_sampling, src/lentil_filter.cpp:74 */
  int32_t samples_override;        /* 0: reference draw-count formula (src/lentil_filter.cpp:177-202);
                                      >0: fixed accepted-draw count per redistributed visit (bench configs) */
  uint32_t xres, yres;             /* buffer size incl. the +1 quirk, src/lentil.h:1079-1080 */
  uint32_t xres_without_region, yres_without_region;
  int32_t region_min_x, region_min_y;

  double sensor_width;
  double focus_distance;
  double aperture_radius;
  double sensor_shift;
  double bidir_add_energy_minimum_luminance;

  float focal_length;
  float bidir_add_energy;
  float bidir_add_energy_transition;
  float abb_spherical;
  float abb_coma;
  float abb_distortion;
  float abb_chromatic;
  float circle_to_square;
  float bokeh_anamorphic;
  float optical_vignetting_distance;
  float optical_vignetting_radius;
  float filter_width;
  float inverse_sample_density;    /* 1/AA^2 as computed at src/lentil_filter.cpp:83-84 (non-adaptive) */
  float lambda_bw;                 /* 0.55, src/lentil_filter.cpp:254 */

  /* AiWorldToCameraMatrix(camera, time) of a static camera, row-vector convention
   * (p' = p * M, translation in row 3), src/lentil_filter.cpp:143-144 */
  float world_to_camera[4][4];
} lentil_params;

/* ------------------------------------------------------------------------------------
 * Lens table: replaces the generated `case <lens>: {...}` bodies spliced in at
 * src/lentil.h:1262,1278,1308,1576 (absent from the reference tree).  One sparse
 * 5-variate polynomial per output; a term is  c * x^e0 * y^e1 * dx^e2 * dy^e3 * lambda^e4
 * with integer powers evaluated like lens_ipow (src/lens.h:226-233), terms summed in
 * table order.  Partial derivatives are derived from these terms inside the library.
 * ---------------------------------------------------------------------------------- */
typedef struct lentil_term {
  double c;
  uint8_t e[5];
  uint8_t pad[3];
} lentil_term;

typedef struct lentil_poly {
  uint32_t first; /* index of first term in lentil_lens_table::terms */
  uint32_t count;
} lentil_poly;

typedef struct lentil_lens_table {
  /* lens constants, src/lentil.h:106-120 */
  double lens_outer_pupil_radius;
  double lens_inner_pupil_radius;
  double lens_length;
  double lens_back_focal_length;
  double lens_effective_focal_length;
  double lens_aperture_pos;
  double lens_aperture_housing_radius;
  double lens_inner_pupil_curvature_radius;
  double lens_outer_pupil_curvature_radius;
  double lens_field_of_view;
  double lens_fstop;
  double lens_aperture_radius_at_fstop;
  int32_t lens_inner_pupil_geometry; /* LENTIL_GEOM_* */
  int32_t lens_outer_pupil_geometry;
  lentil_poly out[5]; /* outer pupil x, y, dx, dy, transmittance (pt_evaluate) */
  lentil_poly ap[4];  /* aperture plane x, y, dx, dy (pt_evaluate_aperture) */
  uint32_t n_terms;
  uint32_t reserved;
  const lentil_term *terms; /* host pointer, copied by lentil_hip_set_lens */
} lentil_lens_table;

/* Bokeh-image importance tables exactly as imageData::bokehProbability leaves them
 * (src/imagebokeh.h:143-338); host pointers, copied by lentil_hip_set_bokeh. */
typedef struct lentil_bokeh_table {
  int32_t x, y;
  const float *cdfRow;          /* y   */
  const int32_t *rowIndices;    /* y   */
  const float *cdfColumn;       /* x*y */
  const int32_t *columnIndices; /* x*y */
} lentil_bokeh_table;

/* ------------------------------------------------------------------------------------
 * Visit stream: what filter_pixel gathers per AOV sample (src/lentil_filter.cpp:115-164,
 * 206-234), as fp32 columns of 4 floats per visit (16-byte rows, so one wave reads
 * 1 KiB per column per load):
 *     rgba          : sample RGBA                                   (:115)
 *     pos_z         : P world space .xyz, Z                         (:116-117)
 *     raydir_time   : lentil_raydir .xyz, lentil_time               (:121,141)
 *     volume_ignore : volume .rgb, lentil_ignore                    (:135,162)
 *     transmission  : transmission RGBA                             (:152)
 *     extra[k]      : AOV k+1 already widened to RGBA               (:214-232)
 * = 80 + 16*K bytes per visit.  Visits are pixel-major in iterator order.
 *
 * Pixel mapping.  visits_per_pixel > 0: uniform footprint; visit v belongs to source
 * pixel p = v / visits_per_pixel with  px = pixel_x0 + p % pixels_per_row,
 * py = pixel_y0 + (p / pixels_per_row) * pixel_row_stride  (region-relative, i.e. after
 * src/lentil_filter.cpp:100-101; row_stride > 1 expresses a row-interleaved multi-GPU
 * partition).  visits_per_pixel == 0: `pixel` holds px | (py << 16) per visit and
 * `inv_density` (optional) the per-visit inverse sample density (adaptive sampling or
 * ragged footprints, src/lentil_filter.cpp:83-84,109).
 * ---------------------------------------------------------------------------------- */
typedef struct lentil_visits {
  uint64_t n;
  uint32_t visits_per_pixel;
  uint32_t pixels_per_row;
  int32_t pixel_x0, pixel_y0;
  uint32_t pixel_row_stride;
  uint32_t n_extra;           /* K: number of extra AOV columns */
  const float *rgba;
  const float *pos_z;
  const float *raydir_time;
  const float *volume_ignore;
  const float *transmission;
  const float *extra[LENTIL_MAX_AOVS - 1];
  const uint32_t *pixel;      /* optional, see above */
  const float *inv_density;   /* optional */
} lentil_visits;

/* counters of the last lentil_hip_redistribute (device-side, fetched on demand) */
typedef struct lentil_counters {
  uint64_t visits;
  uint64_t redistributed_visits; /* visits that entered the draw loop */
  uint64_t attempted_draws;      /* total_samples_taken summed, src/lentil_filter.cpp:248 */
  uint64_t accepted_draws;
  uint64_t worklist_overflow;    /* non-zero => work list too small, results incomplete */
  uint64_t newton_iterations;    /* lane-iterations of the lt_sample_aperture solver (PO draw kernel) */
  uint64_t tries;                /* aperture draws = solves started (attempts x vignetting retries) */
  uint64_t lane_rounds;          /* 64 x scheduler rounds: iteration slots offered (utilisation = iterations / this) */
  uint64_t slow_solves;          /* solves finished by the one-wave-per-solve straggler kernel (LENTIL_SLOW_AT) */
  uint64_t blind_chunks;         /* chunks whose draw rounds were enqueued without waiting for their scan (sized from the previous pass) */
  uint64_t fallback_chunks;      /* ... of which did not fit and were redone with exact sizes */
  uint64_t streamed;             /* 1: the pass ran streamed (one scan launch feeding persistent solve waves; counts as one blind chunk) */
} lentil_counters;

/* Sums over the passes whose end has been looked at since the last reset (lentil_hip_pass_totals): what a caller that pipelines
 * frames reads once, after the frames, instead of lentil_hip_get_counters / _last_timing after every pass (each of which waits
 * for the pass).  Times are HIP-event times as lentil_hip_last_timing reports them. */
typedef struct lentil_pass_totals {
  uint64_t passes;               /* lentil_hip_redistribute calls accounted for */
  uint64_t streamed;             /* ... that ran streamed */
  uint64_t blind_chunks, fallback_chunks;      /* as lentil_counters, summed */
  uint64_t deferred;             /* passes that returned before their end was known (asynchronous end) */
  uint64_t abandoned;            /* ... whose frame was cleared before anybody observed it */
  uint64_t abandoned_incomplete; /* ... and which still needed work at that point (a second round, a redo): the frame nobody
                                  * looked at was not complete.  Their kernel times are in the sums all the same. */
  uint64_t visits, redistributed_visits, attempted_draws, accepted_draws, worklist_overflow;
  uint64_t newton_iterations, tries, lane_rounds, slow_solves;
  uint64_t scan_launches;
  uint64_t rounds_max;           /* most solve/accept rounds any of the passes needed */
  double scan_ms, draw_ms, resolve_ms;
} lentil_pass_totals;

/* one accepted draw, for index-parity tests: (visit, attempt n, linear pixel) */
typedef struct lentil_draw_record {
  uint32_t visit;
  uint32_t attempt;
  uint32_t pixel;
} lentil_draw_record;

typedef struct lentil_hip_ctx lentil_hip_ctx;

/* --- lifetime ----------------------------------------------------------------------
 * replaces `new Camera()` / `delete camera_data` buffer ownership
 * (src/lentil_camera.cpp:58-61,70-75; Camera::destroy_buffers src/lentil.h:1143-1148). */
int lentil_hip_abi_version(void);
int lentil_hip_create(int device, lentil_hip_ctx **out_ctx);
int lentil_hip_destroy(lentil_hip_ctx *ctx);
const char *lentil_hip_last_error(const lentil_hip_ctx *ctx);
/* Diagnostics, no counterpart in the reference: why the last streamed pass of this context that was redone the chunked way
 * (lentil_counters::fallback_chunks) gave up -- which buffer bound or which waiting wave, with the pass's sizes.  "" if none
 * was.  The redo is invisible in the results (the frame is the reference's either way); it costs time. */
const char *lentil_hip_last_redo_note(const lentil_hip_ctx *ctx);

/* --- setup (once per render) -------------------------------------------------------
 * set_params : Camera::get_lentil_camera_params + camera_model_specific_setup results
 *              (src/lentil.h:1189-1243,1568-1670)
 * set_lens   : load_lens_constants.h / load_lt_sample_aperture.h / load_pt_evaluate.h
 *              splices (src/lentil.h:1262,1278,1308,1576)
 * set_bokeh  : imageData tables (src/imagebokeh.h:30-37) after bokehProbability
 * alloc_frame: Camera::setup_filter + AOVData::allocate_regular_buffers
 *              (src/lentil.h:1096-1121, src/aov_data.h:140-143).  AOV 0 is "RGBA" and is
 *              the one that feeds filter_weight_buffer (src/lentil.h:827-828).
 *              xres/yres are taken from the params. */
int lentil_hip_set_params(lentil_hip_ctx *ctx, const lentil_params *params);
int lentil_hip_set_lens(lentil_hip_ctx *ctx, const lentil_lens_table *lens);
int lentil_hip_set_bokeh(lentil_hip_ctx *ctx, const lentil_bokeh_table *bokeh);
/* The reference compiles every lens into the plugin (switch(lensModel) over generated code,
 * src/lentil.h:1262,1278,1308); liblentil_hip likewise carries straight-line kernels for the lens
 * tables it ships (tools/gen_lens_code.py) and recognises them by a hash of the table passed to
 * set_lens.  Any other table runs through the LDS table interpreter (same arithmetic, slower).
 * lens_is_compiled: 1 if the current table has a compiled-in kernel.  set_lens_mode: 0 = auto,
 * 1 = always interpret the table (parity tests compare the two). */
int lentil_hip_lens_is_compiled(lentil_hip_ctx *ctx);
int lentil_hip_set_lens_mode(lentil_hip_ctx *ctx, int mode);
/* A table that has no kernel compiled into the library gets one at run time -- the reference compiles every lens into the plugin
 * (include/auto_generated_lens_includes/load_lt_sample_aperture.h:4-47, used at src/lentil.h:1308): lentil_hip_set_lens starts a
 * thread that emits the lens's straight-line code, compiles it with hiprtc and keeps the code object in a cache on disk
 * (LENTIL_JIT_CACHE, default ~/.cache/lentil_hip); passes run the table interpreter until it is there.  Same results bit for
 * bit.  status: *state 0 nothing to compile (compiled-in lens, thin lens, LENTIL_LENS_JIT=0), 1 compiling, 2 the specialised
 * kernel is in use, -1 compilation failed (the interpreter keeps serving the lens); *compile_seconds: what the compilation took
 * (0 from the cache).  wait: blocks until the state is no longer 1 (timeout_seconds <= 0: no limit); an error if it failed.
 * debug_lens_jit_source: the emitted lens code (tests compare it with tools/gen_lens_code.py's). */
int lentil_hip_lens_jit_status(lentil_hip_ctx *ctx, int *state, double *compile_seconds);
int lentil_hip_lens_jit_wait(lentil_hip_ctx *ctx, double timeout_seconds);
int lentil_hip_debug_lens_jit_source(lentil_hip_ctx *ctx, char *buf, uint64_t capacity, uint64_t *length);
/* ... and without a context or a GPU (hiprtc cross-compiles): pack the table, emit the lens code and, compile != 0, compile the
 * four solve kernels; the emitted source and the compiler's log are copied out (truncated to the capacities). */
int lentil_hip_debug_lens_jit_compile(const lentil_lens_table *t, int compile, char *source, uint64_t source_capacity,
                                      uint64_t *source_length, char *log, uint64_t log_capacity, double *seconds, uint64_t *code_bytes);
int lentil_hip_alloc_frame(lentil_hip_ctx *ctx, uint32_t n_aovs, const uint8_t *aov_filter_kind);

/* A moving camera.  The reference asks Arnold for the matrix at every AOV sample's own time:
 * AiWorldToCameraMatrix(camera_node, lentil_time) (src/lentil_filter.cpp:141-144).  Here the caller hands over n_keys
 * world-to-camera matrices (row-vector convention like lentil_params::world_to_camera) at equidistant times over the
 * camera's shutter, shutter_start + k / (n_keys - 1) * (shutter_end - shutter_start) -- samples of AiWorldToCameraMatrix
 * at those absolute times -- and every visit uses the component-wise interpolation ((b - a) * f) + a of the two keys around
 * its lentil_time (raydir_time column, .w: Arnold's absolute sample time, inside the shutter; times outside it clamp to the
 * first / last key).  n_keys <= 1 or NULL: the static matrix of lentil_params again.  With keys the scan reads the
 * raydir_time column for every visit (80 instead of 64 bytes moved per visit) and runs register-staged.
 * set_camera_shutter: the shutter interval the keys span (Arnold's camera.shutter_start / .shutter_end; a centred shutter
 * is -0.25 ... 0.25); 0 ... 1 until set.  shutter_end must be greater than shutter_start. */
#define LENTIL_MAX_MOTION_KEYS 16
int lentil_hip_set_camera_motion(lentil_hip_ctx *ctx, uint32_t n_keys, const float *world_to_camera);
int lentil_hip_set_camera_shutter(lentil_hip_ctx *ctx, float shutter_start, float shutter_end);

/* Camera::logarithmic_focus_search (src/lentil.h:1445-1460; called per camera update at :1632): the sensor
 * shift, among the 20 001 candidates of logarithmic_values() (src/lens.h:395-407), whose axial ray crosses
 * the optical axis closest in front of focal_distance (mm).  One GPU lane per candidate instead of the
 * reference's sequential loop; the winner is the one that loop keeps (first smallest positive miss).
 * Uses the table of set_lens; lambda in micrometres. */
int lentil_hip_focus_search(lentil_hip_ctx *ctx, double focal_distance, double lambda, double *best_sensor_shift);

/* --- visit stream ------------------------------------------------------------------
 * upload_visits: host columns -> library-owned device memory (what the capturing
 *                filter_pixel hands over once per frame).
 * bind_visits  : columns already resident in HBM (device pointers owned by the caller). */
int lentil_hip_upload_visits(lentil_hip_ctx *ctx, const lentil_visits *host_visits);
int lentil_hip_bind_visits(lentil_hip_ctx *ctx, const lentil_visits *device_visits);

/* The same hand-over piece by piece, while the renderer is still rendering: a capturing filter_pixel
 * (src/lentil_filter.cpp:66-436 runs per pixel on many bucket threads) fills blocks of visits and sends each
 * block as it fills, so that the PCIe transfer (80 B per visit: 0.11 s for a 4K frame in one piece) hides
 * behind the render and the frame end waits for the last block only.
 * host_alloc / host_free : page-locked host memory for those blocks (copies from it are asynchronous DMA;
 *                ordinary memory works too, the copy then returns when the runtime has staged it).
 * visits_begin : starts a frame's stream.  `layout` gives its geometry (visits_per_pixel, pixels_per_row,
 *                pixel_x0/y0, pixel_row_stride, n_extra; n and the column pointers are ignored, except that
 *                inv_density != NULL announces per-visit densities for a ragged stream).  capacity_hint:
 *                expected number of visits (0: unknown; the columns grow by doubling).  Waits for the
 *                previous pass; the previous frame's device columns are reused when they fit.
 * visits_append: part->n visits (host columns) go to the end of the stream, in call order; thread-safe.
 *                Returns at once; *ticket (may be NULL) identifies the copies.
 * visits_wait  : returns when the copies of that ticket are complete -- the block may be refilled.
 * visits_end   : waits for all copies and makes the assembled stream the context's visits (as upload_visits
 *                would have); *n_visits (may be NULL) returns their number. */
int lentil_hip_host_alloc(void **host_ptr, uint64_t bytes);
int lentil_hip_host_free(void *host_ptr);
int lentil_hip_visits_begin(lentil_hip_ctx *ctx, const lentil_visits *layout, uint64_t capacity_hint);
int lentil_hip_visits_append(lentil_hip_ctx *ctx, const lentil_visits *part, uint64_t *ticket);
int lentil_hip_visits_wait(lentil_hip_ctx *ctx, uint64_t ticket);
int lentil_hip_visits_end(lentil_hip_ctx *ctx, uint64_t *n_visits);

/* --- the hot path ------------------------------------------------------------------
 * clear_frame : zero-initialisation done by std::vector::resize (src/lentil.h:1096-1098)
 * redistribute: the filter_pixel visit loop (src/lentil_filter.cpp:91-451) for every bound
 *               visit: predicate + draw count, trace_ray_bw_po / thin-lens draws,
 *               add_to_buffer / filter_and_add_to_buffer_new (src/lentil.h:823-851,938-955).
 *               Accumulates on top of whatever the frame holds (call clear_frame first).
 *               Returns when the pass is complete on the device: it ends with a read-back of the
 *               pass's counters (did everything fit the buffers sized from the previous pass? is any
 *               visit still short of draws?), after which it redoes or continues what is needed.
 *               LENTIL_ERR_NOMEM if the device had to drop work (a frame that is incomplete).
 *               clear_frame, resolve and the row / exchange calls are asynchronous on the context's
 *               stream (lentil_hip_stream); downloads and lentil_hip_sync wait for it.
 * resolve     : driver_process_bucket's normalisation (src/lentil_imager.cpp:112-118,169-186)
 *               into a separate resolved image (the accumulators stay intact).  A streamed pass has
 *               usually done it on its way (beside its second round); the call then costs nothing. */
int lentil_hip_clear_frame(lentil_hip_ctx *ctx);
int lentil_hip_redistribute(lentil_hip_ctx *ctx);
int lentil_hip_resolve(lentil_hip_ctx *ctx);
int lentil_hip_sync(lentil_hip_ctx *ctx);

/* --- results -----------------------------------------------------------------------
 * download_aov     : resolved AOV as xres*yres RGBA floats (what the imager writes into
 *                    bucket_data, src/lentil_imager.cpp:178,182)
 * download_accum   : raw accumulators: AOVData::buffer (xres*yres*4) and
 *                    filter_weight_buffer (xres*yres); either pointer may be NULL */
int lentil_hip_download_aov(lentil_hip_ctx *ctx, uint32_t aov, float *host_rgba);
int lentil_hip_download_accum(lentil_hip_ctx *ctx, uint32_t aov, float *host_rgba, float *host_weight);
/* download_records : every AOV's accumulators and the weight in ONE copy, as the device keeps them: xres*yres records of
 *                    *stride floats -- AOV a's RGBA at floats 4a .. 4a+3, the weight at float 4 * n_aovs, padding behind it.
 *                    host_records holds capacity_floats floats (LENTIL_ERR_INVALID if that is fewer than xres*yres * stride;
 *                    host_records NULL: only *stride is set).  download_accum copies this whole block per call and picks
 *                    one AOV's columns: a caller that looks at all AOVs of a 4K frame wants this instead. */
int lentil_hip_download_records(lentil_hip_ctx *ctx, float *host_records, uint64_t capacity_floats, uint32_t *stride);

/* Thin lens with abb_chromatic > 0 (src/lentil_filter.cpp:393-406): every attempt that passes the optical
 * vignetting test draws its colour channel from xor128 (src/global.h:22-27), whose state the reference
 * keeps in function statics -- one per process, advanced by whichever thread gets there first.  Here the
 * order is the single-threaded one (visits in stream order, attempts in order) and the state is explicit:
 * it starts at the generator's initial constants, every lentil_hip_redistribute continues from where the
 * previous one stopped, and a host that wants another starting point (or to mirror draws it made itself)
 * reads / writes the four words x, y, z, w.  One GPU only. */
int lentil_hip_set_xor128_state(lentil_hip_ctx *ctx, const uint32_t state[4]);
int lentil_hip_get_xor128_state(lentil_hip_ctx *ctx, uint32_t state[4]);

/* Closest-filtered AOVs across GPUs (the reference's single z-buffer, src/lentil.h:832-837).
 * deferred != 0: lentil_hip_redistribute leaves the per-pixel winner keys -- (bits of |Z|) << 32 |
 * (0xFFFFFFFF - frame-wide visit id), empty = all ones -- in lentil_hip_zkey_buffer instead of gathering
 * the winners' values.  The caller takes the unsigned 64-bit minimum of that buffer over all GPUs, then
 * calls lentil_hip_closest_gather on each: a GPU writes the values of the winners it owns and leaves the
 * others zero, so the sum all-reduce of lentil_hip_accum_buffer completes the closest AOVs too.
 * Frame-wide visit ids: uniform streams derive them from pixel_y0 / pixel_row_stride (the id a single
 * process walking the whole frame would give the visit -- row bands and row-interleaved partitions alike);
 * ragged streams use visit_id_base + index. */
int lentil_hip_set_closest_exchange(lentil_hip_ctx *ctx, int deferred, uint32_t visit_id_base);
int lentil_hip_zkey_buffer(lentil_hip_ctx *ctx, void **device_ptr, uint64_t *n_keys);
int lentil_hip_closest_gather(lentil_hip_ctx *ctx);

/* --- multi-GPU, tiled output (BASELINE north_star: "the output frame tiles across the GPUs,
 * cross-tile splat contributions are exchanged") ------------------------------------------------------
 * Every GPU processes the visits of its band of rows into full-frame accumulators; what its draws add
 * outside the band belongs to the band's owner.
 * touched_rows : [row_lo, row_hi) of the frame the last redistribute added anything to (own visits and
 *                splats; 0,0 when nothing).  Also lets the next clear_frame wipe only those rows.
 * merge_rows   : merges n_rows rows, starting at row_begin, of another GPU's accumulator block (device
 *                memory, same record layout: xres * stride floats per row; stride = n_floats of
 *                accum_buffer / (xres * yres)) into this frame: gaussian slots and the weight add up,
 *                closest-filtered slots follow the smaller winner key (dev_key_rows: the matching rows of
 *                the sender's lentil_hip_zkey_buffer, required iff the frame has closest AOVs; the sender
 *                must have gathered its local winners, i.e. not be in deferred mode).
 * pack_rows / merge_packed_rows : the same exchange without the padding of the pixel records: pack_rows writes
 *                n_rows * xres * (4 n_aovs + 1) floats (a record's RGBA values and weight, back to back) to
 *                dev_dst; merge_packed_rows merges rows that arrived in that form.  5 instead of 8 floats per
 *                pixel for a beauty-only frame.
 * compact_rows / merge_sparse : the same exchange for rows that are mostly empty (a band's draws reach ~100 rows into
 *                its neighbours but touch ~2 % of their pixels when highlights are rare): compact_rows turns every
 *                pixel of the rows that holds anything (weight != 0, or a winner key) into one entry -- frame-wide
 *                pixel index (uint32), its 4 n_aovs + 1 floats, its key (uint64, only with closest AOVs) -- in three
 *                device arrays of `capacity` entries and returns the number of entries found (synchronises; if it
 *                exceeds capacity nothing beyond capacity was written: send the rows whole instead).  merge_sparse
 *                merges n such entries of one sender, all inside rows [row_begin, row_begin + n_rows).
 * resolve_rows : lentil_hip_resolve restricted to a band of rows. */
int lentil_hip_touched_rows(lentil_hip_ctx *ctx, int32_t *row_lo, int32_t *row_hi);
int lentil_hip_merge_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, const void *dev_acc_rows,
                          const void *dev_key_rows);
int lentil_hip_pack_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, void *dev_dst);
int lentil_hip_merge_packed_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, const void *dev_packed_rows,
                                 const void *dev_key_rows);
int lentil_hip_compact_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, void *dev_idx, void *dev_vals,
                            void *dev_keys, uint32_t capacity, uint32_t *count);
int lentil_hip_merge_sparse(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, uint32_t n, const void *dev_idx,
                            const void *dev_vals, const void *dev_keys);
int lentil_hip_resolve_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows);

/* --- multi-GPU ---------------------------------------------------------------------
 * accum_buffer: device pointer + float count of the contiguous accumulator block (one record per
 *               pixel: n_aovs x RGBA, the filter weight, padding to a multiple of 8 floats); a sum
 *               all-reduce over it (RCCL) merges the cross-tile splats of all GPUs (SURVEY.md
 *               section 8e).  The reference has no counterpart (single process, shared buffers). */
int lentil_hip_accum_buffer(lentil_hip_ctx *ctx, void **device_ptr, uint64_t *n_floats);
int lentil_hip_stream(lentil_hip_ctx *ctx, void **hip_stream);

/* --- multi-GPU, the exchange itself (RCCL over xGMI; one process per GPU, one context per process) ------
 * The library owns its communicator: librccl.so is loaded on the first of these calls (dlopen; a process
 * that already holds an RCCL gets that same instance), so a renderer linking liblentil_hip.so for one GPU
 * needs no RCCL.  No reference counterpart (src/lentil.h:823-851 adds into buffers all threads share).
 * comm_unique_id : rank 0 fills the 128-byte id (ncclGetUniqueId) and hands it to the other processes by
 *                  whatever channel the host has (the renderer's own job description, a file, MPI ...).
 * comm_init      : collective over `world` processes, rank in [0, world); replaces an earlier communicator.
 * allreduce      : row-interleaved partition, every GPU holds the whole frame.  After lentil_hip_redistribute
 *                  on every rank: closest-AOV winner keys are min-reduced and the winners gathered (requires
 *                  set_closest_exchange(deferred = 1) when the frame has closest AOVs), then one sum all-reduce
 *                  of accum_buffer.  lentil_hip_resolve afterwards gives every rank the whole image.
 * exchange_bands : tiled output.  Rank r's visits are rows [bounds[r], bounds[r+1]) (bounds == NULL: an even split
 *                  of visit_rows); its band of the frame is the same rows, the last band reaching to yres.
 *                  After lentil_hip_redistribute on every rank: the touched rows are all-gathered, what a
 *                  rank added to another's band travels point to point (sparse != 0: as pixel entries when
 *                  at most a quarter of the pixels hold anything, else as packed rows), the owner merges
 *                  the arrivals in rank order and resolves its band; band_lo / band_hi (may be NULL) return it.
 *                  Collective: every rank calls it with the same bounds / visit_rows / sparse. */
int lentil_hip_comm_unique_id(uint8_t id[128]);
int lentil_hip_comm_init(lentil_hip_ctx *ctx, const uint8_t id[128], int rank, int world);
int lentil_hip_comm_destroy(lentil_hip_ctx *ctx);
int lentil_hip_allreduce(lentil_hip_ctx *ctx);
int lentil_hip_exchange_bands(lentil_hip_ctx *ctx, const int32_t *bounds, int32_t visit_rows, int32_t sparse,
                              int32_t *band_lo, int32_t *band_hi);
/* payload bytes this rank sent / received over xGMI in its last lentil_hip_exchange_bands (pixel entries or packed rows,
 * winner keys included) or lentil_hip_allreduce (ring traffic of the reduced buffers); either pointer may be NULL.
 * Instrumentation for the scaling bench; no reference counterpart. */
int lentil_hip_exchange_stats(lentil_hip_ctx *ctx, uint64_t *bytes_sent, uint64_t *bytes_received);
/* lentil_hip_exchange_bands(sparse != 0) since the communicator was made: exchanges that ran in the fixed-capacity form
 * (every message's size known to both ends beforehand, no host wait between compaction and merge) and directed pairs whose
 * entries did not fit their message and followed as whole rows; either pointer may be NULL.  Instrumentation. */
int lentil_hip_exchange_counts(lentil_hip_ctx *ctx, uint64_t *fixed_form, uint64_t *pairs_overflowed);
/* *concurrent = 1 when the four HIP streams a streamed pass keeps kernels resident on run them side by side (they were
 * chosen so at lentil_hip_create: streams that share one of the runtime's hardware queues serialise, and the pass then
 * takes about 1.5 x as long; 2: and a fifth one beside them, on which a pass with cryptomatte AOVs replays the own-pixel
 * adds while its draws go on -- GPU_MAX_HW_QUEUES > 4), 0 when fewer than four hardware queues were to be had or
 * LENTIL_STREAM_PROBE=0.
 * Instrumentation; no reference counterpart. */
int lentil_hip_streams_concurrent(lentil_hip_ctx *ctx, int *concurrent);
/* What this GPU delivers right now, for putting timings of different boxes side by side (bench.py's `box` block; no reference
 * counterpart).  Three short kernels on the context's stream, ~10 ms in all:
 *   probe[0]  fp64 multiply / add rate of dependent chains, three waves per SIMD on every CU -- the form of the Newton solves
 *             (no FMA) -- in TFLOP/s; [1] the shader clock it ran at, MHz (clock64() against the 100 MHz real-time counter);
 *   probe[2]  float4 copy, GB/s read + written; [3] float4 read-only stream, GB/s;
 *   probe[4]  hardware queues the runtime multiplexes this process's streams onto (GPU_MAX_HW_QUEUES as the library sees it;
 *             0: unset, the runtime's default of 4); [5] compute units. */
int lentil_hip_box_probe(lentil_hip_ctx *ctx, double probe[6]);

/* --- cryptomatte AOVs ------------------------------------------------------------------
 * The reference keeps a std::map<float, float> id -> weight and a total weight per pixel for every ranked
 * cryptomatte AOV (AOVData::crypto_hash_map / crypto_total_weight, src/aov_data.h:127-128,145-150).  Each add of a
 * visit to a pixel -- its own pixel (src/lentil.h:952) or the pixel of an accepted draw
 * (src/lentil_filter.cpp:296,443) -- adds the visit's cache {id -> weight} times the sample weight
 * (Camera::add_to_buffer_cryptomatte, src/lentil.h:814-819); the imager writes the pairs at positions rank, rank + 1
 * of a pixel's map sorted by weight (src/lentil_imager.cpp:121-161).
 *
 * lentil_crypto_visits carries, per cryptomatte AOV, the cache of every visit of the bound stream as `entries`
 * (id, weight) pairs (what Camera::cryptomatte_construct_cache, src/lentil.h:781-811, leaves: ids distinct; a pair
 * whose weight has the bits 0xFFFFFFFF is unused; -0 counts as +0, which is one key to the reference's map too).
 * Host arrays with lentil_hip_upload_crypto, device arrays with lentil_hip_bind_crypto; after the visits they belong
 * to (binding other visits makes them stale: the pass then refuses), before lentil_hip_redistribute.  lentil_hip_alloc_crypto follows lentil_hip_alloc_frame (which drops the tables);
 * slots_per_pixel is the number of distinct ids a pixel can hold (0: 16, at most 64) -- a pass that meets more
 * returns LENTIL_ERR_NOMEM, as does one whose accepted draws exceed the draw log the adds are replayed from (sized
 * by the library from the previous pass unless lentil_hip_set_draw_log was called; clear the frame and redistribute
 * again).  lentil_hip_clear_frame empties the tables.  Across GPUs the tables travel with the tiled exchange:
 * lentil_hip_exchange_bands sends the map entries this rank's draws added outside its band to the bands' owners (16-byte
 * records) and adds what arrives; lentil_hip_allreduce (interleaved rows) refuses contexts with cryptomatte AOVs.
 *
 * lentil_hip_download_crypto: np RGBA = (id, weight / total) of positions rank and rank + 1 -- rank 0 / 2 / 4 for
 * the AOVs named ...00 / ...01 / ...02 (:124-126); equal weights stay in id order (what std::sort does to the up
 * to 16 pairs it sorts by insertion).  host_has_rank (np bytes, may be NULL): 0 where the pixel's map has no more
 * than `rank` entries -- the reference stops copying the bucket row at such a pixel (:132-134).
 * lentil_hip_download_crypto_table: the raw tables (np * slots ids as bits, 0xFFFFFFFF = free, and weights; np
 * totals); any pointer may be NULL. */
typedef struct lentil_crypto_visits {
  uint64_t n;                               /* visits (the bound stream's n) */
  uint32_t n_crypto;                        /* cryptomatte AOVs */
  uint32_t entries;                         /* pairs per visit and AOV, 1..64 */
  const float *hash[LENTIL_MAX_CRYPTO];     /* n * entries ids */
  const float *weight[LENTIL_MAX_CRYPTO];   /* n * entries weights */
} lentil_crypto_visits;
int lentil_hip_alloc_crypto(lentil_hip_ctx *ctx, uint32_t n_crypto, uint32_t slots_per_pixel);
int lentil_hip_upload_crypto(lentil_hip_ctx *ctx, const lentil_crypto_visits *c);
int lentil_hip_bind_crypto(lentil_hip_ctx *ctx, const lentil_crypto_visits *c);
/* the caches with a piecewise upload (lentil_hip_visits_begin ... _end above): lentil_hip_visits_begin_crypto right
 * after _visits_begin announces `entries` pairs per visit and cryptomatte AOV; every part then goes through
 * lentil_hip_visits_append_crypto with its caches (caches->n == part->n; page-locked like the part); _visits_end
 * makes them the context's cryptomatte columns */
int lentil_hip_visits_begin_crypto(lentil_hip_ctx *ctx, uint32_t entries);
int lentil_hip_visits_append_crypto(lentil_hip_ctx *ctx, const lentil_visits *part, const lentil_crypto_visits *caches,
                                    uint64_t *ticket);
int lentil_hip_download_crypto(lentil_hip_ctx *ctx, uint32_t crypto, uint32_t rank, float *host_rgba, uint8_t *host_has_rank);
int lentil_hip_download_crypto_table(lentil_hip_ctx *ctx, uint32_t crypto, uint32_t *slots_per_pixel, uint32_t *host_id_bits,
                                     float *host_weight, float *host_total);

/* --- instrumentation ----------------------------------------------------------------
 * timings are HIP-event times on the context's stream for the last redistribute/resolve:
 * ms[0] scan+compaction+direct accumulate, ms[1] draw/splat kernel, ms[2] resolve. */
int lentil_hip_get_counters(lentil_hip_ctx *ctx, lentil_counters *out);
int lentil_hip_last_timing(lentil_hip_ctx *ctx, float ms[3]);
/* kernel launches of the last redistribute: n[0] scan launches (one per chunk of the visit stream; ms[0]
 * of last_timing covers all of them), n[1] solve/accept rounds of the chunk that needed most. */
int lentil_hip_last_launches(lentil_hip_ctx *ctx, uint32_t n[2]);
/* First batches sized from the lens and the frame (round 5).  The reference's loop (src/lentil_filter.cpp:248-299) keeps
 * tracing until `samples` draws of a sample have landed inside the frame, at most 5 x samples attempts of up to
 * vignetting_retries + 1 tries each (src/lentil.h:592-648); this library computes every trace once, in batches, and sizes a
 * sample's FIRST batch from a calibration of the lens (where its rays land, which it vignettes) so that a streamed pass needs
 * no second round of traces.  No result depends on it: surplus traces are never looked at, a shortfall is served by further
 * rounds.  stats[0] calibrations run, [1] passes that ran without a second round in flight, [2] ... of which needed one
 * after all, [3] sixteenths by which the model's margin has been widened since the camera set-up last changed.
 * debug_batch_estimate: the model's answer for n camera-space points (cm, z < 0 in front of the camera) and a draw count:
 * out[n][4] = share of the passing aperture points that land well inside the frame, share that land inside it, share the lens
 * vignettes, traces in the first batch.  LENTIL_ERR_INVALID without a polynomial-optics lens and parameters. */
int lentil_hip_batch_model_stats(lentil_hip_ctx *ctx, uint64_t stats[4]);
/* What the streamed passes of ALL contexts of this process have met (no context needed): stats[0] streamed passes begun,
 * [1] passes whose resident waves gave up waiting (the stuck time-out; the pass is then redone in the chunked form and its
 * result is the same), [2] ... of which were asked for (LENTIL_INJECT_STALL), [3] passes wiped and run again because draws
 * had been accepted by then.  A stall costs time, never results -- so nothing else would ever show one: the GPU test
 * session asserts stats[1] == stats[2] when it ends (tests/conftest.py). */
int lentil_hip_process_stats(uint64_t stats[4]);
/* ... and what the waves that gave up saw (lentil_hip_last_redo_note's text, one line per pass, the first eight that were not
 * asked for), NUL-terminated, truncated to capacity. */
int lentil_hip_process_stall_notes(char *buf, uint64_t capacity);
/* Occlusion probes (round 6).  The reference asks the renderer, before every backward trace, whether anything stands between the
 * sample and the point of the aperture the trace is to go through -- AiTraceProbe along the segment from the sample's world
 * position to cam_to_world * (-aperture * 0.1 / unit) (src/lentil.h:613-629; thin lens: to cam_to_world * (lens / unit),
 * src/lentil_filter.cpp:356-375) -- and a try that is occluded fails like one the lens vignettes (samples the skydome supplied are
 * exempt).  The GPU has no scene.  With a probe set, a pass runs in its round-by-round form, and in every round, once the
 * round's traces are solved and before any of them is accepted, the library hands the host ONE list of segments -- one per try
 * of the round that got through the lens (whether it landed in the frame or not: an occluded try hands its attempt on to the
 * next try) -- and the host answers one byte per segment, non-zero = occluded; the accept then treats those tries as failed.
 * Results are what the reference computes with the same probe (oracle: orc_frame_set_probe; tests/test_gpu_probe.py).
 *   fn(user, n, segments, occluded): called from the thread that called lentil_hip_redistribute, once per round and chunk;
 *   it may spread the n probes over threads of its own.  origin / target are world-space points; the reference's ray is
 *   AiMakeRay(AI_RAY_SHADOW, origin, normalize(target - origin), |target - origin|).
 *   camera_to_world: 16 floats, row-vector convention like lentil_params::world_to_camera (AiCameraToWorldMatrix); NULL: the
 *   inverse of params.world_to_camera (of every motion key, where lentil_hip_set_camera_motion has set keys), computed in fp64.
 *   fn == NULL switches probing off.  Not with abb_chromatic != 0 (LENTIL_ERR_UNSUPPORTED from the pass): the reference draws a
 *   thin-lens attempt's colour channel AFTER its probe, from one generator in visit order.
 * Cost: INTEGRATION.md section 3c (the list travels over PCIe: 24 B out and 1 B back per try that got through the lens). */
typedef struct lentil_probe_segment {
  float origin[3];   /* the sample, world space */
  float target[3];   /* the point on the aperture, world space */
} lentil_probe_segment;
typedef void (*lentil_probe_fn)(void *user, uint64_t n, const lentil_probe_segment *segments, uint8_t *occluded);
int lentil_hip_set_occlusion_probe(lentil_hip_ctx *ctx, lentil_probe_fn fn, void *user, const float *camera_to_world);
/* stats[0] segments handed to the callback since the context was created, [1] of them answered "occluded", [2] callback calls */
int lentil_hip_probe_stats(lentil_hip_ctx *ctx, uint64_t stats[3]);

/* The asynchronous end of a pass (round 6).  lentil_hip_redistribute no longer ends with the host waiting for the device: a
 * streamed pass returns once its kernels are enqueued, and whether it needs more work (buffers that were too small, a draw
 * batch that fell short, a wave that gave up waiting) is found out by the next call that OBSERVES the context -- every entry
 * point except lentil_hip_clear_frame, _bind_visits, _redistribute and _resolve: that call waits for the pass, does what is
 * left and only then proceeds, so lentil_hip_sync, the downloads, _get_counters, _last_timing ... return what they always
 * returned (an error of the pass -- LENTIL_ERR_NOMEM for dropped work -- is reported by that call).  A caller that pipelines
 * frames (clear, redistribute, resolve, clear, ...) keeps the device fed instead: the next frame's clear and scan are
 * enqueued while the pass still runs.  Clearing a frame that nobody has observed abandons its pass: the counters are still read
 * (lentil_hip_pass_totals; they size the next passes), work it still needed is not done, and totals.abandoned_incomplete says
 * so.  The visit columns of a pass must stay valid until the pass has been observed or its frame cleared AND the device has
 * passed it (lentil_hip_sync).  At most two passes are in flight unobserved; a third waits for the oldest.
 * set_async(0): every lentil_hip_redistribute waits for its own end, as before round 6 (also LENTIL_ASYNC_END=0).
 * pass_totals: waits for nothing but the events of passes already observed or abandoned; reset != 0 zeroes the sums afterwards. */
int lentil_hip_set_async(lentil_hip_ctx *ctx, int on);
int lentil_hip_pass_totals(lentil_hip_ctx *ctx, lentil_pass_totals *out, int reset);
int lentil_hip_debug_batch_estimate(lentil_hip_ctx *ctx, uint64_t n, const float *cs_xyz, uint32_t samples, float *out);
int lentil_hip_set_draw_log(lentil_hip_ctx *ctx, uint64_t capacity); /* 0 disables */
int lentil_hip_download_draw_log(lentil_hip_ctx *ctx, lentil_draw_record *out, uint64_t capacity,
                                 uint64_t *n_records);

/* --- single-function device tests (parity of the optics primitives) -----------------
 * Runs n independent evaluations on the GPU; host pointers in/out.
 * lt_sample_aperture: Camera::lens_lt_sample_aperture (src/lentil.h:1296-1313) for
 *    scene[n][3], ap[n][2] -> sensor[n][5], out[n][5], transmittance[n]
 * trace_bw_po: Camera::trace_ray_bw_po (src/lentil.h:573-661) for target[n][3]
 *    (already -P_cs*10), px[n], py[n], attempt[n] -> sensor_xy[n][2], ok[n]
 * aperture_sample: the aperture draw of trace_ray_bw_po (src/lentil.h:596-609) for
 *    seed pairs (a[n] = px*py+px, b[n] = total_samples_taken+tries) -> xy[n][2]
 * y0_intersection: Camera::camera_get_y0_intersection_distance (src/lentil.h:1361-1386) for
 *    sensor_shift[n] -> distance[n], and what lens_pt_sample_aperture / lens_evaluate (:1257-1291)
 *    left on the way: sensor[n][5] (x, y, dx, dy, lambda), out[n][5] (x, y, dx, dy, transmittance);
 *    sensor / out may be NULL */
int lentil_hip_test_y0_intersection(lentil_hip_ctx *ctx, uint64_t n, const double *sensor_shift, double lambda,
                                    double *distance, double *sensor, double *out);
int lentil_hip_test_lt_sample_aperture(lentil_hip_ctx *ctx, uint64_t n, const double *scene,
                                       const double *ap, double lambda, double *sensor,
                                       double *out, double *transmittance);
int lentil_hip_test_trace_bw_po(lentil_hip_ctx *ctx, uint64_t n, const double *target,
                                const int32_t *px, const int32_t *py, const int32_t *attempt,
                                double *sensor_xy, int32_t *ok);
int lentil_hip_test_aperture_sample(lentil_hip_ctx *ctx, uint64_t n, const uint32_t *a,
                                    const uint32_t *b, double *xy);
/* Test hook, needs no GPU.  The scan decides `get_coc_thinlens(z) < 0.4` (src/lentil.h:674-692,
 * src/lentil_filter.cpp:185-190) from the camera-space depth alone wherever that is certain; this returns the
 * intervals it would use for `params`: out[0..1] / out[2..3] lower / upper ends of the (at most two) closed depth
 * intervals on which the circle of confusion is certainly below 0.4, out[4..5] / out[6..7] those outside of which it is
 * certainly not (an interval with lower > upper end is empty).  In between the kernel evaluates the function. */
int lentil_hip_debug_scan_bands(const lentil_params *params, float out[8]);

#ifdef __cplusplus
}
#endif
#endif /* LENTIL_HIP_H */
