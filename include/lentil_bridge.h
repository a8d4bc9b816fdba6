/* lentil_bridge.h -- host side of the drop-in, above the C-ABI of lentil_hip.h.
 *
 * The reference is an Arnold plugin; its node callbacks mix three things: calls into the Arnold SDK
 * (not in this image), plain host logic (parameter mapping, output-string rewiring, per-sample
 * capture, bucket copy) and the hot path.  This library is the plain host logic, SDK-free, with the
 * argument meaning and behaviour of the callbacks it is taken from, so that the plugin glue a
 * maintainer writes against <ai.h> is reduced to fetching values from Arnold and passing them on
 * (INTEGRATION.md).  Every function cites the reference code it stands for (paths relative to the
 * reference repository).
 *
 * C, no C++ types, no Arnold types, no HIP types.  AI_TYPE_* codes are passed as plain ints with the
 * SDK's values (LENTIL_AI_TYPE_* below). */
#ifndef LENTIL_BRIDGE_H
#define LENTIL_BRIDGE_H

#include <stddef.h>
#include <stdint.h>

#include "lentil_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Arnold parameter / pixel type codes (ai_params.h) used by the callbacks mirrored here */
#define LENTIL_AI_TYPE_INT 0x01
#define LENTIL_AI_TYPE_BOOLEAN 0x03
#define LENTIL_AI_TYPE_FLOAT 0x04
#define LENTIL_AI_TYPE_RGB 0x05
#define LENTIL_AI_TYPE_RGBA 0x06
#define LENTIL_AI_TYPE_VECTOR 0x07
#define LENTIL_AI_TYPE_STRING 0x0A
#define LENTIL_AI_TYPE_ENUM 0x0F
#define LENTIL_AI_TYPE_NONE 0xFF

/* ------------------------------------------------------------------------------------
 * lentil_camera node: node_parameters (src/lentil_camera.cpp:19-52) and
 * Camera::get_lentil_camera_params (src/lentil.h:1189-1243)
 * ---------------------------------------------------------------------------------- */
typedef struct lentil_node_param {
  const char *name;
  int type;                       /* LENTIL_AI_TYPE_* */
  double default_value;           /* numbers, bools, enum index */
  const char *default_string;     /* strings only, else NULL */
  const char *const *enum_values; /* NULL-terminated for enums, else NULL */
} lentil_node_param;

/* the camera node's parameters in declaration order; *count receives their number (29) */
const lentil_node_param *lentil_camera_node_parameters(int *count);

/* values of those parameters as a plain struct (field = parameter name) */
typedef struct lentil_camera_node_values {
  int camera_type;               /* 0 ThinLens, 1 PolynomialOptics */
  int bidir_sample_mult;
  int units;                     /* 0 mm, 1 cm, 2 dm, 3 m, 4 automatic */
  float sensor_width;
  int enable_dof;
  float fstop;
  float focus_dist;
  int aperture_blades_lentil;
  float exp;
  int lens_model;
  float wavelength;              /* nm */
  float extra_sensor_shift;
  float focal_length_lentil;
  float optical_vignetting;
  float abb_spherical, abb_distortion, abb_coma, abb_chromatic;
  int abb_chromatic_type;
  float bokeh_circle_to_square, bokeh_anamorphic;
  int bokeh_enable_image;
  const char *bokeh_image_path;
  int vignetting_retries;
  float bidir_add_energy, bidir_add_energy_minimum_luminance, bidir_add_energy_transition;
  int enable_bidir_transmission, enable_skydome;
} lentil_camera_node_values;

void lentil_camera_node_defaults(lentil_camera_node_values *v);

/* get_lentil_camera_params: node values + the two render options the reference reads
 * (options.meters_per_unit for units == automatic, options.ignore_dof) -> the lentil_params fields that
 * come straight from the node (clamps included), plus what the model-specific setup consumes
 * (input_fstop, lambda in micrometres, extra_sensor_shift, exposure).  Resolution / region / filter
 * fields of `out` are left untouched (setup_filter owns them).  Returns 0. */
int lentil_camera_params_from_node(const lentil_camera_node_values *v, float meters_per_unit, int ignore_dof,
                                   lentil_params *out, double *input_fstop, double *lambda_um,
                                   double *extra_sensor_shift, float *exposure);

/* ------------------------------------------------------------------------------------
 * lentil_filter node (src/lentil_filter.cpp:14-63)
 * ---------------------------------------------------------------------------------- */
const char *const *lentil_filter_required_aovs(void);      /* NULL-terminated "TYPE name" list, :16-26 */
float lentil_filter_width(int oidn_imager_present);        /* node_update: 1.0 with an OIDN imager else 1.5, :30-40 */
int lentil_filter_output_type(int input_type);             /* RGBA/RGB/VECTOR/FLOAT -> RGBA, else NONE, :45-63 */

/* the visit prologue of filter_pixel (src/lentil_filter.cpp:79-88): from the number of AOV samples in
 * the pixel's footprint -> inverse sample density, and whether this pixel disables redistribution for
 * the frame (AA below the final AA, or AA_samples < 3).  adaptive sampling: density comes per sample. */
float lentil_filter_inverse_sample_density(int samples_in_footprint, float filter_width, int aa_samples_set_by_user,
                                           int *disable_redistribution);

/* Camera::setup_filter, the resolution / region part (src/lentil.h:1060-1080): region_* as options hold them
 * (INT32_MIN / INT32_MAX = unset -> the whole frame, with region_max = res: the W + 1 quirk); fills xres, yres,
 * xres_without_region, yres_without_region, region_min_x/y and filter_width of `p`. */
void lentil_setup_filter_region(lentil_params *p, int xres, int yres, int region_min_x, int region_min_y, int region_max_x,
                                int region_max_y, float filter_width);

/* The display pass-through of filter_pixel (src/lentil_filter.cpp:453-479): what Arnold shows until the imager
 * overwrites it.  Camera::filter_gaussian_complete (src/lentil.h:738-775): samples within the filter radius weighted
 * by AiFastExp(-2 r) * inverse density (the caller hands the SDK's function in; without one, expf -- display only) -- offsets_xy 2 n,
 * values_rgba 4 n, inv_density n or NULL for a uniform value.  Camera::filter_closest_complete (src/lentil.h:696-735):
 * the value of the sample with the smallest |depth| (later samples win ties; depth 0 re-opens), alpha 1. */
/* lens_model (src/lentil_camera.cpp:13-16,29, src/lentil.h:62-65,1212): the reference's 44 lens ids in its order, then
 * this build's own tables.  _name: the id's string (NULL when out of range); _table: the name of the shipped table that
 * stands in for it, NULL when this build ships none for that id (the camera update then refuses the id by name). */
const char *lentil_lens_model_name(int lens_model);
const char *lentil_lens_model_table(int lens_model);

typedef float (*lentil_exp_fn)(float);   /* the SDK's AiFastExp where there is one (lentil.so passes it); NULL: expf */
void lentil_filter_gaussian_complete(int n, const float *offsets_xy, const float *values_rgba, const float *inv_density,
                                     float uniform_inv_density, float filter_width, lentil_exp_fn fast_exp, float out_rgba[4]);
void lentil_filter_closest_complete(int n, const float *depth, const float *values_rgba, float out_rgba[4]);

/* ------------------------------------------------------------------------------------
 * lentil_operator (src/lentil_operator.cpp:25-127) and the output-string tokens it edits
 * (TokenizedOutputLentil, src/aov_data.h:12-115)
 * ---------------------------------------------------------------------------------- */
typedef struct lentil_output_tokens {
  char camera[128], aov_name[128], aov_type[32], filter[128], driver[128];
  int half_flag;
} lentil_output_tokens;

void lentil_tokenize_output(const char *output_string, lentil_output_tokens *tok);
/* returns the length written (excluding the terminator), or -1 when buf is too small */
int lentil_rebuild_output(const lentil_output_tokens *tok, char *buf, size_t cap);
unsigned lentil_string_to_arnold_type(const char *s);      /* src/global.h:58-65 */

typedef struct lentil_aov_plan {
  lentil_output_tokens to;
  char name[128];
  unsigned type;                  /* LENTIL_AI_TYPE_* or 0 */
  int original_filter;            /* LENTIL_FILTER_GAUSSIAN / _CLOSEST / _VARIANCE */
  int is_duplicate;
  int is_crypto;
  int index;
} lentil_aov_plan;

/* operator_cook's AOV list: for every options.outputs string (with the node-entry name of the filter
 * node it references, e.g. "gaussian_filter") one plan entry -- filter token replaced by
 * "lentil_replaced_filter" where lentil filters the AOV, incompatible filters downgraded to gaussian
 * (a warning line per case is appended to `warnings`), duplicates flagged -- followed by the three
 * AOVs the operator adds as copies of the first output: lentil_debug (FLOAT, closest), lentil_time
 * (FLOAT), lentil_raydir (RGB).  Ranked crypto_* AOVs are skipped like the reference does.
 * Returns the number of plan entries, or -1 when cap is too small / n == 0. */
int lentil_operator_cook(const char *const *outputs, const char *const *filter_entry_names, int n,
                         lentil_aov_plan *plans, int cap, char *warnings, size_t warnings_cap);

/* the `kind` byte lentil_hip_alloc_frame expects for a plan entry: its original filter, except that the AOV named
 * lentil_debug gets LENTIL_FILTER_CLOSEST_DEBUG (own z-buffer and value, src/lentil.h:838-845) */
int lentil_aov_frame_kind(const lentil_aov_plan *plan);

/* sanitize_aov_list + index assignment of rebuild_arnold_outputs_from_list (src/aov_data.h:164-189):
 * drops duplicates and AOVs lentil does not filter, numbers the rest; returns the new count */
int lentil_sanitize_aov_list(lentil_aov_plan *plans, int n);

/* ------------------------------------------------------------------------------------
 * Cryptomatte (Camera::setup_crypto_aovs, cryptomatte_construct_cache, src/lentil.h:781-811,1015-1055; the
 * imager's ranking names, src/lentil_imager.cpp:84-92,124-126)
 * ---------------------------------------------------------------------------------- */
/* The AOV list additions of Camera::setup_crypto_aovs (src/lentil.h:1015-1055), which runs once cryptomatte has added
 * its ranked outputs to options.outputs: every output whose AOV name contains "crypto_" is appended -- the ranked
 * ones (crypto_material00 ...) with is_crypto set and the filter token replaced by "lentil_replaced_filter", the
 * three display AOVs (crypto_material / crypto_asset / crypto_object) as they are.  Returns the number of entries
 * written to plans (to be appended to the camera's list before lentil_sanitize_aov_list), -1 when cap is short. */
int lentil_setup_crypto_aovs(const char *const *outputs, int n, lentil_aov_plan *plans, int cap);
/* Camera::cryptomatte_construct_cache (src/lentil.h:781-811) for one cryptomatte AOV of one AOV sample: its depth
 * samples (opacity RGB 3 n, the AOV's float n; front to back as AiAOVSampleIteratorGetNextDepth yields them) folded
 * into id -> weight.  Writes the pairs in id order, pads ids / weights up to cap with unused pairs (weight bits
 * 0xFFFFFFFF) and returns the number of pairs, or -1 when there are more than cap. */
int lentil_crypto_construct_cache(int n_depth, const float *opacity_rgb, const float *value, float *ids, float *weights,
                                  int cap);
/* position in the weight-sorted map that an AOV of this name shows (and the next): 2 for crypto_material01 /
 * crypto_asset01 / crypto_object01, 4 for ...02, 0 for every other name (src/lentil_imager.cpp:124-126) */
int lentil_crypto_rank_of_name(const char *aov_name);

/* ------------------------------------------------------------------------------------
 * Visit capture: what filter_pixel gathers per AOV sample (src/lentil_filter.cpp:105-165,
 * 206-234) appended to per-thread staging columns instead of being traced on the spot.
 * One slot per render thread; appends to different slots may run concurrently.
 * ---------------------------------------------------------------------------------- */
typedef struct lentil_stage lentil_stage;

typedef struct lentil_sample_capture {
  int px, py;                     /* region-relative pixel, after :100-101 */
  float inverse_sample_density;   /* :83-84 or AiAOVSampleIteratorGetInvDensity (:109) */
  float rgba[4];                  /* AiAOVSampleIteratorGetRGBA */
  float P[3];                     /* AOV "P" */
  float Z;                        /* AOV "Z" */
  float raydir[3];                /* AOV "lentil_raydir" */
  float time;                     /* AOV "lentil_time" */
  float volume[3];                /* AOV "volume" */
  float bidir_ignore;             /* AOV "lentil_bidir_ignore" */
  float transmission[4];          /* AOV "transmission" */
  const float *extra_rgba;        /* n_extra x 4: the other lentil-filtered AOVs widened to RGBA (:214-232) */
  /* stages with cryptomatte AOVs (lentil_stage_set_crypto): the sample's cache per AOV, n_crypto x entries ids and
   * weights as lentil_crypto_construct_cache leaves them (:167-169); NULL otherwise */
  const float *crypto_ids;
  const float *crypto_weights;
} lentil_sample_capture;

int lentil_stage_create(int n_thread_slots, uint32_t n_extra, lentil_stage **out);
/* cryptomatte AOVs ride along as n_crypto x entries (id, weight) pairs per visit; call on an empty stage, before the
 * first append (a streaming stage sends them with its blocks: lentil_hip_visits_begin_crypto / _append_crypto; the
 * context needs its lentil_hip_alloc_crypto first).  lentil_stage_crypto (plain staging): the columns that belong to
 * what lentil_stage_visits returned last (lentil_hip_upload_crypto takes them as they are). */
int lentil_stage_set_crypto(lentil_stage *s, uint32_t n_crypto, uint32_t entries);
int lentil_stage_crypto(lentil_stage *s, lentil_crypto_visits *out);
void lentil_stage_destroy(lentil_stage *s);
void lentil_stage_reset(lentil_stage *s);                         /* new frame */
int lentil_stage_append(lentil_stage *s, int thread_slot, const lentil_sample_capture *c);
uint64_t lentil_stage_size(const lentil_stage *s);
/* concatenates the slots (slot order, append order inside a slot) into contiguous columns owned by the
 * stage and describes them as a ragged visit stream (explicit pixel + inv_density columns) */
int lentil_stage_visits(lentil_stage *s, lentil_visits *out);
/* Streaming mode: instead of keeping the frame's visits on the host until the imager asks, every slot fills
 * page-locked blocks of block_visits visits (0: 16384; allocated when a slot is first used) and sends each full block to `gpu` at once
 * (lentil_hip_visits_append; two blocks per slot, used in turn), so that the PCIe transfer runs while the
 * buckets render.  Call between frames (empty stage); gpu == NULL returns to plain staging.  capacity_hint:
 * expected visits per frame (0: unknown).  lentil_stage_reset starts the next frame's stream;
 * lentil_stage_finish_stream sends the partly filled blocks and makes the stream the context's visits
 * (lentil_imager_process_bucket does that itself); lentil_stage_visits is not available in this mode. */
int lentil_stage_stream_to(lentil_stage *s, lentil_hip_ctx *gpu, uint32_t block_visits, uint64_t capacity_hint);
int lentil_stage_finish_stream(lentil_stage *s, uint64_t *n_visits);
int lentil_stage_is_streaming(const lentil_stage *s);

/* ------------------------------------------------------------------------------------
 * imager_lentil: driver_process_bucket (src/lentil_imager.cpp:66-193).  The first call from any
 * thread runs upload -> clear -> redistribute -> resolve -> download exactly once (the imager runs
 * on the full-frame schedule, :38); every call then copies its bucket out of the downloaded frame.
 * ---------------------------------------------------------------------------------- */
typedef struct lentil_imager lentil_imager;

int lentil_imager_create(lentil_hip_ctx *gpu, lentil_stage *stage, const lentil_params *params, uint32_t n_aovs,
                         lentil_imager **out);
void lentil_imager_destroy(lentil_imager *im);
void lentil_imager_new_frame(lentil_imager *im);                   /* re-arms the once-only trigger */
/* bucket_rgba: bucket_size_x * bucket_size_y RGBA floats, row-major, written in place like bucket_data
 * (:160,178,182); bucket_xo/yo are frame coordinates (region_min is subtracted like :118).  Returns
 * LENTIL_OK or the error of the GPU pass (lentil_imager_last_error). */
int lentil_imager_process_bucket(lentil_imager *im, uint32_t aov, int bucket_xo, int bucket_yo, int bucket_size_x,
                                 int bucket_size_y, float *bucket_rgba);
/* Cryptomatte AOVs of the frame (after lentil_hip_alloc_crypto on the context and lentil_stage_set_crypto on the
 * stage): `ranks[c]` = lentil_crypto_rank_of_name of AOV c.  The once-only pass then also uploads the stage's
 * cryptomatte columns; a pass that reports a short draw log is repeated once (the library has sized its log by
 * then).  lentil_imager_process_crypto_bucket is the cryptomatte branch of driver_process_bucket
 * (src/lentil_imager.cpp:121-161): (id, weight / total) pairs of the AOV's rank; at the first pixel of a bucket
 * row whose map has no more than `rank` entries the rest of that row is left as it is (:132-134). */
int lentil_imager_set_crypto(lentil_imager *im, uint32_t n_crypto, const int *ranks);
int lentil_imager_process_crypto_bucket(lentil_imager *im, uint32_t crypto, int bucket_xo, int bucket_yo, int bucket_size_x,
                                        int bucket_size_y, float *bucket_rgba);
const char *lentil_imager_last_error(const lentil_imager *im);

#ifdef __cplusplus
}
#endif
#endif
