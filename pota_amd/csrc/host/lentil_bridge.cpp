// lentil_bridge.cpp -- SDK-free host logic of the plugin's node callbacks (include/lentil_bridge.h).
// Built with g++ like the reference's own sources; links liblentil_hip.so for the imager path only.
#include "lentil_bridge.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#define BRIDGE_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------
// lentil_camera node
// ---------------------------------------------------------------------------------------
static const char *const kUnits[] = {"mm", "cm", "dm", "m", "automatic", nullptr};
static const char *const kCameraTypes[] = {"ThinLens", "PolynomialOptics", nullptr};
static const char *const kChromaticTypes[] = {"green_magenta", "red_cyan", nullptr};
// the reference splices its lens list in from a generated header that is not in its tree
// (include/auto_generated_lens_includes/pota_cpp_lenses.h); the lenses shipped here take its place
// The reference's lens ids, in its order (include/auto_generated_lens_includes/pota_h_lenses.h:4-47 = the LensModel enum,
// pota_cpp_lenses.h the strings): a scene file written for the reference parses, and an id keeps its number.  Their
// polynomial tables live in zpelgrims/polynomial-optics, which the reference tree does not hold; this build ships
// tables of its own (kLensTableOf below maps the ids they stand in for, every other id is refused by name at camera
// update).  Behind the 44: this build's tables under their own names.
static const char *const kLensModels[] = {
    "angenieux__double_gauss__1953__49mm", "angenieux__double_gauss__1953__85mm", "angenieux__double_gauss__1953__105mm",
    "angenieux__double_gauss__1953__55mm", "asahi__takumar__1969__45mm", "asahi__takumar__1969__50mm",
    "asahi__takumar__1969__65mm", "asahi__takumar__1969__75mm", "asahi__takumar__1969__58mm",
    "asahi__takumar__1969__85mm", "asahi__takumar__1970__28mm", "asahi__takumar__1970__50mm",
    "asahi__takumar__1970__35mm", "canon__retrofocus_wideangle__1982__22mm", "canon__unknown__1956__35mm",
    "canon__unknown__1956__52mm", "cooke__speed_panchro__1920__40mm", "cooke__speed_panchro__1920__75mm",
    "cooke__speed_panchro__1920__100mm", "cooke__speed_panchro__1920__50mm", "kodak__petzval__1948__150mm",
    "kodak__petzval__1948__105mm", "kodak__petzval__1948__85mm", "kodak__petzval__1948__65mm",
    "kodak__petzval__1948__75mm", "kodak__petzval__1948__58mm", "meyer_optik_goerlitz__primoplan__1936__58mm",
    "meyer_optik_goerlitz__primoplan__1936__75mm", "minolta__fisheye__1978__16mm", "minolta__fisheye__1978__22mm",
    "minolta__fisheye__1978__28mm", "nikon__retrofocus_wideangle__1971__28mm", "nikon__retrofocus_wideangle__1971__35mm",
    "nikon__unknown__2014__65mm", "nikon__unknown__2014__40mm", "nikon__unknown__2014__50mm",
    "unknown__petzval__1900__85mm", "unknown__petzval__1900__100mm", "unknown__petzval__1900__75mm",
    "unknown__petzval__1900__65mm", "zeiss__biotar__1927__65mm", "zeiss__biotar__1927__58mm",
    "zeiss__biotar__1927__85mm", "zeiss__biotar__1927__45mm",
    "double_gauss_50mm", "petzval_58mm", "anamorphic_petzval_58mm", nullptr};
static const int kLensModelDefault = 16;      // cooke__speed_panchro__1920__40mm, src/lentil_camera.cpp:29
struct LensStandIn { const char *id, *table; };
static const LensStandIn kLensTableOf[] = {
    {"angenieux__double_gauss__1953__49mm", "double_gauss_50mm"},     // (BASELINE's "double-gauss 50 mm" class)
    {"kodak__petzval__1948__58mm", "petzval_58mm"},
    {"double_gauss_50mm", "double_gauss_50mm"}, {"petzval_58mm", "petzval_58mm"},
    {"anamorphic_petzval_58mm", "anamorphic_petzval_58mm"},
};

#define P_ENUM(n, d, v) {n, LENTIL_AI_TYPE_ENUM, (double)(d), nullptr, v}
#define P_INT(n, d) {n, LENTIL_AI_TYPE_INT, (double)(d), nullptr, nullptr}
#define P_FLT(n, d) {n, LENTIL_AI_TYPE_FLOAT, (double)(d), nullptr, nullptr}
#define P_BOOL(n, d) {n, LENTIL_AI_TYPE_BOOLEAN, (double)(d), nullptr, nullptr}
#define P_STR(n, d) {n, LENTIL_AI_TYPE_STRING, 0.0, d, nullptr}

static const lentil_node_param kCameraParams[] = {     // src/lentil_camera.cpp:20-49, same order
    P_ENUM("camera_type", 0, kCameraTypes),
    P_INT("bidir_sample_mult", 5),
    P_ENUM("units", 1, kUnits),
    P_FLT("sensor_width", 36.0),
    P_BOOL("enable_dof", 1),
    P_FLT("fstop", 0.0),
    P_FLT("focus_dist", 150.0),
    P_INT("aperture_blades_lentil", 0),
    P_FLT("exp", 1.0),
    P_ENUM("lens_model", kLensModelDefault, kLensModels),
    P_FLT("wavelength", 550.0),
    P_FLT("extra_sensor_shift", 0.0),
    P_FLT("focal_length_lentil", 35.0),
    P_FLT("optical_vignetting", 0.0),
    P_FLT("abb_spherical", 0.5),
    P_FLT("abb_distortion", 0.0),
    P_FLT("abb_coma", 0.0),
    P_FLT("abb_chromatic", 0.0),
    P_ENUM("abb_chromatic_type", 0, kChromaticTypes),
    P_FLT("bokeh_circle_to_square", 0.0),
    P_FLT("bokeh_anamorphic", 0.0),
    P_BOOL("bokeh_enable_image", 0),
    P_STR("bokeh_image_path", ""),
    P_INT("vignetting_retries", 15),
    P_FLT("bidir_add_energy", 0.0),
    P_FLT("bidir_add_energy_minimum_luminance", 2.0),
    P_FLT("bidir_add_energy_transition", 1.0),
    P_BOOL("enable_bidir_transmission", 0),
    P_BOOL("enable_skydome", 0),
};

BRIDGE_API const lentil_node_param *lentil_camera_node_parameters(int *count) {
  if (count) *count = (int)(sizeof(kCameraParams) / sizeof(kCameraParams[0]));
  return kCameraParams;
}

BRIDGE_API void lentil_camera_node_defaults(lentil_camera_node_values *v) {
  if (!v) return;
  memset(v, 0, sizeof(*v));
  v->camera_type = 0; v->bidir_sample_mult = 5; v->units = 1; v->sensor_width = 36.0f; v->enable_dof = 1;
  v->fstop = 0.0f; v->focus_dist = 150.0f; v->aperture_blades_lentil = 0; v->exp = 1.0f; v->lens_model = kLensModelDefault;
  v->wavelength = 550.0f; v->extra_sensor_shift = 0.0f; v->focal_length_lentil = 35.0f;
  v->optical_vignetting = 0.0f; v->abb_spherical = 0.5f; v->abb_distortion = 0.0f; v->abb_coma = 0.0f;
  v->abb_chromatic = 0.0f; v->abb_chromatic_type = 0; v->bokeh_circle_to_square = 0.0f; v->bokeh_anamorphic = 0.0f;
  v->bokeh_enable_image = 0; v->bokeh_image_path = ""; v->vignetting_retries = 15; v->bidir_add_energy = 0.0f;
  v->bidir_add_energy_minimum_luminance = 2.0f; v->bidir_add_energy_transition = 1.0f;
  v->enable_bidir_transmission = 0; v->enable_skydome = 0;
}

static inline float clamp_min_f(float in, const float mn) { if (in < mn) in = mn; return in; }        // src/global.h:15-18
static inline float clamp_f(float in, const float mn, const float mx) { if (in < mn) in = mn; if (in > mx) in = mx; return in; }

BRIDGE_API int lentil_camera_params_from_node(const lentil_camera_node_values *v, float meters_per_unit, int ignore_dof,
                                              lentil_params *out, double *input_fstop, double *lambda_um,
                                              double *extra_sensor_shift, float *exposure) {
  if (!v || !out) return LENTIL_ERR_INVALID;
  out->cameraType = v->camera_type;
  int unit = v->units;
  if (unit == 4) {   // "automatic": the float option is compared with double literals (src/lentil.h:1193-1199),
                     // so only 1.0 can match; 0.1f, 0.01f and 0.001f differ from 0.1, 0.01 and 0.001
    if ((double)meters_per_unit == 1.0) unit = 3;
    else if ((double)meters_per_unit == 0.1) unit = 2;
    else if ((double)meters_per_unit == 0.01) unit = 1;
    else if ((double)meters_per_unit == 0.001) unit = 0;
  }
  out->unitModel = unit;
  out->sensor_width = v->sensor_width;
  out->enable_dof = v->enable_dof ? 1 : 0;
  if (ignore_dof) out->enable_dof = 0;
  if (input_fstop) *input_fstop = (double)clamp_min_f(v->fstop, 0.01f);
  out->focus_distance = v->focus_dist;
  out->bokeh_aperture_blades = v->aperture_blades_lentil;
  if (exposure) *exposure = v->exp;
  if (lambda_um) *lambda_um = (double)v->wavelength * 0.001;
  if (extra_sensor_shift) *extra_sensor_shift = (double)v->extra_sensor_shift;
  out->focal_length = clamp_min_f(v->focal_length_lentil, 0.01f);
  out->optical_vignetting_distance = v->optical_vignetting;
  out->optical_vignetting_radius = 1.0f;
  out->abb_spherical = clamp_f(v->abb_spherical, 0.001f, 0.999f);
  out->abb_distortion = v->abb_distortion;
  out->abb_coma = v->abb_coma;
  out->abb_chromatic = v->abb_chromatic;
  out->abb_chromatic_type = v->abb_chromatic_type;
  out->circle_to_square = clamp_f(v->bokeh_circle_to_square, 0.01f, 0.99f);
  out->bokeh_anamorphic = clamp_f((float)(1.0 - (double)v->bokeh_anamorphic), 0.0f, 1.0f);
  out->bokeh_enable_image = v->bokeh_enable_image ? 1 : 0;
  out->bidir_sample_mult = v->bidir_sample_mult;
  out->bidir_add_energy_minimum_luminance = v->bidir_add_energy_minimum_luminance;
  out->bidir_add_energy = v->bidir_add_energy;
  out->bidir_add_energy_transition = v->bidir_add_energy_transition;
  out->vignetting_retries = v->vignetting_retries;
  out->enable_bidir_transmission = v->enable_bidir_transmission ? 1 : 0;
  out->enable_skydome = v->enable_skydome ? 1 : 0;
  return LENTIL_OK;
}

// ---------------------------------------------------------------------------------------
// lentil_filter node
// ---------------------------------------------------------------------------------------
static const char *const kRequiredAovs[] = {"RGBA RGBA", "VECTOR P", "FLOAT Z", "FLOAT lentil_time", "FLOAT lentil_debug",
                                            "RGB lentil_raydir", "RGB opacity", "RGBA transmission",
                                            "FLOAT lentil_bidir_ignore", nullptr};

BRIDGE_API const char *const *lentil_filter_required_aovs(void) { return kRequiredAovs; }
BRIDGE_API float lentil_filter_width(int oidn_imager_present) { return oidn_imager_present ? 1.0f : 1.5f; }

BRIDGE_API int lentil_filter_output_type(int input_type) {
  switch (input_type) {
    case LENTIL_AI_TYPE_RGBA:
    case LENTIL_AI_TYPE_RGB:
    case LENTIL_AI_TYPE_VECTOR:
    case LENTIL_AI_TYPE_FLOAT:
      return LENTIL_AI_TYPE_RGBA;
    default:
      return LENTIL_AI_TYPE_NONE;
  }
}

BRIDGE_API float lentil_filter_inverse_sample_density(int samples_in_footprint, float filter_width, int aa_samples_set_by_user,
                                                      int *disable_redistribution) {
  float AA_samples = std::sqrt(samples_in_footprint) / filter_width;           // std::sqrt(int) is double, :83
  float inverse_sample_density = 1.0 / (AA_samples * AA_samples);              // :84
  if (disable_redistribution)
    *disable_redistribution =
        (static_cast<int>(std::round(AA_samples)) != aa_samples_set_by_user || (aa_samples_set_by_user < 3)) ? 1 : 0;
  return inverse_sample_density;
}

// ---------------------------------------------------------------------------------------
// output strings and the operator's AOV list
// ---------------------------------------------------------------------------------------
static void put(char *dst, size_t cap, const std::string &s) {
  const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
  memcpy(dst, s.data(), n);
  dst[n] = 0;
}

// std::sregex_token_iterator(str, regex(" "), -1) of the reference: fields between single spaces; leading
// and inner empty fields are kept, a trailing empty field is not produced
static std::vector<std::string> split_on_space(const std::string &s) {
  std::vector<std::string> out;
  size_t start = 0;
  while (true) {
    const size_t sp = s.find(' ', start);
    if (sp == std::string::npos) {
      if (start < s.size()) out.push_back(s.substr(start));
      break;
    }
    out.push_back(s.substr(start, sp - start));
    start = sp + 1;
  }
  return out;
}

BRIDGE_API void lentil_tokenize_output(const char *output_string, lentil_output_tokens *tok) {
  if (!tok) return;
  memset(tok, 0, sizeof(*tok));
  const std::vector<std::string> tokens = split_on_space(output_string ? output_string : "");
  std::string c0, c1, c2, c3, c4, c5;                                    // src/aov_data.h:37-56
  if (tokens.size() >= 4) { c0 = tokens[0]; c1 = tokens[1]; c2 = tokens[2]; c3 = tokens[3]; }
  if (tokens.size() >= 5) c4 = tokens[4];
  if (tokens.size() >= 6) c5 = tokens[5];
  const bool no_camera = c4.empty() || c4 == "HALF";
  tok->half_flag = ((no_camera ? c4 : c5) == "HALF") ? 1 : 0;
  put(tok->camera, sizeof(tok->camera), no_camera ? std::string() : c0);
  put(tok->aov_name, sizeof(tok->aov_name), no_camera ? c0 : c1);
  put(tok->aov_type, sizeof(tok->aov_type), no_camera ? c1 : c2);
  put(tok->filter, sizeof(tok->filter), no_camera ? c2 : c3);
  put(tok->driver, sizeof(tok->driver), no_camera ? c3 : c4);
}

static std::string rebuild(const lentil_output_tokens &t) {               // src/aov_data.h:73-91
  std::string s;
  if (t.camera[0]) { s += t.camera; s += " "; }
  s += t.aov_name; s += " ";
  s += t.aov_type; s += " ";
  s += t.filter; s += " ";
  s += t.driver;
  if (t.half_flag) s += " HALF";
  return s;
}

BRIDGE_API int lentil_rebuild_output(const lentil_output_tokens *tok, char *buf, size_t cap) {
  if (!tok || !buf) return -1;
  const std::string s = rebuild(*tok);
  if (s.size() + 1 > cap) return -1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

BRIDGE_API unsigned lentil_string_to_arnold_type(const char *str) {       // src/global.h:58-65
  const std::string s = str ? str : "";
  if (s == "float" || s == "FLOAT" || s == "flt" || s == "FLT") return LENTIL_AI_TYPE_FLOAT;
  if (s == "rgba" || s == "RGBA") return LENTIL_AI_TYPE_RGBA;
  if (s == "rgb" || s == "RGB") return LENTIL_AI_TYPE_RGB;
  if (s == "vector" || s == "vec" || s == "VECTOR" || s == "VEC") return LENTIL_AI_TYPE_VECTOR;
  return 0;
}

static void plan_from_output(const std::string &output, lentil_aov_plan *p) {   // AOVData ctor, src/aov_data.h:136-141
  memset(p, 0, sizeof(*p));
  lentil_tokenize_output(output.c_str(), &p->to);
  put(p->name, sizeof(p->name), p->to.aov_name);
  p->type = lentil_string_to_arnold_type(p->to.aov_type);
  p->original_filter = LENTIL_FILTER_GAUSSIAN;
}

BRIDGE_API int lentil_operator_cook(const char *const *outputs, const char *const *filter_entry_names, int n,
                                    lentil_aov_plan *plans, int cap, char *warnings, size_t warnings_cap) {
  if (!outputs || !filter_entry_names || !plans || n <= 0) return -1;
  if (warnings && warnings_cap) warnings[0] = 0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    lentil_aov_plan aov;
    plan_from_output(outputs[i] ? outputs[i] : "", &aov);
    const std::string fe = filter_entry_names[i] ? filter_entry_names[i] : "";
    if (fe == "gaussian_filter") aov.original_filter = LENTIL_FILTER_GAUSSIAN;            // :51-61
    else if (fe == "closest_filter") aov.original_filter = LENTIL_FILTER_CLOSEST;
    else if (fe == "variance_filter") aov.original_filter = LENTIL_FILTER_VARIANCE;
    else {
      if (warnings && warnings_cap) {
        const size_t used = strlen(warnings);
        snprintf(warnings + used, warnings_cap - used,
                 "[LENTIL] Specified AOV filter (%s) is incompatible with Lentil. Defaulting to gaussian_filter.\n", fe.c_str());
      }
      aov.original_filter = LENTIL_FILTER_GAUSSIAN;
    }
    bool replace_filter = true;
    const std::string ty = aov.to.aov_type, nm = aov.to.aov_name;
    if (ty != "RGBA" && ty != "RGB" && ty != "FLOAT" && ty != "VECTOR") replace_filter = false;       // :66-71
    if (nm == "crypto_material" || nm == "crypto_asset" || nm == "crypto_object") replace_filter = false;   // :75-78
    else if (nm.find("crypto_") != std::string::npos) continue;                                         // :79-82
    if (replace_filter && nm != "lentil_replaced_filter") put(aov.to.filter, sizeof(aov.to.filter), "lentil_replaced_filter");
    for (int j = 0; j < m; ++j)
      if (nm == plans[j].to.aov_name) aov.is_duplicate = 1;                                             // :89-93
    if (m >= cap) return -1;
    plans[m++] = aov;
  }
  if (m == 0) return -1;      // the reference indexes aovs[0] unconditionally (:103)
  // the three AOVs the operator adds as copies of the first output (:99-127); the copy keeps aovs[0]'s
  // original_filter and flags except where the reference overrides them
  struct Extra { const char *name, *type; int closest; };
  const Extra extras[3] = {{"lentil_debug", "FLOAT", 1}, {"lentil_time", "FLOAT", 0}, {"lentil_raydir", "RGB", 0}};
  for (const Extra &e : extras) {
    if (m >= cap) return -1;
    lentil_aov_plan a = plans[0];
    put(a.to.aov_type, sizeof(a.to.aov_type), e.type);
    put(a.to.aov_name, sizeof(a.to.aov_name), e.name);
    const std::string again = rebuild(a.to);            // re-tokenised from the rebuilt string, :107
    lentil_tokenize_output(again.c_str(), &a.to);
    put(a.name, sizeof(a.name), e.name);
    a.type = lentil_string_to_arnold_type(a.to.aov_type);
    if (e.closest) a.original_filter = LENTIL_FILTER_CLOSEST;
    plans[m++] = a;
  }
  return m;
}

BRIDGE_API int lentil_aov_frame_kind(const lentil_aov_plan *plan) {
  if (!plan) return -1;
  if (std::string(plan->name) == "lentil_debug") return LENTIL_FILTER_CLOSEST_DEBUG;
  return plan->original_filter;
}

BRIDGE_API int lentil_sanitize_aov_list(lentil_aov_plan *plans, int n) {
  if (!plans) return -1;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (plans[i].is_duplicate || std::string(plans[i].to.filter) != "lentil_replaced_filter") continue;
    plans[m] = plans[i];
    plans[m].index = m;
    ++m;
  }
  return m;
}

// ---------------------------------------------------------------------------------------
// cryptomatte
// ---------------------------------------------------------------------------------------
BRIDGE_API int lentil_setup_crypto_aovs(const char *const *outputs, int n, lentil_aov_plan *plans, int cap) {
  if (!outputs || !plans || n < 0) return -1;
  int m = 0;
  for (int i = 0; i < n; ++i) {                                            // src/lentil.h:1021-1049
    lentil_aov_plan aov;
    plan_from_output(outputs[i] ? outputs[i] : "", &aov);
    bool replace_filter = true, cryptomatte_aov = false;
    const std::string nm = aov.to.aov_name;
    if (nm == "crypto_material" || nm == "crypto_asset" || nm == "crypto_object") {       // :1032-1036: display only
      replace_filter = false;
      cryptomatte_aov = true;
    } else if (nm.find("crypto_") != std::string::npos) {                                  // :1037-1040
      aov.is_crypto = 1;
      cryptomatte_aov = true;
    }
    if (!cryptomatte_aov) continue;
    if (replace_filter && nm != "lentil_replaced_filter") put(aov.to.filter, sizeof(aov.to.filter), "lentil_replaced_filter");
    if (m >= cap) return -1;
    plans[m++] = aov;
  }
  return m;
}

BRIDGE_API int lentil_crypto_construct_cache(int n_depth, const float *opacity_rgb, const float *value, float *ids,
                                             float *weights, int cap) {
  if (n_depth < 0 || (n_depth && (!opacity_rgb || !value)) || !ids || !weights || cap <= 0) return -1;
  std::map<float, float> cache;                                            // crypto_hashmap_cache[aov.index]
  float iterative_transparency_weight = 1.0f;
  float quota = 1.0;
  float sample_value = 0.0f;
  for (int d = 0; d < n_depth; ++d) {                                      // src/lentil.h:789-801
    // AiColorToGrey: (r + g + b) / 3
    const float sub_sample_opacity = (opacity_rgb[3 * d] + opacity_rgb[3 * d + 1] + opacity_rgb[3 * d + 2]) / 3;
    sample_value = value[d];
    const float sub_sample_weight = sub_sample_opacity * iterative_transparency_weight;
    iterative_transparency_weight *= (1.0f - sub_sample_opacity);
    quota -= sub_sample_weight;
    cache[sample_value] += sub_sample_weight;
  }
  if (quota > 0.0) cache[sample_value] += quota;                           // :804: what is left goes to the last sample
  if ((int)cache.size() > cap) return -1;
  int k = 0;
  for (const auto &e : cache) { ids[k] = e.first; weights[k] = e.second; ++k; }
  const uint32_t unused = 0xFFFFFFFFu;
  for (int j = k; j < cap; ++j) { ids[j] = 0.0f; memcpy(&weights[j], &unused, 4); }
  return k;
}

BRIDGE_API int lentil_crypto_rank_of_name(const char *aov_name) {
  const std::string nm = aov_name ? aov_name : "";
  if (nm == "crypto_material01" || nm == "crypto_asset01" || nm == "crypto_object01") return 2;
  if (nm == "crypto_material02" || nm == "crypto_asset02" || nm == "crypto_object02") return 4;
  return 0;
}

// ---------------------------------------------------------------------------------------
// setup_filter (resolution / region) and the display pass-through filters
// ---------------------------------------------------------------------------------------
BRIDGE_API void lentil_setup_filter_region(lentil_params *p, int xres, int yres, int region_min_x, int region_min_y,
                                           int region_max_x, int region_max_y, float filter_width) {
  if (!p) return;
  auto unset = [](int v) { return v == INT32_MIN || v == INT32_MAX; };
  if (unset(region_min_x) || unset(region_max_x) || unset(region_min_y) || unset(region_max_y)) {     // src/lentil.h:1068-1077
    region_min_x = 0; region_min_y = 0; region_max_x = xres; region_max_y = yres;
  }
  p->xres_without_region = (uint32_t)xres;
  p->yres_without_region = (uint32_t)yres;
  p->region_min_x = region_min_x;
  p->region_min_y = region_min_y;
  p->xres = (uint32_t)(region_max_x - region_min_x + 1);
  p->yres = (uint32_t)(region_max_y - region_min_y + 1);
  p->filter_width = filter_width;
}

// lens_model -> the id's name and the name of the shipped table that stands in for it (NULL: none shipped)
BRIDGE_API const char *lentil_lens_model_name(int lens_model) {
  int count = 0;
  while (kLensModels[count]) ++count;
  return (lens_model >= 0 && lens_model < count) ? kLensModels[lens_model] : nullptr;
}
BRIDGE_API const char *lentil_lens_model_table(int lens_model) {
  const char *id = lentil_lens_model_name(lens_model);
  if (!id) return nullptr;
  for (const LensStandIn &m : kLensTableOf)
    if (strcmp(m.id, id) == 0) return m.table;
  return nullptr;
}

BRIDGE_API void lentil_filter_gaussian_complete(int n, const float *offsets_xy, const float *values_rgba, const float *inv_density,
                                                float uniform_inv_density, float filter_width, lentil_exp_fn fast_exp,
                                                float out_rgba[4]) {
  float aweight = 0.0f, av[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const float invd = inv_density ? inv_density[i] : uniform_inv_density;
    if (invd <= 0.f) continue;
    const float k = 2.0f / filter_width;
    const float r = (k * k) * (offsets_xy[2 * i] * offsets_xy[2 * i] + offsets_xy[2 * i + 1] * offsets_xy[2 * i + 1]);
    if (r > 1.0f) continue;
    const float w = (fast_exp ? fast_exp(2 * -r) : std::exp(2 * -r)) * invd;
    for (int c = 0; c < 4; ++c) av[c] += w * values_rgba[4 * i + c];
    aweight += w;
  }
  if (aweight != 0.0f) for (int c = 0; c < 4; ++c) av[c] *= 1.0f / aweight;
  for (int c = 0; c < 4; ++c) out_rgba[c] = av[c];
}

BRIDGE_API void lentil_filter_closest_complete(int n, const float *depth, const float *values_rgba, float out_rgba[4]) {
  float z = 0.0f;
  out_rgba[0] = out_rgba[1] = out_rgba[2] = out_rgba[3] = 0.0f;
  for (int i = 0; i < n; ++i) {
    if (std::fabs(depth[i]) <= z || z == 0.0f) {
      z = std::fabs(depth[i]);
      out_rgba[0] = values_rgba[4 * i]; out_rgba[1] = values_rgba[4 * i + 1]; out_rgba[2] = values_rgba[4 * i + 2];
      out_rgba[3] = 1.0f;
    }
  }
}

// ---------------------------------------------------------------------------------------
// visit capture
// ---------------------------------------------------------------------------------------
// streaming mode (lentil_stage_stream_to): a page-locked block of `cap` visits, columns one after the other
struct PinnedBlock {
  char *base = nullptr;
  uint32_t n = 0;
  uint64_t ticket = 0;           // of the copies last started from this block
};

struct StageSlot {
  std::vector<float> rgba, pos_z, raydir_time, volume_ignore, transmission, inv_density;
  std::vector<std::vector<float>> extra;
  std::vector<uint32_t> pixel;
  std::vector<std::vector<float>> crypto_ids, crypto_weights;   // per cryptomatte AOV: entries floats per visit
  PinnedBlock blk[2];
  int cur = 0;
  uint64_t sent = 0;             // visits of this slot already on their way to the GPU
};

struct lentil_stage {
  uint32_t n_extra = 0;
  uint32_t n_crypto = 0, crypto_entries = 0;
  std::vector<StageSlot> slots;
  StageSlot all;               // concatenation, built by lentil_stage_visits
  // streaming mode
  lentil_hip_ctx *gpu = nullptr;
  uint32_t block_visits = 0;
  uint64_t capacity_hint = 0;
  bool stream_open = false;
  std::string error;
};

static float *block_col(const lentil_stage *s, const PinnedBlock &b, uint32_t c) {        // column c of a block: cap x 4 floats
  return reinterpret_cast<float *>(b.base + (size_t)c * s->block_visits * 16);
}
static uint32_t *block_pixel(const lentil_stage *s, const PinnedBlock &b) {
  return reinterpret_cast<uint32_t *>(b.base + (size_t)(5 + s->n_extra) * s->block_visits * 16);
}
static float *block_inv(const lentil_stage *s, const PinnedBlock &b) {
  return reinterpret_cast<float *>(b.base + (size_t)(5 + s->n_extra) * s->block_visits * 16 + (size_t)s->block_visits * 4);
}

// bytes of one visit in a page-locked block: the columns, pixel + density, the cryptomatte caches
static size_t block_visit_bytes(const lentil_stage *s) {
  return 16 * (size_t)(5 + s->n_extra) + 8 + (size_t)s->n_crypto * s->crypto_entries * 8;
}
// cryptomatte AOV a's ids (which = 0) / weights (1) of a block: block_visits x entries floats
static float *block_crypto(const lentil_stage *s, const PinnedBlock &b, uint32_t a, int which) {
  const size_t head = ((size_t)(5 + s->n_extra) * 16 + 8) * s->block_visits;
  return reinterpret_cast<float *>(b.base + head + ((size_t)a * 2 + (size_t)which) * s->block_visits * s->crypto_entries * 4);
}

static int stage_begin_stream(lentil_stage *s) {
  lentil_visits lay;
  memset(&lay, 0, sizeof(lay));
  lay.pixel_row_stride = 1;
  lay.n_extra = s->n_extra;
  static const float announce = 0.0f;
  lay.inv_density = &announce;                   // per-visit densities follow
  int rc = lentil_hip_visits_begin(s->gpu, &lay, s->capacity_hint);
  if (rc == LENTIL_OK && s->n_crypto) rc = lentil_hip_visits_begin_crypto(s->gpu, s->crypto_entries);
  s->stream_open = rc == LENTIL_OK;
  if (rc != LENTIL_OK) s->error = lentil_hip_last_error(s->gpu);
  return rc;
}

// sends what block `b` of a slot holds
static int stage_send_block(lentil_stage *s, StageSlot &sl, PinnedBlock &b) {
  if (b.n == 0) return LENTIL_OK;
  lentil_visits part;
  memset(&part, 0, sizeof(part));
  part.n = b.n;
  part.n_extra = s->n_extra;
  part.rgba = block_col(s, b, 0); part.pos_z = block_col(s, b, 1); part.raydir_time = block_col(s, b, 2);
  part.volume_ignore = block_col(s, b, 3); part.transmission = block_col(s, b, 4);
  for (uint32_t k = 0; k < s->n_extra; ++k) part.extra[k] = block_col(s, b, 5 + k);
  part.pixel = block_pixel(s, b);
  part.inv_density = block_inv(s, b);
  int rc;
  if (s->n_crypto) {
    lentil_crypto_visits cv;
    memset(&cv, 0, sizeof(cv));
    cv.n = b.n; cv.n_crypto = s->n_crypto; cv.entries = s->crypto_entries;
    for (uint32_t a = 0; a < s->n_crypto; ++a) { cv.hash[a] = block_crypto(s, b, a, 0); cv.weight[a] = block_crypto(s, b, a, 1); }
    rc = lentil_hip_visits_append_crypto(s->gpu, &part, &cv, &b.ticket);
  } else {
    rc = lentil_hip_visits_append(s->gpu, &part, &b.ticket);
  }
  if (rc != LENTIL_OK) return rc;
  sl.sent += b.n;
  b.n = 0;
  return LENTIL_OK;
}

BRIDGE_API int lentil_stage_create(int n_thread_slots, uint32_t n_extra, lentil_stage **out) {
  if (!out || n_thread_slots <= 0 || n_extra > LENTIL_MAX_AOVS - 1) return LENTIL_ERR_INVALID;
  lentil_stage *s = new (std::nothrow) lentil_stage();
  if (!s) return LENTIL_ERR_NOMEM;
  s->n_extra = n_extra;
  s->slots.resize((size_t)n_thread_slots);
  for (StageSlot &sl : s->slots) sl.extra.resize(n_extra);
  s->all.extra.resize(n_extra);
  *out = s;
  return LENTIL_OK;
}

BRIDGE_API uint64_t lentil_stage_size(const lentil_stage *s);

static void stage_free_blocks(lentil_stage *s) {
  for (StageSlot &sl : s->slots)
    for (PinnedBlock &b : sl.blk) {
      if (b.base) (void)lentil_hip_host_free(b.base);
      b = PinnedBlock();
    }
}

BRIDGE_API void lentil_stage_destroy(lentil_stage *s) {
  if (!s) return;
  stage_free_blocks(s);
  delete s;
}

BRIDGE_API int lentil_stage_stream_to(lentil_stage *s, lentil_hip_ctx *gpu, uint32_t block_visits, uint64_t capacity_hint) {
  if (!s) return LENTIL_ERR_INVALID;
  if (lentil_stage_size(s) != 0) return LENTIL_ERR_INVALID;      // between frames only
  stage_free_blocks(s);
  s->gpu = gpu;
  s->stream_open = false;
  if (!gpu) return LENTIL_OK;                                    // back to plain staging
  s->block_visits = block_visits ? block_visits : (1u << 14);
  s->capacity_hint = capacity_hint;
  const int rc = stage_begin_stream(s);  // a slot's blocks are allocated when a thread first uses the slot
  if (rc != LENTIL_OK) {
    // plain staging again: with `gpu` left set every lentil_stage_append would fail (no open stream) and the frame would
    // lose its redistribution, where the caller was told "the visits are uploaded at the end of the frame"
    stage_free_blocks(s);
    s->gpu = nullptr;
    s->stream_open = false;
  }
  return rc;
}

static void clear_slot(StageSlot &sl) {
  sl.rgba.clear(); sl.pos_z.clear(); sl.raydir_time.clear(); sl.volume_ignore.clear(); sl.transmission.clear();
  sl.inv_density.clear(); sl.pixel.clear();
  for (auto &e : sl.extra) e.clear();
  for (auto &e : sl.crypto_ids) e.clear();
  for (auto &e : sl.crypto_weights) e.clear();
}

BRIDGE_API void lentil_stage_reset(lentil_stage *s) {
  if (!s) return;
  for (StageSlot &sl : s->slots) {
    clear_slot(sl);
    sl.sent = 0; sl.cur = 0;
    for (PinnedBlock &b : sl.blk) { b.n = 0; b.ticket = 0; }      // tickets belong to the previous frame's stream
  }
  clear_slot(s->all);
  if (s->gpu) (void)stage_begin_stream(s);       // a failure shows at the next append
}

static inline void push4(std::vector<float> &v, float a, float b, float c, float d) {
  v.push_back(a); v.push_back(b); v.push_back(c); v.push_back(d);
}

BRIDGE_API int lentil_stage_append(lentil_stage *s, int thread_slot, const lentil_sample_capture *c) {
  if (!s || !c || thread_slot < 0 || (size_t)thread_slot >= s->slots.size()) return LENTIL_ERR_INVALID;
  if (c->px < 0 || c->py < 0 || c->px > 0xFFFF || c->py > 0xFFFF) return LENTIL_ERR_INVALID;
  if (s->n_extra && !c->extra_rgba) return LENTIL_ERR_INVALID;
  if (s->n_crypto && (!c->crypto_ids || !c->crypto_weights)) return LENTIL_ERR_INVALID;
  StageSlot &sl = s->slots[(size_t)thread_slot];
  if (s->gpu) {
    // streaming mode: into the slot's current page-locked block; a full block goes to the GPU and the slot
    // continues in its other block once that one's previous copies are done
    if (!s->stream_open) return LENTIL_ERR_INVALID;
    PinnedBlock *b = &sl.blk[sl.cur];
    if (b->n == s->block_visits) {
      int rc = stage_send_block(s, sl, *b);
      if (rc != LENTIL_OK) return rc;
      sl.cur ^= 1;
      b = &sl.blk[sl.cur];
      if (b->ticket && (rc = lentil_hip_visits_wait(s->gpu, b->ticket)) != LENTIL_OK) return rc;
    }
    if (!b->base) {
      void *p = nullptr;
      const int rc = lentil_hip_host_alloc(&p, (size_t)s->block_visits * block_visit_bytes(s));
      if (rc != LENTIL_OK) return rc;
      b->base = static_cast<char *>(p);
    }
    const uint32_t i = b->n++;
    auto put4 = [&](uint32_t col, float x, float y, float z, float w) {
      float *d = block_col(s, *b, col) + (size_t)i * 4;
      d[0] = x; d[1] = y; d[2] = z; d[3] = w;
    };
    put4(0, c->rgba[0], c->rgba[1], c->rgba[2], c->rgba[3]);
    put4(1, c->P[0], c->P[1], c->P[2], c->Z);
    put4(2, c->raydir[0], c->raydir[1], c->raydir[2], c->time);
    put4(3, c->volume[0], c->volume[1], c->volume[2], c->bidir_ignore);
    put4(4, c->transmission[0], c->transmission[1], c->transmission[2], c->transmission[3]);
    for (uint32_t k = 0; k < s->n_extra; ++k)
      put4(5 + k, c->extra_rgba[k * 4], c->extra_rgba[k * 4 + 1], c->extra_rgba[k * 4 + 2], c->extra_rgba[k * 4 + 3]);
    block_pixel(s, *b)[i] = (uint32_t)c->px | ((uint32_t)c->py << 16);
    block_inv(s, *b)[i] = c->inverse_sample_density;
    for (uint32_t k = 0; k < s->n_crypto; ++k) {
      memcpy(block_crypto(s, *b, k, 0) + (size_t)i * s->crypto_entries, c->crypto_ids + (size_t)k * s->crypto_entries, (size_t)s->crypto_entries * 4);
      memcpy(block_crypto(s, *b, k, 1) + (size_t)i * s->crypto_entries, c->crypto_weights + (size_t)k * s->crypto_entries, (size_t)s->crypto_entries * 4);
    }
    return LENTIL_OK;
  }
  try {
    push4(sl.rgba, c->rgba[0], c->rgba[1], c->rgba[2], c->rgba[3]);
    push4(sl.pos_z, c->P[0], c->P[1], c->P[2], c->Z);
    push4(sl.raydir_time, c->raydir[0], c->raydir[1], c->raydir[2], c->time);
    push4(sl.volume_ignore, c->volume[0], c->volume[1], c->volume[2], c->bidir_ignore);
    push4(sl.transmission, c->transmission[0], c->transmission[1], c->transmission[2], c->transmission[3]);
    for (uint32_t k = 0; k < s->n_extra; ++k)
      push4(sl.extra[k], c->extra_rgba[k * 4], c->extra_rgba[k * 4 + 1], c->extra_rgba[k * 4 + 2], c->extra_rgba[k * 4 + 3]);
    sl.pixel.push_back((uint32_t)c->px | ((uint32_t)c->py << 16));
    sl.inv_density.push_back(c->inverse_sample_density);
    for (uint32_t k = 0; k < s->n_crypto; ++k) {
      const float *ids = c->crypto_ids + (size_t)k * s->crypto_entries, *w = c->crypto_weights + (size_t)k * s->crypto_entries;
      sl.crypto_ids[k].insert(sl.crypto_ids[k].end(), ids, ids + s->crypto_entries);
      sl.crypto_weights[k].insert(sl.crypto_weights[k].end(), w, w + s->crypto_entries);
    }
  } catch (const std::bad_alloc &) {
    return LENTIL_ERR_NOMEM;
  }
  return LENTIL_OK;
}

BRIDGE_API uint64_t lentil_stage_size(const lentil_stage *s) {
  uint64_t n = 0;
  if (s) for (const StageSlot &sl : s->slots) n += sl.pixel.size() + sl.sent + sl.blk[0].n + sl.blk[1].n;
  return n;
}

// streaming mode: the partly filled blocks follow, the stream becomes the context's visits
BRIDGE_API int lentil_stage_finish_stream(lentil_stage *s, uint64_t *n_visits) {
  if (!s || !s->gpu || !s->stream_open) return LENTIL_ERR_INVALID;
  for (StageSlot &sl : s->slots)
    for (PinnedBlock &b : sl.blk) {
      const int rc = stage_send_block(s, sl, b);
      if (rc != LENTIL_OK) return rc;
    }
  s->stream_open = false;
  return lentil_hip_visits_end(s->gpu, n_visits);
}

BRIDGE_API int lentil_stage_is_streaming(const lentil_stage *s) { return s && s->gpu ? 1 : 0; }

BRIDGE_API int lentil_stage_visits(lentil_stage *s, lentil_visits *out) {
  if (!s || !out || s->gpu) return LENTIL_ERR_INVALID;
  StageSlot &a = s->all;
  clear_slot(a);
  try {
    for (const StageSlot &sl : s->slots) {
      a.rgba.insert(a.rgba.end(), sl.rgba.begin(), sl.rgba.end());
      a.pos_z.insert(a.pos_z.end(), sl.pos_z.begin(), sl.pos_z.end());
      a.raydir_time.insert(a.raydir_time.end(), sl.raydir_time.begin(), sl.raydir_time.end());
      a.volume_ignore.insert(a.volume_ignore.end(), sl.volume_ignore.begin(), sl.volume_ignore.end());
      a.transmission.insert(a.transmission.end(), sl.transmission.begin(), sl.transmission.end());
      a.inv_density.insert(a.inv_density.end(), sl.inv_density.begin(), sl.inv_density.end());
      a.pixel.insert(a.pixel.end(), sl.pixel.begin(), sl.pixel.end());
      for (uint32_t k = 0; k < s->n_extra; ++k) a.extra[k].insert(a.extra[k].end(), sl.extra[k].begin(), sl.extra[k].end());
      for (uint32_t k = 0; k < s->n_crypto; ++k) {
        a.crypto_ids[k].insert(a.crypto_ids[k].end(), sl.crypto_ids[k].begin(), sl.crypto_ids[k].end());
        a.crypto_weights[k].insert(a.crypto_weights[k].end(), sl.crypto_weights[k].begin(), sl.crypto_weights[k].end());
      }
    }
  } catch (const std::bad_alloc &) {
    return LENTIL_ERR_NOMEM;
  }
  memset(out, 0, sizeof(*out));
  out->n = a.pixel.size();
  out->visits_per_pixel = 0;
  out->pixel_row_stride = 1;
  out->n_extra = s->n_extra;
  out->rgba = a.rgba.data();
  out->pos_z = a.pos_z.data();
  out->raydir_time = a.raydir_time.data();
  out->volume_ignore = a.volume_ignore.data();
  out->transmission = a.transmission.data();
  for (uint32_t k = 0; k < s->n_extra; ++k) out->extra[k] = a.extra[k].data();
  out->pixel = a.pixel.data();
  out->inv_density = a.inv_density.data();
  return LENTIL_OK;
}

BRIDGE_API int lentil_stage_set_crypto(lentil_stage *s, uint32_t n_crypto, uint32_t entries) {
  if (!s || n_crypto > LENTIL_MAX_CRYPTO || (n_crypto && (entries == 0 || entries > 64))) return LENTIL_ERR_INVALID;
  if (lentil_stage_size(s) != 0) return LENTIL_ERR_INVALID;
  s->n_crypto = n_crypto;
  s->crypto_entries = n_crypto ? entries : 0;
  if (s->gpu) {                        // streaming: the blocks change size, the stream is announced anew
    stage_free_blocks(s);
    const int rc = stage_begin_stream(s);
    if (rc != LENTIL_OK) return rc;
  }
  for (StageSlot &sl : s->slots) { sl.crypto_ids.assign(n_crypto, {}); sl.crypto_weights.assign(n_crypto, {}); }
  s->all.crypto_ids.assign(n_crypto, {});
  s->all.crypto_weights.assign(n_crypto, {});
  return LENTIL_OK;
}

BRIDGE_API int lentil_stage_crypto(lentil_stage *s, lentil_crypto_visits *out) {
  if (!s || !out || s->gpu || !s->n_crypto) return LENTIL_ERR_INVALID;       // (a streaming stage has sent them already)
  memset(out, 0, sizeof(*out));
  out->n = s->all.pixel.size();
  out->n_crypto = s->n_crypto;
  out->entries = s->crypto_entries;
  for (uint32_t k = 0; k < s->n_crypto; ++k) {
    if (s->all.crypto_ids[k].size() != out->n * s->crypto_entries) return LENTIL_ERR_INVALID;   // lentil_stage_visits first
    out->hash[k] = s->all.crypto_ids[k].data();
    out->weight[k] = s->all.crypto_weights[k].data();
  }
  return LENTIL_OK;
}

// ---------------------------------------------------------------------------------------
// imager
// ---------------------------------------------------------------------------------------
struct lentil_imager {
  lentil_hip_ctx *gpu = nullptr;
  lentil_stage *stage = nullptr;
  lentil_params P{};
  uint32_t n_aovs = 0;
  std::once_flag *once = nullptr;
  int rc = LENTIL_OK;
  std::string error;
  std::vector<std::vector<float>> resolved;   // per AOV: xres * yres * 4
  std::vector<int> crypto_ranks;              // per cryptomatte AOV
  std::vector<std::vector<float>> crypto_rgba;      // per cryptomatte AOV: xres * yres * 4
  std::vector<std::vector<uint8_t>> crypto_has;     // ... and whether the pixel's map reaches the AOV's rank
};

BRIDGE_API int lentil_imager_create(lentil_hip_ctx *gpu, lentil_stage *stage, const lentil_params *params, uint32_t n_aovs,
                                    lentil_imager **out) {
  if (!gpu || !stage || !params || !out || n_aovs == 0 || n_aovs > LENTIL_MAX_AOVS) return LENTIL_ERR_INVALID;
  lentil_imager *im = new (std::nothrow) lentil_imager();
  if (!im) return LENTIL_ERR_NOMEM;
  im->gpu = gpu; im->stage = stage; im->P = *params; im->n_aovs = n_aovs;
  im->once = new std::once_flag();
  *out = im;
  return LENTIL_OK;
}

BRIDGE_API void lentil_imager_destroy(lentil_imager *im) {
  if (!im) return;
  delete im->once;
  delete im;
}

BRIDGE_API void lentil_imager_new_frame(lentil_imager *im) {
  if (!im) return;
  delete im->once;
  im->once = new std::once_flag();
  im->rc = LENTIL_OK;
  im->error.clear();
}

static void run_gpu_pass(lentil_imager *im) {
  auto check = [&](int rc, const char *what) {
    if (rc != LENTIL_OK && im->rc == LENTIL_OK) {
      im->rc = rc;
      im->error = std::string("[LENTIL] ") + what + ": " + lentil_hip_last_error(im->gpu);
    }
    return rc == LENTIL_OK;
  };
  if (lentil_stage_is_streaming(im->stage)) {
    // the visits went to the GPU while the buckets rendered; only the partly filled blocks are left
    if (!check(lentil_stage_finish_stream(im->stage, nullptr), "finish_stream")) return;
  } else {
    lentil_visits v;
    if (!check(lentil_stage_visits(im->stage, &v), "stage")) return;
    if (!check(lentil_hip_upload_visits(im->gpu, &v), "upload_visits")) return;
    if (!im->crypto_ranks.empty()) {
      lentil_crypto_visits cv;
      if (!check(lentil_stage_crypto(im->stage, &cv), "stage crypto")) return;
      if (!check(lentil_hip_upload_crypto(im->gpu, &cv), "upload_crypto")) return;
    }
  }
  if (!check(lentil_hip_clear_frame(im->gpu), "clear_frame")) return;
  int rc = lentil_hip_redistribute(im->gpu);
  if (rc == LENTIL_ERR_NOMEM && !im->crypto_ranks.empty() && strstr(lentil_hip_last_error(im->gpu), "draw log")) {
    // the cryptomatte adds are replayed from the pass's draw log, which the library sizes from the pass before:
    // it now knows this frame's count
    if (!check(lentil_hip_clear_frame(im->gpu), "clear_frame")) return;
    rc = lentil_hip_redistribute(im->gpu);
  }
  if (!check(rc, "redistribute")) return;
  if (!check(lentil_hip_resolve(im->gpu), "resolve")) return;
  const size_t np = (size_t)im->P.xres * im->P.yres;
  im->resolved.assign(im->n_aovs, std::vector<float>());
  for (uint32_t a = 0; a < im->n_aovs; ++a) {
    im->resolved[a].resize(np * 4);
    if (!check(lentil_hip_download_aov(im->gpu, a, im->resolved[a].data()), "download_aov")) return;
  }
  const size_t nc = im->crypto_ranks.size();
  im->crypto_rgba.assign(nc, std::vector<float>());
  im->crypto_has.assign(nc, std::vector<uint8_t>());
  for (size_t c = 0; c < nc; ++c) {
    im->crypto_rgba[c].resize(np * 4);
    im->crypto_has[c].resize(np);
    if (!check(lentil_hip_download_crypto(im->gpu, (uint32_t)c, (uint32_t)im->crypto_ranks[c], im->crypto_rgba[c].data(),
                                          im->crypto_has[c].data()), "download_crypto")) return;
  }
}

BRIDGE_API int lentil_imager_process_bucket(lentil_imager *im, uint32_t aov, int bucket_xo, int bucket_yo, int bucket_size_x,
                                            int bucket_size_y, float *bucket_rgba) {
  if (!im || !bucket_rgba || aov >= im->n_aovs || bucket_size_x < 0 || bucket_size_y < 0) return LENTIL_ERR_INVALID;
  std::call_once(*im->once, run_gpu_pass, im);
  if (im->rc != LENTIL_OK) return im->rc;
  const std::vector<float> &img = im->resolved[aov];
  for (int j = 0; j < bucket_size_y; ++j) {
    for (int i = 0; i < bucket_size_x; ++i) {
      const int x = i + bucket_xo - im->P.region_min_x, y = j + bucket_yo - im->P.region_min_y;   // :116-118
      if (x < 0 || y < 0 || x >= im->P.xres || y >= im->P.yres) continue;
      const size_t lin = (size_t)x + (size_t)y * (size_t)im->P.xres;                               // coords_to_linear_pixel
      memcpy(bucket_rgba + ((size_t)j * bucket_size_x + i) * 4, img.data() + lin * 4, 4 * sizeof(float));
    }
  }
  return LENTIL_OK;
}

BRIDGE_API int lentil_imager_set_crypto(lentil_imager *im, uint32_t n_crypto, const int *ranks) {
  if (!im || n_crypto > LENTIL_MAX_CRYPTO || (n_crypto && !ranks)) return LENTIL_ERR_INVALID;
  im->crypto_ranks.assign(ranks, ranks + n_crypto);
  for (int r : im->crypto_ranks) if (r < 0) return LENTIL_ERR_INVALID;
  return LENTIL_OK;
}

BRIDGE_API int lentil_imager_process_crypto_bucket(lentil_imager *im, uint32_t crypto, int bucket_xo, int bucket_yo,
                                                   int bucket_size_x, int bucket_size_y, float *bucket_rgba) {
  if (!im || !bucket_rgba || crypto >= im->crypto_ranks.size() || bucket_size_x < 0 || bucket_size_y < 0) return LENTIL_ERR_INVALID;
  std::call_once(*im->once, run_gpu_pass, im);
  if (im->rc != LENTIL_OK) return im->rc;
  const std::vector<float> &img = im->crypto_rgba[crypto];
  const std::vector<uint8_t> &has = im->crypto_has[crypto];
  for (int j = 0; j < bucket_size_y; ++j) {
    for (int i = 0; i < bucket_size_x; ++i) {
      const int x = i + bucket_xo - im->P.region_min_x, y = j + bucket_yo - im->P.region_min_y;   // :116-118
      if (x < 0 || y < 0 || x >= im->P.xres || y >= im->P.yres) continue;
      const size_t lin = (size_t)x + (size_t)y * (size_t)im->P.xres;
      if (!has[lin]) break;                                         // :132-134: leaves the inner (row) loop
      memcpy(bucket_rgba + ((size_t)j * bucket_size_x + i) * 4, img.data() + lin * 4, 4 * sizeof(float));
    }
  }
  return LENTIL_OK;
}

BRIDGE_API const char *lentil_imager_last_error(const lentil_imager *im) { return im ? im->error.c_str() : "null imager"; }
