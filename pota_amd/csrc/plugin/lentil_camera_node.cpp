// lentil_camera_node.cpp -- the lentil_camera node (src/lentil_camera.cpp): parameters from the bridge's table,
// Camera state as local data, forward rays from the host library, and LentilCamera::setup -- the counterpart of
// Camera::setup_camera (src/lentil.h:211-280) that ends with the GPU context ready for the frame.
#include "lentil_plugin.h"

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>

#include "../generated/lens_tables_host.h"

AI_CAMERA_NODE_EXPORT_METHODS(LentilCameraMtd)

const LentilStrings &lentil_strings() {
  static const LentilStrings s;
  return s;
}

node_parameters {
  // same names, types, defaults and order as src/lentil_camera.cpp:19-52 (lentil_camera_node_parameters());
  // lens_model: the reference's 44 ids in its order, then this build's own tables (lentil_lens_model_table)
  int n = 0;
  const lentil_node_param *p = lentil_camera_node_parameters(&n);
  for (int i = 0; i < n; ++i) {
    switch (p[i].type) {
      case LENTIL_AI_TYPE_INT: AiParameterInt(p[i].name, (int)p[i].default_value) break;
      case LENTIL_AI_TYPE_FLOAT: AiParameterFlt(p[i].name, (float)p[i].default_value) break;
      case LENTIL_AI_TYPE_BOOLEAN: AiParameterBool(p[i].name, p[i].default_value != 0.0) break;
      case LENTIL_AI_TYPE_STRING: AiParameterStr(p[i].name, p[i].default_string ? p[i].default_string : "") break;
      case LENTIL_AI_TYPE_ENUM: AiParameterEnum(p[i].name, (int)p[i].default_value, const_cast<const char **>(p[i].enum_values)) break;
      default: break;
    }
  }
  AiMetaDataSetBool(nentry, nullptr, "force_update", true);
}

node_plugin_initialize { (void)plugin_data; return true; }
node_plugin_cleanup { (void)plugin_data; }

node_initialize {
  AiCameraInitialize(node);
  AiNodeSetLocalData(node, new LentilCamera());
}

node_update {
  LentilCamera *cam = (LentilCamera *)AiNodeGetLocalData(node);
  cam->setup(AiNodeGetUniverse(node));
  AiCameraUpdate(node, false);
}

node_finish {
  delete (LentilCamera *)AiNodeGetLocalData(node);
  AiNodeSetLocalData(node, nullptr);
}

camera_create_ray {
  LentilCamera *cam = (LentilCamera *)AiNodeGetLocalData(node);
  // xor128 state for the vignetting retries: per thread (the reference's generator is a function static shared by
  // all threads, src/global.h:22-27)
  thread_local uint32_t rng[4];
  thread_local bool seeded = false;
  if (!seeded) { lentil_host_xor128_init(rng); seeded = true; }
  (void)tid;
  const float in[6] = {input.sx, input.sy, input.dsx, input.dsy, input.lensx, input.lensy};
  lentil_host_camera_ray r;
  lentil_host_camera_create_ray(&cam->P, cam->host_lens, cam->have_bokeh ? &cam->bokeh : nullptr, rng, cam->lambda_um,
                                cam->exposure, in, &r);
  output.origin = AtVector(r.origin[0], r.origin[1], r.origin[2]);
  output.dir = AtVector(r.dir[0], r.dir[1], r.dir[2]);
  output.weight = AtRGB(r.weight[0], r.weight[1], r.weight[2]);
  output.dOdx = AtVector(r.dOdx[0], r.dOdx[1], r.dOdx[2]);
  output.dOdy = AtVector(r.dOdy[0], r.dOdy[1], r.dOdy[2]);
  output.dDdx = AtVector(r.dDdx[0], r.dDdx[1], r.dDdx[2]);
  output.dDdy = AtVector(r.dDdy[0], r.dDdy[1], r.dDdy[2]);
}

camera_reverse_ray {
  LentilCamera *cam = (LentilCamera *)AiNodeGetLocalData(node);
  (void)relative_time;
  const float po[3] = {Po.x, Po.y, Po.z};
  float ps[2];
  lentil_host_camera_reverse_ray(cam->tan_fov, po, ps);
  Ps.x = ps[0];
  Ps.y = ps[1];
  return true;
}

void registerLentilCamera(AtNodeLib *node) {
  node->methods = (const void *)LentilCameraMtd;
  node->output_type = AI_TYPE_UNDEFINED;
  node->name = "lentil_camera";
  node->node_type = AI_NODE_CAMERA;
  strncpy(node->version, AI_VERSION, AI_MAXSIZE_VERSION - 1);
}

// ---------------------------------------------------------------------------------------------------------------
LentilCamera::~LentilCamera() {
  release_gpu();
  if (host_lens) lentil_host_lens_destroy(host_lens);
}

// The occlusion probe of the backward traces (src/lentil.h:613-629, src/lentil_filter.cpp:356-375): the library hands over the
// segment of every try of a round that got through the lens (include/lentil_hip.h, lentil_hip_set_occlusion_probe); each is
// probed the way the reference probes it -- AiMakeRay(AI_RAY_SHADOW, sample, normalize(lens point - sample), distance) and
// AiTraceProbe -- over the machine's threads, a shader-globals object per thread.
static void arnold_probe(void *, uint64_t n, const lentil_probe_segment *seg, uint8_t *occluded) {
  auto run = [&](uint64_t lo, uint64_t hi) {
    AtShaderGlobals *sg = AiShaderGlobals();
    for (uint64_t i = lo; i < hi; ++i) {
      const AtVector o(seg[i].origin[0], seg[i].origin[1], seg[i].origin[2]), t(seg[i].target[0], seg[i].target[1], seg[i].target[2]);
      const AtVector dir = AiV3Normalize(t - o);
      const AtRay ray = AiMakeRay(AI_RAY_SHADOW, o, &dir, AiV3Dist(t, o), sg);
      occluded[i] = AiTraceProbe(ray, sg) ? 1 : 0;
    }
    AiShaderGlobalsDestroy(sg);
  };
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 64) nt = 64;
  if (n < (1u << 14) || nt < 2) { run(0, n); return; }
  std::vector<std::thread> th;
  const uint64_t per = (n + nt - 1) / nt;
  for (unsigned k = 0; k < nt && (uint64_t)k * per < n; ++k)
    th.emplace_back(run, (uint64_t)k * per, std::min(n, (uint64_t)(k + 1) * per));
  for (std::thread &t : th) t.join();
}

void LentilCamera::release_gpu() {
  if (imager) { lentil_imager_destroy(imager); imager = nullptr; }
  if (stage) { lentil_stage_destroy(stage); stage = nullptr; }
  if (gpu) { lentil_hip_destroy(gpu); gpu = nullptr; }
}

int LentilCamera::aov_index(const char *name) const {
  for (size_t i = 0; i < aovs.size(); ++i)
    if (strcmp(aovs[i].name, name) == 0) return (int)i;
  return -1;
}

int LentilCamera::crypto_index(const char *name) const {
  for (size_t i = 0; i < crypto_aovs.size(); ++i)
    if (strcmp(crypto_aovs[i].name, name) == 0) return (int)i;
  return -1;
}

// Every camera object numbers its setups process-wide (a new object at a recycled address never repeats a number), and a
// thread's cached slot belongs to one of them: setup() hands the slots out afresh, so a render thread that outlives a
// camera update -- or meets another camera at the same address -- must not keep a slot a new thread may be given too
// (two threads appending to one staging slot, unlocked).
static std::atomic<uint64_t> g_slot_generations{0};

int LentilCamera::thread_slot() {
  thread_local int slot = -1;
  thread_local uint64_t generation = 0;
  const uint64_t now = slot_generation.load(std::memory_order_acquire);
  if (slot < 0 || generation != now) { slot = next_slot++; generation = now; }
  return slot;
}

void LentilCamera::setup(AtUniverse *universe) {
  std::lock_guard<std::mutex> guard(setup_mutex);
  const LentilStrings &S = lentil_strings();
  options_node = AiUniverseGetOptions(universe);
  camera_node = AiUniverseGetCamera(universe);
  redistribution = false;

  // ---- get_lentil_camera_params (src/lentil.h:1189-1243): one getter per parameter of the table
  lentil_camera_node_values v;
  lentil_camera_node_defaults(&v);
  auto I = [&](const char *n) { return AiNodeGetInt(camera_node, AtString(n)); };
  auto F = [&](const char *n) { return AiNodeGetFlt(camera_node, AtString(n)); };
  auto B = [&](const char *n) { return AiNodeGetBool(camera_node, AtString(n)) ? 1 : 0; };
  v.camera_type = I("camera_type"); v.bidir_sample_mult = I("bidir_sample_mult"); v.units = I("units");
  v.sensor_width = F("sensor_width"); v.enable_dof = B("enable_dof"); v.fstop = F("fstop"); v.focus_dist = F("focus_dist");
  v.aperture_blades_lentil = I("aperture_blades_lentil"); v.exp = F("exp"); v.lens_model = I("lens_model");
  v.wavelength = F("wavelength"); v.extra_sensor_shift = F("extra_sensor_shift"); v.focal_length_lentil = F("focal_length_lentil");
  v.optical_vignetting = F("optical_vignetting"); v.abb_spherical = F("abb_spherical"); v.abb_distortion = F("abb_distortion");
  v.abb_coma = F("abb_coma"); v.abb_chromatic = F("abb_chromatic"); v.abb_chromatic_type = I("abb_chromatic_type");
  v.bokeh_circle_to_square = F("bokeh_circle_to_square"); v.bokeh_anamorphic = F("bokeh_anamorphic");
  v.bokeh_enable_image = B("bokeh_enable_image");
  const AtString image_path = AiNodeGetStr(camera_node, AtString("bokeh_image_path"));
  v.bokeh_image_path = image_path.c_str();
  v.vignetting_retries = I("vignetting_retries"); v.bidir_add_energy = F("bidir_add_energy");
  v.bidir_add_energy_minimum_luminance = F("bidir_add_energy_minimum_luminance");
  v.bidir_add_energy_transition = F("bidir_add_energy_transition");
  v.enable_bidir_transmission = B("enable_bidir_transmission"); v.enable_skydome = B("enable_skydome");
  lentil_camera_params_from_node(&v, AiNodeGetFlt(options_node, AtString("meters_per_unit")),
                                 AiNodeGetBool(options_node, AtString("ignore_dof")) ? 1 : 0, &P, &input_fstop, &lambda_um,
                                 &extra_sensor_shift, &exposure);
  P.adaptive_sampling = AiNodeGetBool(options_node, AtString("enable_adaptive_sampling")) ? 1 : 0;
  if (const char *e = getenv("LENTIL_SAMPLES_OVERRIDE")) P.samples_override = atoi(e);     // bench / test knob, 0 upstream

  // ---- the lens of the enum (the reference switches over its compiled-in lens bodies, src/lentil.h:1576)
  if (host_lens) { lentil_host_lens_destroy(host_lens); host_lens = nullptr; }
  lens_table = nullptr;
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS) {
    const char *id = lentil_lens_model_name(v.lens_model);
    const char *want = lentil_lens_model_table(v.lens_model);
    for (int i = 0; want && i < kShippedLensCount; ++i)
      if (strcmp(kShippedLenses[i].name, want) == 0) lens_table = kShippedLenses[i].table;
    if (!lens_table) {
      AiMsgError("[LENTIL] no polynomial table shipped for lens_model %s (%d): this build carries tables for "
                 "angenieux__double_gauss__1953__49mm and kodak__petzval__1948__58mm (self-fitted stand-ins)", id ? id : "?", v.lens_model);
      AiRenderAbort();
      return;
    }
    host_lens = lentil_host_lens_create(lens_table);
  }
  // ---- camera_model_specific_setup (src/lentil.h:1568-1670); its focus search runs on the GPU when there is one
  // (LENTIL_GPU_FOCUS_SEARCH=0: the sequential loop on the host; both give the same shift bit for bit)
  lentil_focus_search_fn focus_fn = nullptr;
  const char *gf = getenv("LENTIL_GPU_FOCUS_SEARCH");
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS && !(gf && gf[0] == '0')) {
    if (!gpu && lentil_hip_create(0, &gpu) != LENTIL_OK) gpu = nullptr;
    if (gpu && lentil_hip_set_lens(gpu, lens_table) == LENTIL_OK)
      focus_fn = [](void *user, double focal_distance, double lambda, double *shift) -> int {
        return lentil_hip_focus_search(static_cast<lentil_hip_ctx *>(user), focal_distance, lambda, shift);
      };
  }
  if (lentil_host_camera_model_specific_setup_with(&P, host_lens, input_fstop, lambda_um * 1000.0, extra_sensor_shift, &tan_fov,
                                                   focus_fn, gpu) != 0) {
    AiMsgError("[LENTIL] camera setup failed");
    AiRenderAbort();
    return;
  }
  P.lambda_bw = (float)lambda_um;
  // The reference asks for the matrix at every sample's own time, AiWorldToCameraMatrix(camera_node, lentil_time)
  // (src/lentil_filter.cpp:141-144) -- Arnold's absolute sample time, which lies inside the camera's shutter
  // [shutter_start, shutter_end] (0 ... 1, a centred -0.25 ... 0.25, ...).  The GPU path gets the matrix sampled at
  // equidistant times over exactly that interval and interpolates per visit (lentil_hip_set_camera_motion / _shutter); a
  // shutter of zero length has one time and one matrix.
  shutter_start = AiNodeGetFlt(camera_node, AtString("shutter_start"));
  shutter_end = AiNodeGetFlt(camera_node, AtString("shutter_end"));
  AtMatrix w2c;
  AiWorldToCameraMatrix(camera_node, shutter_start, w2c);
  memcpy(P.world_to_camera, w2c.data, sizeof P.world_to_camera);
  motion_keys.clear();
  if (shutter_end > shutter_start) {
    int nkeys = 1;
    if (AtArray *ma = AiNodeGetArray(camera_node, AtString("matrix"))) nkeys = (int)AiArrayGetNumKeys(ma);
    // A moving camera (two keys or more) is sampled LENTIL_MAX_MOTION_KEYS times over the shutter, whatever its own key
    // count: the camera's keys span its MOTION range, which need not be the shutter (two keys over 0 ... 1 under a centred
    // shutter -0.25 ... 0.25: clamped before 0, linear after), so its knots fall between samples taken at `nkeys`
    // equidistant times; sixteen samples bound the deviation from AiWorldToCameraMatrix(camera, time) to one sixteenth of
    // the shutter around each knot (round-4 ADVICE).
    const int ns = nkeys >= 2 ? LENTIL_MAX_MOTION_KEYS : 1;
    for (int k = 0; ns >= 2 && k < ns; ++k) {
      AtMatrix mk;
      AiWorldToCameraMatrix(camera_node, shutter_start + ((float)k / (float)(ns - 1)) * (shutter_end - shutter_start), mk);
      motion_keys.insert(motion_keys.end(), &mk.data[0][0], &mk.data[0][0] + 16);
    }
  }

  have_bokeh = false;
  if (P.bokeh_enable_image) {
    // the reference reads the image through AiTextureLoad (src/imagebokeh.h:16-18): not part of this path's scope
    AiMsgError("[LENTIL CAMERA PO] Couldn't open bokeh image!");
    AiRenderAbort();
    return;
  }

  // ---- get_bidirectional_status + setup_lentil_aovs + setup_filter (src/lentil.h:1008-1122): the operator's AOV
  // list, sanitised; resolution and region; filter width
  // get_bidirectional_status (src/lentil.h:1151-1174)
  if (!P.enable_dof) {
    AiMsgWarning("[LENTIL BIDIRECTIONAL] Depth of field is disabled, therefore disabling bidirectional sampling.");
    release_gpu();
    return;
  }
  if (P.bidir_sample_mult == 0) {
    AiMsgWarning("[LENTIL BIDIRECTIONAL] Bidirectional samples are set to 0, filter will not execute.");
    release_gpu();
    return;
  }
  if (AiNodeGetBool(options_node, AtString("enable_progressive_render"))) {
    AiMsgError("[LENTIL BIDIRECTIONAL] Progressive rendering is not supported.");
    AiRenderAbort();
    return;
  }
  // setup_lentil_aovs (src/lentil.h:988-1012): the lentil_operator node, by its node entry
  AtNode *op = nullptr;
  AtNodeIterator *iter = AiUniverseGetNodeIterator(universe, AI_NODE_ALL);
  while (!AiNodeIteratorFinished(iter)) {
    AtNode *c = AiNodeIteratorGetNext(iter);
    if (AiNodeEntryGetNameAtString(AiNodeGetNodeEntry(c)) == AtString("lentil_operator")) { op = c; break; }
  }
  AiNodeIteratorDestroy(iter);
  LentilOperatorData *od = op ? (LentilOperatorData *)AiNodeGetLocalData(op) : nullptr;
  if (!od || !od->cooked) {
    AiMsgError("[LENTIL] Since Lentil 2.5, lentil requires an operator (lentil_operator) to function. Please insert this operator.");
    release_gpu();
    return;
  }
  aovs = od->aovs;
  // cryptomatte (src/lentil.h:241-278): its node sits in options.aov_shaders; once it has added its ranked outputs
  // (crypto_material00 ...) to options.outputs, setup_crypto_aovs puts lentil's filter on them.  The reference polls
  // CryptomatteData::is_setup_completed of that node's local data for up to 25 s; that struct is CryptomatteArnold's
  // (not part of this build), so the outputs are taken as they are when the camera updates.
  crypto_aovs.clear();
  cryptomatte_lentil = false;
  {
    AtArray *aov_shaders = AiNodeGetArray(options_node, AtString("aov_shaders"));
    const uint32_t ns = aov_shaders ? AiArrayGetNumElements(aov_shaders) : 0;
    for (uint32_t i = 0; i < ns; ++i) {
      AtNode *sh = static_cast<AtNode *>(AiArrayGetPtr(aov_shaders, i));
      if (sh && AiNodeEntryGetNameAtString(AiNodeGetNodeEntry(sh)) == AtString("cryptomatte")) cryptomatte_lentil = true;
    }
  }
  {
    // setup_crypto_aovs (src/lentil.h:1015-1055): every crypto_* output joins the list, options.outputs is rebuilt
    AtArray *outputs = AiNodeGetArray(options_node, S.outputs);
    const uint32_t n_out = outputs ? AiArrayGetNumElements(outputs) : 0;
    std::vector<std::string> strings(n_out);
    std::vector<const char *> ptrs(n_out);
    for (uint32_t i = 0; i < n_out; ++i) { strings[i] = AiArrayGetStr(outputs, i).c_str(); ptrs[i] = strings[i].c_str(); }
    std::vector<lentil_aov_plan> extra(n_out + 1);
    const int m = lentil_setup_crypto_aovs(ptrs.data(), (int)n_out, extra.data(), (int)extra.size());
    if (m > 0) {
      // the outputs the operator left, with the cryptomatte ones now carrying lentil's filter (a list that already
      // holds them -- a second update of the same scene -- is left alone)
      AtArray *rebuilt = AiArrayAllocate(n_out, 1, AI_TYPE_STRING);
      for (uint32_t i = 0; i < n_out; ++i) {
        lentil_output_tokens tok;
        lentil_tokenize_output(strings[i].c_str(), &tok);
        std::string out = strings[i];
        for (int k = 0; k < m; ++k)
          if (strcmp(extra[(size_t)k].to.aov_name, tok.aov_name) == 0 && strcmp(extra[(size_t)k].to.driver, tok.driver) == 0) {
            char buf[1024];
            if (lentil_rebuild_output(&extra[(size_t)k].to, buf, sizeof buf) >= 0) out = buf;
          }
        AiArraySetStr(rebuilt, i, AtString(out.c_str()));
      }
      AiNodeSetArray(options_node, S.outputs, rebuilt);
      for (int k = 0; k < m; ++k) {
        bool dup = false;
        for (const lentil_aov_plan &a : aovs) dup = dup || strcmp(a.to.aov_name, extra[(size_t)k].to.aov_name) == 0;
        if (!dup) aovs.push_back(extra[(size_t)k]);
      }
    }
  }
  const int n_aovs = lentil_sanitize_aov_list(aovs.data(), (int)aovs.size());
  aovs.resize((size_t)(n_aovs < 0 ? 0 : n_aovs));
  for (size_t i = 0; i < aovs.size();)
    if (aovs[i].is_crypto) { crypto_aovs.push_back(aovs[i]); aovs.erase(aovs.begin() + (long)i); } else ++i;
  if (!cryptomatte_lentil) crypto_aovs.clear();        // (no cache is built without the node, src/lentil_filter.cpp:169)
  if (crypto_aovs.size() > LENTIL_MAX_CRYPTO) { AiMsgError("[LENTIL] more than %d cryptomatte AOVs", LENTIL_MAX_CRYPTO); AiRenderAbort(); return; }
  if (aovs.empty() || strcmp(aovs[0].name, "RGBA") != 0) {
    AiMsgWarning("[LENTIL] the first lentil-filtered output must be RGBA; redistribution is off");
    release_gpu();
    return;
  }
  // the bookkeeping AOV lentil_time adds nothing to the images (the imager skips it, src/lentil_imager.cpp:109)
  for (size_t i = 0; i < aovs.size();)
    if (strcmp(aovs[i].name, "lentil_time") == 0) aovs.erase(aovs.begin() + (long)i); else ++i;
  if (aovs.size() > LENTIL_MAX_AOVS) { AiMsgError("[LENTIL] more than %d redistributed AOVs", LENTIL_MAX_AOVS); AiRenderAbort(); return; }

  const AtNodeEntry *oidn = AiNodeEntryLookUp(AtString("imager_denoiser_oidn"));
  lentil_setup_filter_region(&P, AiNodeGetInt(options_node, AtString("xres")), AiNodeGetInt(options_node, AtString("yres")),
                             AiNodeGetInt(options_node, AtString("region_min_x")), AiNodeGetInt(options_node, AtString("region_min_y")),
                             AiNodeGetInt(options_node, AtString("region_max_x")), AiNodeGetInt(options_node, AtString("region_max_y")),
                             lentil_filter_width(AiNodeEntryGetCount(oidn) != 0));
  const int aa = AiNodeGetInt(options_node, AtString("AA_samples"));
  P.inverse_sample_density = 1.0f / ((float)aa * (float)aa);

  // ---- the GPU side: context, lens, frame, staging, imager
  auto check = [&](int rc, const char *what) {
    if (rc == LENTIL_OK) return true;
    AiMsgError("[LENTIL] %s: %s", what, lentil_hip_last_error(gpu));
    AiRenderAbort();
    return false;
  };
  if (!gpu && !check(lentil_hip_create(0, &gpu), "lentil_hip_create")) return;
  if (!motion_keys.empty() && !check(lentil_hip_set_camera_shutter(gpu, shutter_start, shutter_end), "lentil_hip_set_camera_shutter")) return;
  if (!check(lentil_hip_set_camera_motion(gpu, (uint32_t)(motion_keys.size() / 16), motion_keys.empty() ? nullptr : motion_keys.data()),
             "lentil_hip_set_camera_motion")) return;
  if (!check(lentil_hip_set_params(gpu, &P), "set_params")) return;
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS && !check(lentil_hip_set_lens(gpu, lens_table), "set_lens")) return;
  if (!check(lentil_hip_set_bokeh(gpu, nullptr), "set_bokeh")) return;
  std::vector<uint8_t> kinds;
  for (const lentil_aov_plan &a : aovs) kinds.push_back((uint8_t)lentil_aov_frame_kind(&a));
  if (!check(lentil_hip_alloc_frame(gpu, (uint32_t)aovs.size(), kinds.data()), "alloc_frame")) return;
  // scene occlusion along every backward trace, as the reference asks the renderer (LENTIL_OCCLUSION_PROBES=0: none; the
  // camera-to-world matrices are the inverses of the world-to-camera keys above).  Not with a thin lens's chromatic
  // aberration, whose colour draw the reference orders behind the probe (the library refuses the pair): said once, not probed.
  {
    const char *op = getenv("LENTIL_OCCLUSION_PROBES");
    bool probes = !(op && op[0] == '0');
    if (probes && P.abb_chromatic != 0.0f) {
      AiMsgWarning("[LENTIL BIDIRECTIONAL] abb_chromatic is set: scene occlusion along the redistributed rays is not probed");
      probes = false;
    }
    if (!check(lentil_hip_set_occlusion_probe(gpu, probes ? arnold_probe : nullptr, nullptr, nullptr), "set_occlusion_probe")) return;
  }
  if (imager) { lentil_imager_destroy(imager); imager = nullptr; }
  if (stage) { lentil_stage_destroy(stage); stage = nullptr; }
  stage_slots = 256;
  next_slot = 0;
  slot_generation.store(++g_slot_generations, std::memory_order_release);      // (cached slots of the threads are void)
  if (lentil_stage_create(stage_slots, (uint32_t)aovs.size() - 1, &stage) != LENTIL_OK ||
      lentil_imager_create(gpu, stage, &P, (uint32_t)aovs.size(), &imager) != LENTIL_OK) {
    AiMsgError("[LENTIL] could not set up the visit staging");
    AiRenderAbort();
    return;
  }
  if (!crypto_aovs.empty()) {
    // per-pixel id tables on the GPU, the samples' caches as extra columns of the stage, the ranks by AOV name
    if (const char *e = getenv("LENTIL_CRYPTO_ENTRIES")) crypto_entries = atoi(e) > 0 ? atoi(e) : crypto_entries;
    const char *sl = getenv("LENTIL_CRYPTO_SLOTS");
    std::vector<int> ranks;
    for (const lentil_aov_plan &a : crypto_aovs) ranks.push_back(lentil_crypto_rank_of_name(a.name));
    if (!check(lentil_hip_alloc_crypto(gpu, (uint32_t)crypto_aovs.size(), sl ? (uint32_t)atoi(sl) : 0), "alloc_crypto")) return;
    if (lentil_stage_set_crypto(stage, (uint32_t)crypto_aovs.size(), (uint32_t)crypto_entries) != LENTIL_OK ||
        lentil_imager_set_crypto(imager, (uint32_t)crypto_aovs.size(), ranks.data()) != LENTIL_OK) {
      AiMsgError("[LENTIL] could not set up the cryptomatte staging");
      AiRenderAbort();
      return;
    }
  }
  // the visits (and their cryptomatte caches) travel to the GPU while the buckets render (LENTIL_STREAM_UPLOAD=0: one
  // upload at the frame end)
  const char *su = getenv("LENTIL_STREAM_UPLOAD");
  if (!(su && su[0] == '0')) {
    const double per_pixel = P.inverse_sample_density > 0.0f ? 1.0 / P.inverse_sample_density : 1.0;      // AA^2
    const uint64_t expect = (uint64_t)((double)P.xres * P.yres * per_pixel);
    if (lentil_stage_stream_to(stage, gpu, 0, expect) != LENTIL_OK)
      AiMsgWarning("[LENTIL] streaming upload unavailable (%s); the visits are uploaded at the end of the frame", lentil_hip_last_error(gpu));
  }
  for (const lentil_aov_plan &a : aovs)
    AiMsgInfo("[LENTIL BIDIRECTIONAL] Driver '%s' -- Adding aov %s of type %s", a.to.driver, a.to.aov_name, a.to.aov_type);
  for (const lentil_aov_plan &a : crypto_aovs)
    AiMsgInfo("[LENTIL BIDIRECTIONAL] Driver '%s' -- Adding aov %s of type %s", a.to.driver, a.to.aov_name, a.to.aov_type);
  imager_print_once_only = false;
  redistribution = true;
}
