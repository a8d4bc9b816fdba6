// fake_arnold.cpp -- libai_fake.so: an in-memory stand-in for the Arnold core behind tests/fake_arnold/ai.h, plus the
// harness (fa_* functions) that plays the renderer for the plugin under test.  TEST INFRASTRUCTURE ONLY.
//
// What the harness does in fa_render, in Arnold's order: operators are initialised and cooked
// (lentil_operator rewires options.outputs and creates its filter node), every node is initialised and updated
// (camera first), filter_pixel is called for every pixel and every output whose filter node is a plugin filter --
// from n_threads threads at once, rows interleaved -- with an AOV sample iterator over the samples of that pixel,
// then every imager (driver with metadata subtype = "imager") gets driver_process_bucket for every bucket, again
// from n_threads threads, and its in-place edits land in the output images.
#include "ai.h"

#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#define FA_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------------------------------
// data model
// ---------------------------------------------------------------------------------------------------------------
struct Param {
  uint8_t type = AI_TYPE_NONE;
  int i = 0;
  float f = 0.f;
  bool b = false;
  std::string s;
  AtArray *arr = nullptr;
  void *p = nullptr;
  std::vector<std::string> enum_values;
};
typedef std::vector<std::pair<std::string, Param>> ParamList;
struct AtList { ParamList *l; };

struct AtNodeEntry {
  std::string name;
  int node_type = 0;
  uint8_t output_type = AI_TYPE_NONE;
  const AtNodeMethods *m = nullptr;
  ParamList params;
  std::map<std::string, std::string> meta;     // "param|key" -> value ("|key" for node metadata)
  int count = 0;
  void *plugin_data = nullptr;
  bool plugin_inited = false;
};

struct AtArray {
  uint8_t type = AI_TYPE_NONE;
  uint8_t nkeys = 1;
  std::vector<float> mats;                 // AI_TYPE_MATRIX: nkeys x 16 floats
  std::vector<std::string> strs;
  std::vector<void *> ptrs;
};

struct AtRenderSession { std::map<std::string, int> hints; };

struct AtNode {
  AtNodeEntry *e = nullptr;
  AtUniverse *u = nullptr;
  std::string name;
  std::map<std::string, Param> p;
  void *local = nullptr;
  float filter_width = 0.f;
  std::vector<std::string> required_aovs;
  std::map<std::string, AtNode *> links;
  bool initialized = false;
  void *op_user_data = nullptr;
};

struct SampleStore {
  int n = 0;
  std::vector<int> px, py;
  std::vector<float> ox, oy, inv_density;
  std::map<std::string, std::pair<uint8_t, std::vector<float>>> aov;    // name -> (type, 4 floats per sample)
  std::vector<std::vector<int>> per_pixel;
  // deep samples: sample i holds depth_count[i] depth entries starting at depth_start[i]; AOVs that have per-depth
  // values (opacity, the cryptomatte ids) are read from depth_aov while an iterator walks a sample's depths
  std::vector<int> depth_count, depth_start;
  std::map<std::string, std::vector<float>> depth_aov;                   // name -> 4 floats per depth entry
};

struct AtUniverse {
  std::vector<AtNode *> nodes;
  AtNode *options = nullptr, *camera = nullptr;
  AtRenderSession session;
  SampleStore samples;
  int xres = 0, yres = 0;
  std::map<std::string, std::vector<float>> images;     // output AOV name -> xres * yres * 4
  std::map<std::string, std::vector<float>> display;    // the same right after filter_pixel, before the imagers edit the buckets
  std::vector<AtArray *> arrays;
  std::vector<std::string> late_outputs;                // appended to options.outputs after the operators cooked
};

struct AtAOVSampleIterator {
  AtUniverse *u;
  int x, y;
  const std::vector<int> *ids;
  int cur;                 // -1 before the first GetNext
  const char *aov_name;    // interned
  int depth = -1;          // -1: not inside a sample's depth entries
};

struct OutputSlot { const char *name; int type; float *bucket; };
struct AtOutputIterator { std::vector<OutputSlot> slots; size_t cur; };

static std::mutex g_mutex;
static std::set<std::string> g_strings;
static std::map<std::string, AtNodeEntry *> g_entries;
static std::vector<AtNodeEntry *> g_plugin_entries;        // in NodeLoader order
static std::string g_messages;
static std::atomic<int> g_errors{0}, g_aborts{0};

const char *AiFakeIntern(const char *s) {
  std::lock_guard<std::mutex> g(g_mutex);
  return g_strings.insert(s ? s : "").first->c_str();
}

static AtNodeEntry *entry(const std::string &name, int node_type = AI_NODE_UNDEFINED) {
  std::lock_guard<std::mutex> g(g_mutex);
  auto it = g_entries.find(name);
  if (it != g_entries.end()) return it->second;
  AtNodeEntry *e = new AtNodeEntry();
  e->name = name;
  e->node_type = node_type;
  g_entries[name] = e;
  return e;
}

static void builtin_entries() {
  static bool done = false;
  if (done) return;
  done = true;
  entry("options", AI_NODE_OPTIONS);
  for (const char *f : {"gaussian_filter", "closest_filter", "variance_filter", "box_filter"}) entry(f, AI_NODE_FILTER);
  entry("driver_exr", AI_NODE_DRIVER);
  entry("imager_denoiser_oidn", AI_NODE_DRIVER);
  for (const char *s : {"aov_write_float", "aov_write_rgb", "state_float", "state_vector"}) entry(s, AI_NODE_SHADER);
  entry("cryptomatte", AI_NODE_SHADER);            // another plugin's nodes, present by name only
  entry("cryptomatte_filter", AI_NODE_FILTER);
}

static void msg(const char *level, const char *format, va_list ap) {
  char buf[2048];
  vsnprintf(buf, sizeof buf, format, ap);
  std::lock_guard<std::mutex> g(g_mutex);
  g_messages += level;
  g_messages += buf;
  g_messages += "\n";
}
void AiMsgInfo(const char *format, ...) { va_list ap; va_start(ap, format); msg("INFO  ", format, ap); va_end(ap); }
void AiMsgWarning(const char *format, ...) { va_list ap; va_start(ap, format); msg("WARN  ", format, ap); va_end(ap); }
void AiMsgError(const char *format, ...) { va_list ap; va_start(ap, format); msg("ERROR ", format, ap); va_end(ap); ++g_errors; }
void AiRenderAbort() { ++g_aborts; }
const char *AiParamGetTypeName(uint8_t type) {
  switch (type) {
    case AI_TYPE_INT: return "INT"; case AI_TYPE_BOOLEAN: return "BOOL"; case AI_TYPE_FLOAT: return "FLOAT";
    case AI_TYPE_RGB: return "RGB"; case AI_TYPE_RGBA: return "RGBA"; case AI_TYPE_VECTOR: return "VECTOR";
    case AI_TYPE_STRING: return "STRING"; case AI_TYPE_ENUM: return "ENUM"; default: return "?";
  }
}

// ---------------------------------------------------------------------------------------------------------------
// parameter declaration, metadata
// ---------------------------------------------------------------------------------------------------------------
static Param &declare(AtList *params, const char *pname, uint8_t type) {
  params->l->push_back(std::make_pair(std::string(pname), Param()));
  Param &p = params->l->back().second;
  p.type = type;
  return p;
}
void AiNodeParamInt(AtList *params, int, const char *pname, int d) { declare(params, pname, AI_TYPE_INT).i = d; }
void AiNodeParamFlt(AtList *params, int, const char *pname, float d) { declare(params, pname, AI_TYPE_FLOAT).f = d; }
void AiNodeParamBool(AtList *params, int, const char *pname, bool d) { declare(params, pname, AI_TYPE_BOOLEAN).b = d; }
void AiNodeParamStr(AtList *params, int, const char *pname, const char *d) { declare(params, pname, AI_TYPE_STRING).s = d ? d : ""; }
void AiNodeParamEnum(AtList *params, int, const char *pname, int d, const char **e) {
  Param &p = declare(params, pname, AI_TYPE_ENUM);
  p.i = d;
  for (; e && *e; ++e) p.enum_values.push_back(*e);
}
bool AiMetaDataSetBool(AtNodeEntry *nentry, const char *param, const char *name, bool value) {
  nentry->meta[std::string(param ? param : "") + "|" + name] = value ? "true" : "false";
  return true;
}
bool AiMetaDataSetStr(AtNodeEntry *nentry, const char *param, const char *name, const char *value) {
  nentry->meta[std::string(param ? param : "") + "|" + name] = value ? value : "";
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// nodes, universe, arrays
// ---------------------------------------------------------------------------------------------------------------
static AtNode *make_node(AtUniverse *u, AtNodeEntry *e, const std::string &name) {
  AtNode *n = new AtNode();
  n->e = e; n->u = u; n->name = name;
  for (auto &kv : e->params) n->p[kv.first] = kv.second;
  ++e->count;
  u->nodes.push_back(n);
  return n;
}
static Param *find(const AtNode *node, AtString param) {
  if (!node) return nullptr;
  auto it = const_cast<AtNode *>(node)->p.find(param.c_str());
  return it == node->p.end() ? nullptr : &it->second;
}
AtUniverse *AiNodeGetUniverse(const AtNode *node) { return node ? node->u : nullptr; }
AtNode *AiUniverseGetOptions(const AtUniverse *u) { return u ? u->options : nullptr; }
AtNode *AiUniverseGetCamera(const AtUniverse *u) { return u ? u->camera : nullptr; }
AtRenderSession *AiUniverseGetRenderSession(const AtUniverse *u) { return u ? const_cast<AtRenderSession *>(&u->session) : nullptr; }
bool AiRenderSetHintInt(AtRenderSession *s, AtString hint, int value) { if (!s) return false; s->hints[hint.c_str()] = value; return true; }
AtNode *AiNode(AtUniverse *u, AtString nentry_name, AtString name) {
  if (!u) return nullptr;
  auto it = g_entries.find(nentry_name.c_str());
  if (it == g_entries.end()) { AiMsgError("[fake] unknown node entry '%s'", nentry_name.c_str()); return nullptr; }
  return make_node(u, it->second, name.c_str());
}
AtNode *AiNodeLookUpByName(const AtUniverse *u, AtString name) {
  if (!u) return nullptr;
  for (AtNode *n : u->nodes) if (n->name == name.c_str()) return n;
  return nullptr;
}
struct AtNodeIterator { const AtUniverse *u; unsigned mask; size_t i; };
static void skip(AtNodeIterator *it) { while (it->i < it->u->nodes.size() && !(it->u->nodes[it->i]->e->node_type & it->mask) && it->mask != AI_NODE_ALL) ++it->i; }
AtNodeIterator *AiUniverseGetNodeIterator(const AtUniverse *u, unsigned int mask) { AtNodeIterator *it = new AtNodeIterator{u, mask, 0}; skip(it); return it; }
bool AiNodeIteratorFinished(const AtNodeIterator *it) { return it->i >= it->u->nodes.size(); }
AtNode *AiNodeIteratorGetNext(AtNodeIterator *it) { if (it->i >= it->u->nodes.size()) return nullptr; AtNode *n = it->u->nodes[it->i++]; skip(it); return n; }
void AiNodeIteratorDestroy(AtNodeIterator *it) { delete it; }
const char *AiNodeGetName(const AtNode *node) { return node ? node->name.c_str() : ""; }
const AtNodeEntry *AiNodeGetNodeEntry(const AtNode *node) { return node ? node->e : nullptr; }
AtString AiNodeEntryGetNameAtString(const AtNodeEntry *e) { return AtString(e ? e->name.c_str() : ""); }
const AtNodeEntry *AiNodeEntryLookUp(AtString name) { auto it = g_entries.find(name.c_str()); return it == g_entries.end() ? nullptr : it->second; }
int AiNodeEntryGetCount(const AtNodeEntry *e) { return e ? e->count : 0; }
bool AiNodeIs(const AtNode *node, AtString str) { return node && node->e->name == str.c_str(); }
void AiNodeSetLocalData(AtNode *node, void *data) { if (node) node->local = data; }
void *AiNodeGetLocalData(const AtNode *node) { return node ? node->local : nullptr; }
int AiNodeGetInt(const AtNode *node, AtString param) { Param *p = find(node, param); return p ? p->i : 0; }
bool AiNodeGetBool(const AtNode *node, AtString param) { Param *p = find(node, param); return p ? p->b : false; }
float AiNodeGetFlt(const AtNode *node, AtString param) { Param *p = find(node, param); return p ? p->f : 0.f; }
AtString AiNodeGetStr(const AtNode *node, AtString param) { Param *p = find(node, param); return AtString(p ? p->s.c_str() : ""); }
AtArray *AiNodeGetArray(const AtNode *node, AtString param) { Param *p = find(node, param); return p ? p->arr : nullptr; }
void AiNodeSetStr(AtNode *node, AtString param, AtString value) { if (node) { Param &p = node->p[param.c_str()]; p.type = AI_TYPE_STRING; p.s = value.c_str(); } }
bool AiNodeSetArray(AtNode *node, AtString param, AtArray *array) { if (!node) return false; Param &p = node->p[param.c_str()]; p.type = AI_TYPE_ARRAY; p.arr = array; return true; }
bool AiNodeLink(AtNode *src, AtString input, AtNode *target) { if (!src || !target) return false; target->links[input.c_str()] = src; return true; }
uint8_t AiArrayGetNumKeys(const AtArray *a) { return a ? a->nkeys : 0; }
uint32_t AiArrayGetNumElements(const AtArray *a) { return a ? (uint32_t)(a->type == AI_TYPE_STRING ? a->strs.size() : a->ptrs.size()) : 0; }
AtString AiArrayGetStr(const AtArray *a, uint32_t i) { return AtString(a && i < a->strs.size() ? a->strs[i].c_str() : ""); }
void *AiArrayGetPtr(const AtArray *a, uint32_t i) { return a && i < a->ptrs.size() ? a->ptrs[i] : nullptr; }
AtArray *AiArrayAllocate(uint32_t n, uint8_t, uint8_t type) {
  AtArray *a = new AtArray();
  a->type = type;
  if (type == AI_TYPE_STRING) a->strs.resize(n); else a->ptrs.resize(n);
  return a;
}
bool AiArraySetStr(AtArray *a, uint32_t i, AtString v) { if (!a || i >= a->strs.size()) return false; a->strs[i] = v.c_str(); return true; }
bool AiArraySetPtr(AtArray *a, uint32_t i, void *p) { if (!a || i >= a->ptrs.size()) return false; a->ptrs[i] = p; return true; }
void AiArrayResize(AtArray *a, uint32_t n, uint8_t) { if (!a) return; if (a->type == AI_TYPE_STRING) a->strs.resize(n); else a->ptrs.resize(n); }

void AiCameraInitialize(AtNode *) {}
void AiCameraUpdate(AtNode *, bool) {}
float AiCameraGetShutterStart() { return 0.f; }
float AiCameraGetShutterEnd() { return 0.f; }
void AiFilterInitialize(AtNode *node, bool, const char **required_aovs) {
  node->required_aovs.clear();
  for (; required_aovs && *required_aovs; ++required_aovs) node->required_aovs.push_back(*required_aovs);
}
void AiFilterUpdate(AtNode *node, float width) { node->filter_width = width; }
void AiDriverInitialize(AtNode *, bool) {}
static void identity(AtMatrix &m) { memset(&m, 0, sizeof m); for (int i = 0; i < 4; ++i) m.data[i][i] = 1.f; }
// The stand-in's camera carries WORLD-TO-CAMERA keys in its "matrix" array (fa_camera_set_matrix_keys); between keys the
// matrix is interpolated component-wise, ((b - a) * f) + a -- what this repo defines a moving camera's matrix to be
// (include/lentil_hip.h, lentil_hip_set_camera_motion; how Arnold itself interpolates is not in the reference tree).
void AiWorldToCameraMatrix(const AtNode *node, float time, AtMatrix &out) {
  identity(out);
  const AtArray *a = node ? AiNodeGetArray(node, AtString("matrix")) : nullptr;
  if (!a || a->mats.size() < 16) return;
  const int n = a->nkeys;
  if (n < 2) { memcpy(out.data, a->mats.data(), 64); return; }
  // `time` is absolute; the array's keys lie at equidistant times over the node's motion range (motion_start ... motion_end,
  // 0 ... 1 unless set, as in Arnold)
  float ms = AiNodeGetFlt(node, AtString("motion_start")), me = AiNodeGetFlt(node, AtString("motion_end"));
  if (!(me > ms)) { ms = 0.f; me = 1.f; }
  float t = (time - ms) / (me - ms);
  t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
  const float sc = t * (float)(n - 1);
  int i0 = (int)sc;
  if (i0 > n - 2) i0 = n - 2;
  const float f = sc - (float)i0;
  const float *ka = a->mats.data() + (size_t)i0 * 16, *kb = ka + 16;
  for (int i = 0; i < 16; ++i) (&out.data[0][0])[i] = ((kb[i] - ka[i]) * f) + ka[i];
}
void AiCameraToWorldMatrix(const AtNode *, float, AtMatrix &out) { identity(out); }

// ---- ray probes: the stand-in's whole scene is one sphere (fa_set_sphere_occluder; radius <= 0: nothing).  A probe along
// origin + t * dir, 0 <= t <= maxdist, is occluded when the segment passes through it -- the closest point of the segment to
// the centre lies inside, in fp64 from the ray's fp32 members (the formula of oracle/lentil_oracle.cpp's orc_sphere_occluder,
// there on the segment's end points).
struct AtShaderGlobals { int unused; };
static float g_sphere[4] = {0.f, 0.f, 0.f, 0.f};
static std::atomic<uint64_t> g_probes{0}, g_probe_hits{0};
AtVector AiV3Normalize(const AtVector &a) {
  const float len = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  if (len == 0.f) return AtVector(0.f, 0.f, 0.f);
  const float inv = 1.0f / len;
  return AtVector(a.x * inv, a.y * inv, a.z * inv);
}
float AiV3Dist(const AtVector &a, const AtVector &b) {
  const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
  return sqrtf(x * x + y * y + z * z);
}
AtShaderGlobals *AiShaderGlobals() { return new AtShaderGlobals(); }
void AiShaderGlobalsDestroy(AtShaderGlobals *sg) { delete sg; }
AtRay AiMakeRay(uint8_t type, const AtVector &origin, const AtVector *dir, float maxdist, const AtShaderGlobals *) {
  AtRay r;
  r.type = type;
  r.origin = origin;
  r.dir = dir ? *dir : AtVector(0.f, 0.f, 0.f);
  r.mindist = 0.f;
  r.maxdist = maxdist;
  return r;
}
bool AiTraceProbe(const AtRay &ray, const AtShaderGlobals *) {
  g_probes.fetch_add(1, std::memory_order_relaxed);
  if (!(g_sphere[3] > 0.f)) return false;
  const double o[3] = {ray.origin.x, ray.origin.y, ray.origin.z};
  const double d[3] = {(double)ray.dir.x * (double)ray.maxdist, (double)ray.dir.y * (double)ray.maxdist, (double)ray.dir.z * (double)ray.maxdist};
  const double c[3] = {g_sphere[0] - o[0], g_sphere[1] - o[1], g_sphere[2] - o[2]};
  const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  double t = dd > 0.0 ? (c[0] * d[0] + c[1] * d[1] + c[2] * d[2]) / dd : 0.0;
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  const double q[3] = {c[0] - t * d[0], c[1] - t * d[1], c[2] - t * d[2]};
  const bool hit = (q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) < (double)g_sphere[3] * (double)g_sphere[3];
  if (hit) g_probe_hits.fetch_add(1, std::memory_order_relaxed);
  return hit;
}

// ---------------------------------------------------------------------------------------------------------------
// iterators
// ---------------------------------------------------------------------------------------------------------------
bool AiAOVSampleIteratorGetNext(AtAOVSampleIterator *it) { it->depth = -1; if (it->cur + 1 >= (int)it->ids->size()) return false; ++it->cur; return true; }
void AiAOVSampleIteratorReset(AtAOVSampleIterator *it) { it->cur = -1; it->depth = -1; }
// walks the depth entries of the current sample; after the last one the iterator has moved on to the next sample
// (what the reference's comment at src/lentil.h:806-807 reports of the SDK), so callers reset to their sample
bool AiAOVSampleIteratorGetNextDepth(AtAOVSampleIterator *it) {
  const SampleStore &s = it->u->samples;
  if (it->cur < 0 || it->ids->empty() || s.depth_count.empty()) return false;
  const int id = (*it->ids)[(size_t)it->cur];
  if (it->depth + 1 < s.depth_count[(size_t)id]) { ++it->depth; return true; }
  it->depth = -1;
  if (it->cur + 1 < (int)it->ids->size()) ++it->cur;
  return false;
}
float AiFastExp(float x) { return exp2f(x * 1.44269504088896340736f); }       // (see ai.h: recognisably not expf)
void AiAOVSampleIteratorGetPixel(AtAOVSampleIterator *it, int &x, int &y) { x = it->x; y = it->y; }
static int sid(const AtAOVSampleIterator *it) { return (*it->ids)[(size_t)(it->cur < 0 ? 0 : it->cur)]; }
AtVector2 AiAOVSampleIteratorGetOffset(AtAOVSampleIterator *it) { const int s = sid(it); return AtVector2{it->u->samples.ox[s], it->u->samples.oy[s]}; }
float AiAOVSampleIteratorGetInvDensity(AtAOVSampleIterator *it) { return it->u->samples.inv_density[sid(it)]; }
AtString AiAOVSampleIteratorGetAOVName(AtAOVSampleIterator *it) { return AtString(it->aov_name); }
static const float *aov4(AtAOVSampleIterator *it, const char *name) {
  static const float zero[4] = {0, 0, 0, 0};
  if (it->depth >= 0 && !it->ids->empty()) {
    auto d = it->u->samples.depth_aov.find(name);
    if (d != it->u->samples.depth_aov.end())
      return d->second.data() + ((size_t)it->u->samples.depth_start[(size_t)(*it->ids)[(size_t)it->cur]] + (size_t)it->depth) * 4;
  }
  auto f = it->u->samples.aov.find(name);
  if (f == it->u->samples.aov.end() || it->ids->empty()) return zero;
  return f->second.second.data() + (size_t)sid(it) * 4;
}
AtRGBA AiAOVSampleIteratorGetAOVRGBA(AtAOVSampleIterator *it, AtString name) { const float *v = aov4(it, name.c_str()); return AtRGBA(v[0], v[1], v[2], v[3]); }
AtRGB AiAOVSampleIteratorGetAOVRGB(AtAOVSampleIterator *it, AtString name) { const float *v = aov4(it, name.c_str()); return AtRGB(v[0], v[1], v[2]); }
AtVector AiAOVSampleIteratorGetAOVVec(AtAOVSampleIterator *it, AtString name) { const float *v = aov4(it, name.c_str()); return AtVector(v[0], v[1], v[2]); }
float AiAOVSampleIteratorGetAOVFlt(AtAOVSampleIterator *it, AtString name) { return aov4(it, name.c_str())[0]; }
AtRGBA AiAOVSampleIteratorGetRGBA(AtAOVSampleIterator *it) { return AiAOVSampleIteratorGetAOVRGBA(it, AtString(it->aov_name)); }
AtRGB AiAOVSampleIteratorGetRGB(AtAOVSampleIterator *it) { return AiAOVSampleIteratorGetAOVRGB(it, AtString(it->aov_name)); }
AtVector AiAOVSampleIteratorGetVec(AtAOVSampleIterator *it) { return AiAOVSampleIteratorGetAOVVec(it, AtString(it->aov_name)); }
float AiAOVSampleIteratorGetFlt(AtAOVSampleIterator *it) { return AiAOVSampleIteratorGetAOVFlt(it, AtString(it->aov_name)); }
void AiOutputIteratorReset(AtOutputIterator *it) { it->cur = 0; }
bool AiOutputIteratorGetNext(AtOutputIterator *it, AtString *output_name, int *pixel_type, const void **bucket_data) {
  if (it->cur >= it->slots.size()) return false;
  const OutputSlot &s = it->slots[it->cur++];
  if (output_name) *output_name = AtString(s.name);
  if (pixel_type) *pixel_type = s.type;
  if (bucket_data) *bucket_data = s.bucket;
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// harness
// ---------------------------------------------------------------------------------------------------------------
FA_API int fa_load_plugin(const char *path) {
  builtin_entries();
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "fa_load_plugin: %s\n", dlerror()); return -1; }
  typedef bool (*Loader)(int, AtNodeLib *);
  Loader loader = (Loader)dlsym(h, "NodeLoader");
  if (!loader) return -2;
  int n = 0;
  for (;; ++n) {
    AtNodeLib lib;
    memset(&lib, 0, sizeof lib);
    if (!loader(n, &lib)) break;
    AtNodeEntry *e = entry(lib.name, lib.node_type);
    if (e->m) continue;                      // loaded before (a second fa_load_plugin)
    e->node_type = lib.node_type;
    e->output_type = lib.output_type;
    e->m = (const AtNodeMethods *)lib.methods;
    e->meta["|ai_version"] = lib.version;
    AtList l{&e->params};
    e->m->cmethods->Parameters(&l, e);
    g_plugin_entries.push_back(e);
  }
  return n;
}
FA_API int fa_entry_count() { return (int)g_plugin_entries.size(); }
FA_API const char *fa_entry_name(int i) { return i >= 0 && i < (int)g_plugin_entries.size() ? g_plugin_entries[i]->name.c_str() : ""; }
FA_API int fa_entry_node_type(int i) { return i >= 0 && i < (int)g_plugin_entries.size() ? g_plugin_entries[i]->node_type : -1; }
FA_API const char *fa_entry_meta(const char *entry_name, const char *param, const char *key) {
  auto it = g_entries.find(entry_name);
  if (it == g_entries.end()) return "";
  auto m = it->second->meta.find(std::string(param ? param : "") + "|" + key);
  return m == it->second->meta.end() ? "" : m->second.c_str();
}
FA_API int fa_entry_param_count(const char *entry_name) {
  auto it = g_entries.find(entry_name);
  return it == g_entries.end() ? -1 : (int)it->second->params.size();
}
// name, type, numeric default, string default / "a|b|c" for enums
FA_API int fa_entry_param(const char *entry_name, int i, char *name, int name_cap, int *type, double *num, char *str, int str_cap) {
  auto it = g_entries.find(entry_name);
  if (it == g_entries.end() || i < 0 || i >= (int)it->second->params.size()) return -1;
  const auto &kv = it->second->params[(size_t)i];
  snprintf(name, (size_t)name_cap, "%s", kv.first.c_str());
  *type = kv.second.type;
  *num = kv.second.type == AI_TYPE_FLOAT ? kv.second.f : (kv.second.type == AI_TYPE_BOOLEAN ? (kv.second.b ? 1 : 0) : kv.second.i);
  std::string s = kv.second.s;
  for (size_t k = 0; k < kv.second.enum_values.size(); ++k) s += (k ? "|" : "") + kv.second.enum_values[k];
  snprintf(str, (size_t)str_cap, "%s", s.c_str());
  return 0;
}

FA_API AtUniverse *fa_universe_create(int xres, int yres, int aa_samples) {
  builtin_entries();
  AtUniverse *u = new AtUniverse();
  u->xres = xres; u->yres = yres;
  AtNode *o = make_node(u, entry("options"), "options");
  auto seti = [&](const char *k, int v) { Param p; p.type = AI_TYPE_INT; p.i = v; o->p[k] = p; };
  auto setb = [&](const char *k, bool v) { Param p; p.type = AI_TYPE_BOOLEAN; p.b = v; o->p[k] = p; };
  auto setf = [&](const char *k, float v) { Param p; p.type = AI_TYPE_FLOAT; p.f = v; o->p[k] = p; };
  seti("xres", xres); seti("yres", yres); seti("AA_samples", aa_samples);
  for (const char *k : {"region_min_x", "region_min_y", "region_max_x", "region_max_y"}) seti(k, INT32_MIN);
  setb("enable_adaptive_sampling", false); setb("ignore_dof", false); setf("meters_per_unit", 1.0f);
  seti("threads", 0);
  Param outs; outs.type = AI_TYPE_ARRAY; outs.arr = AiArrayAllocate(0, 1, AI_TYPE_STRING); o->p["outputs"] = outs;
  Param shaders; shaders.type = AI_TYPE_ARRAY; shaders.arr = AiArrayAllocate(0, 1, AI_TYPE_NODE); o->p["aov_shaders"] = shaders;
  u->options = o;
  return u;
}
FA_API AtNode *fa_node(AtUniverse *u, const char *entry_name, const char *name) { return AiNode(u, AtString(entry_name), AtString(name)); }
FA_API void fa_set_camera(AtUniverse *u, AtNode *n) { u->camera = n; }
FA_API void fa_node_set_int(AtNode *n, const char *k, int v) { Param &p = n->p[k]; if (p.type == AI_TYPE_NONE) p.type = AI_TYPE_INT; p.i = v; }
FA_API void fa_node_set_flt(AtNode *n, const char *k, float v) { Param &p = n->p[k]; p.type = AI_TYPE_FLOAT; p.f = v; }
FA_API void fa_node_set_bool(AtNode *n, const char *k, int v) { Param &p = n->p[k]; p.type = AI_TYPE_BOOLEAN; p.b = v != 0; }
FA_API void fa_node_set_str(AtNode *n, const char *k, const char *v) { Param &p = n->p[k]; p.type = AI_TYPE_STRING; p.s = v; }
FA_API AtNode *fa_options(AtUniverse *u) { return u->options; }
FA_API void fa_add_output(AtUniverse *u, const char *output) {
  AtArray *a = u->options->p["outputs"].arr;
  a->strs.push_back(output);
}
FA_API int fa_output_count(AtUniverse *u) { return (int)u->options->p["outputs"].arr->strs.size(); }
FA_API const char *fa_output(AtUniverse *u, int i) { return u->options->p["outputs"].arr->strs[(size_t)i].c_str(); }

FA_API void fa_set_samples(AtUniverse *u, int n, const int *px, const int *py, const float *ox, const float *oy, const float *inv_density) {
  SampleStore &s = u->samples;
  s.n = n;
  s.px.assign(px, px + n); s.py.assign(py, py + n);
  s.ox.assign(ox, ox + n); s.oy.assign(oy, oy + n);
  s.inv_density.assign(inv_density, inv_density + n);
  s.per_pixel.assign((size_t)u->xres * u->yres, std::vector<int>());
  for (int i = 0; i < n; ++i)
    if (px[i] >= 0 && py[i] >= 0 && px[i] < u->xres && py[i] < u->yres) s.per_pixel[(size_t)py[i] * u->xres + px[i]].push_back(i);
}
FA_API void fa_set_depths(AtUniverse *u, const int *counts) {
  SampleStore &s = u->samples;
  s.depth_count.assign(counts, counts + s.n);
  s.depth_start.resize((size_t)s.n);
  int at = 0;
  for (int i = 0; i < s.n; ++i) { s.depth_start[(size_t)i] = at; at += counts[i]; }
}
FA_API void fa_set_depth_aov(AtUniverse *u, const char *name, const float *data4) {
  const SampleStore &s = u->samples;
  const size_t total = s.n ? (size_t)s.depth_start.back() + (size_t)s.depth_count.back() : 0;
  u->samples.depth_aov[name].assign(data4, data4 + total * 4);
}
FA_API void fa_add_late_output(AtUniverse *u, const char *output) { u->late_outputs.push_back(output); }
FA_API void fa_add_aov_shader(AtUniverse *u, AtNode *n) { u->options->p["aov_shaders"].arr->ptrs.push_back(n); }
FA_API void fa_set_aov(AtUniverse *u, const char *name, int type, const float *data4) {
  auto &slot = u->samples.aov[name];
  slot.first = (uint8_t)type;
  slot.second.assign(data4, data4 + (size_t)u->samples.n * 4);
}

struct ParsedOutput { std::string aov, type, filter, driver; };
static ParsedOutput parse_output(const std::string &s) {
  std::vector<std::string> t;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && s[i] == ' ') ++i;
    size_t j = i;
    while (j < s.size() && s[j] != ' ') ++j;
    if (j > i) t.push_back(s.substr(i, j - i));
    i = j;
  }
  ParsedOutput o;
  // [camera] aov type filter driver [HALF]
  const size_t off = (t.size() >= 5 && t.back() != "HALF") || t.size() >= 6 ? 1 : 0;
  if (t.size() >= off + 4) { o.aov = t[off]; o.type = t[off + 1]; o.filter = t[off + 2]; o.driver = t[off + 3]; }
  return o;
}
static int type_code(const std::string &t) {
  if (t == "RGBA") return AI_TYPE_RGBA;
  if (t == "RGB") return AI_TYPE_RGB;
  if (t == "FLOAT") return AI_TYPE_FLOAT;
  if (t == "VECTOR") return AI_TYPE_VECTOR;
  return AI_TYPE_NONE;
}

// operators only (what Arnold does before the nodes are updated); returns how many cooked successfully
FA_API int fa_cook_operators(AtUniverse *u) {
  int ok = 0;
  for (size_t k = 0; k < u->nodes.size(); ++k) {
    AtNode *n = u->nodes[k];
    if (n->e->node_type != AI_NODE_OPERATOR || !n->e->m) continue;
    const AtOperatorNodeMethods *om = (const AtOperatorNodeMethods *)n->e->m->dmethods;
    if (!n->initialized) { n->e->m->cmethods->Initialize(n); om->Init(n, &n->op_user_data); n->initialized = true; }
    if (om->Cook(nullptr, n, nullptr, n->op_user_data, nullptr, nullptr)) ++ok;
    om->PostCook(n, n->op_user_data);
  }
  return ok;
}

FA_API int fa_render(AtUniverse *u, int n_threads, int bucket_size) {
  if (n_threads < 1) n_threads = 1;
  fa_node_set_int(u->options, "threads", n_threads);
  // 1. operators
  fa_cook_operators(u);
  // (outputs other plugins add in their own update, after the operators: cryptomatte's ranked AOVs)
  for (const std::string &o : u->late_outputs) u->options->p["outputs"].arr->strs.push_back(o);
  u->late_outputs.clear();
  // 2. node initialise / update: camera first
  std::vector<AtNode *> order;
  if (u->camera) order.push_back(u->camera);
  for (AtNode *n : u->nodes) if (n != u->camera && n->e->m && n->e->node_type != AI_NODE_OPERATOR) order.push_back(n);
  for (AtNode *n : order) {
    if (!n->e->plugin_inited) { n->e->m->cmethods->PluginInitialize(&n->e->plugin_data); n->e->plugin_inited = true; }
    if (!n->initialized) { n->e->m->cmethods->Initialize(n); n->initialized = true; }
    n->e->m->cmethods->Update(n);
  }
  if (g_aborts) return -1;
  // 3. outputs filtered by a plugin filter node
  struct Out { ParsedOutput po; AtNode *filter; int in_type, out_type; const char *aov; };
  std::vector<Out> outs;
  AtArray *oa = u->options->p["outputs"].arr;
  for (const std::string &s : oa->strs) {
    Out o;
    o.po = parse_output(s);
    o.filter = AiNodeLookUpByName(u, AtString(o.po.filter.c_str()));
    if (!o.filter || !o.filter->e->m || o.filter->e->node_type != AI_NODE_FILTER) continue;
    o.in_type = type_code(o.po.type);
    o.out_type = ((const AtFilterNodeMethods *)o.filter->e->m->dmethods)->FilterOutputType(o.filter, (uint8_t)o.in_type);
    o.aov = AiFakeIntern(o.po.aov.c_str());
    if (o.out_type == AI_TYPE_NONE) continue;
    u->images[o.po.aov].assign((size_t)u->xres * u->yres * 4, 0.f);
    outs.push_back(o);
  }
  // 4. filter_pixel, rows interleaved over the threads
  {
    std::vector<std::thread> ts;
    for (int t = 0; t < n_threads; ++t) {
      ts.emplace_back([&, t]() {
        for (int y = t; y < u->yres; y += n_threads)
          for (int x = 0; x < u->xres; ++x)
            for (const Out &o : outs) {
              AtAOVSampleIterator it{u, x, y, &u->samples.per_pixel[(size_t)y * u->xres + x], -1, o.aov};
              float data[4] = {0, 0, 0, 0};
              ((const AtFilterNodeMethods *)o.filter->e->m->dmethods)->FilterPixel(o.filter, &it, data, (uint8_t)o.out_type);
              float *dst = u->images[o.po.aov].data() + ((size_t)y * u->xres + x) * 4;
              const int nc = o.out_type == AI_TYPE_RGBA ? 4 : (o.out_type == AI_TYPE_FLOAT ? 1 : 3);
              for (int c = 0; c < nc; ++c) dst[c] = data[c];
            }
      });
    }
    for (auto &t : ts) t.join();
  }
  u->display = u->images;
  // 5. imagers: driver_process_bucket per bucket, in place
  for (AtNode *n : u->nodes) {
    if (!n->e->m || n->e->node_type != AI_NODE_DRIVER) continue;
    auto sub = n->e->meta.find("|subtype");
    if (sub == n->e->meta.end() || sub->second != "imager") continue;
    const AtDriverNodeMethods *dm = (const AtDriverNodeMethods *)n->e->m->dmethods;
    struct B { int xo, yo, sx, sy; };
    std::vector<B> buckets;
    for (int yo = 0; yo < u->yres; yo += bucket_size)
      for (int xo = 0; xo < u->xres; xo += bucket_size)
        buckets.push_back(B{xo, yo, std::min(bucket_size, u->xres - xo), std::min(bucket_size, u->yres - yo)});
    std::atomic<size_t> next{0};
    std::vector<std::thread> ts;
    for (int t = 0; t < n_threads; ++t) {
      ts.emplace_back([&, t]() {
        for (;;) {
          const size_t b = next++;
          if (b >= buckets.size()) break;
          const B &bk = buckets[b];
          if (!dm->DriverNeedsBucket(n, bk.xo, bk.yo, bk.sx, bk.sy, (uint16_t)t)) continue;
          dm->DriverPrepareBucket(n, bk.xo, bk.yo, bk.sx, bk.sy, (uint16_t)t);
          std::vector<std::vector<float>> data(outs.size());
          AtOutputIterator it;
          it.cur = 0;
          for (size_t k = 0; k < outs.size(); ++k) {
            data[k].resize((size_t)bk.sx * bk.sy * 4);
            const float *img = u->images[outs[k].po.aov].data();
            for (int j = 0; j < bk.sy; ++j)
              memcpy(data[k].data() + (size_t)j * bk.sx * 4, img + ((size_t)(bk.yo + j) * u->xres + bk.xo) * 4, (size_t)bk.sx * 16);
            it.slots.push_back(OutputSlot{outs[k].aov, AI_TYPE_RGBA, data[k].data()});
          }
          dm->DriverProcessBucket(n, &it, nullptr, bk.xo, bk.yo, bk.sx, bk.sy, (uint16_t)t);
          for (size_t k = 0; k < outs.size(); ++k) {
            float *img = u->images[outs[k].po.aov].data();
            for (int j = 0; j < bk.sy; ++j)
              memcpy(img + ((size_t)(bk.yo + j) * u->xres + bk.xo) * 4, data[k].data() + (size_t)j * bk.sx * 4, (size_t)bk.sx * 16);
          }
        }
      });
    }
    for (auto &t : ts) t.join();
  }
  return g_aborts ? -1 : 0;
}

FA_API int fa_get_image(AtUniverse *u, const char *aov, float *dst) {
  auto it = u->images.find(aov);
  if (it == u->images.end()) return -1;
  memcpy(dst, it->second.data(), it->second.size() * sizeof(float));
  return 0;
}
// what filter_pixel returned (the display pass-through), before the imagers ran
FA_API int fa_get_display_image(AtUniverse *u, const char *aov, float *dst) {
  auto it = u->display.find(aov);
  if (it == u->display.end()) return -1;
  memcpy(dst, it->second.data(), it->second.size() * sizeof(float));
  return 0;
}
// the camera node's own entry points, through its method table (what Arnold calls per camera sample):
// in = sx, sy, dsx, dsy, lensx, lensy, relative_time; out = origin, dir, dOdx, dOdy, dDdx, dDdy, weight (21 floats)
// the scene's occluder (process-wide, like the render the SDK's AiShaderGlobals() belongs to), the probes counted
FA_API void fa_set_sphere_occluder(float cx, float cy, float cz, float r) { g_sphere[0] = cx; g_sphere[1] = cy; g_sphere[2] = cz; g_sphere[3] = r; }
FA_API void fa_probe_counts(uint64_t out[2], int reset) {
  out[0] = g_probes.load(); out[1] = g_probe_hits.load();
  if (reset) { g_probes.store(0); g_probe_hits.store(0); }
}
// A lentil_probe_fn (include/lentil_hip.h) for the oracle's side of a plugin test: every segment probed the way the plugin's
// callback probes it (pota_amd/csrc/plugin/lentil_camera_node.cpp, arnold_probe) -- AiMakeRay(AI_RAY_SHADOW, origin,
// normalize(target - origin), dist(target, origin)) then AiTraceProbe.
struct FaSegment { float origin[3], target[3]; };
FA_API void fa_probe_segments(void *, uint64_t n, const FaSegment *seg, uint8_t *occluded) {
  AtShaderGlobals *sg = AiShaderGlobals();
  for (uint64_t i = 0; i < n; ++i) {
    const AtVector o(seg[i].origin[0], seg[i].origin[1], seg[i].origin[2]), t(seg[i].target[0], seg[i].target[1], seg[i].target[2]);
    const AtVector dir = AiV3Normalize(t - o);
    const AtRay ray = AiMakeRay(AI_RAY_SHADOW, o, &dir, AiV3Dist(t, o), sg);
    occluded[i] = AiTraceProbe(ray, sg) ? 1 : 0;
  }
  AiShaderGlobalsDestroy(sg);
}
FA_API int fa_camera_create_ray(AtUniverse *u, const float in[7], float out[21], int tid) {
  if (!u->camera || !u->camera->e->m || u->camera->e->node_type != AI_NODE_CAMERA) return -1;
  const AtCameraNodeMethods *cm = (const AtCameraNodeMethods *)u->camera->e->m->dmethods;
  AtCameraInput ci{in[0], in[1], in[2], in[3], in[4], in[5], in[6]};
  AtCameraOutput co{};
  cm->CreateRay(u->camera, ci, co, (uint16_t)tid);
  const AtVector *v[6] = {&co.origin, &co.dir, &co.dOdx, &co.dOdy, &co.dDdx, &co.dDdy};
  for (int k = 0; k < 6; ++k) { out[3 * k] = v[k]->x; out[3 * k + 1] = v[k]->y; out[3 * k + 2] = v[k]->z; }
  out[18] = co.weight.r; out[19] = co.weight.g; out[20] = co.weight.b;
  return 0;
}
FA_API int fa_camera_reverse_ray(AtUniverse *u, const float po[3], float relative_time, float ps[2]) {
  if (!u->camera || !u->camera->e->m || u->camera->e->node_type != AI_NODE_CAMERA) return -1;
  const AtCameraNodeMethods *cm = (const AtCameraNodeMethods *)u->camera->e->m->dmethods;
  AtVector2 r{0.f, 0.f};
  const bool ok = cm->ReverseRay(u->camera, AtVector(po[0], po[1], po[2]), relative_time, r);
  ps[0] = r.x; ps[1] = r.y;
  return ok ? 1 : 0;
}
// a moving camera: n world-to-camera matrices (row-vector convention) over the shutter
FA_API int fa_camera_set_matrix_keys(AtUniverse *u, int n, const float *mats) {
  if (!u->camera || n < 1 || n > 255) return -1;
  AtArray *a = new AtArray();
  a->type = AI_TYPE_MATRIX;
  a->nkeys = (uint8_t)n;
  a->mats.assign(mats, mats + (size_t)n * 16);
  u->arrays.push_back(a);
  AiNodeSetArray(u->camera, AtString("matrix"), a);
  return 0;
}
FA_API int fa_filter_width_x1000(AtUniverse *u, const char *node_name) {
  AtNode *n = AiNodeLookUpByName(u, AtString(node_name));
  return n ? (int)lroundf(n->filter_width * 1000.f) : -1;
}
FA_API int fa_render_hint(AtUniverse *u, const char *hint) {
  auto it = u->session.hints.find(hint);
  return it == u->session.hints.end() ? -12345 : it->second;
}
FA_API int fa_node_exists(AtUniverse *u, const char *name) { return AiNodeLookUpByName(u, AtString(name)) ? 1 : 0; }
FA_API int fa_aov_shader_count(AtUniverse *u) { return (int)u->options->p["aov_shaders"].arr->ptrs.size(); }
FA_API void fa_universe_destroy(AtUniverse *u) {
  for (AtNode *n : u->nodes) {
    if (!n->e->m) continue;
    if (n->e->node_type == AI_NODE_OPERATOR) {
      if (n->initialized) ((const AtOperatorNodeMethods *)n->e->m->dmethods)->Cleanup(n, n->op_user_data);
    } else if (n->initialized) {
      n->e->m->cmethods->Finish(n);
    }
  }
  for (AtNode *n : u->nodes) { --n->e->count; delete n; }
  delete u;
}
FA_API int fa_messages(char *buf, int cap) {
  std::lock_guard<std::mutex> g(g_mutex);
  snprintf(buf, (size_t)cap, "%s", g_messages.c_str());
  return (int)g_messages.size();
}
FA_API void fa_messages_clear() { std::lock_guard<std::mutex> g(g_mutex); g_messages.clear(); g_errors = 0; g_aborts = 0; }
FA_API int fa_error_count() { return g_errors; }
