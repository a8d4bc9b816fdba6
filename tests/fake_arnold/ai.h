// ai.h -- a stand-in for the Arnold 7.x SDK header, TEST INFRASTRUCTURE ONLY.
//
// The Arnold SDK is not in this image.  This header declares the slice of its API that the plugin glue under
// pota_amd/csrc/plugin/ uses, with the 7.x signatures as SURVEY.md (section 8b, appendix A) lists them: universe-aware
// getters, the node-method export macros, imagers as drivers with subtype "imager", the AOV sample and output
// iterators.  libai_fake.so (fake_arnold.cpp) implements it over an in-memory universe plus a small harness
// (fa_* functions, C linkage) that plays the renderer: loads the plugin through NodeLoader, cooks operators, updates
// nodes, calls filter_pixel from several threads and driver_process_bucket per bucket.
// Written from the SDK's documented surface as recalled -- type codes, struct layouts and macro bodies are this
// file's own; nothing here is Autodesk's code.  The real plugin is compiled against the real <ai.h>.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define AI_VERSION "7.2.0.0"
#define AI_MAXSIZE_VERSION 32
#define AI_EXPORT_LIB extern "C" __attribute__((visibility("default")))
#define AI_API __attribute__((visibility("default")))

// parameter / pixel type codes
#define AI_TYPE_BYTE 0x00
#define AI_TYPE_INT 0x01
#define AI_TYPE_UINT 0x02
#define AI_TYPE_BOOLEAN 0x03
#define AI_TYPE_FLOAT 0x04
#define AI_TYPE_RGB 0x05
#define AI_TYPE_RGBA 0x06
#define AI_TYPE_VECTOR 0x07
#define AI_TYPE_VECTOR2 0x09
#define AI_TYPE_STRING 0x0A
#define AI_TYPE_POINTER 0x0B
#define AI_TYPE_NODE 0x0C
#define AI_TYPE_ARRAY 0x0D
#define AI_TYPE_MATRIX 0x0E
#define AI_TYPE_ENUM 0x0F
#define AI_TYPE_NONE 0xFF
#define AI_TYPE_UNDEFINED 0xFF

// node types
#define AI_NODE_UNDEFINED 0x0000
#define AI_NODE_OPTIONS 0x0001
#define AI_NODE_CAMERA 0x0002
#define AI_NODE_SHADER 0x0010
#define AI_NODE_DRIVER 0x0040
#define AI_NODE_FILTER 0x0080
#define AI_NODE_OPERATOR 0x1000
#define AI_NODE_ALL 0xFFFF

#define AI_PI 3.14159265358979323846f
// The SDK's fast exp is an exported approximation whose bits only the SDK has.  The stand-in is deliberately NOT expf
// (2^(x log2 e) through exp2f), so that a test can tell whether the plugin went through this function or through libm's.
AI_API float AiFastExp(float x);
#define AI_EPSILON 1.0e-4f
#define AI_BIG 1.0e12f
#define AI_INFINITE 1.0e30f

struct AtNode;
struct AtNodeEntry;
struct AtUniverse;
struct AtArray;
struct AtList;
struct AtRenderSession;
struct AtAOVSampleIterator;
struct AtOutputIterator;
struct AtCookContext;
struct AtNodeIterator;

AI_API const char *AiFakeIntern(const char *s);

struct AtString {
  const char *s;
  AtString() : s(nullptr) {}
  explicit AtString(const char *str) : s(str ? AiFakeIntern(str) : nullptr) {}
  const char *c_str() const { return s ? s : ""; }
  bool empty() const { return !s || !s[0]; }
  bool operator==(const AtString &o) const { return s == o.s || (c_str()[0] == 0 && o.c_str()[0] == 0); }
  bool operator!=(const AtString &o) const { return !(*this == o); }
};

struct AtVector2 { float x, y; };
struct AtVector {
  float x, y, z;
  AtVector() : x(0), y(0), z(0) {}
  AtVector(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct AtRGB {
  float r, g, b;
  AtRGB() : r(0), g(0), b(0) {}
  AtRGB(float a, float c, float d) : r(a), g(c), b(d) {}
};
struct AtRGBA {
  float r, g, b, a;
  AtRGBA() : r(0), g(0), b(0), a(0) {}
  AtRGBA(float x, float y, float z, float w) : r(x), g(y), b(z), a(w) {}
};
struct AtMatrix { float data[4][4]; };
struct AtBBox2 { int minx, miny, maxx, maxy; };

struct AtCameraInput { float sx, sy, dsx, dsy, lensx, lensy, relative_time; };
struct AtCameraOutput { AtVector origin, dir; AtVector dOdx, dOdy, dDdx, dDdy; AtRGB weight; };

struct AtNodeLib {
  const void *methods;
  uint8_t output_type;
  const char *name;
  int node_type;
  char version[AI_MAXSIZE_VERSION];
};

// ---- method tables --------------------------------------------------------------------------------------
struct AtCommonMethods {
  bool (*PluginInitialize)(void **plugin_data);
  void (*PluginCleanup)(void *plugin_data);
  void (*Parameters)(AtList *params, AtNodeEntry *nentry);
  void (*Initialize)(AtNode *node);
  void (*Update)(AtNode *node);
  void (*Finish)(AtNode *node);
};
struct AtNodeMethods {
  const AtCommonMethods *cmethods;
  const void *dmethods;
};
struct AtCameraNodeMethods {
  void (*CreateRay)(const AtNode *node, const AtCameraInput &input, AtCameraOutput &output, uint16_t tid);
  bool (*ReverseRay)(const AtNode *node, const AtVector &Po, float relative_time, AtVector2 &Ps);
};
struct AtFilterNodeMethods {
  uint8_t (*FilterOutputType)(const AtNode *node, uint8_t input_type);
  void (*FilterPixel)(AtNode *node, AtAOVSampleIterator *iterator, void *data_out, uint8_t data_type);
};
struct AtDriverNodeMethods {
  bool (*DriverSupportsPixelType)(const AtNode *node, uint8_t pixel_type);
  const char **(*DriverExtension)();
  void (*DriverOpen)(AtNode *node, AtOutputIterator *iterator, AtBBox2 display_window, AtBBox2 data_window, int bucket_size);
  bool (*DriverNeedsBucket)(AtNode *node, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid);
  void (*DriverPrepareBucket)(AtNode *node, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid);
  void (*DriverProcessBucket)(AtNode *node, AtOutputIterator *iterator, AtAOVSampleIterator *sample_iterator, int bucket_xo,
                              int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid);
  void (*DriverWriteBucket)(AtNode *node, AtOutputIterator *iterator, AtAOVSampleIterator *sample_iterator, int bucket_xo,
                            int bucket_yo, int bucket_size_x, int bucket_size_y);
  void (*DriverClose)(AtNode *node, AtOutputIterator *iterator);
};
struct AtOperatorNodeMethods {
  bool (*Init)(AtNode *op, void **user_data);
  bool (*Cleanup)(AtNode *op, void *user_data);
  bool (*Cook)(AtNode *node, AtNode *op, void *child_data, void *user_data, const AtArray *matching_params, AtCookContext *cook_context);
  bool (*PostCook)(AtNode *op, void *user_data);
};

// ---- node-method export macros ----------------------------------------------------------------------------
#define AI_FAKE_COMMON_METHODS                                                                               \
  static bool PluginInitialize(void **plugin_data);                                                          \
  static void PluginCleanup(void *plugin_data);                                                              \
  static void Parameters(AtList *params, AtNodeEntry *nentry);                                               \
  static void Initialize(AtNode *node);                                                                      \
  static void Update(AtNode *node);                                                                          \
  static void Finish(AtNode *node);                                                                          \
  static const AtCommonMethods ai_common_mtds = {PluginInitialize, PluginCleanup, Parameters, Initialize, Update, Finish};

#define AI_CAMERA_NODE_EXPORT_METHODS(tag)                                                                   \
  AI_FAKE_COMMON_METHODS                                                                                     \
  static void CameraCreateRay(const AtNode *node, const AtCameraInput &input, AtCameraOutput &output, uint16_t tid); \
  static bool CameraReverseRay(const AtNode *node, const AtVector &Po, float relative_time, AtVector2 &Ps);  \
  static const AtCameraNodeMethods ai_cam_mtds = {CameraCreateRay, CameraReverseRay};                        \
  static const AtNodeMethods ai_node_mtds = {&ai_common_mtds, &ai_cam_mtds};                                 \
  const AtNodeMethods *tag = &ai_node_mtds;

#define AI_FILTER_NODE_EXPORT_METHODS(tag)                                                                   \
  AI_FAKE_COMMON_METHODS                                                                                     \
  static uint8_t FilterOutputType(const AtNode *node, uint8_t input_type);                                   \
  static void FilterPixel(AtNode *node, AtAOVSampleIterator *iterator, void *data_out, uint8_t data_type);   \
  static const AtFilterNodeMethods ai_filter_mtds = {FilterOutputType, FilterPixel};                         \
  static const AtNodeMethods ai_node_mtds = {&ai_common_mtds, &ai_filter_mtds};                              \
  const AtNodeMethods *tag = &ai_node_mtds;

#define AI_DRIVER_NODE_EXPORT_METHODS(tag)                                                                   \
  AI_FAKE_COMMON_METHODS                                                                                     \
  static bool DriverSupportsPixelType(const AtNode *node, uint8_t pixel_type);                               \
  static const char **DriverExtension();                                                                     \
  static void DriverOpen(AtNode *node, AtOutputIterator *iterator, AtBBox2 display_window, AtBBox2 data_window, int bucket_size); \
  static bool DriverNeedsBucket(AtNode *node, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid); \
  static void DriverPrepareBucket(AtNode *node, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid); \
  static void DriverProcessBucket(AtNode *node, AtOutputIterator *iterator, AtAOVSampleIterator *sample_iterator,               \
                                  int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid);           \
  static void DriverWriteBucket(AtNode *node, AtOutputIterator *iterator, AtAOVSampleIterator *sample_iterator,                 \
                                int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y);                           \
  static void DriverClose(AtNode *node, AtOutputIterator *iterator);                                         \
  static const AtDriverNodeMethods ai_driver_mtds = {DriverSupportsPixelType, DriverExtension, DriverOpen, DriverNeedsBucket,  \
                                                     DriverPrepareBucket, DriverProcessBucket, DriverWriteBucket, DriverClose}; \
  static const AtNodeMethods ai_node_mtds = {&ai_common_mtds, &ai_driver_mtds};                              \
  const AtNodeMethods *tag = &ai_node_mtds;

#define AI_OPERATOR_NODE_EXPORT_METHODS(tag)                                                                 \
  AI_FAKE_COMMON_METHODS                                                                                     \
  static bool OperatorInit(AtNode *op, void **user_data);                                                    \
  static bool OperatorCleanup(AtNode *op, void *user_data);                                                  \
  static bool OperatorCook(AtNode *node, AtNode *op, void *child_data, void *user_data, const AtArray *matching_params,        \
                           AtCookContext *cook_context);                                                     \
  static bool OperatorPostCook(AtNode *op, void *user_data);                                                 \
  static const AtOperatorNodeMethods ai_op_mtds = {OperatorInit, OperatorCleanup, OperatorCook, OperatorPostCook};             \
  static const AtNodeMethods ai_node_mtds = {&ai_common_mtds, &ai_op_mtds};                                  \
  const AtNodeMethods *tag = &ai_node_mtds;

#define node_parameters static void Parameters(AtList *params, AtNodeEntry *nentry)
#define node_plugin_initialize static bool PluginInitialize(void **plugin_data)
#define node_plugin_cleanup static void PluginCleanup(void *plugin_data)
#define node_initialize static void Initialize(AtNode *node)
#define node_update static void Update(AtNode *node)
#define node_finish static void Finish(AtNode *node)
#define camera_create_ray static void CameraCreateRay(const AtNode *node, const AtCameraInput &input, AtCameraOutput &output, uint16_t tid)
#define camera_reverse_ray static bool CameraReverseRay(const AtNode *node, const AtVector &Po, float relative_time, AtVector2 &Ps)
#define filter_output_type static uint8_t FilterOutputType(const AtNode *node, uint8_t input_type)
#define filter_pixel static void FilterPixel(AtNode *node, AtAOVSampleIterator *iterator, void *data_out, uint8_t data_type)
#define driver_supports_pixel_type static bool DriverSupportsPixelType(const AtNode *node, uint8_t pixel_type)
#define driver_extension static const char **DriverExtension()
#define driver_open static void DriverOpen(AtNode *node, AtOutputIterator *iterator, AtBBox2 display_window, AtBBox2 data_window, int bucket_size)
#define driver_needs_bucket static bool DriverNeedsBucket(AtNode *node, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid)
#define driver_prepare_bucket static void DriverPrepareBucket(AtNode *node, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid)
#define driver_process_bucket static void DriverProcessBucket(AtNode *node, AtOutputIterator *iterator, AtAOVSampleIterator *sample_iterator, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y, uint16_t tid)
#define driver_write_bucket static void DriverWriteBucket(AtNode *node, AtOutputIterator *iterator, AtAOVSampleIterator *sample_iterator, int bucket_xo, int bucket_yo, int bucket_size_x, int bucket_size_y)
#define driver_close static void DriverClose(AtNode *node, AtOutputIterator *iterator)
#define operator_init static bool OperatorInit(AtNode *op, void **user_data)
#define operator_cleanup static bool OperatorCleanup(AtNode *op, void *user_data)
#define operator_cook static bool OperatorCook(AtNode *node, AtNode *op, void *child_data, void *user_data, const AtArray *matching_params, AtCookContext *cook_context)
#define operator_post_cook static bool OperatorPostCook(AtNode *op, void *user_data)
#define node_loader AI_EXPORT_LIB bool NodeLoader(int i, AtNodeLib *node)

// ---- parameter declaration (inside node_parameters) -------------------------------------------------------
AI_API void AiNodeParamInt(AtList *params, int varoffset, const char *pname, int pdefault);
AI_API void AiNodeParamFlt(AtList *params, int varoffset, const char *pname, float pdefault);
AI_API void AiNodeParamBool(AtList *params, int varoffset, const char *pname, bool pdefault);
AI_API void AiNodeParamStr(AtList *params, int varoffset, const char *pname, const char *pdefault);
AI_API void AiNodeParamEnum(AtList *params, int varoffset, const char *pname, int pdefault, const char **enum_type);
#define AiParameterInt(n, c) AiNodeParamInt(params, -1, n, c);
#define AiParameterFlt(n, c) AiNodeParamFlt(params, -1, n, c);
#define AiParameterBool(n, c) AiNodeParamBool(params, -1, n, c);
#define AiParameterStr(n, c) AiNodeParamStr(params, -1, n, c);
#define AiParameterEnum(n, c, e) AiNodeParamEnum(params, -1, n, c, e);
AI_API bool AiMetaDataSetBool(AtNodeEntry *nentry, const char *param, const char *name, bool value);
AI_API bool AiMetaDataSetStr(AtNodeEntry *nentry, const char *param, const char *name, const char *value);

// ---- nodes, universe, arrays --------------------------------------------------------------------------------
AI_API AtUniverse *AiNodeGetUniverse(const AtNode *node);
AI_API AtNode *AiUniverseGetOptions(const AtUniverse *universe);
AI_API AtNode *AiUniverseGetCamera(const AtUniverse *universe);
AI_API AtRenderSession *AiUniverseGetRenderSession(const AtUniverse *universe);
AI_API bool AiRenderSetHintInt(AtRenderSession *session, AtString hint, int value);
AI_API void AiRenderAbort();
AI_API AtNode *AiNode(AtUniverse *universe, AtString nentry_name, AtString name);
AI_API AtNode *AiNodeLookUpByName(const AtUniverse *universe, AtString name);
AI_API AtNodeIterator *AiUniverseGetNodeIterator(const AtUniverse *universe, unsigned int node_mask);
AI_API bool AiNodeIteratorFinished(const AtNodeIterator *iter);
AI_API AtNode *AiNodeIteratorGetNext(AtNodeIterator *iter);
AI_API void AiNodeIteratorDestroy(AtNodeIterator *iter);
AI_API const char *AiNodeGetName(const AtNode *node);
AI_API const AtNodeEntry *AiNodeGetNodeEntry(const AtNode *node);
AI_API AtString AiNodeEntryGetNameAtString(const AtNodeEntry *nentry);
AI_API const AtNodeEntry *AiNodeEntryLookUp(AtString name);
AI_API int AiNodeEntryGetCount(const AtNodeEntry *nentry);
AI_API bool AiNodeIs(const AtNode *node, AtString str);
AI_API void AiNodeSetLocalData(AtNode *node, void *data);
AI_API void *AiNodeGetLocalData(const AtNode *node);
AI_API int AiNodeGetInt(const AtNode *node, AtString param);
AI_API bool AiNodeGetBool(const AtNode *node, AtString param);
AI_API float AiNodeGetFlt(const AtNode *node, AtString param);
AI_API AtString AiNodeGetStr(const AtNode *node, AtString param);
AI_API AtArray *AiNodeGetArray(const AtNode *node, AtString param);
AI_API void AiNodeSetStr(AtNode *node, AtString param, AtString value);
AI_API bool AiNodeSetArray(AtNode *node, AtString param, AtArray *array);
AI_API bool AiNodeLink(AtNode *src, AtString input, AtNode *target);
AI_API uint32_t AiArrayGetNumElements(const AtArray *array);
AI_API uint8_t AiArrayGetNumKeys(const AtArray *array);
AI_API AtString AiArrayGetStr(const AtArray *array, uint32_t i);
AI_API void *AiArrayGetPtr(const AtArray *array, uint32_t i);
AI_API AtArray *AiArrayAllocate(uint32_t nelements, uint8_t nkeys, uint8_t type);
AI_API bool AiArraySetStr(AtArray *array, uint32_t i, AtString value);
AI_API bool AiArraySetPtr(AtArray *array, uint32_t i, void *ptr);
AI_API void AiArrayResize(AtArray *array, uint32_t nelements, uint8_t nkeys);

// ---- node-type helpers ----------------------------------------------------------------------------------------
AI_API void AiCameraInitialize(AtNode *node);
AI_API void AiCameraUpdate(AtNode *node, bool plane_distance);
AI_API float AiCameraGetShutterStart();
AI_API float AiCameraGetShutterEnd();
AI_API void AiFilterInitialize(AtNode *node, bool requires_depth, const char **required_aovs);
AI_API void AiFilterUpdate(AtNode *node, float width);
AI_API void AiDriverInitialize(AtNode *node, bool supports_multiple_outputs);
AI_API void AiWorldToCameraMatrix(const AtNode *node, float time, AtMatrix &out);
AI_API void AiCameraToWorldMatrix(const AtNode *node, float time, AtMatrix &out);

// ---- ray probes (ai_ray.h, ai_shaderglobals.h, ai_vector.h): what the occlusion probe of the backward trace needs
// (src/lentil.h:613-629).  AiV3Normalize / AiV3Dist are inline in the SDK; here they are functions of the stand-in so that
// the plugin and the test's own segment walker (fa_probe_segments) run the same machine code.
struct AtShaderGlobals;
#define AI_RAY_UNDEFINED 0x00
#define AI_RAY_CAMERA 0x01
#define AI_RAY_SHADOW 0x02
struct AtRay {
  uint8_t type;
  AtVector origin, dir;
  float mindist, maxdist;
};
inline AtVector operator-(const AtVector &a, const AtVector &b) { return AtVector(a.x - b.x, a.y - b.y, a.z - b.z); }
AI_API AtVector AiV3Normalize(const AtVector &a);
AI_API float AiV3Dist(const AtVector &a, const AtVector &b);
AI_API AtShaderGlobals *AiShaderGlobals();
AI_API void AiShaderGlobalsDestroy(AtShaderGlobals *sg);
AI_API AtRay AiMakeRay(uint8_t type, const AtVector &origin, const AtVector *dir, float maxdist, const AtShaderGlobals *sg);
AI_API bool AiTraceProbe(const AtRay &ray, const AtShaderGlobals *sg);

// ---- iterators -------------------------------------------------------------------------------------------------
AI_API bool AiAOVSampleIteratorGetNext(AtAOVSampleIterator *iter);
AI_API bool AiAOVSampleIteratorGetNextDepth(AtAOVSampleIterator *iter);
AI_API void AiAOVSampleIteratorReset(AtAOVSampleIterator *iter);
AI_API void AiAOVSampleIteratorGetPixel(AtAOVSampleIterator *iter, int &x, int &y);
AI_API AtVector2 AiAOVSampleIteratorGetOffset(AtAOVSampleIterator *iter);
AI_API float AiAOVSampleIteratorGetInvDensity(AtAOVSampleIterator *iter);
AI_API AtString AiAOVSampleIteratorGetAOVName(AtAOVSampleIterator *iter);
AI_API AtRGBA AiAOVSampleIteratorGetRGBA(AtAOVSampleIterator *iter);
AI_API AtRGB AiAOVSampleIteratorGetRGB(AtAOVSampleIterator *iter);
AI_API AtVector AiAOVSampleIteratorGetVec(AtAOVSampleIterator *iter);
AI_API float AiAOVSampleIteratorGetFlt(AtAOVSampleIterator *iter);
AI_API AtRGBA AiAOVSampleIteratorGetAOVRGBA(AtAOVSampleIterator *iter, AtString name);
AI_API AtRGB AiAOVSampleIteratorGetAOVRGB(AtAOVSampleIterator *iter, AtString name);
AI_API AtVector AiAOVSampleIteratorGetAOVVec(AtAOVSampleIterator *iter, AtString name);
AI_API float AiAOVSampleIteratorGetAOVFlt(AtAOVSampleIterator *iter, AtString name);
AI_API void AiOutputIteratorReset(AtOutputIterator *iter);
AI_API bool AiOutputIteratorGetNext(AtOutputIterator *iter, AtString *output_name, int *pixel_type, const void **bucket_data);

// ---- messages ----------------------------------------------------------------------------------------------------
AI_API void AiMsgInfo(const char *format, ...);
AI_API void AiMsgWarning(const char *format, ...);
AI_API void AiMsgError(const char *format, ...);
AI_API const char *AiParamGetTypeName(uint8_t type);
