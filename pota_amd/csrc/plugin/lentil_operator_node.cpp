// lentil_operator_node.cpp -- lentil_operator (src/lentil_operator.cpp): at cook time, routes every output lentil can
// filter through one lentil_filter node ("lentil_replaced_filter"), remembers what filter each had, adds the three
// bookkeeping AOVs (lentil_debug, lentil_time, lentil_raydir) and the two aov_write shaders that feed them.  The AOV
// planning itself is lentil_operator_cook of liblentil_bridge.so; what stays here are the calls into Arnold.
#include "lentil_plugin.h"

#include <cstring>
#include <string>

AI_OPERATOR_NODE_EXPORT_METHODS(LentilOperatorMtd);

node_parameters {
  (void)params;
  AiMetaDataSetBool(nentry, nullptr, "force_update", true);
}

node_plugin_initialize { (void)plugin_data; return true; }
node_plugin_cleanup { (void)plugin_data; }
node_initialize { (void)node; }
node_update { (void)node; }
node_finish { (void)node; }

operator_init {
  (void)user_data;
  AiNodeSetLocalData(op, new LentilOperatorData());
  return true;
}

operator_cook {
  (void)node; (void)child_data; (void)user_data; (void)matching_params; (void)cook_context;
  LentilOperatorData *data = (LentilOperatorData *)AiNodeGetLocalData(op);
  AtUniverse *universe = AiNodeGetUniverse(op);
  AtNode *camera_node = AiUniverseGetCamera(universe);
  if (!camera_node || AiNodeEntryGetNameAtString(AiNodeGetNodeEntry(camera_node)) != AtString("lentil_camera")) return false;

  AtNode *filter_node = AiNodeLookUpByName(universe, AtString("lentil_replaced_filter"));
  if (!filter_node) filter_node = AiNode(universe, AtString("lentil_filter"), AtString("lentil_replaced_filter"));

  AtNode *options = AiUniverseGetOptions(universe);
  AtArray *outputs = AiNodeGetArray(options, AtString("outputs"));
  const uint32_t n = AiArrayGetNumElements(outputs);
  if (n == 0) return false;
  std::vector<std::string> strings(n), entries(n);
  std::vector<const char *> out_ptrs(n), entry_ptrs(n);
  for (uint32_t i = 0; i < n; ++i) {
    strings[i] = AiArrayGetStr(outputs, i).c_str();
    lentil_output_tokens tok;
    lentil_tokenize_output(strings[i].c_str(), &tok);
    AtNode *orig = AiNodeLookUpByName(universe, AtString(tok.filter));
    entries[i] = orig ? AiNodeEntryGetNameAtString(AiNodeGetNodeEntry(orig)).c_str() : "";
    out_ptrs[i] = strings[i].c_str();
    entry_ptrs[i] = entries[i].c_str();
  }
  std::vector<lentil_aov_plan> plans(n + 8);
  char warnings[4096];
  const int m = lentil_operator_cook(out_ptrs.data(), entry_ptrs.data(), (int)n, plans.data(), (int)plans.size(), warnings, sizeof warnings);
  if (warnings[0]) AiMsgWarning("%s", warnings);
  if (m <= 0) return false;
  plans.resize((size_t)m);
  data->aovs = plans;

  // options.outputs: the rewired strings of every output plus the three added AOVs (rebuild_arnold_outputs_from_list,
  // src/aov_data.h:164-189)
  AtArray *rebuilt = AiArrayAllocate((uint32_t)m, 1, AI_TYPE_STRING);
  for (int i = 0; i < m; ++i) {
    char buf[1024];
    if (lentil_rebuild_output(&plans[(size_t)i].to, buf, sizeof buf) < 0) return false;
    AiArraySetStr(rebuilt, (uint32_t)i, AtString(buf));
  }
  AiNodeSetArray(options, AtString("outputs"), rebuilt);

  // the two shaders that write the bookkeeping AOVs, appended to options.aov_shaders (:133-165)
  AtArray *aov_shaders = AiNodeGetArray(options, AtString("aov_shaders"));
  uint32_t n_shaders = AiArrayGetNumElements(aov_shaders);
  auto add_writer = [&](const char *write_entry, const char *write_name, const char *read_entry, const char *read_name,
                        const char *variable, const char *aov) {
    if (AiNodeLookUpByName(universe, AtString(write_name))) return;        // cooked before
    AtNode *w = AiNode(universe, AtString(write_entry), AtString(write_name));
    AtNode *r = AiNode(universe, AtString(read_entry), AtString(read_name));
    AiNodeSetStr(r, AtString("variable"), AtString(variable));
    AiNodeSetStr(w, AtString("aov_name"), AtString(aov));
    AiNodeLink(r, AtString("aov_input"), w);
    n_shaders += 1;
    AiArrayResize(aov_shaders, n_shaders, 1);
    AiArraySetPtr(aov_shaders, n_shaders - 1, (void *)w);
    AiNodeSetArray(options, AtString("aov_shaders"), aov_shaders);
  };
  add_writer("aov_write_float", "lentil_time_write", "state_float", "lentil_time_read", "time", "lentil_time");
  add_writer("aov_write_rgb", "lentil_raydir_write", "state_vector", "lentil_raydir_read", "Rd", "lentil_raydir");
  data->cooked = true;
  return true;
}

operator_post_cook { (void)op; (void)user_data; return true; }

operator_cleanup {
  (void)user_data;
  delete (LentilOperatorData *)AiNodeGetLocalData(op);
  AiNodeSetLocalData(op, nullptr);
  return true;
}

void registerLentilOperator(AtNodeLib *node) {
  node->methods = (const void *)LentilOperatorMtd;
  node->output_type = AI_TYPE_NONE;
  node->name = "lentil_operator";
  node->node_type = AI_NODE_OPERATOR;
  strncpy(node->version, AI_VERSION, AI_MAXSIZE_VERSION - 1);
}
