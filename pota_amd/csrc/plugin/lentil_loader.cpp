// lentil_loader.cpp -- the DSO entry point: four nodes, in the reference's order and under its names
// (src/lentil_loader.cpp:11-28): lentil_camera, lentil_filter, imager_lentil, lentil_operator.
#include <ai.h>

#include <cstring>

void registerLentilCamera(AtNodeLib *node);
void registerLentilFilter(AtNodeLib *node);
void registerLentilImager(AtNodeLib *node);
void registerLentilOperator(AtNodeLib *node);

node_loader {
  // (The pass keeps four HIP streams busy and uses a fifth where the runtime has one: GPU_MAX_HW_QUEUES=8 in the renderer's
  // launch environment, INTEGRATION.md.  Not set from here: setenv in a plugin races with getenv on the host's other threads.)
  typedef void (*Register)(AtNodeLib *);
  static const Register nodes[] = {registerLentilCamera, registerLentilFilter, registerLentilImager, registerLentilOperator};
  if (i < 0 || i >= (int)(sizeof nodes / sizeof nodes[0])) return false;
  strncpy(node->version, AI_VERSION, AI_MAXSIZE_VERSION - 1);
  nodes[i](node);
  return true;
}
