// lentil_loader.cpp -- the DSO entry point: four nodes, in the reference's order and under its names
// (src/lentil_loader.cpp:11-28): lentil_camera, lentil_filter, imager_lentil, lentil_operator.
#include <ai.h>

#include <cstring>

void registerLentilCamera(AtNodeLib *node);
void registerLentilFilter(AtNodeLib *node);
void registerLentilImager(AtNodeLib *node);
void registerLentilOperator(AtNodeLib *node);

node_loader {
  // (Nothing is asked of the renderer's environment: the pass keeps four HIP streams busy, the runtime's default of four hardware
  // queues holds them, and lentil_hip_create probes that they do run side by side -- a context whose streams do not takes the
  // chunked form of the pass.  A process with other busy GPU streams may want GPU_MAX_HW_QUEUES=8: INTEGRATION.md section 3a.)
  typedef void (*Register)(AtNodeLib *);
  static const Register nodes[] = {registerLentilCamera, registerLentilFilter, registerLentilImager, registerLentilOperator};
  if (i < 0 || i >= (int)(sizeof nodes / sizeof nodes[0])) return false;
  strncpy(node->version, AI_VERSION, AI_MAXSIZE_VERSION - 1);
  nodes[i](node);
  return true;
}
