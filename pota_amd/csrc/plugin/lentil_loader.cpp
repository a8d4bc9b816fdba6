// lentil_loader.cpp -- the DSO entry point: four nodes, in the reference's order and under its names
// (src/lentil_loader.cpp:11-28): lentil_camera, lentil_filter, imager_lentil, lentil_operator.
#include <ai.h>

#include <cstdlib>
#include <cstring>

void registerLentilCamera(AtNodeLib *node);
void registerLentilFilter(AtNodeLib *node);
void registerLentilImager(AtNodeLib *node);
void registerLentilOperator(AtNodeLib *node);

node_loader {
  // The pass keeps four HIP streams busy at once and uses a fifth where it can (cryptomatte replay beside the draws); the
  // ROCm runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 unless told otherwise, and reads
  // the variable when it initialises -- which a renderer that loads this plugin has normally not done yet.  Never overrides
  // what the user has set.
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  typedef void (*Register)(AtNodeLib *);
  static const Register nodes[] = {registerLentilCamera, registerLentilFilter, registerLentilImager, registerLentilOperator};
  if (i < 0 || i >= (int)(sizeof nodes / sizeof nodes[0])) return false;
  strncpy(node->version, AI_VERSION, AI_MAXSIZE_VERSION - 1);
  nodes[i](node);
  return true;
}
