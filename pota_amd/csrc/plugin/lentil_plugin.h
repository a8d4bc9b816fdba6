// lentil_plugin.h -- state shared by the four Arnold nodes of lentil.so (this build's plugin DSO).
//
// The reference's `struct Camera` (src/lentil.h:92-196) owns everything: parameters, lens, AOV list, buffers.  Here
// the same object -- LentilCamera, local data of the lentil_camera node, borrowed by the filter and the imager through
// AiNodeGetLocalData(AiUniverseGetCamera(universe)) like the reference does (src/lentil_filter.cpp:68-70,
// src/lentil_imager.cpp:70-72) -- holds the parameters and three handles: the host lens (liblentil_host.so), the
// visit stage and the imager (liblentil_bridge.so) and, through them, the GPU context (liblentil_hip.so).  The node
// callbacks only fetch values from Arnold and pass them on.
#pragma once
#include <ai.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "lentil_bridge.h"
#include "lentil_host.h"

// what lentil_operator leaves for the camera (the reference's OperatorData, src/operator_data.h)
struct LentilOperatorData {
  std::vector<lentil_aov_plan> aovs;
  bool cooked = false;
};

struct LentilCamera {
  std::mutex setup_mutex;                 // the reference's l_critsec around setup_camera (src/lentil.h:212,235)
  lentil_params P{};
  float exposure = 1.0f;
  double input_fstop = 0.0, lambda_um = 0.55, extra_sensor_shift = 0.0, tan_fov = 0.0;
  AtNode *camera_node = nullptr, *options_node = nullptr;

  lentil_host_lens *host_lens = nullptr;
  const lentil_lens_table *lens_table = nullptr;
  std::vector<float> bokeh_cdf_row, bokeh_cdf_col;
  std::vector<int32_t> bokeh_row_idx, bokeh_col_idx;
  lentil_bokeh_table bokeh{};
  bool have_bokeh = false;

  std::vector<lentil_aov_plan> aovs;      // sanitised: the AOVs lentil filters, RGBA first
  // the ranked cryptomatte AOVs (AOVData::is_crypto, src/lentil.h:1037-1040) in their own list: they take no column of
  // the frame but a per-pixel id table each; crypto_entries = ids one AOV sample's depth entries may hold
  std::vector<lentil_aov_plan> crypto_aovs;
  bool cryptomatte_lentil = false;
  int crypto_entries = 8;
  std::atomic<bool> redistribution{false};
  bool imager_print_once_only = false;

  lentil_hip_ctx *gpu = nullptr;
  lentil_stage *stage = nullptr;
  lentil_imager *imager = nullptr;
  int stage_slots = 0;
  std::vector<float> motion_keys;           // world-to-camera sampled over the shutter (16 floats each; empty: static)
  float shutter_start = 0.0f, shutter_end = 0.0f;     // the camera's shutter: where the AOV samples' lentil_time values lie
  std::atomic<int> next_slot{0};
  std::atomic<uint64_t> slot_generation{0};   // advanced by every setup(): render threads that outlive a camera update take a new slot

  ~LentilCamera();
  void release_gpu();
  // Camera::setup_camera (src/lentil.h:211-280): parameters, model-specific setup, AOV list, filter set-up, GPU
  void setup(AtUniverse *universe);
  int thread_slot();                      // one staging slot per render thread
  int aov_index(const char *name) const;
  int crypto_index(const char *name) const;
};

// AtString constants, interned once
struct LentilStrings {
  AtString rgba{"RGBA"}, p{"P"}, z{"Z"}, time{"lentil_time"}, raydir{"lentil_raydir"}, debug{"lentil_debug"},
      volume{"volume"}, transmission{"transmission"}, ignore{"lentil_ignore"}, outputs{"outputs"}, opacity{"opacity"};
};
const LentilStrings &lentil_strings();
