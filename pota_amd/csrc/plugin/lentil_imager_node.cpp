// lentil_imager_node.cpp -- imager_lentil (src/lentil_imager.cpp): a driver with subtype "imager" on the full-frame
// schedule.  The first driver_process_bucket call of a frame, from whichever thread, runs the GPU pass once
// (upload -> clear -> redistribute -> resolve -> download, lentil_imager_process_bucket); every call then copies its
// bucket of every lentil-filtered AOV out of the downloaded images, in place of the reference's per-pixel resolve
// (:112-118,169-186).
#include "lentil_plugin.h"

#include <cstring>

AI_DRIVER_NODE_EXPORT_METHODS(LentilImagerMtd);

node_parameters {
  AiMetaDataSetStr(nentry, nullptr, "subtype", "imager");
  AiParameterBool("enable", true);
}

node_plugin_initialize { (void)plugin_data; return true; }
node_plugin_cleanup { (void)plugin_data; }

node_initialize { AiDriverInitialize(node, false); }

node_update {
  AtRenderSession *session = AiUniverseGetRenderSession(AiNodeGetUniverse(node));
  AiRenderSetHintInt(session, AtString("imager_padding"), 0);
  AiRenderSetHintInt(session, AtString("imager_schedule"), 0x02);      // full frame: every bucket after the whole render
}

driver_supports_pixel_type {
  (void)node;
  return pixel_type == AI_TYPE_RGBA || pixel_type == AI_TYPE_RGB || pixel_type == AI_TYPE_FLOAT || pixel_type == AI_TYPE_VECTOR;
}

driver_open { (void)node; (void)iterator; (void)display_window; (void)data_window; (void)bucket_size; }

driver_extension {
  static const char *extensions[] = {nullptr};
  return extensions;
}

driver_needs_bucket {
  (void)node; (void)bucket_xo; (void)bucket_yo; (void)bucket_size_x; (void)bucket_size_y; (void)tid;
  return true;
}

driver_prepare_bucket { (void)node; (void)bucket_xo; (void)bucket_yo; (void)bucket_size_x; (void)bucket_size_y; (void)tid; }

driver_process_bucket {
  (void)sample_iterator; (void)tid;
  AiOutputIteratorReset(iterator);
  LentilCamera *cam = (LentilCamera *)AiNodeGetLocalData(AiUniverseGetCamera(AiNodeGetUniverse(node)));
  if (!cam->redistribution || !cam->imager) {
    if (!cam->imager_print_once_only) {
      AiMsgInfo("[LENTIL IMAGER] Skipping imager");
      cam->imager_print_once_only = true;
    }
    return;
  }
  AtString aov_name;
  int aov_type = 0;
  const void *bucket_data = nullptr;
  while (AiOutputIteratorGetNext(iterator, &aov_name, &aov_type, &bucket_data)) {
    const int idx = cam->aov_index(aov_name.c_str());
    if (idx < 0) {
      // a ranked cryptomatte AOV (src/lentil_imager.cpp:121-161)?
      const int cidx = cam->crypto_index(aov_name.c_str());
      if (cidx < 0) continue;                 // not one of lentil's AOVs (lentil_time was dropped from the list)
      const int rc = lentil_imager_process_crypto_bucket(cam->imager, (uint32_t)cidx, bucket_xo, bucket_yo, bucket_size_x,
                                                         bucket_size_y, (float *)const_cast<void *>(bucket_data));
      if (rc != LENTIL_OK) {
        if (!cam->imager_print_once_only) {
          AiMsgError("%s", lentil_imager_last_error(cam->imager));
          AiRenderAbort();
        }
        cam->imager_print_once_only = true;
        return;
      }
      continue;
    }
    // the filter turned every type into RGBA (filter_output_type), so bucket_data is AtRGBA[sx * sy] (:100,160,178)
    const int rc = lentil_imager_process_bucket(cam->imager, (uint32_t)idx, bucket_xo, bucket_yo, bucket_size_x, bucket_size_y,
                                                (float *)const_cast<void *>(bucket_data));
    if (rc != LENTIL_OK) {
      if (!cam->imager_print_once_only) {
        AiMsgError("%s", lentil_imager_last_error(cam->imager));
        AiRenderAbort();
      }
      cam->imager_print_once_only = true;
      return;
    }
  }
  cam->imager_print_once_only = true;
}

driver_write_bucket { (void)node; (void)iterator; (void)sample_iterator; (void)bucket_xo; (void)bucket_yo; (void)bucket_size_x; (void)bucket_size_y; }
driver_close { (void)node; (void)iterator; }
node_finish { (void)node; }

void registerLentilImager(AtNodeLib *node) {
  node->methods = (const void *)LentilImagerMtd;
  node->output_type = AI_TYPE_NONE;
  node->name = "imager_lentil";
  node->node_type = AI_NODE_DRIVER;
  strncpy(node->version, AI_VERSION, AI_MAXSIZE_VERSION - 1);
}
