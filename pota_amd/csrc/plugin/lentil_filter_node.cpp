// lentil_filter_node.cpp -- the lentil_filter node (src/lentil_filter.cpp).  filter_pixel is called by Arnold for
// every pixel and output from many threads; for the RGBA output it used to trace every redistributed sample through
// the lens on the spot (:237-448).  Here that branch only CAPTURES what it gathers per AOV sample (:105-165,
// 206-234) into the calling thread's staging columns; the imager's first bucket hands them to the GPU.  The
// pass-through filtering for the display (:453-479) is unchanged in behaviour.
#include "lentil_plugin.h"

#include <cmath>
#include <cstring>
#include <vector>

AI_FILTER_NODE_EXPORT_METHODS(LentilFilterDataMtd);

node_parameters {
  (void)params;
  AiMetaDataSetBool(nentry, nullptr, "force_update", true);
}

node_plugin_initialize { (void)plugin_data; return true; }
node_plugin_cleanup { (void)plugin_data; }

node_initialize {
  AiFilterInitialize(node, false, const_cast<const char **>(lentil_filter_required_aovs()));
}

node_update {
  const AtNodeEntry *oidn = AiNodeEntryLookUp(AtString("imager_denoiser_oidn"));
  AiFilterUpdate(node, lentil_filter_width(AiNodeEntryGetCount(oidn) != 0));
}

filter_output_type {
  (void)node;
  return (uint8_t)lentil_filter_output_type(input_type);
}

filter_pixel {
  AtUniverse *universe = AiNodeGetUniverse(node);
  LentilCamera *cam = (LentilCamera *)AiNodeGetLocalData(AiUniverseGetCamera(universe));
  const LentilStrings &S = lentil_strings();
  AtNode *options = AiUniverseGetOptions(universe);
  const int aa_samples_set_by_user = AiNodeGetInt(options, AtString("AA_samples"));
  const bool rgba_aov = AiAOVSampleIteratorGetAOVName(iterator) == S.rgba;     // early out for non-primary AOV samples
  const bool adaptive = AiNodeGetBool(options, AtString("enable_adaptive_sampling"));
  float inverse_sample_density = 0.0f;

  // the visit prologue (:79-88): the footprint's sample count gives the AA level; below the final level (or AA < 3)
  // redistribution is off for the rest of the frame
  if (!adaptive && rgba_aov) {
    int count = 0;
    while (AiAOVSampleIteratorGetNext(iterator)) ++count;
    AiAOVSampleIteratorReset(iterator);
    int disable = 0;
    inverse_sample_density = lentil_filter_inverse_sample_density(count, cam->P.filter_width, aa_samples_set_by_user, &disable);
    if (disable) cam->redistribution = false;
  }

  if (cam->redistribution && rgba_aov && cam->stage) {
    int px, py;
    AiAOVSampleIteratorGetPixel(iterator, px, py);
    px -= cam->P.region_min_x;
    py -= cam->P.region_min_y;
    const int slot = cam->thread_slot();
    const size_t n_extra = cam->aovs.size() - 1;
    float extra[4 * LENTIL_MAX_AOVS];
    const size_t n_crypto = cam->crypto_aovs.size();
    const int entries = cam->crypto_entries;
    std::vector<float> crypto_ids(n_crypto * (size_t)entries), crypto_w(n_crypto * (size_t)entries);
    for (int sampleid = 0; AiAOVSampleIteratorGetNext(iterator); ++sampleid) {
      lentil_sample_capture c;
      memset(&c, 0, sizeof c);
      c.px = px; c.py = py;
      c.inverse_sample_density = adaptive ? AiAOVSampleIteratorGetInvDensity(iterator) : inverse_sample_density;
      const AtRGBA rgba = AiAOVSampleIteratorGetRGBA(iterator);
      c.rgba[0] = rgba.r; c.rgba[1] = rgba.g; c.rgba[2] = rgba.b; c.rgba[3] = rgba.a;
      const AtVector P = AiAOVSampleIteratorGetAOVVec(iterator, S.p);
      c.P[0] = P.x; c.P[1] = P.y; c.P[2] = P.z;
      c.Z = AiAOVSampleIteratorGetAOVFlt(iterator, S.z);
      const AtVector rd = AiAOVSampleIteratorGetAOVVec(iterator, S.raydir);
      c.raydir[0] = rd.x; c.raydir[1] = rd.y; c.raydir[2] = rd.z;
      c.time = AiAOVSampleIteratorGetAOVFlt(iterator, S.time);
      const AtRGB vol = AiAOVSampleIteratorGetAOVRGB(iterator, S.volume);
      c.volume[0] = vol.r; c.volume[1] = vol.g; c.volume[2] = vol.b;
      c.bidir_ignore = AiAOVSampleIteratorGetAOVFlt(iterator, S.ignore);
      const AtRGBA tr = AiAOVSampleIteratorGetAOVRGBA(iterator, S.transmission);
      c.transmission[0] = tr.r; c.transmission[1] = tr.g; c.transmission[2] = tr.b; c.transmission[3] = tr.a;
      // the other lentil-filtered AOVs, widened to RGBA (:214-232); lentil_debug takes no column (its value is the
      // visit's draw count, recomputed on the device)
      for (size_t k = 0; k < n_extra; ++k) {
        const lentil_aov_plan &a = cam->aovs[k + 1];
        float *e = extra + 4 * k;
        e[0] = e[1] = e[2] = 0.f; e[3] = 1.f;
        if (strcmp(a.name, "lentil_debug") == 0) { e[3] = 0.f; continue; }
        const AtString name(a.name);
        switch (a.type) {
          case AI_TYPE_RGBA: { const AtRGBA v = AiAOVSampleIteratorGetAOVRGBA(iterator, name); e[0] = v.r; e[1] = v.g; e[2] = v.b; e[3] = v.a; } break;
          case AI_TYPE_RGB: { const AtRGB v = AiAOVSampleIteratorGetAOVRGB(iterator, name); e[0] = v.r; e[1] = v.g; e[2] = v.b; } break;
          case AI_TYPE_FLOAT: { const float v = AiAOVSampleIteratorGetAOVFlt(iterator, name); e[0] = e[1] = e[2] = v; } break;
          case AI_TYPE_VECTOR: { const AtVector v = AiAOVSampleIteratorGetAOVVec(iterator, name); e[0] = v.x; e[1] = v.y; e[2] = v.z; } break;
          default: break;
        }
      }
      c.extra_rgba = n_extra ? extra : nullptr;
      // cryptomatte_construct_cache (src/lentil.h:781-811): per cryptomatte AOV the sample's depth entries -- opacity
      // and the AOV's id -- folded into id -> weight; the iterator has left the sample after its last depth entry
      // and is put back (reset_iterator_to_id, :1178-1186)
      bool crypto_ok = true;
      for (size_t k = 0; k < n_crypto; ++k) {
        float opacity[3 * 64], value[64];
        int nd = 0;
        const AtString name(cam->crypto_aovs[k].name);
        while (AiAOVSampleIteratorGetNextDepth(iterator)) {
          if (nd < 64) {
            const AtRGB o = AiAOVSampleIteratorGetAOVRGB(iterator, S.opacity);
            opacity[3 * nd] = o.r; opacity[3 * nd + 1] = o.g; opacity[3 * nd + 2] = o.b;
            value[nd] = AiAOVSampleIteratorGetAOVFlt(iterator, name);
            ++nd;
          } else crypto_ok = false;
        }
        AiAOVSampleIteratorReset(iterator);
        for (int i = 0; AiAOVSampleIteratorGetNext(iterator); ++i) if (i == sampleid) break;
        if (lentil_crypto_construct_cache(nd, opacity, value, crypto_ids.data() + k * (size_t)entries,
                                          crypto_w.data() + k * (size_t)entries, entries) < 0) crypto_ok = false;
      }
      if (!crypto_ok) {
        AiMsgError("[LENTIL] an AOV sample holds more cryptomatte ids than LENTIL_CRYPTO_ENTRIES (%d)", entries);
        cam->redistribution = false;
        break;
      }
      c.crypto_ids = n_crypto ? crypto_ids.data() : nullptr;
      c.crypto_weights = n_crypto ? crypto_w.data() : nullptr;
      if (slot >= cam->stage_slots || lentil_stage_append(cam->stage, slot, &c) != LENTIL_OK) {
        AiMsgError("[LENTIL] could not stage an AOV sample (thread slot %d)", slot);
        cam->redistribution = false;
        break;
      }
    }
  }

  // regular filtering (pass-through) for display purposes (:453-479)
  AiAOVSampleIteratorReset(iterator);
  float offs[2 * 256], vals[4 * 256], dens[256], depth[256];
  int n = 0;
  while (n < 256 && AiAOVSampleIteratorGetNext(iterator)) {
    const AtVector2 o = AiAOVSampleIteratorGetOffset(iterator);
    offs[2 * n] = o.x; offs[2 * n + 1] = o.y;
    dens[n] = adaptive ? AiAOVSampleIteratorGetInvDensity(iterator) : inverse_sample_density;
    depth[n] = AiAOVSampleIteratorGetAOVFlt(iterator, S.z);
    float *v = vals + 4 * n;
    v[0] = v[1] = v[2] = 0.f; v[3] = 1.f;
    switch (data_type) {
      case AI_TYPE_RGBA: { const AtRGBA s = AiAOVSampleIteratorGetRGBA(iterator); v[0] = s.r; v[1] = s.g; v[2] = s.b; v[3] = s.a; } break;
      case AI_TYPE_RGB: { const AtRGB s = AiAOVSampleIteratorGetRGB(iterator); v[0] = s.r; v[1] = s.g; v[2] = s.b; } break;
      case AI_TYPE_VECTOR: { const AtVector s = AiAOVSampleIteratorGetVec(iterator); v[0] = s.x; v[1] = s.y; v[2] = s.z; } break;
      case AI_TYPE_FLOAT: { const float s = AiAOVSampleIteratorGetFlt(iterator); v[0] = v[1] = v[2] = s; } break;
      default: break;
    }
    ++n;
  }
  float out[4] = {0, 0, 0, 0};
  switch (data_type) {
    case AI_TYPE_RGBA:
      lentil_filter_gaussian_complete(n, offs, vals, dens, 0.f, cam->P.filter_width, AiFastExp, out);
      *((AtRGBA *)data_out) = AtRGBA(out[0], out[1], out[2], out[3]);
      break;
    case AI_TYPE_RGB:
      lentil_filter_gaussian_complete(n, offs, vals, dens, 0.f, cam->P.filter_width, AiFastExp, out);
      *((AtRGB *)data_out) = AtRGB(out[0], out[1], out[2]);
      break;
    case AI_TYPE_VECTOR:
      lentil_filter_closest_complete(n, depth, vals, out);
      *((AtVector *)data_out) = AtVector(out[0], out[1], out[2]);
      break;
    case AI_TYPE_FLOAT:
      lentil_filter_closest_complete(n, depth, vals, out);
      *((float *)data_out) = out[0];
      break;
    default: break;
  }
}

node_finish { (void)node; }

void registerLentilFilter(AtNodeLib *node) {
  node->methods = (const void *)LentilFilterDataMtd;
  node->output_type = AI_TYPE_NONE;
  node->name = "lentil_filter";
  node->node_type = AI_NODE_FILTER;
  strncpy(node->version, AI_VERSION, AI_MAXSIZE_VERSION - 1);
}
