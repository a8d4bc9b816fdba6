// lentil_crypto.h -- cryptomatte AOVs through the bidirectional pass (lentil_hip_alloc_crypto / _upload_crypto /
// _bind_crypto / _download_crypto / _download_crypto_table; include/lentil_hip.h).  Included by lentil_hip.hip.
//
// The reference keeps, per cryptomatte AOV and pixel, a std::map<float, float> id -> weight and a total weight
// (AOVData::crypto_hash_map / crypto_total_weight, src/aov_data.h:127-128).  Every add of a visit to a pixel -- its
// own pixel when it is not redistributed (src/lentil.h:952), the pixel of each accepted draw otherwise
// (src/lentil_filter.cpp:296,443) -- adds `sample_weight` to the total and `cache weight * sample_weight` to the map
// entry of every id in the visit's cache (add_to_buffer_cryptomatte, src/lentil.h:814-819); the cache is the visit's
// depth samples folded by opacity (cryptomatte_construct_cache, :781-811: host work on the renderer's iterator,
// lentil_crypto_construct_cache in liblentil_bridge.so).  The imager sorts a pixel's map by weight and writes the
// pairs at positions rank, rank + 1 (src/lentil_imager.cpp:121-161).
//
// Here: the maps are open-addressed tables of `slots` (id bits, fp32 weight) pairs per pixel in HBM -- a pixel's
// table is one or two 64-byte lines -- filled with atomicCAS on the id and atomicAdd on the weight.  The adds do not
// ride in the draw kernels (whose register budgets are what the pass's speed hangs on): they are replayed after the
// pass from what it leaves behind -- which visits it redistributed (the scan's work lists, as one bit per visit:
// flag_bits_kernel; ragged streams recompute the decision with visit_redistributes, the function the scan uses), for
// the visits that stay in their pixel, and the pass's draw log (visit, pixel per accepted draw; the log the parity
// tests compare with the oracle's in every mode of the pass) for the others.  The weight of
// a draw is the visit's inverse density times 1 / draw count, recomputed like load_work_visit does.
#pragma once

constexpr uint32_t kCryptoEmpty = 0xFFFFFFFFu;     // id bits of a free slot (a NaN pattern; cryptomatte ids never are)
constexpr uint32_t kCryptoMaxSlots = 64;

struct CryptoDev {
  uint32_t n_crypto, entries, slots;
  uint64_t np;
  const float *hash[LENTIL_MAX_CRYPTO];     // per visit: `entries` ids ...
  const float *weight[LENTIL_MAX_CRYPTO];   // ... and weights (bits 0xFFFFFFFF: pair unused)
  uint32_t *keys;                           // [n_crypto][np][slots]
  float *wts;                               // [n_crypto][np][slots]
  float *total;                             // [n_crypto][np]
  unsigned long long *overflow;             // adds that found their pixel's table full
};

struct LentilCrypto {
  CryptoDev D{};
  bool have_columns = false;
  bool from_upload = false;                 // the columns belong to the context's piecewise upload (lentil_hip_visits_*_crypto)
  uint64_t n_visits = 0;
  uint64_t visits_gen = 0;                  // lentil_hip_ctx::visits_gen when the columns were handed over
  std::vector<void *> owned;                // device columns of lentil_hip_upload_crypto
  bool tables_clear = false;                // nothing has been added since lentil_hip_clear_frame
  bool clear_pending = false;               // ... and the tables have not been wiped yet either: crypto_flush_clear, or a pass whose
                                            // own-pixel kernel writes every line whole (crypto_direct_tile_kernel<2>)
  uint32_t *d_flag_bits = nullptr;          // one bit per visit of the stream: redistributed by the last pass (flag_bits_kernel)
  uint64_t flag_words = 0;
  float *d_rank = nullptr;                  // download staging: np RGBA + np flags
  uint8_t *d_has = nullptr;
  // (whether the draw log is this module's own, and what the last pass needed, live in the context --
  // lentil_hip_ctx::crypto_auto_log / _hint --: alloc_frame drops this object on every camera update, the log stays)
};

// std::map<float, float> compares ids as floats: +0 and -0 are one key.  The table compares bits.
LD_DEV uint32_t crypto_key_bits(float id) {
  const uint32_t b = __float_as_uint(id);
  return b == 0x80000000u ? 0u : b;
}

// where the probing for `key` starts in a table of `slots` entries (a power of two takes the mask: a 32-bit modulo by a
// run-time divisor is ~40 instructions, and the own-pixel kernel does one per map entry of every visit)
LD_DEV uint32_t crypto_first_slot(uint32_t key, uint32_t slots) {
  const uint32_t h = key * 2654435761u >> 16;
  return (slots & (slots - 1u)) == 0u ? (h & (slots - 1u)) : h % slots;
}

// map[key] += val for pixel pix of cryptomatte AOV c (key: crypto_key_bits), with atomics
LD_DEV void crypto_table_add(const CryptoDev &C, uint32_t c, uint64_t pix, uint32_t key, float val) {
  uint32_t *K = C.keys + ((uint64_t)c * C.np + pix) * C.slots;
  float *Wt = C.wts + ((uint64_t)c * C.np + pix) * C.slots;
  uint32_t s = crypto_first_slot(key, C.slots);
  bool placed = false;
  for (uint32_t i = 0; i < C.slots && !placed; ++i) {
    uint32_t cur = __hip_atomic_load(K + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kCryptoEmpty) cur = atomicCAS(K + s, kCryptoEmpty, key);
    if (cur == kCryptoEmpty || cur == key) { atomicAdd(Wt + s, val); placed = true; }
    else s = s + 1u == C.slots ? 0u : s + 1u;
  }
  if (!placed) atomicAdd(C.overflow, 1ull);
}

// add_to_buffer_cryptomatte, src/lentil.h:814-819, for visit v and every cryptomatte AOV
LD_DEV void crypto_add_visit(const CryptoDev &C, uint64_t pix, uint64_t v, float sample_weight) {
  for (uint32_t c = 0; c < C.n_crypto; ++c) {
    atomicAdd(C.total + (uint64_t)c * C.np + pix, sample_weight);                       // :815
    const float *h = C.hash[c] + v * C.entries, *w = C.weight[c] + v * C.entries;
    for (uint32_t e = 0; e < C.entries; ++e) {
      const float cw = w[e];
      if (__float_as_uint(cw) == kCryptoEmpty) continue;
      crypto_table_add(C, c, pix, crypto_key_bits(h[e]), cw * sample_weight);           // :817
    }
  }
}

// ---- between GPUs (lentil_hip_exchange_bands): what this rank's draws added to the maps of pixels in another rank's
// band travels as a list of 16-byte records -- (pixel, AOV, id bits, weight) per map entry, (pixel, AOV | kCryptoTotal, 0,
// total weight) per pixel and AOV -- and is added to the owner's tables there.
constexpr uint32_t kCryptoTotal = 0x80000000u;
// records of rows [p_begin, p_begin + n_pix): counted (out == nullptr) or written
__global__ __launch_bounds__(256) void crypto_band_records_kernel(CryptoDev C, uint64_t p_begin, uint64_t n_pix, uint4 *out,
                                                                  unsigned long long cap, unsigned long long *count) {
  const uint64_t cells = n_pix * C.n_crypto;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
    const uint32_t c = (uint32_t)(i / n_pix);
    const uint64_t pix = p_begin + (i - (uint64_t)c * n_pix);
    const uint32_t *K = C.keys + ((uint64_t)c * C.np + pix) * C.slots;
    const float *Wt = C.wts + ((uint64_t)c * C.np + pix) * C.slots;
    const float total = C.total[(uint64_t)c * C.np + pix];
    uint32_t n = total != 0.0f ? 1u : 0u;
    for (uint32_t s = 0; s < C.slots; ++s) n += K[s] != kCryptoEmpty ? 1u : 0u;
    if (!n) continue;
    unsigned long long at = atomicAdd(count, (unsigned long long)n);
    if (!out) continue;
    if (total != 0.0f) { if (at < cap) out[at] = make_uint4((uint32_t)pix, c | kCryptoTotal, 0u, __float_as_uint(total)); ++at; }
    for (uint32_t s = 0; s < C.slots; ++s)
      if (K[s] != kCryptoEmpty) { if (at < cap) out[at] = make_uint4((uint32_t)pix, c, K[s], __float_as_uint(Wt[s])); ++at; }
  }
}
__global__ __launch_bounds__(256) void crypto_merge_records_kernel(CryptoDev C, const uint4 *rec, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint4 r = rec[i];
    const uint32_t c = r.y & ~kCryptoTotal;
    if (c >= C.n_crypto || r.x >= C.np) continue;
    if (r.y & kCryptoTotal) atomicAdd(C.total + (uint64_t)c * C.np + r.x, __uint_as_float(r.w));
    else crypto_table_add(C, c, r.x, r.z, __uint_as_float(r.w));
  }
}

// visits that stay in their own pixel: filter_and_add_to_buffer_new, src/lentil.h:938-955 (:952), called at
// src/lentil_filter.cpp:243-246 / :306-309 -- sample weight = inverse sample density
__global__ __launch_bounds__(256) void crypto_direct_kernel(CryptoDev C, VisitsDev V, lentil_params P, double lens_length) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V.n; v += stride) {
    const float invd = V.inv_density ? V.inv_density[v] : P.inverse_sample_density;
    if (visit_redistributes(P, lens_length, V.pos_z[v], V.volume_ignore[v], V.transmission[v], invd,
                            [&]() { return V.raydir_time[v]; }, V.cam))
      continue;
    int px, py;
    visit_pixel(V, v, px, py);
    crypto_add_visit(C, (uint64_t)P.xres * (uint32_t)py + (uint32_t)px, v, invd);
  }
}

// The same for pixel-major streams (visits_per_pixel > 0): one lane owns a pixel and walks its visits in stream order,
// which is the order the reference adds them in.  Nothing else touches the tables while this kernel runs (the draws
// are replayed behind it), so the owner reads and writes its pixel's table without atomics, and a pixel no draw lands
// on ends up with the reference's sums bit for bit.
__global__ __launch_bounds__(256) void crypto_direct_owner_kernel(CryptoDev C, VisitsDev V, lentil_params P, double lens_length) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t n_pix = V.n / V.visits_per_pixel;
  for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pix; q += stride) {
    int px, py;
    visit_pixel(V, q * V.visits_per_pixel, px, py);
    const uint64_t pix = (uint64_t)P.xres * (uint32_t)py + (uint32_t)px;
    for (uint32_t c = 0; c < C.n_crypto; ++c) {
      uint32_t *K = C.keys + ((uint64_t)c * C.np + pix) * C.slots;
      float *Wt = C.wts + ((uint64_t)c * C.np + pix) * C.slots;
      float total = C.total[(uint64_t)c * C.np + pix];
      for (uint32_t m = 0; m < V.visits_per_pixel; ++m) {
        const uint64_t v = q * V.visits_per_pixel + m;
        const float invd = V.inv_density ? V.inv_density[v] : P.inverse_sample_density;
        if (visit_redistributes(P, lens_length, V.pos_z[v], V.volume_ignore[v], V.transmission[v], invd,
                                [&]() { return V.raydir_time[v]; }, V.cam))
          continue;
        total += invd;                                                                    // :815
        const float *h = C.hash[c] + v * C.entries, *w = C.weight[c] + v * C.entries;
        for (uint32_t e = 0; e < C.entries; ++e) {
          const float cw = w[e];
          if (__float_as_uint(cw) == kCryptoEmpty) continue;
          const uint32_t key = crypto_key_bits(h[e]);
          uint32_t s = crypto_first_slot(key, C.slots);
          bool placed = false;
          for (uint32_t i = 0; i < C.slots && !placed; ++i) {
            const uint32_t cur = K[s];
            if (cur == kCryptoEmpty) K[s] = key;
            if (cur == kCryptoEmpty || cur == key) { Wt[s] += cw * invd; placed = true; }     // :817
            else s = s + 1u == C.slots ? 0u : s + 1u;
          }
          if (!placed) atomicAdd(C.overflow, 1ull);
        }
      }
      C.total[(uint64_t)c * C.np + pix] = total;
    }
  }
}

// The tiled form of the owner kernel, the one pixel-major streams normally get: a block of 128 lanes takes 128
// consecutive pixels of the stream.  All lanes first read the tile's visits the way they lie in memory (lane = visit:
// the three columns the decision needs, then per cryptomatte AOV the visits' pairs as one flat range) into LDS, and
// the tile's table lines likewise; then lane = pixel walks its visits out of LDS, in stream order, into its LDS copy
// of the table; the tables go back as whole lines.  HBM-bound: one bit per visit (48 B without the bitmap), entries * 8 B
// per visit and AOV, the table lines written once per AOV (and read first unless the frame was just cleared).
struct CryptoTile {
  uint32_t tp;            // pixels per tile (= block size)
  uint32_t off_w, off_h, off_cw, off_k, off_wt, off_tot, off_pix;     // LDS offsets in 4-byte words
  const uint32_t *flagged;     // bit v: visit v was redistributed by the pass (its work lists, flag_bits_kernel); null: decide here
};

// The visits the pass has redistributed, as a bitmap: one bit per visit from the scan's work lists instead of the three
// columns (48 B per visit) the decision would have to read again.
__global__ __launch_bounds__(256) void flag_bits_kernel(const uint2 *work, const DevCounters *ctr, uint64_t cap, uint32_t *bits) {
  const uint64_t n = ctr->work_count < cap ? ctr->work_count : cap;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t v = work[i].x;
    atomicOr(bits + (v >> 5), 1u << (v & 31u));
  }
}

// kTables 1: the tables hold nothing yet (a pass straight after lentil_hip_clear_frame, the usual one): their lines are
// not read, only written -- and quads of free slots not even that.  kTables 2 (round 5): the same pass when the clear itself
// was put off (LentilCrypto::clear_pending) and the stream covers every pixel of the frame -- every line is written whole,
// free slots and all, and the 1.06 GB per AOV that the clear would have written (0.16 ms ahead of the scan) are not.
// kTables 0: lines are read, added to, written.
template <int kTables>
__global__ __launch_bounds__(128) void crypto_direct_tile_kernel(CryptoDev C, VisitsDev V, lentil_params P, double lens_length,
                                                                 CryptoTile T) {
  extern __shared__ uint32_t crypto_lds[];
  float *s_w = reinterpret_cast<float *>(crypto_lds + T.off_w);        // [tp * M] inverse density, or the unused pattern
  float *s_h = reinterpret_cast<float *>(crypto_lds + T.off_h);        // [tp * M * entries]
  float *s_cw = reinterpret_cast<float *>(crypto_lds + T.off_cw);
  uint32_t *s_k = crypto_lds + T.off_k;                                // [tp][slots + 1]
  float *s_wt = reinterpret_cast<float *>(crypto_lds + T.off_wt);      // [tp][slots + 1]
  float *s_tot = reinterpret_cast<float *>(crypto_lds + T.off_tot);    // [tp]
  uint32_t *s_pix = crypto_lds + T.off_pix;                            // [tp] frame pixel of the tile's entries
  constexpr bool kCleared = kTables != 0;
  const uint32_t M = V.visits_per_pixel, E = C.entries, SL = C.slots, SP = SL + 1u;
  const uint64_t n_pix = V.n / M;
  const uint64_t n_tiles = (n_pix + T.tp - 1) / T.tp;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t q0 = tile * T.tp;
    const uint32_t np_tile = (uint32_t)(n_pix - q0 < T.tp ? n_pix - q0 : T.tp);
    const uint64_t v0 = q0 * M;
    const uint32_t nv = np_tile * M;
    __syncthreads();                       // the previous tile's LDS is done with
    // (three visits' columns in flight per lane before the first decision: at 6 waves per CU the loads of one
    // iteration alone do not cover the HBM latency)
    if (T.flagged) {
      for (uint32_t i = threadIdx.x; i < nv; i += blockDim.x) {
        const uint64_t v = v0 + i;
        const bool redistributed = (T.flagged[v >> 5] >> (v & 31u)) & 1u;
        s_w[i] = redistributed ? __uint_as_float(kCryptoEmpty) : (V.inv_density ? V.inv_density[v] : P.inverse_sample_density);
      }
    } else
    for (uint32_t i0 = threadIdx.x; i0 < nv; i0 += 3u * blockDim.x) {
      float4 pz[3], vi[3], tr[3];
#pragma unroll
      for (uint32_t u = 0; u < 3; ++u) {
        const uint32_t i = i0 + u * blockDim.x;
        if (i < nv) { pz[u] = nt_load(V.pos_z + v0 + i); vi[u] = nt_load(V.volume_ignore + v0 + i); tr[u] = nt_load(V.transmission + v0 + i); }
      }
#pragma unroll
      for (uint32_t u = 0; u < 3; ++u) {
        const uint32_t i = i0 + u * blockDim.x;
        if (i >= nv) continue;
        const uint64_t v = v0 + i;
        const float invd = V.inv_density ? V.inv_density[v] : P.inverse_sample_density;
        const bool redistributed = visit_redistributes(P, lens_length, pz[u], vi[u], tr[u], invd, [&]() { return V.raydir_time[v]; }, V.cam);
        s_w[i] = redistributed ? __uint_as_float(kCryptoEmpty) : invd;
      }
    }
    int px = 0, py = 0;
    uint64_t pix = 0;
    const bool own = threadIdx.x < np_tile;
    if (own) {
      visit_pixel(V, (q0 + threadIdx.x) * M, px, py);
      pix = (uint64_t)P.xres * (uint32_t)py + (uint32_t)px;
      s_pix[threadIdx.x] = (uint32_t)pix;
    }
    for (uint32_t c = 0; c < C.n_crypto; ++c) {
      __syncthreads();
      const float *gh = C.hash[c] + v0 * E, *gw = C.weight[c] + v0 * E;
      if (((nv * E) & 3u) == 0 && ((v0 * E) & 3u) == 0) {          // 16 bytes per lane (full tiles always are)
#pragma unroll 4
        for (uint32_t i = threadIdx.x; i < nv * E / 4; i += blockDim.x) {
          const float4 a = nt_load(reinterpret_cast<const float4 *>(gh) + i), b = nt_load(reinterpret_cast<const float4 *>(gw) + i);
          *reinterpret_cast<float4 *>(s_h + 4 * i) = a;
          *reinterpret_cast<float4 *>(s_cw + 4 * i) = b;
        }
      } else {
        for (uint32_t i = threadIdx.x; i < nv * E; i += blockDim.x) { s_h[i] = __builtin_nontemporal_load(gh + i); s_cw[i] = __builtin_nontemporal_load(gw + i); }
      }
      // the tile's table lines: element (pixel j, slot s); the pixel of tile entry j from its stream position
      if (kCleared) {
        for (uint32_t i = threadIdx.x; i < np_tile * SP; i += blockDim.x) { s_k[i] = kCryptoEmpty; s_wt[i] = 0.0f; }
      } else if ((SL & 3u) == 0) {
        const uint32_t QL = SL / 4;
#pragma unroll 4
        for (uint32_t i = threadIdx.x; i < np_tile * QL; i += blockDim.x) {
          const uint32_t j = i / QL, sl = (i - j * QL) * 4;
          const uint64_t at = ((uint64_t)c * C.np + s_pix[j]) * SL + sl;
          const uint4 k4 = *reinterpret_cast<const uint4 *>(C.keys + at);
          const float4 w4 = *reinterpret_cast<const float4 *>(C.wts + at);
          uint32_t *dk = s_k + j * SP + sl; float *dw = s_wt + j * SP + sl;
          dk[0] = k4.x; dk[1] = k4.y; dk[2] = k4.z; dk[3] = k4.w;
          dw[0] = w4.x; dw[1] = w4.y; dw[2] = w4.z; dw[3] = w4.w;
        }
      } else {
        for (uint32_t i = threadIdx.x; i < np_tile * SL; i += blockDim.x) {
          const uint32_t j = i / SL, sl = i - j * SL;
          const uint64_t at = ((uint64_t)c * C.np + s_pix[j]) * SL + sl;
          s_k[j * SP + sl] = C.keys[at];
          s_wt[j * SP + sl] = C.wts[at];
        }
      }
      if (own) s_tot[threadIdx.x] = kCleared ? 0.0f : C.total[(uint64_t)c * C.np + pix];
      __syncthreads();
      if (own) {
        uint32_t *K = s_k + threadIdx.x * SP;
        float *Wt = s_wt + threadIdx.x * SP;
        float total = s_tot[threadIdx.x];
        // The pixel's visits mostly carry the same few ids (one object covers the pixel): the two ids met last are kept in
        // registers with their slot and the slot's running weight, so that a pair whose id is one of them costs one add instead
        // of a chain of dependent LDS reads and writes (probe, compare, read-modify-write) -- which is what this kernel's waves
        // spent 60 % of their cycles waiting for (profiles/r04_pmc_crypto.txt).  A slot's weight still receives the same adds in
        // the same order (it is loaded when the id enters the registers, written back when it leaves them): bit for bit the
        // sequential sums, as asserted by tests/test_crypto.py.
        uint32_t k0 = kCryptoEmpty, k1 = kCryptoEmpty, s0 = 0, s1 = 0;
        float a0 = 0.0f, a1 = 0.0f;
        bool newer0 = false;                 // which of the two was used last
        for (uint32_t m = 0; m < M; ++m) {
          const float w = s_w[threadIdx.x * M + m];
          if (__float_as_uint(w) == kCryptoEmpty) continue;
          total += w;                                                                       // :815
          for (uint32_t e = 0; e < E; ++e) {
            const float cw = s_cw[(threadIdx.x * M + m) * E + e];
            if (__float_as_uint(cw) == kCryptoEmpty) continue;
            const uint32_t key = crypto_key_bits(s_h[(threadIdx.x * M + m) * E + e]);
            if (key == kCryptoEmpty) {
              // (an id whose bits are the "free slot" pattern -- a NaN: as ever, its weight lands in the first free slot of its
              // probe sequence, which no register holds)
              uint32_t sl = crypto_first_slot(key, SL);
              bool placed = false;
              for (uint32_t i = 0; i < SL && !placed; ++i) {
                if (K[sl] == kCryptoEmpty) { Wt[sl] += cw * w; placed = true; }
                else sl = sl + 1u == SL ? 0u : sl + 1u;
              }
              if (!placed) atomicAdd(C.overflow, 1ull);
              continue;
            }
            if (key == k0) { a0 += cw * w; newer0 = true; continue; }                        // :817
            if (key == k1) { a1 += cw * w; newer0 = false; continue; }
            // another id: the older of the two leaves the registers, this one is looked up (or entered) in the table
            if (newer0) { if (k1 != kCryptoEmpty) Wt[s1] = a1; } else { if (k0 != kCryptoEmpty) Wt[s0] = a0; }
            uint32_t sl = crypto_first_slot(key, SL);
            bool placed = false;
            for (uint32_t i = 0; i < SL && !placed; ++i) {
              const uint32_t cur = K[sl];
              if (cur == kCryptoEmpty) K[sl] = key;
              if (cur == kCryptoEmpty || cur == key) placed = true;
              else sl = sl + 1u == SL ? 0u : sl + 1u;
            }
            if (!placed) { atomicAdd(C.overflow, 1ull); if (newer0) k1 = kCryptoEmpty; else k0 = kCryptoEmpty; continue; }
            const float acc = Wt[sl] + cw * w;
            if (newer0) { k1 = key; s1 = sl; a1 = acc; newer0 = false; } else { k0 = key; s0 = sl; a0 = acc; newer0 = true; }
          }
        }
        if (k0 != kCryptoEmpty) Wt[s0] = a0;
        if (k1 != kCryptoEmpty) Wt[s1] = a1;
        s_tot[threadIdx.x] = total;
      }
      __syncthreads();
      if ((SL & 3u) == 0) {
        const uint32_t QL = SL / 4;
        for (uint32_t i = threadIdx.x; i < np_tile * QL; i += blockDim.x) {
          const uint32_t j = i / QL, sl = (i - j * QL) * 4;
          const uint64_t at = ((uint64_t)c * C.np + s_pix[j]) * SL + sl;
          const uint32_t *sk = s_k + j * SP + sl; const float *sw = s_wt + j * SP + sl;
          // (straight after a clear the table holds "free" everywhere already: four free slots need not be written again --
          // most of a pixel's sixteen are, and the lines this kernel writes are half of what it moves)
          if (kTables == 1 && (sk[0] & sk[1] & sk[2] & sk[3]) == kCryptoEmpty) continue;
          *reinterpret_cast<uint4 *>(C.keys + at) = make_uint4(sk[0], sk[1], sk[2], sk[3]);
          *reinterpret_cast<float4 *>(C.wts + at) = make_float4(sw[0], sw[1], sw[2], sw[3]);
        }
      } else {
        for (uint32_t i = threadIdx.x; i < np_tile * SL; i += blockDim.x) {
          const uint32_t j = i / SL, sl = i - j * SL;
          const uint64_t at = ((uint64_t)c * C.np + s_pix[j]) * SL + sl;
          C.keys[at] = s_k[j * SP + sl];
          C.wts[at] = s_wt[j * SP + sl];
        }
      }
      if (own) C.total[(uint64_t)c * C.np + pix] = s_tot[threadIdx.x];
    }
  }
}

// accepted draws (src/lentil_filter.cpp:296 polynomial optics, :443 thin lens; with abb_chromatic every channel's
// draw is a log record of its own, like it is an add of its own there) -- sample weight = inverse density / draws
__global__ __launch_bounds__(256) void crypto_draws_kernel(CryptoDev C, VisitsDev V, lentil_params P, double lens_length,
                                                           const lentil_draw_record *log, const unsigned long long *log_count, uint64_t log_cap) {
  // (the pass's own count, read here: the host does not wait for the pass before it enqueues this; a log that did not
  // fit is found out afterwards, and the frame is void then)
  const uint64_t n_log = *log_count < log_cap ? *log_count : log_cap;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_log; i += stride) {
    const uint32_t v = log[i].visit;
    const float invd = V.inv_density ? V.inv_density[v] : P.inverse_sample_density;
    const VisitInfo I = visit_prologue(P, lens_length, V.rgba[v], V.pos_z[v], V.raydir_time[v], V.volume_ignore[v],
                                       V.transmission[v], invd, V.cam);
    const float inv_samples = (float)(1.0 / (double)(float)(int)I.samples);            // src/lentil_filter.cpp:199
    crypto_add_visit(C, log[i].pixel, v, invd * inv_samples);
  }
}

// src/lentil_imager.cpp:121-161 for one AOV: the map's pairs ordered by weight, largest first (std::sort with
// compareTail on the map's id-ordered pairs: for the up to 16 pairs libstdc++ sorts by insertion, equal weights stay
// in id order -- the order used here for any count), positions rank and rank + 1 written as (id, weight / total).
// has[p] = 0 where the map has no more than `rank` entries (the reference stops copying the bucket row there, :132-134).
__global__ __launch_bounds__(256) void crypto_rank_kernel(CryptoDev C, uint32_t c, uint32_t rank, float4 *out, uint8_t *has) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < C.np; p += stride) {
    const uint32_t *K = C.keys + ((uint64_t)c * C.np + p) * C.slots;
    const float *Wt = C.wts + ((uint64_t)c * C.np + p) * C.slots;
    uint32_t count = 0;
    for (uint32_t i = 0; i < C.slots; ++i) count += K[i] != kCryptoEmpty;
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (count > rank) {
      const float total = C.total[(uint64_t)c * C.np + p];
      for (uint32_t i = 0; i < C.slots; ++i) {
        if (K[i] == kCryptoEmpty) continue;
        const float wi = Wt[i], ki = __uint_as_float(K[i]);
        uint32_t pos = 0;                    // pairs sorted before pair i
        for (uint32_t j = 0; j < C.slots; ++j) {
          if (j == i || K[j] == kCryptoEmpty) continue;
          const float wj = Wt[j], kj = __uint_as_float(K[j]);
          pos += (wj > wi) || (!(wi > wj) && kj < ki);
        }
        if (pos == rank) { o.x = ki; o.y = wi / total; }
        else if (pos == rank + 1u) { o.z = ki; o.w = wi / total; }
      }
    }
    out[p] = o;
    has[p] = count > rank;
  }
}

// lentil_hip_set_draw_log: a log of the caller's is the caller's to size (an overflow is reported, the log stays)
static void crypto_log_set_by_caller(lentil_hip_ctx *ctx) {
  if (ctx->crypto_auto_log > ctx->crypto_auto_log_hint) ctx->crypto_auto_log_hint = ctx->crypto_auto_log;
  ctx->crypto_auto_log = 0;
}

static uint32_t crypto_count(const lentil_hip_ctx *ctx) { return ctx->crypto ? ctx->crypto->D.n_crypto : 0; }

static void crypto_free_columns(LentilCrypto *k) {
  for (void *p : k->owned) (void)hipFree(p);
  k->owned.clear();
  for (uint32_t c = 0; c < LENTIL_MAX_CRYPTO; ++c) k->D.hash[c] = k->D.weight[c] = nullptr;
  k->have_columns = false;
  k->from_upload = false;
  k->n_visits = 0;
}

// the piecewise upload (lentil_upload.h) frees or reallocates the columns it lent to this module
static void crypto_columns_gone(lentil_hip_ctx *ctx) {
  LentilCrypto *k = ctx->crypto;
  if (!k || !k->from_upload) return;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  crypto_free_columns(k);
}

// lentil_hip_visits_end of a stream that carried cryptomatte caches
static void crypto_bind_uploaded(lentil_hip_ctx *ctx, uint32_t n_crypto, uint32_t entries, uint64_t n, float *const *hash, float *const *weight) {
  LentilCrypto *k = ctx->crypto;
  if (!k || k->D.n_crypto != n_crypto) return;       // (alloc_crypto changed under the stream: redistribute reports the mismatch)
  crypto_free_columns(k);
  for (uint32_t a = 0; a < n_crypto; ++a) { k->D.hash[a] = hash[a]; k->D.weight[a] = weight[a]; }
  k->D.entries = entries;
  k->n_visits = n;
  k->visits_gen = ctx->visits_gen;
  k->have_columns = true;
  k->from_upload = true;
}

static void crypto_destroy(lentil_hip_ctx *ctx) {
  LentilCrypto *k = ctx->crypto;
  if (!k) return;
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  crypto_free_columns(k);
  (void)hipFree(k->D.keys); (void)hipFree(k->D.wts); (void)hipFree(k->D.total); (void)hipFree(k->D.overflow);
  (void)hipFree(k->d_rank); (void)hipFree(k->d_has); (void)hipFree(k->d_flag_bits);
  delete k;
  ctx->crypto = nullptr;
}

LENTIL_API int lentil_hip_alloc_crypto(lentil_hip_ctx *ctx, uint32_t n_crypto, uint32_t slots_per_pixel) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "alloc_crypto needs alloc_frame first");
  if (n_crypto > LENTIL_MAX_CRYPTO) return fail(ctx, LENTIL_ERR_INVALID, "too many cryptomatte AOVs");
  if (slots_per_pixel == 0) slots_per_pixel = 16;
  if (slots_per_pixel > kCryptoMaxSlots) return fail(ctx, LENTIL_ERR_INVALID, "at most 64 table slots per pixel");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  crypto_destroy(ctx);
  if (n_crypto == 0) return LENTIL_OK;
  LentilCrypto *k = new LentilCrypto();
  ctx->crypto = k;
  k->D.n_crypto = n_crypto;
  k->D.slots = slots_per_pixel;
  k->D.np = ctx->F.np;
  const uint64_t cells = (uint64_t)n_crypto * k->D.np * slots_per_pixel;
  HIP_TRY(ctx, hipMalloc(&k->D.keys, cells * sizeof(uint32_t)));
  HIP_TRY(ctx, hipMalloc(&k->D.wts, cells * sizeof(float)));
  HIP_TRY(ctx, hipMalloc(&k->D.total, (uint64_t)n_crypto * k->D.np * sizeof(float)));
  HIP_TRY(ctx, hipMalloc(&k->D.overflow, sizeof(unsigned long long)));
  HIP_TRY(ctx, hipMalloc(&k->d_rank, k->D.np * sizeof(float4)));
  HIP_TRY(ctx, hipMalloc(&k->d_has, k->D.np));
  HIP_TRY(ctx, hipMemsetAsync(k->D.keys, 0xFF, cells * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(k->D.wts, 0, cells * sizeof(float), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(k->D.total, 0, (uint64_t)n_crypto * k->D.np * sizeof(float), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(k->D.overflow, 0, sizeof(unsigned long long), ctx->stream));
  k->tables_clear = true;
  return LENTIL_OK;
}

// lentil_hip_clear_frame: AOVData::allocate_cryptomatte_buffers, src/aov_data.h:145-150
static int crypto_clear(lentil_hip_ctx *ctx) {
  LentilCrypto *k = ctx->crypto;
  if (!k) return LENTIL_OK;
  // Round 5, an option: the wipe of the tables (1.06 GB per AOV at sixteen ids per pixel: 0.16 ms ahead of the pass's scan) is put off.
  // The usual next thing is a pass over a stream that covers the whole frame, whose own-pixel kernel then writes every line
  // whole instead of finding it wiped (crypto_direct_tile_kernel<2>); anything else that looks at the tables first wipes them
  // then (crypto_flush_clear).
  // (Measured, same box, headline frame, clear + pass: +1.39-1.47 ms for one AOV and +1.26 per AOV for three either way --
  // the lines the kernel then writes whole cost what the wipe cost; the pass alone reads +1.36 / +1.24 instead of +1.07 / +1.05.
  // Off; LENTIL_CRYPTO_LAZY_CLEAR=1 turns it on.)
  const char *lazy_env = getenv("LENTIL_CRYPTO_LAZY_CLEAR");      // (read per call: the tests switch it)
  const bool lazy = lazy_env && lazy_env[0] == '1';
  if (lazy) {
    HIP_TRY(ctx, hipMemsetAsync(k->D.overflow, 0, sizeof(unsigned long long), ctx->stream));
    k->tables_clear = true;
    k->clear_pending = true;
    return LENTIL_OK;
  }
  k->clear_pending = false;
  const uint64_t cells = (uint64_t)k->D.n_crypto * k->D.np * k->D.slots;
  HIP_TRY(ctx, hipMemsetAsync(k->D.keys, 0xFF, cells * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(k->D.wts, 0, cells * sizeof(float), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(k->D.total, 0, (uint64_t)k->D.n_crypto * k->D.np * sizeof(float), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(k->D.overflow, 0, sizeof(unsigned long long), ctx->stream));
  k->tables_clear = true;
  return LENTIL_OK;
}

// the wipe lentil_hip_clear_frame put off, on stream `st` (ahead of whatever reads or adds to the tables there)
static int crypto_flush_clear(lentil_hip_ctx *ctx, hipStream_t st) {
  LentilCrypto *k = ctx->crypto;
  if (!k || !k->clear_pending) return LENTIL_OK;
  const uint64_t cells = (uint64_t)k->D.n_crypto * k->D.np * k->D.slots;
  HIP_TRY(ctx, hipMemsetAsync(k->D.keys, 0xFF, cells * sizeof(uint32_t), st));
  HIP_TRY(ctx, hipMemsetAsync(k->D.wts, 0, cells * sizeof(float), st));
  HIP_TRY(ctx, hipMemsetAsync(k->D.total, 0, (uint64_t)k->D.n_crypto * k->D.np * sizeof(float), st));
  k->clear_pending = false;
  return LENTIL_OK;
}

static int crypto_check_columns(lentil_hip_ctx *ctx, const lentil_crypto_visits *c) {
  if (!ctx->crypto) return fail(ctx, LENTIL_ERR_INVALID, "no cryptomatte AOVs allocated (lentil_hip_alloc_crypto)");
  if (!c) return fail(ctx, LENTIL_ERR_INVALID, "crypto columns are null");
  if (c->n_crypto != ctx->crypto->D.n_crypto) return fail(ctx, LENTIL_ERR_INVALID, "crypto columns for a different number of AOVs");
  if (c->entries == 0 || c->entries > 64) return fail(ctx, LENTIL_ERR_INVALID, "crypto entries per visit must be 1..64");
  for (uint32_t a = 0; a < c->n_crypto; ++a)
    if (c->n && (!c->hash[a] || !c->weight[a])) return fail(ctx, LENTIL_ERR_INVALID, "a crypto column is null");
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_bind_crypto(lentil_hip_ctx *ctx, const lentil_crypto_visits *c) {
  CHECK_CTX(ctx);
  { const int rc = crypto_check_columns(ctx, c); if (rc) return rc; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  LentilCrypto *k = ctx->crypto;
  crypto_free_columns(k);
  for (uint32_t a = 0; a < c->n_crypto; ++a) { k->D.hash[a] = c->hash[a]; k->D.weight[a] = c->weight[a]; }
  k->D.entries = c->entries;
  k->n_visits = c->n;
  k->visits_gen = ctx->visits_gen;
  k->have_columns = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_upload_crypto(lentil_hip_ctx *ctx, const lentil_crypto_visits *c) {
  CHECK_CTX(ctx);
  { const int rc = crypto_check_columns(ctx, c); if (rc) return rc; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  LentilCrypto *k = ctx->crypto;
  crypto_free_columns(k);
  const size_t bytes = (size_t)c->n * c->entries * sizeof(float);
  for (uint32_t a = 0; a < c->n_crypto && bytes; ++a) {
    for (int which = 0; which < 2; ++which) {
      void *d = nullptr;
      HIP_TRY(ctx, hipMalloc(&d, bytes));
      k->owned.push_back(d);
      HIP_TRY(ctx, hipMemcpyAsync(d, which ? (const void *)c->weight[a] : (const void *)c->hash[a], bytes, hipMemcpyHostToDevice, ctx->stream));
      (which ? k->D.weight[a] : k->D.hash[a]) = (const float *)d;
    }
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  k->D.entries = c->entries;
  k->n_visits = c->n;
  k->visits_gen = ctx->visits_gen;
  k->have_columns = true;
  return LENTIL_OK;
}

// lentil_hip_redistribute, before the pass: the replay needs the pass's draw log
static int crypto_before_pass(lentil_hip_ctx *ctx) {
  LentilCrypto *k = ctx->crypto;
  if (!k) return LENTIL_OK;
  if (!k->have_columns || k->n_visits != ctx->V.n || k->visits_gen != ctx->visits_gen)
    return fail(ctx, LENTIL_ERR_INVALID, "the cryptomatte columns do not belong to the bound visit stream (lentil_hip_upload_crypto / _bind_crypto after the visits)");
  if (ctx->closest_deferred)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "cryptomatte AOVs travel with the tiled exchange only (lentil_hip_exchange_bands)");
  if (ctx->log_cap == 0) {
    // no log asked for by the caller: one sized from the last pass, or 4 Mi records (48 MB; LENTIL_CRYPTO_LOG overrides)
    uint64_t want = 4ull << 20;
    if (const char *e = getenv("LENTIL_CRYPTO_LOG")) { const long long v = atoll(e); if (v > 0) want = (uint64_t)v; }
    if (ctx->crypto_auto_log_hint > want) want = ctx->crypto_auto_log_hint;
    const int rc = lentil_hip_set_draw_log(ctx, want);
    if (rc) return rc;
    ctx->crypto_auto_log = want;
  }
  // one bit per visit for crypto_enqueue_direct (which runs while the pass is in flight: it must not allocate)
  if (ctx->V.n <= 0xFFFFFFFFull) {
    const uint64_t words = (ctx->V.n + 31) / 32;
    if (words > k->flag_words) {
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      if (k->d_flag_bits) (void)hipFree(k->d_flag_bits);
      k->d_flag_bits = nullptr; k->flag_words = 0;
      HIP_TRY(ctx, hipMalloc(&k->d_flag_bits, words * sizeof(uint32_t)));
      k->flag_words = words;
    }
  }
  return LENTIL_OK;
}

// ... and after it (on the main stream, behind everything the pass enqueued)
// The adds of the visits that stay in their own pixel, on stream `st` (behind the pass's scans: all this needs of the pass
// is its work lists).  A streamed pass enqueues it on the context's spare stream as soon as its scan is, where there is
// one (lentil_hip_ctx::aux_stream), so that it runs beside the draws: they leave HBM idle.
static int crypto_enqueue_direct(lentil_hip_ctx *ctx, hipStream_t st) {
  LentilCrypto *k = ctx->crypto;
  if (!k || !ctx->V.n) return LENTIL_OK;
  const double lens_length = ctx->have_lens ? ctx->hlens.length : 0.0;
  const unsigned blocks = (unsigned)ctx->num_cu * 8;
  // a wipe that was put off: the tile kernel can stand in for it where its owner lanes write every line of the frame
  bool whole_lines = false;
  if (k->clear_pending) {
    const VisitsDev &V = ctx->V;
    const bool whole_frame = V.visits_per_pixel && V.n == (uint64_t)k->D.np * V.visits_per_pixel &&
                             V.pixels_per_row == (uint32_t)ctx->P.xres && V.pixel_x0 <= 0 && V.pixel_y0 <= 0 && V.pixel_row_stride <= 1;
    const char *force = getenv("LENTIL_CRYPTO_TILE");
    const size_t lds_words = (size_t)((128u * V.visits_per_pixel + 3u) & ~3u) + 2u * ((128u * V.visits_per_pixel * k->D.entries + 3u) & ~3u) +
                             2u * 128u * (k->D.slots + 1u) + 2u * 128u;
    whole_lines = whole_frame && k->tables_clear && (k->D.slots & 3u) == 0u && lds_words * 4u <= 64u * 1024u && !(force && force[0] == '0');
    if (!whole_lines) { const int rc = crypto_flush_clear(ctx, st); if (rc) return rc; }
  }
  // pixel-major streams whose pixels are all distinct (one pass per clear: the tables hold nothing yet that another
  // lane could be adding to): owner lanes, no atomics
  if (ctx->V.visits_per_pixel && ctx->V.n % ctx->V.visits_per_pixel == 0) {
    CryptoTile T{};
    T.tp = 128;
    const uint32_t M = ctx->V.visits_per_pixel, E = k->D.entries, SP = k->D.slots + 1;
    uint32_t at = 0;
    T.off_w = at; at += (T.tp * M + 3u) & ~3u;
    T.off_h = at; at += (T.tp * M * E + 3u) & ~3u;          // (16-byte aligned: filled with float4 stores)
    T.off_cw = at; at += (T.tp * M * E + 3u) & ~3u;
    T.off_k = at; at += T.tp * SP;
    T.off_wt = at; at += T.tp * SP;
    T.off_tot = at; at += T.tp;
    T.off_pix = at; at += T.tp;
    const size_t lds = (size_t)at * 4;
    const char *force = getenv("LENTIL_CRYPTO_TILE");
    if (lds <= 64 * 1024 && !(force && force[0] == '0')) {
      const uint64_t n_tiles = (ctx->V.n / M + T.tp - 1) / T.tp;
      // (LENTIL_CRYPTO_TILE_BLOCKS: blocks per CU of the launch -- the kernel walks the tiles with its grid's stride)
      static const int tile_blocks = getenv("LENTIL_CRYPTO_TILE_BLOCKS") ? atoi(getenv("LENTIL_CRYPTO_TILE_BLOCKS")) : 0;
      const uint64_t max_blocks = (uint64_t)ctx->num_cu * (uint64_t)(tile_blocks >= 1 && tile_blocks <= 64 ? tile_blocks : ctx->crypto_tile_blocks);
      // which visits the pass redistributed: from its work lists (LENTIL_CRYPTO_FLAGS=0: decided again from the columns)
      static const bool use_flags = !(getenv("LENTIL_CRYPTO_FLAGS") && getenv("LENTIL_CRYPTO_FLAGS")[0] == '0');
      if (use_flags && ctx->V.n <= 0xFFFFFFFFull) {
        // (sized by crypto_before_pass: nothing is allocated or freed while the pass's kernels are being enqueued -- hipFree
        // waits for the device, and the publishers and solves of a streamed pass are launched after this)
        const uint64_t words = (ctx->V.n + 31) / 32;
        if (words > k->flag_words) return fail(ctx, LENTIL_ERR_INVALID, "cryptomatte: the redistributed-visit flags were not sized for this stream");
        HIP_TRY(ctx, hipMemsetAsync(k->d_flag_bits, 0, words * sizeof(uint32_t), st));
        for (int ci = 0; ci < ctx->n_chunks; ++ci) {
          const lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
          if (ch.v_end <= ch.v_begin) continue;
          hipLaunchKernelGGL(flag_bits_kernel, dim3(64), dim3(256), 0, st, ctx->d_work + ch.v_begin, ctx->d_ctr + ci,
                             ch.v_end - ch.v_begin, k->d_flag_bits);
        }
        HIP_TRY(ctx, hipGetLastError());
        T.flagged = k->d_flag_bits;
      }
      const dim3 grid((unsigned)(n_tiles < max_blocks ? n_tiles : max_blocks));
      if (whole_lines)
        hipLaunchKernelGGL(crypto_direct_tile_kernel<2>, grid, dim3(128), lds, st, k->D, ctx->V, ctx->P, lens_length, T);
      else if (k->tables_clear)
        hipLaunchKernelGGL(crypto_direct_tile_kernel<1>, grid, dim3(128), lds, st, k->D, ctx->V, ctx->P, lens_length, T);
      else
        hipLaunchKernelGGL(crypto_direct_tile_kernel<0>, grid, dim3(128), lds, st, k->D, ctx->V, ctx->P, lens_length, T);
      if (whole_lines) k->clear_pending = false;
    } else {
      hipLaunchKernelGGL(crypto_direct_owner_kernel, dim3(blocks), dim3(256), 0, st, k->D, ctx->V, ctx->P, lens_length);
    }
  } else
    hipLaunchKernelGGL(crypto_direct_kernel, dim3(blocks), dim3(256), 0, st, k->D, ctx->V, ctx->P, lens_length);
  HIP_TRY(ctx, hipGetLastError());
  k->tables_clear = false;
  return LENTIL_OK;
}

static int crypto_after_pass(lentil_hip_ctx *ctx) {
  LentilCrypto *k = ctx->crypto;
  if (!k || !ctx->V.n) return LENTIL_OK;
  const unsigned long long *d_n_log =
      reinterpret_cast<const unsigned long long *>((char *)(ctx->d_ctr + ctx->n_chunks) + offsetof(DevCounters, log_count));
  const double lens_length = ctx->have_lens ? ctx->hlens.length : 0.0;
  const unsigned blocks = (unsigned)ctx->num_cu * 8;
  if (ctx->crypto_direct_enqueued) {
    // (the pass has put the own-pixel adds beside its draws: the draws' replay follows them)
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_crypto, 0));
    ctx->crypto_direct_enqueued = false;
  } else {
    const int rc = crypto_enqueue_direct(ctx, ctx->stream);
    if (rc) return rc;
  }
  if (ctx->log_cap) {
    hipLaunchKernelGGL(crypto_draws_kernel, dim3(blocks), dim3(256), 0, ctx->stream, k->D, ctx->V, ctx->P, lens_length,
                       ctx->d_log, d_n_log, (uint64_t)ctx->log_cap);
    HIP_TRY(ctx, hipGetLastError());
  }
  unsigned long long full = 0, n_log = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n_log, d_n_log, sizeof(n_log), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(&full, k->D.overflow, sizeof(full), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n_log > ctx->log_cap) {
    const uint64_t had = ctx->log_cap;
    if (ctx->crypto_auto_log) {       // this module's own log: the next pass gets one that fits
      (void)lentil_hip_set_draw_log(ctx, 0);
      ctx->crypto_auto_log_hint = n_log + n_log / 4;
    }
    return fail(ctx, LENTIL_ERR_NOMEM, "the draw log (" + std::to_string(had) + " records) is too small for the cryptomatte AOVs of this pass (" +
                                           std::to_string(n_log) + " accepted draws): clear the frame and redistribute again");
  }
  if (full)
    return fail(ctx, LENTIL_ERR_NOMEM, std::to_string(full) + " cryptomatte adds found their pixel's table full (" + std::to_string(k->D.slots) +
                                           " ids per pixel): allocate more slots (lentil_hip_alloc_crypto)");
  return LENTIL_OK;
}

// lentil_hip_exchange_bands, behind the accumulators' exchange: bands[2 q], bands[2 q + 1] = rank q's rows; [lo, hi) =
// the rows this rank's draws touched.  Counts first (one small all-gather), then one send / receive per pair, then the
// arrivals are added in rank order.
static int crypto_exchange_bands(lentil_hip_ctx *ctx, const int32_t *bands, int32_t lo, int32_t hi) {
  LentilCrypto *k = ctx->crypto;
  LentilComm *cm = ctx->comm;
  if (!k || !cm) return LENTIL_OK;
  const int world = cm->world, rank = cm->rank;
  const uint32_t xres = ctx->P.xres;
  const unsigned blocks = (unsigned)ctx->num_cu * 8;
  int rc;
  if ((rc = crypto_flush_clear(ctx, ctx->stream))) return rc;      // (a frame cleared and exchanged without a pass)
  void *p;
  if ((rc = comm_scratch(ctx, cm, (size_t)world * 12 + 0, (size_t)world * sizeof(unsigned long long), &p))) return rc;
  unsigned long long *d_counts = (unsigned long long *)p;
  HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, (size_t)world * sizeof(unsigned long long), ctx->stream));
  std::vector<int32_t> s_lo((size_t)world, 0), s_hi((size_t)world, 0);
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    s_lo[(size_t)q] = lo > bands[2 * q] ? lo : bands[2 * q];
    s_hi[(size_t)q] = hi < bands[2 * q + 1] ? hi : bands[2 * q + 1];
    if (s_hi[(size_t)q] <= s_lo[(size_t)q]) continue;
    hipLaunchKernelGGL(crypto_band_records_kernel, dim3(blocks), dim3(256), 0, ctx->stream, k->D, (uint64_t)s_lo[(size_t)q] * xres,
                       (uint64_t)(s_hi[(size_t)q] - s_lo[(size_t)q]) * xres, (uint4 *)nullptr, 0ull, d_counts + q);
  }
  HIP_TRY(ctx, hipGetLastError());
  std::vector<unsigned long long> n_out((size_t)world, 0);
  HIP_TRY(ctx, hipMemcpyAsync(n_out.data(), d_counts, (size_t)world * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // the lists themselves
  std::vector<uint4 *> out((size_t)world, nullptr);
  HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, (size_t)world * sizeof(unsigned long long), ctx->stream));
  for (int q = 0; q < world; ++q) {
    if (!n_out[(size_t)q]) continue;
    if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 10, (size_t)n_out[(size_t)q] * sizeof(uint4), &p))) return rc;
    out[(size_t)q] = (uint4 *)p;
    hipLaunchKernelGGL(crypto_band_records_kernel, dim3(blocks), dim3(256), 0, ctx->stream, k->D, (uint64_t)s_lo[(size_t)q] * xres,
                       (uint64_t)(s_hi[(size_t)q] - s_lo[(size_t)q]) * xres, out[(size_t)q], n_out[(size_t)q], d_counts + q);
  }
  HIP_TRY(ctx, hipGetLastError());
  // who sends how much to whom: row r of the gathered matrix is rank r's list sizes per destination
  std::vector<int64_t> mine((size_t)world, 0);
  for (int q = 0; q < world; ++q) mine[(size_t)q] = (int64_t)n_out[(size_t)q];
  HIP_TRY(ctx, hipMemcpyAsync(cm->d_meta_mine, mine.data(), (size_t)world * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  RCCL_TRY(ctx, g_rccl.AllGather(cm->d_meta_mine, cm->d_meta_all, (size_t)world, kNcclInt64, cm->comm, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(cm->h_meta_all, cm->d_meta_all, (size_t)world * (size_t)world * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<uint64_t> n_in((size_t)world, 0);
  std::vector<uint4 *> in((size_t)world, nullptr);
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    n_in[(size_t)q] = (uint64_t)cm->h_meta_all[(size_t)q * (size_t)world + (size_t)rank];
    if (!n_in[(size_t)q]) continue;
    if ((rc = comm_scratch(ctx, cm, (size_t)q * 12 + 11, (size_t)n_in[(size_t)q] * sizeof(uint4), &p))) return rc;
    in[(size_t)q] = (uint4 *)p;
  }
  RCCL_TRY(ctx, g_rccl.GroupStart());
  int g_err = 0;
  for (int q = 0; q < world; ++q) {
    if (q == rank) continue;
    if (n_out[(size_t)q] && !g_err) g_err = g_rccl.Send(out[(size_t)q], (size_t)n_out[(size_t)q] * 16, kNcclUint8, q, cm->comm, ctx->stream);
    if (n_in[(size_t)q] && !g_err) g_err = g_rccl.Recv(in[(size_t)q], (size_t)n_in[(size_t)q] * 16, kNcclUint8, q, cm->comm, ctx->stream);
    cm->last_sent += n_out[(size_t)q] * 16ull; cm->last_received += n_in[(size_t)q] * 16ull;
  }
  {
    const int r_end = g_rccl.GroupEnd();
    if (g_err) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclSend / ncclRecv (cryptomatte): ") + g_rccl.GetErrorString(g_err));
    if (r_end) return fail(ctx, LENTIL_ERR_HIP, std::string("ncclGroupEnd (cryptomatte): ") + g_rccl.GetErrorString(r_end));
  }
  for (int q = 0; q < world; ++q) {
    if (!n_in[(size_t)q]) continue;
    hipLaunchKernelGGL(crypto_merge_records_kernel, dim3(blocks), dim3(256), 0, ctx->stream, k->D, in[(size_t)q], n_in[(size_t)q]);
  }
  HIP_TRY(ctx, hipGetLastError());
  k->tables_clear = false;
  unsigned long long full = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&full, k->D.overflow, sizeof(full), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (full)
    return fail(ctx, LENTIL_ERR_NOMEM, std::to_string(full) + " cryptomatte adds found their pixel's table full (" + std::to_string(k->D.slots) +
                                           " ids per pixel): allocate more slots (lentil_hip_alloc_crypto)");
  return LENTIL_OK;
}

static int crypto_check_index(lentil_hip_ctx *ctx, uint32_t crypto) {
  if (!ctx->crypto) return fail(ctx, LENTIL_ERR_INVALID, "no cryptomatte AOVs allocated");
  if (crypto >= ctx->crypto->D.n_crypto) return fail(ctx, LENTIL_ERR_INVALID, "cryptomatte AOV index out of range");
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_crypto(lentil_hip_ctx *ctx, uint32_t crypto, uint32_t rank, float *host_rgba, uint8_t *host_has_rank) {
  CHECK_CTX(ctx);
  { const int rc = crypto_check_index(ctx, crypto); if (rc) return rc; }
  if (!host_rgba) return fail(ctx, LENTIL_ERR_INVALID, "host_rgba is null");
  LentilCrypto *k = ctx->crypto;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  { const int rc_ = crypto_flush_clear(ctx, ctx->stream); if (rc_) return rc_; }
  hipLaunchKernelGGL(crypto_rank_kernel, dim3((unsigned)ctx->num_cu * 8), dim3(256), 0, ctx->stream, k->D, crypto, rank,
                     reinterpret_cast<float4 *>(k->d_rank), k->d_has);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(host_rgba, k->d_rank, k->D.np * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  if (host_has_rank) HIP_TRY(ctx, hipMemcpyAsync(host_has_rank, k->d_has, k->D.np, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_crypto_table(lentil_hip_ctx *ctx, uint32_t crypto, uint32_t *slots_per_pixel, uint32_t *host_id_bits,
                                                float *host_weight, float *host_total) {
  CHECK_CTX(ctx);
  { const int rc = crypto_check_index(ctx, crypto); if (rc) return rc; }
  LentilCrypto *k = ctx->crypto;
  if (slots_per_pixel) *slots_per_pixel = k->D.slots;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  { const int rc_ = crypto_flush_clear(ctx, ctx->stream); if (rc_) return rc_; }
  const uint64_t cells = k->D.np * k->D.slots;
  if (host_id_bits) HIP_TRY(ctx, hipMemcpyAsync(host_id_bits, k->D.keys + crypto * cells, cells * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (host_weight) HIP_TRY(ctx, hipMemcpyAsync(host_weight, k->D.wts + crypto * cells, cells * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (host_total) HIP_TRY(ctx, hipMemcpyAsync(host_total, k->D.total + crypto * k->D.np, k->D.np * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LENTIL_OK;
}
