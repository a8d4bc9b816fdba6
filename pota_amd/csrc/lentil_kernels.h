// lentil_kernels.h -- the gfx950 kernels of the redistribution path.
//
//   scan_*_kernel      K1+K2+K6: reads the visit columns (80+16K B/visit, HBM-bound), evaluates the
//                      redistribute predicate and the draw count, wave-ballot/prefix-sum compacts the
//                      redistributed visits ("items") into a work list and accumulates the
//                      non-redistributed visits of each source pixel in reference order.
//   prep_items_kernel  per item: camera-space target, seed, first batch of solve tasks.
//   solve_*_kernel     K3/K4/K5: one backward trace per (item, m) -- see "solve once" below.
//   accept_kernel      ordered acceptance of the traced draws + fp32 atomic splat; schedules the next
//                      batch of solves for items that still miss accepted draws.
//   resolve_kernel     K7: weight normalisation.
#pragma once
#include "lentil_device.h"
#include "lentil_batch_model.h"
#include "generated/lens_registry.h"

using namespace lentil;

// ---------------------------------------------------------------------------------------
// device-side bookkeeping
// ---------------------------------------------------------------------------------------
struct DevCounters {
  unsigned long long work_count;     // items pushed by the scan kernel
  unsigned long long sum_samples;    // sum of the items' draw counts
  unsigned long long redistributed;
  unsigned long long attempted;
  unsigned long long accepted;
  unsigned long long overflow;       // work list / task list / result pool overflow (results incomplete)
  unsigned long long log_count;
  unsigned long long newton_iters;   // lane-iterations of the Newton solver
  unsigned long long tries;          // solves started
  unsigned long long lane_rounds;    // 64 x scheduler rounds (iteration slots offered)
  unsigned long long slow_solves;    // solves handed over to solve_slow_kernel (stragglers)
  unsigned long long fallback;       // blind prep: the chunk did not fit the buffers sized from the last pass
  unsigned long long rounds_used;    // 1 + the last round whose accept kernel found items to process
  // per-round queues, double buffered by round parity
  unsigned int n_tasks[2];
  unsigned int task_head[2];
  unsigned int n_active[2];
  unsigned int active_head[2];
  unsigned long long pool_used[2];
  unsigned int n_slow[2];            // stragglers parked by this round's solve kernel
  unsigned int slow_head[2];
  unsigned int waves_started[2], waves_done[2];   // live straggler queue: solve waves of the round that have begun / exited
  unsigned int accept_done[2];       // accept blocks of the round that have emitted everything they will (DrawArgs::emit_live)
  // rows of the frame this pass has added to (zero-initialised: "none"): max over (INT_MAX - row), max over (row + 1)
  unsigned int inv_row_min, row_max_p1;
  // streamed pass: scan blocks that have published everything they found (solve_po_kernel<.., kStream> polls it)
  unsigned int scan_blocks_done, publishers_done;
  unsigned int n_ranges, range_head;   // work-list ranges the scan has handed to publish_kernel / tickets drawn on them
  unsigned int stuck, tile_next;       // a wave gave up waiting for its queue slot (kStuckTicks): the pass is void; scan_dma_kernel's tile cursor
  // a candidate for a closest-filtered AOV came with |Z| == 0 or NaN (closest_key_of): the pass is refused, see there
  unsigned int degenerate_depth, pad_;
  // Streamed pass with extension (ItemLive): items whose current batch is not complete yet / the task queue has its end
  // markers / tasks and items the solve waves added themselves
  unsigned int items_open, queue_final, ext_tasks, ext_items;
  unsigned int ext_n, ext_head, pad2_[2];        // the extension's own task queue (DrawArgs::ext_q): tasks appended / tickets drawn
  // diagnostics of a stall (lentil_hip_last_redo_note): accept blocks that have begun, per round parity; what the wave that
  // gave up first saw -- round, parity, the queue's n_tasks, accept_done[0], accept_started[0], its slot's tag word, block
  unsigned int accept_started[2];
  unsigned int stuck_info[8];
  // items an accept with DrawArgs::emit_live has finished / its queue has its end markers (what the solve kernel beside it
  // waits for does not depend on blocks of the accept that have not begun)
  unsigned int accept_items_done[2], accept_final[2];
  // LENTIL_DISPATCH_PROBE=1 in a library built with -DLENTIL_PROBE_BUILD (development aid, tools/dispatch_probe.sh): per XCD, blocks of the first accept that have begun and waves of the second
  // round's solve kernel / its stragglers / the first round's stragglers that are resident; and what those read when the
  // accept's last item was finished with blocks of its grid still not begun (lentil_hip_last_redo_note prints it)
  unsigned int probe_accept_xcc[8], probe_res_xcc[3][8];
  unsigned int probe_snap[1 + 8 + 24 + 2];
  // DrawArgs::early_accept: items whose first batch is complete (every result delivered, parked solves counted as delivered),
  // pushed by the solve wave that delivered the last one / tickets drawn on that queue by accept_kernel<4>
  unsigned int n_ready, ready_head;
};

LD_DEV uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }      // HW_REG_XCC_ID[3:0]

// per-thread running row range -> one pair of atomics per wave
LD_DEV void flush_row_range(DevCounters *ctr, uint32_t rmin, uint32_t rmax_p1) {
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t a = __shfl_down(rmin, off), b = __shfl_down(rmax_p1, off);
    rmin = a < rmin ? a : rmin;
    rmax_p1 = b > rmax_p1 ? b : rmax_p1;
  }
  if ((threadIdx.x & 63u) == 0u && rmax_p1) {
    atomicMax(&ctr->inv_row_min, 0x7FFFFFFFu - rmin);
    atomicMax(&ctr->row_max_p1, rmax_p1);
  }
}

struct VisitsDev {
  uint64_t n;
  uint32_t visits_per_pixel, pixels_per_row;
  int32_t pixel_x0, pixel_y0;
  uint32_t pixel_row_stride, n_extra;
  const float4 *rgba, *pos_z, *raydir_time, *volume_ignore, *transmission;
  const float4 *extra[LENTIL_MAX_AOVS - 1];
  const uint32_t *pixel;
  const float *inv_density;
  uint32_t id_base;       // ragged streams: frame-wide id of visit 0 (multi-GPU partitions)
  CamMotion cam;          // a moving camera's matrix keys (lentil_hip_set_camera_motion); n < 2: lentil_params::world_to_camera
};

// Frame-wide visit id: the position of a visit in the order a single process would walk the whole frame
// in.  Closest-filtered AOVs break depth ties by it (the later visit wins, src/lentil.h:833), so every
// GPU of a row-interleaved partition must number its visits like the undivided stream does.
LD_DEV uint32_t visit_gid(const VisitsDev &V, uint32_t v) {
  if (V.visits_per_pixel == 0 || V.pixel_y0 < 0) return V.id_base + v;
  const uint32_t row_visits = V.pixels_per_row * V.visits_per_pixel;
  const uint32_t ly = v / row_visits;
  return ((uint32_t)V.pixel_y0 + ly * V.pixel_row_stride) * row_visits + (v - ly * row_visits);
}
// inverse; false when the visit belongs to another GPU's partition
LD_DEV bool visit_from_gid(const VisitsDev &V, uint32_t gid, uint32_t &v) {
  if (V.visits_per_pixel == 0 || V.pixel_y0 < 0) {
    v = gid - V.id_base;
    return gid >= V.id_base && (uint64_t)v < V.n;
  }
  const uint32_t row_visits = V.pixels_per_row * V.visits_per_pixel;
  const uint32_t py = gid / row_visits;
  if (py < (uint32_t)V.pixel_y0 || (py - (uint32_t)V.pixel_y0) % V.pixel_row_stride) return false;
  v = ((py - (uint32_t)V.pixel_y0) / V.pixel_row_stride) * row_visits + (gid - py * row_visits);
  return (uint64_t)v < V.n;
}

struct FrameDev {
  // One record per pixel: n_aovs x RGBA (AOVData::buffer), then filter_weight_buffer, padded to a multiple
  // of 8 floats (32 B).  Everything one accepted draw touches is contiguous, so the accept kernel can
  // issue its fp32 atomics "transposed" (consecutive lanes = consecutive floats of one record): the
  // memory side executes atomics per 64-B request, and a draw then costs 1-3 requests instead of 5-37.
  float *acc;        // [np][stride]
  // What scan_dma_kernel adds up for a pixel out of its own visits (same record layout): stored whole, never read
  // by the scan and never touched by a splat, so that the accept kernel's atomics on `acc` may run while the scan
  // still does.  The resolve adds the two; anything else that looks at the accumulators folds `dir` into `acc`
  // first (fold_direct_kernel).  Null while it holds nothing.
  float *dir;
  // One byte per 64 pixel records of `acc`, set by the accept kernel for every record a draw is added to.  Non-null
  // only while nothing but splats has been added since the last clear (a pass whose scan stores to `dir`): the
  // resolve then reads `acc` only where something was splatted -- 2 % of the frame in the scan-dominated regime --
  // and the next clear wipes only that.
  uint8_t *touched;
  // ... and one byte per pixel record, set with the group's (same values; non-null together with `touched`): behind the pass's
  // whole-frame resolve only the pixels a draw reached are resolved again, and the next clear wipes only those -- a petzval
  // frame's draws reach every 64-pixel group and an eighth of the pixels (round 6: resolve_touched_kernel 386 -> see DESIGN 4.2c).
  uint8_t *touched_px;
  uint32_t stride;   // floats per record
  // closest-filter AOVs (src/lentil.h:832-837): per pixel the winning candidate as one 64-bit key,
  // (bits of |Z|) << 32 | (0xFFFFFFFF - visit), reduced with atomicMin: smallest depth wins, equal depths
  // go to the later visit -- exactly what the reference's sequential "depth <= zbuffer" test leaves.
  unsigned long long *zkey;   // [np], null when the frame has no closest AOV
  unsigned long long *zkey_dbg;   // [np], lentil_debug's own z-buffer (src/lentil.h:838-845): redistributed draws only
  uint32_t debug_mask;        // bit k: AOV k is lentil_debug (also set in closest_mask)
  uint32_t n_aovs;
  uint32_t closest_mask;      // bit k: AOV k is closest-filtered (never bit 0)
  uint64_t np;       // xres*yres
  LD_DEV float4 *aov(uint64_t p, uint32_t a) const { return reinterpret_cast<float4 *>(acc + p * stride + 4u * a); }
  LD_DEV float *wt(uint64_t p) const { return acc + p * stride + 4u * n_aovs; }
};

LD_DEV unsigned long long closest_key(float depth, uint32_t visit) {
  return ((unsigned long long)__float_as_uint(fabsf(depth)) << 32) | (unsigned long long)(0xFFFFFFFFu - visit);
}
// The key of a candidate (a pixel's own visit in a scan kernel, an item in an accept kernel).  The reference's z-buffer
// (src/lentil.h:832-837: `abs(depth) <= zbuffer || zbuffer == 0`) uses 0 as "empty": in its sequential order a candidate at
// |Z| == 0 wins and at once re-opens the pixel for whatever comes next, and a NaN written into an empty pixel is never
// replaced -- outcomes that depend on the order of the candidates at a pixel, which a reduction over keys cannot give.  Such
// a candidate is therefore flagged, and lentil_hip_redistribute refuses the pass (LENTIL_ERR_UNSUPPORTED) instead of
// returning a frame that differs from the reference's: never a silently different image.  (A renderer's Z is neither for
// anything in front of the camera; samples that hit nothing carry AI_INFINITE, which the keys order like the reference.)
LD_DEV unsigned long long closest_key_of(DevCounters *ctr, float depth, uint32_t visit) {
  if (!(fabsf(depth) > 0.0f)) ctr->degenerate_depth = 1u;
  return closest_key(depth, visit);
}

struct ItemHdr {          // written by prep_items_kernel / publish_item; a 128-byte line of its own (streamed pass: never
                          // in an L2 before it is written)
  double tx, ty, tz;      // PO: -P_cs * 10 (src/lentil_filter.cpp:271); thin lens: P_cs (floats, exactly)
  uint32_t seed_a;        // (unsigned)(px*py+px)
  int32_t px_py;          // px | py << 16
  // DrawArgs::item_ready (streamed pass, lean tail): solves of this item parked for solve_slow_kernel so far / finished by it.
  // Zeroed by publish_item with the header; the solve kernel counts a solve when it parks it, the straggler wave when its result is
  // in the record; accept_kernel<3> -- behind the solve kernel, beside the stragglers -- takes an item whose two counts agree and
  // leaves the others to the accept behind the stragglers.
  uint32_t parked, parked_done;
  // DrawArgs::early_accept: results the item's first batch holds (one per try and channel) / results delivered so far.  One
  // 8-byte word: the solve wave that delivers results adds to the high half with ONE returning atomic and reads the low half
  // from what it returns; the wave whose add makes the two equal pushes the item onto the ready queue.
  uint32_t issued, delivered;
  uint32_t pad[20];
};
static_assert(sizeof(ItemHdr) == 128, "ItemHdr is a 128-byte line");

struct ItemProg {         // 48 B, progress of an item across rounds
  uint32_t n_done;        // attempts resolved so far (all earlier attempts are final)
  uint32_t accepted;
  uint32_t m_lo, m_hi;    // R(m) of the current round covers [m_lo, m_hi); any R(m < m_lo) still needed is FAIL ...
  uint32_t res_off;       // offset of R(m_lo) in the current round's result pool
  uint32_t last_ok;       // highest accepted attempt index
  uint32_t splats;        // chromatic mode: accepted (attempt, channel) pairs so far
  // ... unless it lies in [p_lo, p_hi): the batch before, at p_off in the OTHER parity's pool.  Only an item whose first
  // accept ran while some of its solves were still parked (uacc > 0 of them were met; accept_item<1>) carries one: the
  // accept behind it (accept_item<2>) walks the item again from attempt 0, knows from n_end1 and the pending marks
  // which draws the first one has added already, and adds the rest.
  uint32_t uacc;
  uint32_t p_lo, p_hi, p_off;
  uint32_t n_end1;        // attempts the first accept looked at
};

// Streamed pass, "extension" (round 4).  An item near the frame's edge loses attempts to draws that land outside the frame
// and needs more than its first batch -- 33 of the headline frame's 1 168 items, 3 % more solves -- and the round that
// used to serve them (first accept -> tasks -> solves -> their stragglers -> second accept) was 0.4 ms of latency at the
// end of a 2 ms pass.  Now the first round's solve kernel looks after them itself: every result it delivers is counted per
// item (one returning atomic per item and wave flush), and the wave that delivers the LAST result of an item's batch
// compares the successes with what the item needs; if they fall short it appends the next batch's tasks to the very queue
// it is working on -- sized like accept_next_batch would, from the item's own success rate -- and the item stays open.
// The queue's end markers are written when the last publisher has signed off AND no item is open (whoever sees both).
// The result pool holds 5 x samples + retries slots per item from the start, so an item's batches stay contiguous; the
// first accept takes the batch's end from here.  What it then finds missing (an estimate that was too kind) still goes
// the old way.  One 32-byte record per item, written through (another CU's waves read it while the kernel runs).
// MEASURED (round 4, headline frame) AND OFF BY DEFAULT (LENTIL_EXTEND=1 switches it on): draw lists stay bit-identical, the
// second round disappears (811 appended tasks, no item short in the accept) -- and the pass takes 2.30 ms instead of 2.00.
// The second round was never idle time: the item found last by the scan needs its first batch (0.15 ms), the batch
// behind it (0.15 ms) and that batch's slowest solves (100 Newton iterations: 0.25-0.35 ms on a straggler wave) one after
// the other whoever schedules them, and the old layout runs the first accept of the other 1 100 items BESIDE that chain,
// this one behind it.
// The appended batches have a queue of their own (DrawArgs::ext_q), served by the solve kernel's first `ext_keeper_blocks`
// blocks once the main queue has ended: the other blocks leave as they always did -- the straggler kernel's waves are only
// placed when solve waves leave -- and the main queue's end does not wait for the items near the frame's edge.
struct ItemLive {
  unsigned long long cnt;        // results delivered so far: count (bits 0-20), pixels (21-41), outside the frame (42-62)
  unsigned long long hi_s;       // R(m) issued so far: [0, m_hi) (low word), samples (high word)
  uint32_t res_off, pad[3];
};
static_assert(sizeof(ItemLive) == 32, "ItemLive is 32 bytes");

struct Task {             // up to 64 consecutive m of one item (and one wavelength channel)
  uint32_t item, m_base, res_off, count;   // count: bits 0-7 number of m, bits 8-9 channel
};


// ---------------------------------------------------------------------------------------
// Streamed pass: the scan kernels publish every item they find -- header, progress record and the solve
// tasks of its first batch -- themselves, while they run; persistent waves of solve_po_kernel<.., kStream>
// draw tickets on the task queue and poll their slot.  Hand-off between CUs without fences
// (MI355X_MICROARCH.md, "Valid forms", 8-byte granules): every handed-off byte is stored write-through (agent-scope
// relaxed store = global_store sc1) and waited for (vmcnt(0)) before the word that publishes it, and read with a
// returning atomic (ld_coherent64).  A slot is valid when its tag equals this pass's epoch, so the queues are never
// cleared; the end of a queue is a run of end markers behind its last entry, one for every wave that may hold a
// ticket: nobody polls a shared word.
// ---------------------------------------------------------------------------------------
struct StreamPub {
  uint32_t epoch;              // 22 bits, never 0
  uint32_t n_channels;
  int32_t retries;
  uint32_t extra_num, extra_const;
  uint32_t item_cap, task_cap;
  uint64_t pool_cap;
  ItemHdr *hdr;
  ItemProg *prog;
  uint32_t *active0;
  Task *tasks0;
  ItemLive *live;              // non-null: extension (ItemLive)
  Task *ext_q;                 // ... its task queue and the waves that serve it (end markers)
  uint32_t ext_keepers;
  BatchModelDev model;         // land non-null: first batches sized from the lens and the frame (lentil_batch_model.h)
};

// Hand-off words are written with atomics as well (exchange, nothing returned): "8-byte agent-scope atomics on both
// sides" is a form MI355X_MICROARCH.md lists as valid; a write-through store on one side and an atomic on the
// other is not, and was seen to lose a word once in a few hundred passes (an end marker that its reader, polling
// with atomics, never saw).
LD_DEV void st_agent64(void *p, uint64_t v) {
  (void)__hip_atomic_exchange(reinterpret_cast<uint64_t *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
LD_DEV void st_agent32(void *p, uint32_t v) {
  (void)__hip_atomic_exchange(reinterpret_cast<uint32_t *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Reads of words another CU writes while this kernel runs are returning atomics (x | 0), not loads: an agent-scope
// load (sc1) goes past this CU's L1 but is served by this XCD's L2, which may hold the line from before the other
// XCD wrote it through -- measured: end-of-queue words polled with sc1 loads were seen 0.4-0.9 ms late once the
// scan's traffic no longer swept the L2s.  Atomics execute at the memory side (MI355X_MICROARCH.md).
// A compare-and-swap whose comparand never occurs (all ones: no record, header word or counter is that): it
// returns the word and never writes.  (Not `x | 0`: the optimiser turns that into an atomic load, volatile or not --
// and as an instruction it writes the old value back.)
LD_DEV uint64_t ld_coherent64(const void *p) {
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  uint64_t old;
  const u64x2 swap_cmp = {~0ull, ~0ull};
  asm volatile("global_atomic_cmpswap_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(p), "v"(swap_cmp) : "memory");
  return old;
}
LD_DEV uint32_t ld_coherent32(const void *p) {
  uint32_t old;
  const uint64_t swap_cmp = ~0ull;
  asm volatile("global_atomic_cmpswap %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(p), "v"(swap_cmp) : "memory");
  return old;
}
// Do two streams run kernels at the same time?  (lentil_hip_create, pick_concurrent_streams: streams that share one of
// the runtime's hardware queues run their kernels one after the other.)  The waiting kernel gives up after 2 ms.
__global__ void probe_wait_kernel(uint32_t *flag, uint32_t *seen) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (ld_coherent32(flag) == 0u) {
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) { *seen = 0u; return; }
    __builtin_amdgcn_s_sleep(32);
  }
  *seen = 1u;
}
// The pass's counters straight into the host's (pinned, device-visible) copy, then a sequence number behind them: the host
// reads them as soon as they have crossed the bus, without a copy command's completion signal and the wake-up behind it
// (redistribute_streamed; ~40 us of a 2 ms pass).  One block.
__global__ __launch_bounds__(256) void report_counters_kernel(const uint32_t *src, uint32_t *host_dst, uint32_t n_words,
                                                              uint32_t *host_seq, uint32_t seq) {
  for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x)
    __hip_atomic_store(host_dst + i, __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// LENTIL_A_FIRST=1 (experiment): ahead of a streamed pass's scan on its stream -- the scan's blocks are dispatched when the
// resident solve kernel's waves have their registers (waves_started), so that those lie in one piece at the bottom of
// every SIMD's file and what the scan's waves give back is one piece too (a third solve block fits it, LENTIL_SOLVE_B).
__global__ void wait_waves_kernel(const DevCounters *ctr, uint32_t want, uint64_t max_ticks) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (ld_coherent32(&ctr->waves_started[0]) < want && __builtin_amdgcn_s_memrealtime() - t0 < max_ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ void probe_set_kernel(uint32_t *flag) { (void)__hip_atomic_exchange(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr uint32_t kEndCount = 0xFFu;       // Task::count of the end-of-queue markers behind the last task
constexpr uint32_t kEndRange = 0x3FFu;      // ... and the count field of the end markers of the range queue
// every wait on a queue slot is bounded: 250 ms of the 100 MHz real-time counter, then DevCounters::stuck (the
// host redoes the draws of the pass the chunked way)
constexpr uint64_t kStuckTicks = 25000000ull;
constexpr uint32_t kTaskTagShift = 10;    // Task::count bits 10..31: epoch of the pass that published the slot

// get_coc_thinlens(P, cz) < 0.4f (src/lentil_filter.cpp:185-190), decided from cz where that is certain: the exact circle of
// confusion is |A| |c0 - c1 / cz|, below 0.4 (1 - eps) on the `in` intervals and above 0.4 (1 + eps) outside the `out`
// intervals (closed; an empty one has lo > hi); in between -- and where the host found nothing usable -- the function decides.
struct ScanBands {
  float in_lo[2], in_hi[2];
  float out_lo[2], out_hi[2];
};

struct ScanArgs {
  lentil_params P;
  double lens_length;
  VisitsDev V;
  FrameDev F;
  uint2 *work;       // (visit, samples) per item
  uint64_t work_cap;
  DevCounters *ctr;
  uint32_t ppt;      // pixels per wave tile (uniform mode)
  uint32_t tv_pad;   // staging entries per wave (>= ppt * visits_per_pixel)
  uint64_t tile_begin, tile_end;   // uniform mode: this launch's range of pixel tiles
  uint64_t v_begin, v_end;         // ragged mode: this launch's range of visits
  // streamed pass: every flush of a wave queue is announced to publish_kernel as a (first item, count) range
  uint64_t *ranges;                // null: not a streamed pass
  uint32_t range_cap, epoch;
  uint32_t end_ranges;             // waves of publish_kernel: end markers behind the last range
  uint32_t flush_each_tile;        // announce at the end of every tile (few items per pass: latency matters, atomics do not)
  uint32_t ring;                   // scan_dma_multi_kernel: slots per wave
  uint32_t outside_in;             // scan_dma*_kernel: tiles from both ends of the range inwards (scan_order)
  float4 *dummy;                   // scan_dma_multi_kernel: 64 x 16 B that lanes without a record store to
  ScanBands bands;                 // scan_dma2_kernel: where the circle-of-confusion test is decided by the depth alone
  uint32_t ppr_magic, ppr_shift;   // scan_dma2_kernel: pixel / pixels_per_row = __umulhi(pixel, ppr_magic) >> ppr_shift
  uint32_t skip_blocks;            // scan_dma2_kernel: blocks at the front of the grid that leave at once (see there)
};

LD_DEV void visit_pixel(const VisitsDev &V, uint64_t v, int &px, int &py) {
  if (V.visits_per_pixel) {
    const uint64_t p = v / V.visits_per_pixel;
    px = V.pixel_x0 + (int)(p % V.pixels_per_row);
    py = V.pixel_y0 + (int)(p / V.pixels_per_row) * (int)V.pixel_row_stride;
  } else {
    const uint32_t q = V.pixel[v];
    px = (int)(q & 0xFFFFu);
    py = (int)(q >> 16);
  }
}

LD_DEV float4 nt_load(const float4 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

LD_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Development aid (-DLENTIL_TIMELINE, tools/timeline.py; never in the shipped build): what the kernels of a pass do
// when -- events counted into 20 us buckets of the chip-wide 100 MHz counter.  One lane of a wave calls tl_add.
#ifdef LENTIL_TIMELINE
constexpr int kTlChannels = 10, kTlBuckets = 1024, kTlSub = 64;
// (every bucket exists kTlSub times, 40 KB apart, chosen by block: thousands of atomics per microsecond on ONE word
// throttle the kernels that issue them -- the first version of this slowed the scan from 1.1 to 4 ms)
__device__ unsigned int g_timeline[kTlSub][kTlChannels][kTlBuckets];
LD_DEV void tl_add(int ch, uint32_t n) {
  if (n) atomicAdd(&g_timeline[(blockIdx.x * 4u + (threadIdx.x >> 6)) & (kTlSub - 1)][ch][(uint32_t)(__builtin_amdgcn_s_memrealtime() >> 11) & (kTlBuckets - 1)], n);
}
__device__ unsigned long long g_dbg[32];       // plain event counts (tools/timeline.py prints them)
LD_DEV void dbg_add(int i, unsigned long long n) { atomicAdd(&g_dbg[i], n); }
// ... and when every kernel of the pass begins and ends: first block in, last block out, in ticks of the same counter
__device__ unsigned long long g_span[32][2];
struct TlSpan {
  int id;
  LD_DEV explicit TlSpan(int i) : id(i) {
    if (threadIdx.x == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      atomicMin(&g_span[id][0], t);
    }
  }
  LD_DEV ~TlSpan() { if (threadIdx.x == 0) atomicMax(&g_span[id][1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); }
};
#define LENTIL_TL_SPAN(ID) TlSpan tl_span_guard_(ID)
#else
LD_DEV void tl_add(int, uint32_t) {}
LD_DEV void dbg_add(int, unsigned long long) {}
#define LENTIL_TL_SPAN(ID) do {} while (0)
#endif
enum { SPAN_SCAN = 0, SPAN_PUBLISH, SPAN_SOLVE_R0, SPAN_SOLVE_R1, SPAN_SOLVE_R2, SPAN_SLOW_R0, SPAN_SLOW_R1, SPAN_SLOW_R2,
       SPAN_ACCEPT0, SPAN_ACCEPT1, SPAN_ACCEPT2, SPAN_RESOLVE, SPAN_RESOLVE_TOUCHED, SPAN_CLEAR, SPAN_RESET, SPAN_PREP };
enum { TL_SCAN_TILES = 0, TL_TASKS_PUBLISHED, TL_ITERS_A, TL_ITERS_B, TL_ITERS_LATER, TL_ITERS_SLOW, TL_ITEMS_ACCEPTED,
       TL_POLLS_EMPTY, TL_TASKS_TAKEN, TL_PARKED };

// lentil_debug (src/lentil_filter.cpp:209-212): the AOV's value, samples * redistribute, is taken BEFORE the "sample can't be
// inside of lens" test clears redistribute (:240).  A visit that fails only that test therefore reaches the direct path
// (src/lentil.h:938-955) with a non-zero value and competes for lentil_debug's own z-buffer (src/lentil.h:838-845) at its own
// pixel, exactly like an accepted draw does.  True for such a visit: the decision with the lens test left out (lens length 0).
template <class RaydirLoad>
LD_DEV bool visit_feeds_debug_directly(const lentil_params &P, float4 pos_z, float4 volume_ignore, float4 transmission,
                                       float inv_density, RaydirLoad load_raydir, const CamMotion &cm) {
  return visit_redistributes(P, 0.0, pos_z, volume_ignore, transmission, inv_density, load_raydir, cm);
}

struct ItemVisit {
  uint32_t visit, samples;
  int px, py;
  VisitInfo I;
  float4 rgba;
  float w;
};

// everything the draw kernels need to know about a work-list entry (visit, samples)
LD_DEV ItemVisit load_work_visit(const lentil_params &P, const VisitsDev &V, uint2 wi, double lens_length) {
  ItemVisit h;
  h.visit = wi.x;
  h.samples = wi.y;
  const uint32_t v = h.visit;
  h.rgba = V.rgba[v];
  const float invd = V.inv_density ? V.inv_density[v] : P.inverse_sample_density;
  h.I = visit_prologue(P, lens_length, h.rgba, V.pos_z[v], V.raydir_time[v], V.volume_ignore[v], V.transmission[v], invd, V.cam);
  visit_pixel(V, v, h.px, h.py);
  const float inv_samples = (float)(1.0 / (double)(float)(int)h.samples);
  h.w = 1.0f * invd * inv_samples;              // src/lentil_filter.cpp:297
  return h;
}

LD_DEV ItemHdr make_item_hdr(const lentil_params &P, const float cs[3], int px, int py) {
  ItemHdr hd{};
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS) {
    hd.tx = -(double)cs[0] * 10.0; hd.ty = -(double)cs[1] * 10.0; hd.tz = -(double)cs[2] * 10.0;
  } else {
    hd.tx = (double)cs[0]; hd.ty = (double)cs[1]; hd.tz = (double)cs[2];
  }
  hd.seed_a = (uint32_t)(px * py + px);
  hd.px_py = (px & 0xFFFF) | (py << 16);
  return hd;
}

// first batch of an item: R(0 .. samples - 1 + retries), plus the over-provisioned share
LD_DEV uint32_t first_batch_hi(uint32_t samples, uint32_t retries, uint32_t extra_num, uint32_t extra_const) {
  const uint32_t m_limit = samples * 5u + retries;
  uint32_t m_hi = samples + retries + (uint32_t)(((unsigned long long)samples * extra_num) >> 8) + extra_const;
  return m_hi > m_limit ? m_limit : m_hi;
}

// Streamed pass: one lane publishes one item (work-list entry `item`): header, progress record, result space and
// the tasks of its first batch.  An item that does not fit (item buffers, result pool, task queue: sized from the
// previous pass) raises DevCounters::fallback -- the host then redoes the draws of the pass with exact sizes -- and
// still fills the task slots it reserved (with empty tasks), so that no ticket waits for a slot that never comes.
LD_DEV void publish_item(const lentil_params &P, const VisitsDev &V, const StreamPub &S, DevCounters *ctr, uint32_t item, uint2 wi,
                         uint32_t count, const float cs[3]) {
  const uint32_t retries = (uint32_t)S.retries, nch = S.n_channels;
  const uint32_t nt = (count + 63u) / 64u;
  // (extension: room for every R(m) the item can ever ask for, so that later batches lie behind the first)
  const uint32_t reserve = S.live ? wi.y * 5u + retries : count;
  const unsigned long long off = atomicAdd(&ctr->pool_used[0], (unsigned long long)reserve * nch);
  const uint32_t tb = atomicAdd(&ctr->n_tasks[0], nt * nch);
  const bool ok = item < S.item_cap && off + (unsigned long long)reserve * nch <= S.pool_cap &&
                  (unsigned long long)tb + nt * nch <= S.task_cap;
  // (which bound it was, for lentil_hip_last_redo_note: 1 items, 2 result pool, 4 task queue)
  if (!ok) atomicOr(&ctr->fallback, (item < S.item_cap ? 0ull : 1ull) |
                                    (off + (unsigned long long)reserve * nch <= S.pool_cap ? 0ull : 2ull) |
                                    ((unsigned long long)tb + nt * nch <= S.task_cap ? 0ull : 4ull));
  if (item < S.item_cap) {
    const uint32_t v = wi.x;
    int px, py;
    visit_pixel(V, v, px, py);
    const ItemHdr hd = make_item_hdr(P, cs, px, py);
    uint64_t *d = reinterpret_cast<uint64_t *>(S.hdr + item);
    st_agent64(d + 0, (uint64_t)__double_as_longlong(hd.tx));
    st_agent64(d + 1, (uint64_t)__double_as_longlong(hd.ty));
    st_agent64(d + 2, (uint64_t)__double_as_longlong(hd.tz));
    st_agent64(d + 3, (uint64_t)hd.seed_a | ((uint64_t)(uint32_t)hd.px_py << 32));
    st_agent64(d + 4, 0ull);          // ItemHdr::parked / parked_done
    st_agent64(d + 5, ok ? (uint64_t)count * nch : 0ull);      // ItemHdr::issued / delivered
    ItemProg pg{};
    pg.m_lo = 0;
    pg.m_hi = ok ? count : 0u;
    pg.res_off = (uint32_t)off;
    S.prog[item] = pg;
    S.active0[item] = item;
    if (S.live && ok) {
      ItemLive *L = S.live + item;
      st_agent64(&L->cnt, 0ull);
      st_agent64(&L->hi_s, (uint64_t)count | ((uint64_t)(wi.y & 0xFFFFu) << 32));
      st_agent32(&L->res_off, (uint32_t)off);
      atomicAdd(&ctr->items_open, 1u);
    }
  }
  // the header must have arrived before a task that names the item can be seen
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (uint32_t i = 0; i < nt * nch; ++i) {
    const uint32_t slot = tb + i, t = i % nt;
    if (slot < S.task_cap) st_agent64(S.tasks0 + slot, (uint64_t)item | ((uint64_t)(t * 64u) << 32));
  }
  // ... and a slot's first half before the half that carries its tag
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (uint32_t i = 0; i < nt * nch; ++i) {
    const uint32_t slot = tb + i, c = i / nt, t = i - c * nt;
    if (slot >= S.task_cap) continue;
    const uint32_t n = ok ? ((count - t * 64u) < 64u ? (count - t * 64u) : 64u) : 0u;
    st_agent64(reinterpret_cast<uint64_t *>(S.tasks0 + slot) + 1,
               (uint64_t)((uint32_t)off + c * count + t * 64u) |
                   ((uint64_t)(n | (c << 8) | (S.epoch << kTaskTagShift)) << 32));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tl_add(TL_TASKS_PUBLISHED, nt * nch);
}

// K2: wave-ballot + prefix-sum compaction of flagged lanes into the work list.  Each wave collects
// its items in a private 128-entry LDS queue (slot = popcount of the ballot below the lane) and
// reserves space in the global list with ONE returning atomic per flush (queue more than half full,
// or end of the wave) -- a single hot counter only sustains ~90 returning atomics per microsecond.
constexpr uint32_t kWaveQueue = 128;
// scan_uniform_multi_kernel: float4 slots per AOV plane of a wave's staging area.  65, not 64: the summing lanes
// (pixel, aov) read plane aov at the same offset -- a stride of 64 float4 put all planes on the same LDS banks
constexpr uint32_t kMultiPlane = 65;

// (streamed pass: what a flush added to the work list is published at the end of the tile, where nothing of the
// tile is live any more -- up to kWavePending flushes are remembered behind the queue, in the same LDS block)
constexpr uint32_t kWavePending = 16;
constexpr uint32_t kWaveQueueLds = kWaveQueue + kWavePending;     // uint2 entries per wave

struct WaveQueue {
  uint2 *q;                 // wave-private LDS, kWaveQueueLds entries
  uint32_t n;               // wave-uniform fill
  uint32_t n_pend;          // wave-uniform: flushed (base, count) ranges not published yet
  unsigned long long sum_samples, n_items;   // per-lane partial sums, reduced in finish()

  LD_DEV void init(uint2 *lds) { q = lds; n = 0; n_pend = 0; sum_samples = 0; n_items = 0; }

  LD_DEV void flush(const ScanArgs &a) {
    if (n == 0) return;
    const uint32_t lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0) {
      base = atomicAdd(&a.ctr->work_count, (unsigned long long)n);
      if (a.ranges) {
        atomicAdd(&a.ctr->n_active[0], n);
        if (n_pend < kWavePending) q[kWaveQueue + n_pend] = make_uint2((uint32_t)base, n);
        else atomicOr(&a.ctr->fallback, 8ull);   // (a flush holds more than 64 entries, a tile fewer than 16 x 64 visits)
      }
    }
    if (a.ranges && n_pend < kWavePending) ++n_pend;
    base = __shfl(base, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    for (uint32_t i = lane; i < n; i += 64u) {
      if (base + i < a.work_cap) {
        // streamed pass: read by publish_kernel on another CU while this kernel runs -- written through
        if (a.ranges) st_agent64(a.work + base + i, (uint64_t)q[i].x | ((uint64_t)q[i].y << 32));
        else a.work[base + i] = q[i];
      } else {
        atomicAdd(&a.ctr->overflow, 1ull);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    n = 0;
  }

  LD_DEV void push(bool flagged, uint32_t visit, uint32_t samples, const ScanArgs &a) {
    const unsigned long long mask = __ballot(flagged);
    if (mask == 0ull) return;
    const uint32_t lane = lane_id();
    if (flagged) {
      q[n + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = make_uint2(visit, samples);
      sum_samples += samples;
      n_items += 1;
    }
    n += (uint32_t)__builtin_popcountll(mask);
    if (n > kWaveQueue - 64u) flush(a);
  }

  // announce the flushed ranges: one 8-byte record each, (first item | count << 32 | epoch tag << 42)
  LD_DEV void publish_pending(const ScanArgs &a) {
    if (n_pend == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's work-list stores have arrived
    const uint32_t lane = lane_id();
    uint32_t k = 0;
    if (lane == 0) k = atomicAdd(&a.ctr->n_ranges, n_pend);
    k = __shfl(k, 0);
    if (lane < n_pend) {
      const uint2 pr = q[kWaveQueue + lane];
      if (k + lane < a.range_cap)
        st_agent64(a.ranges + k + lane, (uint64_t)pr.x | ((uint64_t)(pr.y | (a.epoch << kTaskTagShift)) << 32));
      else
        atomicOr(&a.ctr->fallback, 16ull);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    n_pend = 0;
  }

  // streamed pass: what a tile found goes out when the tile is done, not when the wave is (few items per pass:
  // latency matters, the extra atomics do not)
  LD_DEV void end_tile(const ScanArgs &a) {
    if (!a.ranges) return;
    if (a.flush_each_tile) flush(a);
    publish_pending(a);
  }

  LD_DEV void finish(const ScanArgs &a) {
    flush(a);
    if (a.ranges) publish_pending(a);
    for (int off = 32; off > 0; off >>= 1) {
      sum_samples += __shfl_down(sum_samples, off);
      n_items += __shfl_down(n_items, off);
    }
    if (lane_id() == 0 && n_items) {
      atomicAdd(&a.ctr->sum_samples, sum_samples);
      atomicAdd(&a.ctr->redistributed, n_items);
    }
  }
};

// streamed pass: the last thing a scan block does -- after every wave's items and tasks have arrived
LD_DEV void scan_block_done(const ScanArgs &a) {
  if (!a.ranges) return;
  __shared__ uint32_t s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(&a.ctr->scan_blocks_done, 1u) == gridDim.x - 1u ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // the last block to sign off: the range queue is complete.  One end marker for every publisher wave.
  const uint32_t n = ld_coherent32(&a.ctr->n_ranges);
  for (uint32_t i = threadIdx.x; i < a.end_ranges; i += blockDim.x)
    if ((uint64_t)n + i < a.range_cap)
      st_agent64(a.ranges + n + i, (uint64_t)(kEndRange | (a.epoch << kTaskTagShift)) << 32);
}

// ---------------------------------------------------------------------------------------
// K1+K2+K6, uniform footprints.  One wave owns a tile of `ppt` consecutive source pixels
// (= ppt*M consecutive visits, read as fully coalesced 1 KiB column loads), stages the weighted
// contributions in wave-private LDS, then lane p adds up pixel p's M entries in iterator order --
// the order the reference accumulates them (filter_and_add_to_buffer_new, src/lentil.h:938-955).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_uniform_kernel(ScanArgs a) {
  extern __shared__ float4 smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t waves_per_block = blockDim.x >> 6;
  float4 *sval = smem + (size_t)wave * a.tv_pad;
  float *sw = reinterpret_cast<float *>(smem + (size_t)waves_per_block * a.tv_pad) + (size_t)wave * a.tv_pad;
  // wave queues live behind the staging area: [4 x tv_pad float4][4 x tv_pad float][4 x kWaveQueue uint2]
  uint2 *qmem = reinterpret_cast<uint2 *>(reinterpret_cast<float *>(smem + (size_t)waves_per_block * a.tv_pad) +
                                          (size_t)waves_per_block * a.tv_pad);
  WaveQueue wq;
  wq.init(qmem + (size_t)wave * kWaveQueueLds);

  const VisitsDev &V = a.V;
  const uint32_t M = V.visits_per_pixel;
  const uint32_t ppt = a.ppt;
  const uint32_t TV = ppt * M;
  const uint64_t n_pixels = (V.n + M - 1) / M;
  const uint64_t n_tiles = a.tile_end;
  const uint64_t wave_global = a.tile_begin + (uint64_t)blockIdx.x * waves_per_block + wave;
  const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;
  const uint32_t xres = a.P.xres;

  for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_stride) {
    const uint64_t pix0 = tile * ppt;
    const uint64_t v0 = pix0 * M;
    // this lane's pixel record: requested now, used after the tile's visits have been staged
    const uint64_t pix = pix0 + lane;
    const bool own = (lane < ppt) && (pix < n_pixels);
    uint64_t lin = 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float ws = 0.f;
    if (own) {
      const int px = V.pixel_x0 + (int)(pix % V.pixels_per_row);
      const int py = V.pixel_y0 + (int)(pix / V.pixels_per_row) * (int)V.pixel_row_stride;
      lin = (uint64_t)px + (uint64_t)py * xres;
      s = *a.F.aov(lin, 0);
      ws = *a.F.wt(lin);
    }
    // two groups of 64 visits per step: eight 1 KiB column loads in flight per wave before the first use
    for (uint32_t eb = 0; eb < TV; eb += 128) {
      bool valid[2], flagged[2];
      float4 rgba[2], pz[2], vi[2], tr[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t e = eb + 64u * g + lane;
        const uint64_t v = v0 + e;
        valid[g] = (e < TV) && (v < V.n);
        const uint64_t vl = valid[g] ? v : v0;      // lanes past the end re-read the tile's first visit (unused)
        // read once: nontemporal, so that the stream does not push the pixel records and work lists out of L2
        rgba[g] = nt_load(V.rgba + vl); pz[g] = nt_load(V.pos_z + vl); vi[g] = nt_load(V.volume_ignore + vl);
        tr[g] = nt_load(V.transmission + vl);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t e = eb + 64u * g + lane;
        const uint64_t v = v0 + e;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        float w = 0.f;
        int samples = 0;
        flagged[g] = false;
        if (valid[g]) {
          const float invd = V.inv_density ? V.inv_density[v] : a.P.inverse_sample_density;
          flagged[g] = visit_redistributes(a.P, a.lens_length, pz[g], vi[g], tr[g], invd, [&]() { return V.raydir_time[v]; }, V.cam);
          if (flagged[g]) {
            // a few visits in 10^5: the draw count (same function as the draw kernels use)
            samples = visit_prologue(a.P, a.lens_length, rgba[g], pz[g], V.raydir_time[v], vi[g], tr[g], invd, V.cam).samples;
          } else {
            w = 1.0f * invd;                              // filter_weight * inv_density, lentil.h:949-953
            val = make_float4((rgba[g].x + 0.0f) * w, (rgba[g].y + 0.0f) * w, (rgba[g].z + 0.0f) * w, (rgba[g].w + 0.0f) * w);
          }
        }
        wq.push(flagged[g], (uint32_t)v, (uint32_t)samples, a);
        if (e < TV) { sval[e] = val; sw[e] = w; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

    if (own) {
      // a visit that was redistributed left zeros here; adding +0 changes nothing (an accumulator is never -0)
      for (uint32_t j = 0; j < M; ++j) {
        const float4 c = sval[lane * M + j];
        const float cw = sw[lane * M + j];
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; ws += cw;
      }
      *a.F.aov(lin, 0) = s;
      *a.F.wt(lin) = ws;
      if (a.F.zkey) {
        // closest AOVs: this pixel's own (non-redistributed) visits compete with their depth
        unsigned long long kmin = ~0ull;
        for (uint32_t j = 0; j < M; ++j) {
          if (sw[lane * M + j] != 0.0f) {
            const uint64_t vv = v0 + (uint64_t)lane * M + j;
            const unsigned long long key = closest_key_of(a.ctr, V.pos_z[vv].w, visit_gid(V, (uint32_t)vv));
            if (key < kmin) kmin = key;
          }
        }
        if (kmin != ~0ull) atomicMin(a.F.zkey + lin, kmin);
      }
      if (a.F.zkey_dbg) {
        // lentil_debug: own visits that failed nothing but the inside-the-lens test (visit_feeds_debug_directly)
        unsigned long long kmin = ~0ull;
        for (uint32_t j = 0; j < M; ++j) {
          if (sw[lane * M + j] != 0.0f) {
            const uint64_t vv = v0 + (uint64_t)lane * M + j;
            const float invd = V.inv_density ? V.inv_density[vv] : a.P.inverse_sample_density;
            if (visit_feeds_debug_directly(a.P, V.pos_z[vv], V.volume_ignore[vv], V.transmission[vv], invd, [&]() { return V.raydir_time[vv]; }, V.cam)) {
              const unsigned long long key = closest_key_of(a.ctr, V.pos_z[vv].w, visit_gid(V, (uint32_t)vv));
              if (key < kmin) kmin = key;
            }
          }
        }
        if (kmin != ~0ull) atomicMin(a.F.zkey_dbg + lin, kmin);
      }
    }
    // extra AOVs: same weights, one column at a time through the same staging area
    for (uint32_t k = 0; k < V.n_extra; ++k) {
      if (a.F.closest_mask & (2u << k)) continue;       // closest AOVs are gathered from the winners later
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      // four independent 16 B loads in flight per lane (one column has no other source of memory parallelism)
      const float4 *col = V.extra[k];
      for (uint32_t eb = 0; eb < TV; eb += 256) {
        float4 c[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t e = eb + 64u * u + lane;
          const uint64_t v = v0 + e;
          w[u] = (e < TV) ? sw[e] : 0.0f;
          c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < TV && v < V.n && w[u] != 0.0f) c[u] = col[v];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t e = eb + 64u * u + lane;
          if (e < TV)
            sval[e] = (w[u] != 0.0f) ? make_float4((c[u].x + 0.0f) * w[u], (c[u].y + 0.0f) * w[u],
                                                   (c[u].z + 0.0f) * w[u], (c[u].w + 0.0f) * w[u])
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      if (own) {
        float4 *dst = a.F.aov(lin, k + 1);
        float4 s = *dst;
        for (uint32_t j = 0; j < M; ++j) {
          if (sw[lane * M + j] != 0.0f) {
            const float4 c = sval[lane * M + j];
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
          }
        }
        *dst = s;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    wq.end_tile(a);
  }
  wq.finish(a);
  scan_block_done(a);
}

// ---------------------------------------------------------------------------------------
// K1+K2+K6 for beauty-only frames with a uniform footprint -- the LDS-DMA form.  The register-staged kernel
// above keeps its bytes in flight in VGPRs (104 of them, three waves per SIMD): beside it a SIMD holds ONE wave of
// the solve kernel (168), which alone runs at 60 % of what two or three reach.  Here the columns go from HBM
// straight into LDS (global_load_lds_dwordx4, 1 KiB per instruction, no destination registers): two waves per SIMD
// with up to 18 KiB in flight each, in under 88 VGPRs -- which leaves room for two solve waves.
//   tile    64 pixels = M groups of 64 visits (M = visits per pixel)
//   LDS     per wave: the tile's rgba column (M KiB, read again by the ordered sums), a ring of kDmaRing slots for
//           the three columns the decision reads (3 KiB per group), 64 per-pixel counts of visits that do not add
//           to their own pixel, the work queue
//   order   rgba DMAs first, then the ring: vmcnt retires in issue order, so when a ring slot has landed the whole
//           rgba tile has.  Group g is awaited with s_waitcnt vmcnt(3 x groups issued after it).
//   sums    lane p adds pixel p's M entries in iterator order (src/lentil.h:938-955) and STORES the record to
//           FrameDev::dir: nothing is read back, nothing is shared with the splats.
//   tiles   handed out four at a time from DevCounters::tile_next (blocks are placed as the solve kernels leave
//           room: a static split would wait for the slowest CU).
// ---------------------------------------------------------------------------------------
constexpr uint32_t kDmaRing = 2;      // (3 slots: 79 KB per block at M = 9, two blocks then leave the solve kernel no LDS)

LD_DEV void lds_dma16_cached(const float4 *g, float4 *lds_wave_base) {      // the same without the nontemporal hint
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)(const void *)g, (lptr_t)(void *)lds_wave_base, 16, 0, 0);
}
LD_DEV void lds_dma16(const float4 *g, float4 *lds_wave_base) {
  // 64 lanes x 16 B -> lds_wave_base[lane]; aux 2 = nontemporal (read once)
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)(const void *)g, (lptr_t)(void *)lds_wave_base, 16, 0, 2);
}

// Which run of `run` tiles the k-th draw from DevCounters::tile_next stands for (t = k * run, relative to tile_begin).
// ScanArgs::outside_in: the runs alternate between the two ends of the range and meet in the middle.  A streamed pass
// ends when the solves of the items found LAST have ended, and the items near the frame's top and bottom edge -- where the
// lens vignettes: solves of 20-40 iterations, most of them failing, hundreds parked for the straggler kernel -- are the
// slow ones: scanned first, they are through long before the scan is; what the scan finds last lies mid-frame.
// (An experiment kept as a switch, off by default: the step does not change -- the parked solves are only served once the
// scan's waves have made room for the straggler kernel's.)
LD_DEV uint32_t scan_order(const ScanArgs &a, uint32_t t, uint32_t run) {
  if (!a.outside_in) return t;
  const uint32_t n_runs = (uint32_t)((a.tile_end - a.tile_begin + run - 1u) / run);
  const uint32_t k = t / run;
  return ((k & 1u) ? n_runs - 1u - (k >> 1) : (k >> 1)) * run;
}

LD_DEV void wait_vmcnt(uint32_t n) {       // n is wave-uniform; the instruction takes an immediate
#define LENTIL_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    LENTIL_VMCNT_CASE(0) LENTIL_VMCNT_CASE(1) LENTIL_VMCNT_CASE(2) LENTIL_VMCNT_CASE(3) LENTIL_VMCNT_CASE(4)
    LENTIL_VMCNT_CASE(5) LENTIL_VMCNT_CASE(6) LENTIL_VMCNT_CASE(7) LENTIL_VMCNT_CASE(8) LENTIL_VMCNT_CASE(9)
    LENTIL_VMCNT_CASE(10) LENTIL_VMCNT_CASE(11) LENTIL_VMCNT_CASE(12) LENTIL_VMCNT_CASE(13) LENTIL_VMCNT_CASE(14)
    LENTIL_VMCNT_CASE(15) LENTIL_VMCNT_CASE(16) LENTIL_VMCNT_CASE(17) LENTIL_VMCNT_CASE(18) LENTIL_VMCNT_CASE(19)
    LENTIL_VMCNT_CASE(20) LENTIL_VMCNT_CASE(21) LENTIL_VMCNT_CASE(22) LENTIL_VMCNT_CASE(23) LENTIL_VMCNT_CASE(24)
    LENTIL_VMCNT_CASE(25) LENTIL_VMCNT_CASE(26) LENTIL_VMCNT_CASE(27) LENTIL_VMCNT_CASE(28) LENTIL_VMCNT_CASE(29)
    LENTIL_VMCNT_CASE(30) LENTIL_VMCNT_CASE(31) LENTIL_VMCNT_CASE(32) LENTIL_VMCNT_CASE(33) LENTIL_VMCNT_CASE(34)
    LENTIL_VMCNT_CASE(35) LENTIL_VMCNT_CASE(36) LENTIL_VMCNT_CASE(37) LENTIL_VMCNT_CASE(38) LENTIL_VMCNT_CASE(39)
    LENTIL_VMCNT_CASE(40) LENTIL_VMCNT_CASE(41) LENTIL_VMCNT_CASE(42) LENTIL_VMCNT_CASE(43) LENTIL_VMCNT_CASE(44)
    LENTIL_VMCNT_CASE(45) LENTIL_VMCNT_CASE(46) LENTIL_VMCNT_CASE(47) LENTIL_VMCNT_CASE(48) LENTIL_VMCNT_CASE(49)
    LENTIL_VMCNT_CASE(50) LENTIL_VMCNT_CASE(51) LENTIL_VMCNT_CASE(52) LENTIL_VMCNT_CASE(53) LENTIL_VMCNT_CASE(54)
    LENTIL_VMCNT_CASE(55) LENTIL_VMCNT_CASE(56) LENTIL_VMCNT_CASE(57) LENTIL_VMCNT_CASE(58) LENTIL_VMCNT_CASE(59)
    LENTIL_VMCNT_CASE(60) LENTIL_VMCNT_CASE(61) LENTIL_VMCNT_CASE(62) LENTIL_VMCNT_CASE(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef LENTIL_VMCNT_CASE
}

__host__ __device__ constexpr uint32_t dma_wave_f4(uint32_t M, uint32_t ring = kDmaRing) { return M * 64u + ring * 192u + 16u; }

// The three columns of one ring slot, lane's visit.  In assembly because hipcc puts s_waitcnt vmcnt(0) in front of
// every LDS read it can see while an LDS-DMA may be in flight -- which would wait for the groups behind this one too
// (the caller has waited for exactly this group's DMAs).  The reads are waited for here: the compiler does not
// track them.
template <uint32_t kColBytes = 1024u>
LD_DEV void lds_read_slot(const float4 *slot_lane, float4 &c0, float4 &c1, float4 &c2) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) void *lptr_t;
  const uint32_t addr = (uint32_t)(size_t)(lptr_t)(const void *)slot_lane;
  v4f x, y, z;
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%4\n\tds_read_b128 %2, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(x), "=&v"(y), "=&v"(z)
               : "v"(addr), "n"(kColBytes), "n"(2u * kColBytes)
               : "memory");
  c0 = make_float4(x.x, x.y, x.z, x.w);
  c1 = make_float4(y.x, y.y, y.z, y.w);
  c2 = make_float4(z.x, z.y, z.z, z.w);
}

__global__ __launch_bounds__(256) void scan_dma_kernel(ScanArgs a) {
  LENTIL_TL_SPAN(SPAN_SCAN);
  extern __shared__ float4 smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const VisitsDev &V = a.V;
  const uint32_t M = V.visits_per_pixel;
  const uint32_t R = a.ring >= 2u && a.ring <= 8u ? a.ring : kDmaRing;       // ring slots (ScanArgs::ring; 0: kDmaRing)
  const uint32_t wave_f4 = dma_wave_f4(M, R);
  float4 *srgba = smem + (size_t)wave * wave_f4;                   // [M][64]
  float4 *ring = srgba + (size_t)M * 64u;                          // [R][3][64]
  uint32_t *nskip = reinterpret_cast<uint32_t *>(ring + R * 192u);   // [64]
  uint2 *qmem = reinterpret_cast<uint2 *>(smem + (size_t)4u * wave_f4);
  WaveQueue wq;
  wq.init(qmem + (size_t)wave * kWaveQueueLds);

  // beside solve waves that would issue fp64 arithmetic in every cycle: the scan's few instructions go first
  __builtin_amdgcn_s_setprio(3);
  const uint64_t n_pixels = V.n / M;                               // (the host checks that the stream holds whole pixels)
  const uint64_t n_tiles = a.tile_end;
  const uint32_t xres = a.P.xres;
  const float w = 1.0f * a.P.inverse_sample_density;               // filter_weight * inv_density, lentil.h:949-953
  float4 *dir4 = reinterpret_cast<float4 *>(a.F.dir);
  const uint64_t v_last = V.n - 1;

  nskip[lane] = 0u;
  auto issue_ring = [&](uint64_t v0, uint32_t g) {
    const uint64_t v = v0 + (uint64_t)g * 64u + lane;
    const uint64_t vl = v < V.n ? v : v_last;                      // lanes past the end re-read the last visit (unused)
    float4 *slot = ring + (size_t)(g % R) * 192u;
    lds_dma16(V.pos_z + vl, slot);
    lds_dma16(V.volume_ignore + vl, slot + 64);
    lds_dma16(V.transmission + vl, slot + 128);
  };

  while (true) {
    uint32_t t4 = 0;
    if (lane == 0) t4 = atomicAdd(&a.ctr->tile_next, 4u);
    t4 = __builtin_amdgcn_readfirstlane(t4);
    if (a.tile_begin + (uint64_t)t4 >= n_tiles) break;
    const uint64_t tile4 = a.tile_begin + (uint64_t)scan_order(a, t4, 4u);
    const uint64_t tile4_end = tile4 + 4u < n_tiles ? tile4 + 4u : n_tiles;
    for (uint64_t tile = tile4; tile < tile4_end; ++tile) {
      const uint64_t pix0 = tile * 64u;
      const uint64_t v0 = pix0 * M;
      for (uint32_t g = 0; g < M; ++g) {
        const uint64_t v = v0 + (uint64_t)g * 64u + lane;
        lds_dma16(V.rgba + (v < V.n ? v : v_last), srgba + (size_t)g * 64u);
      }
      for (uint32_t g = 0; g < R && g < M; ++g) issue_ring(v0, g);
      for (uint32_t g = 0; g < M; ++g) {
        // ring groups issued after g's: g+1 .. min(g + kDmaRing - 1, M - 1)
        const uint32_t behind = (M - 1u - g) < (R - 1u) ? (M - 1u - g) : (R - 1u);
        wait_vmcnt(3u * behind);
        float4 pz, vi, tr;
        lds_read_slot(ring + (size_t)(g % R) * 192u + lane, pz, vi, tr);
        const uint32_t e = g * 64u + lane;
        const uint64_t v = v0 + e;
        const bool valid = v < V.n;
        bool flagged = false;
        int samples = 0;
        if (valid) {
          flagged = visit_redistributes(a.P, a.lens_length, pz, vi, tr, a.P.inverse_sample_density,
                                        [&]() { return V.raydir_time[v]; }, V.cam);
          // a few visits in 10^5: the draw count (same function as the draw kernels use)
          if (flagged)
            samples = visit_prologue(a.P, a.lens_length, srgba[e], pz, V.raydir_time[v], vi, tr, a.P.inverse_sample_density, V.cam).samples;
        }
        if (flagged || !valid) {
          // adds nothing to its own pixel: +0 values (x + (+0) changes no bit, an accumulator is never -0), one weight fewer
          srgba[e] = make_float4(0.f, 0.f, 0.f, 0.f);
          atomicAdd(&nskip[e / M], 1u);
        }
        wq.push(flagged, (uint32_t)v, (uint32_t)samples, a);
        if (g + R < M) {
          // the slot's values are in registers (the decision above has used them): refill it
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
          issue_ring(v0, g + R);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      const uint64_t pix = pix0 + lane;
      const uint32_t skipped = nskip[lane];
      nskip[lane] = 0u;        // for the next tile, here: an LDS access at the top of a tile would wait for this tile's stores
      if (pix < n_pixels) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t j = 0; j < M; ++j) {
          const float4 c = srgba[lane * M + j];
          s.x += (c.x + 0.0f) * w; s.y += (c.y + 0.0f) * w; s.z += (c.z + 0.0f) * w; s.w += (c.w + 0.0f) * w;
        }
        float ws = 0.f;
        const uint32_t cnt = M - skipped;
        for (uint32_t j = 0; j < cnt; ++j) ws += w;
        const int px = V.pixel_x0 + (int)(pix % V.pixels_per_row);
        const int py = V.pixel_y0 + (int)(pix / V.pixels_per_row) * (int)V.pixel_row_stride;
        const uint64_t lin = (uint64_t)px + (uint64_t)py * xres;
        dir4[lin * 2u] = s;
        dir4[lin * 2u + 1u] = make_float4(ws, 0.f, 0.f, 0.f);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      wq.end_tile(a);
    }
    if (lane == 0) tl_add(TL_SCAN_TILES, (uint32_t)(tile4_end - tile4));
  }
  wq.finish(a);
  scan_block_done(a);
}

// ---------------------------------------------------------------------------------------
// scan_dma2_kernel (round 4) -- scan_dma_kernel with the tiles of a wave pipelined into one another, and a third of
// its vector instructions.  What the measurements said: with one block per CU -- all that fits
// beside two resident solve blocks -- scan_dma_kernel takes 1.35-1.47 ms whether the solve waves beside it work or
// sleep (two blocks per CU: 1.05): every tile ends with nothing in flight, and a wave alone on its SIMD has nobody to
// cover the memory latency that follows.  And its 122 vector instructions per 64 visits are 0.24 ms of the chip's
// vector issue time in a pass whose other tenant, the solves, is bound by exactly that.
//   pipeline  the rgba column of the NEXT tile is requested group by group while this tile's groups are decided (a
//             second rgba buffer: 2 M + 6 KiB of LDS per wave), and the ring of decision columns simply runs on into
//             the next tile: in the steady state every step issues the same four DMAs -- ring group G + 2 (three
//             columns), rgba group g of the next tile -- and waits for `vmcnt(5)`: the five LOADS issued behind the
//             group it is about to read.  (Only loads are counted: they return in order among themselves; the two record
//             stores of a tile's end may complete early or late -- the wait then merely includes a load or two more.)
//   tiles     a wave's first 7/8 are its share of a static interleaved split (runs of four tiles; the next tile is known
//             without asking anybody), the rest come from DevCounters::tile_next in runs of four (a returning atomic
//             and a drained pipeline per run, where the CUs' different speeds need evening out).
//   decision  visit_redistributes, with its two fp32 divisions (get_coc_thinlens < 0.4) replaced by comparisons of the
//             camera-space depth with precomputed intervals wherever those decide it with certainty (ScanBands, host:
//             scan_bands() -- the comparison of the exact circle of confusion with 0.4 (1 +- 1e-3)); a wave with a visit
//             inside a band, at infinite depth under a skydome, or behind a moving camera calls the function itself.
//   sums      as before (lane p adds pixel p's M entries in iterator order), the entries read with inline ds_read_b128
//             (a load the compiler can see makes it wait for every DMA in flight), packed fp32 arithmetic, the record's
//             linear index from a multiply-high instead of a division.
// Same decisions, same sums bit for bit as scan_dma_kernel (tests: every parity case that runs a beauty-only frame).
// ---------------------------------------------------------------------------------------
__host__ __device__ constexpr uint32_t dma2_wave_f4(uint32_t M) { return 2u * M * 64u + 2u * 192u + 16u; }
LD_DEV float4 lds_read_f4(const float4 *p);       // (below, with the other untracked reads)

LD_DEV void dma2_col(const float4 *ubase, uint32_t lane16, float4 *lds_uniform) {
  // 64 lanes x 16 B at ubase + lane -> lds_uniform[lane]; the base is wave-uniform (scalar registers), the lane part a
  // 32-bit offset; aux 2 = nontemporal
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)(const void *)(reinterpret_cast<const char *>(ubase) + lane16), (lptr_t)(void *)lds_uniform, 16, 0, 2);
}
// three consecutive float4 of the wave's LDS with one wait, in one asm block: nothing may touch the destination registers
// between a read and the wait
LD_DEV void lds_read_3f4(const float4 *p, float4 &c0, float4 &c1, float4 &c2) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) void *lptr_t;
  const uint32_t addr = (uint32_t)(size_t)(lptr_t)(const void *)p;
  v4f x, y, z;
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(x), "=&v"(y), "=&v"(z)
               : "v"(addr)
               : "memory");
  c0 = make_float4(x.x, x.y, x.z, x.w); c1 = make_float4(y.x, y.y, y.z, y.w); c2 = make_float4(z.x, z.y, z.z, z.w);
}

// get_coc_thinlens(P, cz) < 0.4f from cz alone: 1 = certainly, 0 = certainly not, 2 = ask the function
LD_DEV int coc_below_by_bands(const ScanBands &B, float cz) {
  const bool in0 = cz >= B.in_lo[0] && cz <= B.in_hi[0];
  const bool in1 = cz >= B.in_lo[1] && cz <= B.in_hi[1];
  const bool out0 = cz >= B.out_lo[0] && cz <= B.out_hi[0];
  const bool out1 = cz >= B.out_lo[1] && cz <= B.out_hi[1];
  if (in0 || in1) return 1;
  if (out0 || out1) return 2;
  return 0;       // (NaN, zero and the infinities come here: the function's NaN / infinity is "not below" as well)
}

__global__ __launch_bounds__(256) void scan_dma2_kernel(ScanArgs a) {
  LENTIL_TL_SPAN(SPAN_SCAN);
  extern __shared__ float4 smem[];
  // (the wave's number as a scalar: tile numbers, LDS bases and every branch on them stay in scalar registers)
  // ScanArgs::skip_blocks: the first blocks of the grid leave at once -- their CUs' registers and LDS go to a third resident
  // solve block (lentil_hip.hip, scan_cus_pct).  They are part of the grid, not left out of it, so that whichever of the two
  // kernels the dispatcher places first, every block that does scan finds a CU with room: a CU holds either this kernel's
  // block and two solve blocks or three solve blocks, and the blocks that leave free exactly the CUs the others need.
  if (blockIdx.x < a.skip_blocks) { scan_block_done(a); return; }
  const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const VisitsDev &V = a.V;
  const uint32_t M = V.visits_per_pixel;
  const uint32_t wave_f4 = dma2_wave_f4(M);
  float4 *rg = smem + (size_t)wave * wave_f4;                      // [2][M][64]
  float4 *ring = rg + (size_t)2u * M * 64u;                        // [2][3][64]
  uint32_t *nskip = reinterpret_cast<uint32_t *>(ring + 384u);     // [64]
  uint2 *qmem = reinterpret_cast<uint2 *>(smem + (size_t)4u * wave_f4);
  WaveQueue wq;
  wq.init(qmem + (size_t)wave * kWaveQueueLds);
  __builtin_amdgcn_s_setprio(3);

  const uint32_t lane16 = lane * 16u;
  const uint32_t TV = 64u * M;                                     // visits per tile
  const uint64_t full_end = (V.n / TV) < a.tile_end ? (V.n / TV) : a.tile_end;     // tiles whose 64 pixels all exist
  const float w = 1.0f * a.P.inverse_sample_density;               // filter_weight * inv_density, lentil.h:949-953
  float ws_full = 0.f;
  for (uint32_t j = 0; j < M; ++j) ws_full += w;                   // the weight of a pixel none of whose visits is redistributed
  float4 *dir4 = reinterpret_cast<float4 *>(a.F.dir);
  const uint32_t xres = a.P.xres;
  // decision constants (a resting camera: this kernel is not chosen otherwise)
  const float mz0 = a.P.world_to_camera[0][2], mz1 = a.P.world_to_camera[1][2], mz2 = a.P.world_to_camera[2][2], mz3 = a.P.world_to_camera[3][2];
  float scale = 1.0f;
  if (a.P.unitModel == LENTIL_UNIT_MM) scale = 0.1f;
  else if (a.P.unitModel == LENTIL_UNIT_DM) scale = 10.0f;
  else if (a.P.unitModel == LENTIL_UNIT_M) scale = 100.0f;
  const bool never = a.P.adaptive_sampling && a.P.inverse_sample_density > 0.2f;
  const double inside_lens = a.lens_length * 0.1;
  const bool po = a.P.cameraType == LENTIL_POLYNOMIAL_OPTICS;

  nskip[lane] = 0u;
  auto issue_ring = [&](uint64_t v_group, uint32_t slot) {
    float4 *s = ring + (size_t)slot * 192u;
    dma2_col(V.pos_z + v_group, lane16, s);
    dma2_col(V.volume_ignore + v_group, lane16, s + 64);
    dma2_col(V.transmission + v_group, lane16, s + 128);
  };

  // ---- one tile.  Its rgba column has been requested into rg[buf], its first two ring groups into slots par, par ^ 1.
  // has_next: the tile that follows is `next_tile` -- its rgba goes to rg[buf ^ 1], its first ring groups behind this
  // tile's last.
  auto do_tile = [&](uint64_t tile, bool has_next, uint64_t next_tile, uint32_t buf, uint32_t par) {
    const uint64_t v0 = tile * TV;
    const uint64_t nv0 = next_tile * TV;
    float4 *srgba = rg + (size_t)buf * M * 64u;
    float4 *nrgba = rg + (size_t)(buf ^ 1u) * M * 64u;
    bool tile_any = false;
    for (uint32_t g = 0; g < M; ++g) {
      if (has_next) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (g + 1u < M) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t slot = (par + g) & 1u;
      float4 pz, vi, tr;
      lds_read_slot(ring + (size_t)slot * 192u + lane, pz, vi, tr);
      // the slot's values are in registers: refill it, and ask for the next tile's rgba group
      if (g + 2u < M) issue_ring(v0 + (uint64_t)(g + 2u) * 64u, slot);
      else if (has_next) issue_ring(nv0 + (uint64_t)(g + 2u - M) * 64u, slot);
      if (has_next) dma2_col(V.rgba + nv0 + (uint64_t)g * 64u, lane16, nrgba + (size_t)g * 64u);
      // ---- the decision (visit_redistributes, src/lentil_filter.cpp:105-165,240)
      const float wx = pz.x, wy = pz.y, wz = pz.z, depth = pz.w;
      const bool small = fabsf(wx) < kAiEpsilon && fabsf(wy) < kAiEpsilon && fabsf(wz) < kAiEpsilon;
      const bool far = (depth == kAiInfinite) || small;
      const float cz = (wx * mz0 + wy * mz1 + wz * mz2 + mz3) * scale;
      const int below = coc_below_by_bands(a.bands, cz);
      bool flagged;
      if (__ballot((far && a.P.enable_skydome) || below == 2) == 0ull) {
        bool r = !never && !far;
        if (fmaxf(fmaxf(vi.x, vi.y), vi.z) > 0.0f) r = false;
        if (!a.P.enable_bidir_transmission && fmaxf(fmaxf(tr.x, tr.y), tr.z) > 0.0f) r = false;
        if (vi.w > 0.0f) r = false;
        if (below == 1) r = false;
        if (po && (double)fabsf(cz) < inside_lens) r = false;
        flagged = r;
      } else {
        const uint64_t v = v0 + (uint64_t)g * 64u + lane;
        flagged = visit_redistributes(a.P, a.lens_length, pz, vi, tr, a.P.inverse_sample_density,
                                      [&]() { return V.raydir_time[v]; }, V.cam);
      }
      if (__ballot(flagged)) {
        // a few visits in 10^5: the draw count (same function as the draw kernels use); the visit adds nothing to its
        // own pixel: +0 values (x + (+0) changes no bit, an accumulator is never -0), one weight fewer
        tile_any = true;
        const uint32_t e = g * 64u + lane;
        const uint64_t v = v0 + e;
        int samples = 0;
        if (flagged) {
          samples = visit_prologue(a.P, a.lens_length, lds_read_f4(srgba + e), pz, V.raydir_time[v], vi, tr, a.P.inverse_sample_density, V.cam).samples;
          srgba[e] = make_float4(0.f, 0.f, 0.f, 0.f);
          atomicAdd(&nskip[e / M], 1u);
        }
        wq.push(flagged, (uint32_t)v, (uint32_t)samples, a);
      }
    }
    // ---- the ordered sums of the tile's 64 pixels (filter_and_add_to_buffer_new, src/lentil.h:938-955)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
    const v2f w2 = {w, w};
    const float4 *mine = srgba + (size_t)lane * M;
    uint32_t j = 0;
    for (; j + 3u <= M; j += 3u) {
      float4 c0, c1, c2;
      lds_read_3f4(mine + j, c0, c1, c2);
      v2f p;
      p = (v2f){c0.x, c0.y} * w2; s01 += p; p = (v2f){c0.z, c0.w} * w2; s23 += p;
      p = (v2f){c1.x, c1.y} * w2; s01 += p; p = (v2f){c1.z, c1.w} * w2; s23 += p;
      p = (v2f){c2.x, c2.y} * w2; s01 += p; p = (v2f){c2.z, c2.w} * w2; s23 += p;
    }
    for (; j < M; ++j) {
      const float4 c0 = lds_read_f4(mine + j);
      v2f p;
      p = (v2f){c0.x, c0.y} * w2; s01 += p; p = (v2f){c0.z, c0.w} * w2; s23 += p;
    }
    float ws = ws_full;
    if (tile_any) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      // (the zeros the flagged visits left in the rgba tile were stored ahead of the sums' reads: LDS runs in order)
      const uint32_t skipped = nskip[lane];
      nskip[lane] = 0u;
      ws = 0.f;
      for (uint32_t q = 0; q < M - skipped; ++q) ws += w;
    }
    {
      const uint32_t pix = (uint32_t)(tile * 64u) + lane;        // (the host checks that the stream has fewer than 2^32 pixels)
      const uint32_t row = __umulhi(pix, a.ppr_magic) >> a.ppr_shift;     // pix / pixels_per_row
      const uint32_t col = pix - row * V.pixels_per_row;
      const int px = V.pixel_x0 + (int)col;
      const int py = V.pixel_y0 + (int)row * (int)V.pixel_row_stride;
      const uint64_t lin = (uint64_t)px + (uint64_t)py * xres;
      dir4[lin * 2u] = make_float4(s01.x, s01.y, s23.x, s23.y);
      dir4[lin * 2u + 1u] = make_float4(ws, 0.f, 0.f, 0.f);
    }
    if (tile_any) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      wq.end_tile(a);
    }
  };

  // ---- a wave's tiles: runs of four, claimed from DevCounters::tile_next in SEQUENCES that shrink as the scan goes on
  // (round 5; until then 7/8 of the runs were split statically between the waves of the grid, which made the scan depend on
  // every one of its blocks being resident at once: a block the dispatcher had not placed -- its CU full of resident solve
  // blocks -- kept its share of the frame unscanned while the solve waves waited for the scan's end: a pass stuck for good,
  // seen as the odd 250 ms time-out.  Now any subset of the blocks finishes the frame.)
  // With R = runs per wave: a ticket of level 0 stands for R/2 runs, of level 1 for R/4, then R/8, R/16, then single runs;
  // every level hands out one ticket per wave of the grid, and a ticket's runs lie W runs apart (ticket t of a level: runs
  // base + t, base + t + W, ...), so that the waves still move through the frame side by side, in scan order, and a
  // sequence is one pipelined chain of tiles as the static share was: a wave pays the bubble between two sequences five
  // times and then once per single run, on the last sixteenth of the frame.
  const uint64_t n_full = full_end > a.tile_begin ? full_end - a.tile_begin : 0ull;
  const uint32_t n_runs = (uint32_t)((n_full + 3u) / 4u);
  const uint32_t W = (gridDim.x - a.skip_blocks) * 4u;
  const uint32_t R = n_runs / W;
  // (scalars, not arrays: everything here is wave-uniform and should stay in SGPRs)
  const uint32_t n_lvl = (R >= 2u ? 1u : 0u) + (R >= 4u ? 1u : 0u) + (R >= 8u ? 1u : 0u) + (R >= 16u ? 1u : 0u);
  const uint32_t lb1 = (R >> 1) * W, lb2 = lb1 + (R >> 2) * W, lb3 = lb2 + (R >> 3) * W, lb4 = lb3 + (R >> 4) * W;
  // (ScanArgs::outside_in: the runs alternate between the two ends of the range and meet in the middle, scan_order)
  auto run_first = [&](uint32_t r) { return a.tile_begin + (uint64_t)(a.outside_in ? ((r & 1u) ? n_runs - 1u - (r >> 1) : (r >> 1)) : r) * 4u; };
  auto run_len = [&](uint64_t first) { return (uint32_t)(full_end - first < 4u ? full_end - first : 4u); };
  uint32_t buf = 0, par = 0;
  while (true) {
    // the next sequence of runs (nothing of this wave's is in flight here: the returning atomic does not disturb the
    // counted waits inside a sequence)
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&a.ctr->tile_next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane(t);
    uint32_t r, runs_left, stride;
    if (t < n_lvl * W) {
      const uint32_t l = t / W;
      r = (l == 0u ? 0u : (l == 1u ? lb1 : (l == 2u ? lb2 : lb3))) + (t - l * W);
      runs_left = R >> (l + 1u);
      stride = W;
    } else {
      r = lb4 + (t - n_lvl * W);
      runs_left = 1u;
      stride = 0u;
      if (r >= n_runs) break;
    }
    uint64_t tile = run_first(r);
    uint32_t left = run_len(tile);
    // prologue: this tile's rgba and first two ring groups, and everything landed
    {
      const uint64_t v0 = tile * TV;
      for (uint32_t g = 0; g < M; ++g) dma2_col(V.rgba + v0 + (uint64_t)g * 64u, lane16, rg + (size_t)buf * M * 64u + (size_t)g * 64u);
      issue_ring(v0, par);
      if (M > 1u) issue_ring(v0 + 64u, par ^ 1u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    while (true) {
      // which tile follows?
      bool has_next = false;
      uint64_t next_tile = tile;
      uint32_t next_left = 0;
      if (left > 1u) { has_next = true; next_tile = tile + 1u; next_left = left - 1u; }
      else if (runs_left > 1u) {
        has_next = true;
        next_tile = run_first(r + stride);
        next_left = run_len(next_tile);
      }
      do_tile(tile, has_next, next_tile, buf, par);
      if (lane == 0) tl_add(TL_SCAN_TILES, 1u);
      buf ^= 1u;
      par = (par + M) & 1u;
      if (left == 1u) { r += stride; --runs_left; }
      if (!has_next) break;
      tile = next_tile; left = next_left;
    }
  }
  // the stream's last, partial tile (fewer than 64 pixels): one wave, straight from global memory
  if (a.tile_end > full_end && blockIdx.x == a.skip_blocks && wave == 0) {
    const uint64_t tile = full_end;
    const uint64_t n_pixels = V.n / M;
    const uint64_t pix = tile * 64u + lane;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float ws = 0.f;
    for (uint32_t j = 0; j < M; ++j) {
      const uint64_t v = pix * M + j;
      const bool valid = pix < n_pixels;
      bool flagged = false;
      int samples = 0;
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        const float4 pz = V.pos_z[v], vi = V.volume_ignore[v], tr = V.transmission[v];
        c = V.rgba[v];
        flagged = visit_redistributes(a.P, a.lens_length, pz, vi, tr, a.P.inverse_sample_density, [&]() { return V.raydir_time[v]; }, V.cam);
        if (flagged) samples = visit_prologue(a.P, a.lens_length, c, pz, V.raydir_time[v], vi, tr, a.P.inverse_sample_density, V.cam).samples;
      }
      if (valid && !flagged) { s.x += c.x * w; s.y += c.y * w; s.z += c.z * w; s.w += c.w * w; ws += w; }
      wq.push(flagged, (uint32_t)v, (uint32_t)samples, a);
    }
    if (pix < n_pixels) {
      const int px = V.pixel_x0 + (int)(pix % V.pixels_per_row);
      const int py = V.pixel_y0 + (int)(pix / V.pixels_per_row) * (int)V.pixel_row_stride;
      const uint64_t lin = (uint64_t)px + (uint64_t)py * xres;
      dir4[lin * 2u] = s;
      dir4[lin * 2u + 1u] = make_float4(ws, 0.f, 0.f, 0.f);
    }
    wq.end_tile(a);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wq.finish(a);
  scan_block_done(a);
}

// ---------------------------------------------------------------------------------------
// K1+K2+K6 for frames with extra (gaussian) AOVs and a uniform footprint -- the LDS-DMA form of
// scan_uniform_multi_kernel, as scan_dma_kernel is of scan_uniform_kernel: 80 + 16 K bytes per visit go from HBM
// straight into LDS, the ordered sums are stored to FrameDev::dir whole (no record is read), and the kernel stays
// under 100 VGPRs, so that two solve blocks per CU fit beside it: frames with extra AOVs can run streamed.
//   group   ppt = 64 / M whole pixels = ppt * M <= 64 visits (M = visits per pixel): everything a pixel's sums need
//           lies in one group.  Per group 4 + K columns of 1 KiB: pos_z, volume_ignore, transmission, rgba, extra[k].
//   ring    R = 2 or 3 slots (ScanArgs::ring) of 4 + K columns (1 KiB + 16 B of padding each, kDmaMultiCol) per wave; groups g + 1 .. g + R - 1 are in flight while g is decided and summed,
//           g + R is issued when g's slot is free.  vmcnt counts loads and stores in issue order, so the wait for group g names
//           what was issued after its loads: the next group's columns and the record stores of the group before.
//           (Every LDS read of a slot is inline assembly: the compiler puts vmcnt(0) before any LDS read it sees.)
//   sums    lane (pixel, float4 j of its record) adds that pixel's M entries of AOV j in iterator order
//           (src/lentil.h:938-955) from the raw columns, four entries per LDS round trip; which entries count comes
//           from one ballot per group (a redistributed visit's entries count as +0 and add no weight).  j = n_aovs is
//           the record's weight.  A group's records lie side by side in FrameDev::dir, so a pass of 64 lanes is ONE
//           store instruction of 1 KiB without holes; lanes without a record write a scratch line, so that every
//           pass is exactly one instruction: that keeps the vmcnt arithmetic exact.
//   runs    a wave draws 16 groups at a time from DevCounters::tile_next.
// ---------------------------------------------------------------------------------------
LD_DEV float4 lds_read_f4(const float4 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) void *lptr_t;
  const uint32_t addr = (uint32_t)(size_t)(lptr_t)(const void *)p;
  v4f x;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(addr) : "memory");
  return make_float4(x.x, x.y, x.z, x.w);
}
// four consecutive float4 with one wait (the ordered sums: a pixel's entries lie side by side)
LD_DEV void lds_read_4f4(const float4 *p, float4 c[4]) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(3))) void *lptr_t;
  const uint32_t addr = (uint32_t)(size_t)(lptr_t)(const void *)p;
  v4f x, y, z, w;
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w)
               : "v"(addr)
               : "memory");
  c[0] = make_float4(x.x, x.y, x.z, x.w); c[1] = make_float4(y.x, y.y, y.z, y.w);
  c[2] = make_float4(z.x, z.y, z.z, z.w); c[3] = make_float4(w.x, w.y, w.z, w.w);
}
// A global read the compiler does not track (waited for here): a load it can see makes it wait for vmcnt(0)
// wherever the destination registers are written next, e.g. in the middle of the next group's DMAs.
LD_DEV float4 global_read_f4_untracked(const float4 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f x;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
  return make_float4(x.x, x.y, x.z, x.w);
}
constexpr uint32_t kDmaMultiRun = 16;      // groups a wave draws at a time
// A column of a slot is 64 float4 and one of padding: columns 1 KiB apart would put the sum lanes of one pixel -- same
// entry, different AOV -- on the same LDS banks (nine ways at nine AOVs); 65 spreads them evenly.
constexpr uint32_t kDmaMultiCol = 65;
__host__ __device__ constexpr uint32_t dma_multi_wave_f4(uint32_t n_extra, uint32_t ring) { return ring * (4u + n_extra) * kDmaMultiCol; }

__global__ __launch_bounds__(256) void scan_dma_multi_kernel(ScanArgs a) {
  LENTIL_TL_SPAN(SPAN_SCAN);
  extern __shared__ float4 smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const VisitsDev &V = a.V;
  const uint32_t M = V.visits_per_pixel, K = V.n_extra, n_aovs = a.F.n_aovs;
  const uint32_t n_cols = 4u + K;
  const uint32_t ppt = a.ppt, TV = ppt * M;
  constexpr uint32_t CS = kDmaMultiCol;
  const uint32_t R = a.ring & 0xFFu;                                // slots per wave (2 or 3)
  float4 *ring = smem + (size_t)wave * dma_multi_wave_f4(K, R);    // [R][n_cols][CS]
  uint2 *qmem = reinterpret_cast<uint2 *>(smem + (size_t)4u * dma_multi_wave_f4(K, R));
  WaveQueue wq;
  wq.init(qmem + (size_t)wave * kWaveQueueLds);
  __builtin_amdgcn_s_setprio(3);
  const uint64_t n_pixels = V.n / M;
  const uint64_t n_tiles = a.tile_end;
  const uint32_t xres = a.P.xres, q = a.F.stride >> 2;
  const float w_in = 1.0f * a.P.inverse_sample_density;
  float4 *dir4 = reinterpret_cast<float4 *>(a.F.dir);
  float4 *dummy = a.dummy + lane;                                   // where lanes without a record put their stores
  const uint64_t v_last = V.n - 1;
  // record stores per group: the sum lanes come in passes of 64 (pixel, float4 of the record) pairs, one instruction each
  const uint32_t n_sum = ppt * q, passes = (n_sum + 63u) / 64u, store_ops = passes;

  auto issue = [&](uint64_t tile, uint32_t si) {
    const uint64_t v = tile * (uint64_t)TV + lane;
    const uint64_t vl = (lane < TV && v < V.n) ? v : v_last;        // lanes past the group / the end re-read the last visit (unused)
    float4 *slot = ring + (size_t)si * n_cols * CS;
    if (a.ring & 0x100u) {
      lds_dma16_cached(V.pos_z + vl, slot);
      lds_dma16_cached(V.volume_ignore + vl, slot + CS);
      lds_dma16_cached(V.transmission + vl, slot + 2u * CS);
      lds_dma16_cached(V.rgba + vl, slot + 3u * CS);
      for (uint32_t k = 0; k < K; ++k) lds_dma16_cached(V.extra[k] + vl, slot + (4u + k) * CS);
      return;
    }
    lds_dma16(V.pos_z + vl, slot);
    lds_dma16(V.volume_ignore + vl, slot + CS);
    lds_dma16(V.transmission + vl, slot + 2u * CS);
    lds_dma16(V.rgba + vl, slot + 3u * CS);
    for (uint32_t k = 0; k < K; ++k) lds_dma16(V.extra[k] + vl, slot + (4u + k) * CS);
  };

#ifdef LENTIL_TIMELINE
  uint64_t tacc[6] = {0, 0, 0, 0, 0, 0};
#endif
  while (true) {
    uint32_t t0 = 0;
    if (lane == 0) t0 = atomicAdd(&a.ctr->tile_next, kDmaMultiRun);
    t0 = __builtin_amdgcn_readfirstlane(t0);
    if (a.tile_begin + (uint64_t)t0 >= n_tiles) break;
    const uint64_t run0 = a.tile_begin + (uint64_t)scan_order(a, t0, kDmaMultiRun);
    const uint64_t run1 = run0 + kDmaMultiRun < n_tiles ? run0 + kDmaMultiRun : n_tiles;
    for (uint32_t g = 0; g < R && run0 + g < run1; ++g) issue(run0 + g, g);
    uint32_t si = 0;                                                // the slot of `tile`: (tile - run0) mod R
    for (uint64_t tile = run0; tile < run1; ++tile) {
      // what was issued after this group's loads: the columns of the R - 1 groups behind it, the record stores of the
      // R - 1 groups before it
      const uint32_t behind = (uint32_t)(run1 - 1u - tile) < R - 1u ? (uint32_t)(run1 - 1u - tile) : R - 1u;
      const uint32_t before = (uint32_t)(tile - run0) < R - 1u ? (uint32_t)(tile - run0) : R - 1u;
#ifdef LENTIL_TIMELINE
      const uint64_t tp0 = clock64();
#endif
      wait_vmcnt(behind * n_cols + before * store_ops);
#ifdef LENTIL_TIMELINE
      const uint64_t tp1 = clock64();
#endif
      const float4 *slot = ring + (size_t)si * n_cols * CS;
      const uint64_t pix0 = tile * ppt;
      const uint64_t v = tile * (uint64_t)TV + lane;
      const bool valid = lane < TV && v < V.n;
      float4 pz, vi, tr;
      lds_read_slot<CS * 16u>(slot + lane, pz, vi, tr);
      bool flagged = false;
      int samples = 0;
      if (valid) {
        // (no camera motion keys here: the host sends such streams to scan_uniform_multi_kernel)
        flagged = visit_redistributes(a.P, a.lens_length, pz, vi, tr, a.P.inverse_sample_density,
                                      [&]() { return global_read_f4_untracked(V.raydir_time + v); });
        if (flagged)
          samples = visit_prologue(a.P, a.lens_length, lds_read_f4(slot + 3u * CS + lane), pz,
                                   global_read_f4_untracked(V.raydir_time + v), vi, tr, a.P.inverse_sample_density).samples;
      }
      // bit e: the group's entry e adds to its own pixel (weight w_in; the others count as +0 and add no weight)
      const unsigned long long counts = __ballot(valid && !flagged);
      const bool all_count = counts == __ballot(lane < TV) && tile * (uint64_t)TV + TV <= V.n;
      wq.push(flagged, (uint32_t)v, (uint32_t)samples, a);
#ifdef LENTIL_TIMELINE
      const uint64_t tp2 = clock64();
#endif
      // ---- sums: lane (pixel, float4 of its record)
      for (uint32_t pass = 0; pass < passes; ++pass) {
        const uint32_t idx = pass * 64u + lane;
        const uint32_t pi = idx / q, k = idx - pi * q;
        const bool mine = idx < n_sum && pix0 + pi < n_pixels;
        const uint32_t e0 = (mine ? pi : 0u) * M;
        const float4 *col = slot + (3u + (k < n_aovs ? k : 0u)) * CS + e0;       // rgba, extra[k - 1]
        const unsigned long long bits = counts >> e0;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        float ws = 0.f;
        auto add = [&](const float4 &c, bool on) {
          // (an entry that does not count is +0: it changes no bit of the sum)
          const float tx = sum.x + (c.x + 0.0f) * w_in, ty = sum.y + (c.y + 0.0f) * w_in;
          const float tz = sum.z + (c.z + 0.0f) * w_in, tw = sum.w + (c.w + 0.0f) * w_in;
          sum.x = on ? tx : sum.x; sum.y = on ? ty : sum.y; sum.z = on ? tz : sum.z; sum.w = on ? tw : sum.w;
          ws = on ? ws + w_in : ws;
        };
        uint32_t j0 = 0;
        if (all_count) {
          // nearly every group: nothing of it is redistributed, every entry counts
          for (; j0 + 4u <= M; j0 += 4u) {
            float4 c[4];
            lds_read_4f4(col + j0, c);
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
              sum.x += (c[u].x + 0.0f) * w_in; sum.y += (c[u].y + 0.0f) * w_in;
              sum.z += (c[u].z + 0.0f) * w_in; sum.w += (c[u].w + 0.0f) * w_in;
              ws += w_in;
            }
          }
          for (; j0 < M; ++j0) {
            const float4 c = lds_read_f4(col + j0);
            sum.x += (c.x + 0.0f) * w_in; sum.y += (c.y + 0.0f) * w_in; sum.z += (c.z + 0.0f) * w_in; sum.w += (c.w + 0.0f) * w_in;
            ws += w_in;
          }
        } else {
          for (; j0 + 4u <= M; j0 += 4u) {
            float4 c[4];
            lds_read_4f4(col + j0, c);
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) add(c[u], (bits >> (j0 + u)) & 1ull);
          }
          for (; j0 < M; ++j0) add(lds_read_f4(col + j0), (bits >> j0) & 1ull);
        }
        // the float4 after the AOVs holds the weight, what follows it (padding of the record) nothing
        if (k == n_aovs) sum = make_float4(ws, 0.f, 0.f, 0.f);
        else if (k > n_aovs) sum = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 *dst = dummy;
        if (mine) {
          const uint64_t pp = pix0 + pi;
          const int px = V.pixel_x0 + (int)(pp % V.pixels_per_row);
          const int py = V.pixel_y0 + (int)(pp / V.pixels_per_row) * (int)V.pixel_row_stride;
          dst = dir4 + ((uint64_t)px + (uint64_t)py * xres) * q + k;
        }
        *dst = sum;
      }
      // this group's slot is free (its values are summed): the group after the next goes there
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
#ifdef LENTIL_TIMELINE
      const uint64_t tp3 = clock64();
#endif
      if (tile + R < run1) issue(tile + R, si);
      si = si + 1u == R ? 0u : si + 1u;
#ifdef LENTIL_TIMELINE
      const uint64_t tp4 = clock64();
#endif
      wq.end_tile(a);
#ifdef LENTIL_TIMELINE
      tacc[0] += tp1 - tp0; tacc[1] += tp2 - tp1; tacc[2] += tp3 - tp2; tacc[3] += tp4 - tp3; tacc[4] += clock64() - tp4; tacc[5] += 1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#ifdef LENTIL_TIMELINE
  if (lane == 0) for (int i = 0; i < 6; ++i) dbg_add(20 + i, tacc[i]);
#endif
  wq.finish(a);
  scan_block_done(a);
}

// K1+K2+K6 for ragged footprints (explicit per-visit pixel): lane per visit, fp32 atomics for the
// direct accumulation.
// K1+K2+K6 for frames with extra AOVs (visits_per_pixel <= 64).  A wave takes ppt = 64 / M whole pixels =
// at most 64 visits per step: every lane issues all 5 + K column loads of its visit at once (the only way a
// wide record gets enough bytes in flight), weights them, and parks one float4 per gaussian AOV in LDS.
// Lane (pixel, aov) then adds that pixel's M entries in iterator order -- the reference's summation order --
// into a record tile, and the tile's records go to the accumulators with coalesced float4 accesses.
__global__ __launch_bounds__(256) void scan_uniform_multi_kernel(ScanArgs a) {
  extern __shared__ float4 smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t waves_per_block = blockDim.x >> 6;
  const VisitsDev &V = a.V;
  const uint32_t M = V.visits_per_pixel;
  const uint32_t ppt = a.ppt;                 // pixels per step, ppt * M <= 64
  const uint32_t TV = ppt * M;
  const uint32_t n_aovs = a.F.n_aovs;
  const uint32_t q = a.F.stride >> 2;         // float4 per record
  // per wave: [n_aovs][kMultiPlane] float4 staged values, [64] float weights, [ppt][stride] record tile, then the queue
  const size_t wave_f4 = (size_t)n_aovs * kMultiPlane + 16u + (size_t)ppt * q;
  float4 *sval = smem + (size_t)wave * wave_f4;
  float *sw = reinterpret_cast<float *>(sval + (size_t)n_aovs * kMultiPlane);
  float4 *srec = sval + (size_t)n_aovs * kMultiPlane + 16u;
  uint2 *qmem = reinterpret_cast<uint2 *>(smem + (size_t)waves_per_block * wave_f4);
  WaveQueue wq;
  wq.init(qmem + (size_t)wave * kWaveQueueLds);

  const uint64_t n_pixels = (V.n + M - 1) / M;
  const uint64_t n_tiles = a.tile_end;
  const uint64_t wave_global = a.tile_begin + (uint64_t)blockIdx.x * waves_per_block + wave;
  const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;
  const uint32_t xres = a.P.xres;
  float4 *acc4 = reinterpret_cast<float4 *>(a.F.acc);

  for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_stride) {
    const uint64_t pix0 = tile * ppt;
    const uint64_t v = pix0 * M + lane;
    const bool valid = (lane < TV) && (v < V.n);
    bool flagged = false;
    int samples = 0;
    float w = 0.f;
    float depth = 0.f;
    bool feeds_dbg = false;       // lentil_debug: a visit that failed only the inside-the-lens test (visit_feeds_debug_directly)
    float4 val[LENTIL_MAX_AOVS];
    // the tile's pixel records (ppt * q float4; the first two per lane -- all of them at 9 visits per pixel and
    // 9 AOVs): requested together with the visit columns, used at the end of the step
    const uint64_t left = n_pixels - pix0;
    const uint32_t np_tile = (uint32_t)(left < ppt ? left : ppt);
    float4 *rdst[2] = {nullptr, nullptr};
    float4 rcur[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t i = lane + 64u * u;
      rcur[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < np_tile * q) {
        const uint32_t pi = i / q, j = i - pi * q;
        const uint64_t pp = pix0 + pi;
        const int px = V.pixel_x0 + (int)(pp % V.pixels_per_row);
        const int py = V.pixel_y0 + (int)(pp / V.pixels_per_row) * (int)V.pixel_row_stride;
        rdst[u] = acc4 + ((uint64_t)px + (uint64_t)py * xres) * q + j;
        rcur[u] = *rdst[u];
      }
    }
    if (valid) {
      const float4 rgba = V.rgba[v];
      const float4 pz = V.pos_z[v];
      const float4 vi = V.volume_ignore[v];
      const float4 tr = V.transmission[v];
#pragma unroll
      for (uint32_t k = 1; k < LENTIL_MAX_AOVS; ++k)
        if (k < n_aovs && !(a.F.closest_mask & (1u << k))) val[k] = V.extra[k - 1][v];
      const float invd = V.inv_density ? V.inv_density[v] : a.P.inverse_sample_density;
      depth = pz.w;
      if (visit_redistributes(a.P, a.lens_length, pz, vi, tr, invd, [&]() { return V.raydir_time[v]; }, V.cam)) {
        flagged = true;
        samples = visit_prologue(a.P, a.lens_length, rgba, pz, V.raydir_time[v], vi, tr, invd, V.cam).samples;
      } else {
        w = 1.0f * invd;                                // filter_weight * inv_density, lentil.h:949-953
        val[0] = rgba;
        if (a.F.zkey_dbg) feeds_dbg = visit_feeds_debug_directly(a.P, pz, vi, tr, invd, [&]() { return V.raydir_time[v]; }, V.cam);
      }
    }
    wq.push(flagged, (uint32_t)v, (uint32_t)samples, a);
    if (lane < TV) {
      sw[lane] = w;
#pragma unroll
      for (uint32_t k = 0; k < LENTIL_MAX_AOVS; ++k) {
        if (k < n_aovs && !(a.F.closest_mask & (1u << k))) {
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (w != 0.0f) x = make_float4((val[k].x + 0.0f) * w, (val[k].y + 0.0f) * w, (val[k].z + 0.0f) * w, (val[k].w + 0.0f) * w);
          sval[(size_t)k * kMultiPlane + lane] = x;
        }
      }
    }
    // closest AOVs: a pixel's own (non-redistributed) visits compete with their depth
    if (a.F.zkey && valid && w != 0.0f) {
      const uint64_t pp = pix0 + lane / M;
      const int px = V.pixel_x0 + (int)(pp % V.pixels_per_row);
      const int py = V.pixel_y0 + (int)(pp / V.pixels_per_row) * (int)V.pixel_row_stride;
      atomicMin(a.F.zkey + ((uint64_t)px + (uint64_t)py * xres), closest_key_of(a.ctr, depth, visit_gid(V, (uint32_t)v)));
    }
    if (a.F.zkey_dbg && valid && w != 0.0f && feeds_dbg) {
      const uint64_t pp = pix0 + lane / M;
      const int px = V.pixel_x0 + (int)(pp % V.pixels_per_row);
      const int py = V.pixel_y0 + (int)(pp / V.pixels_per_row) * (int)V.pixel_row_stride;
      atomicMin(a.F.zkey_dbg + ((uint64_t)px + (uint64_t)py * xres), closest_key_of(a.ctr, depth, visit_gid(V, (uint32_t)v)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

    // sums: one lane per (pixel, AOV), M entries in iterator order
    for (uint32_t idx = lane; idx < np_tile * q; idx += 64u) srec[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    for (uint32_t idx = lane; idx < np_tile * n_aovs; idx += 64u) {
      const uint32_t pi = idx / n_aovs, k = idx - pi * n_aovs;
      if (a.F.closest_mask & (1u << k)) continue;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      float ws = 0.f;
      // entries of visits that do not add to their own pixel are +0 (weight and values): adding them changes nothing
      for (uint32_t j = 0; j < M; ++j) {
        const float4 c = sval[(size_t)k * kMultiPlane + pi * M + j];
        sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
        ws += sw[pi * M + j];
      }
      srec[(size_t)pi * q + k] = sum;
      if (k == 0) reinterpret_cast<float *>(srec + (size_t)pi * q)[4u * n_aovs] = ws;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

    // records -> accumulators (a row of the stream ends after pixels_per_row pixels, the frame row is wider)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t i = lane + 64u * u;
      if (i < np_tile * q) {
        const float4 add = srec[i];
        float4 cur = rcur[u];
        cur.x += add.x; cur.y += add.y; cur.z += add.z; cur.w += add.w;
        *rdst[u] = cur;
      }
    }
    // tiles of more than two float4 per lane (few visits per pixel, many AOVs): the rest the plain way
    for (uint32_t i = lane + 128u; i < np_tile * q; i += 64u) {
      const uint32_t pi = i / q, j = i - pi * q;
      const uint64_t pp = pix0 + pi;
      const int px = V.pixel_x0 + (int)(pp % V.pixels_per_row);
      const int py = V.pixel_y0 + (int)(pp / V.pixels_per_row) * (int)V.pixel_row_stride;
      float4 *dst = acc4 + ((uint64_t)px + (uint64_t)py * xres) * q + j;
      const float4 add = srec[i];
      float4 cur = *dst;
      cur.x += add.x; cur.y += add.y; cur.z += add.z; cur.w += add.w;
      *dst = cur;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    wq.end_tile(a);
  }
  wq.finish(a);
  scan_block_done(a);
}

__global__ __launch_bounds__(256) void scan_ragged_kernel(ScanArgs a) {
  const VisitsDev &V = a.V;
  __shared__ uint2 s_queue[4 * kWaveQueueLds];
  WaveQueue wq;
  wq.init(s_queue + (threadIdx.x >> 6) * kWaveQueueLds);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t n_round = a.v_begin + ((a.v_end - a.v_begin + 63ull) & ~63ull);
  uint32_t rmin = 0x7FFFFFFFu, rmax_p1 = 0u;
  for (uint64_t v = a.v_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_round; v += stride) {
    bool flagged = false;
    int samples = 0;
    if (v < a.v_end) {
      const float4 rgba = V.rgba[v];
      const float invd = V.inv_density ? V.inv_density[v] : a.P.inverse_sample_density;
      const float4 pz = V.pos_z[v], vi = V.volume_ignore[v], tr = V.transmission[v];
      if (visit_redistributes(a.P, a.lens_length, pz, vi, tr, invd, [&]() { return V.raydir_time[v]; }, V.cam)) {
        flagged = true;
        samples = visit_prologue(a.P, a.lens_length, rgba, pz, V.raydir_time[v], vi, tr, invd, V.cam).samples;
      } else {
        int px, py;
        visit_pixel(V, v, px, py);
        rmin = (uint32_t)py < rmin ? (uint32_t)py : rmin;
        rmax_p1 = (uint32_t)py + 1u > rmax_p1 ? (uint32_t)py + 1u : rmax_p1;
        const uint64_t lin = (uint64_t)px + (uint64_t)py * a.P.xres;
        const float w = 1.0f * invd;
        float *d = reinterpret_cast<float *>(a.F.aov(lin, 0));
        atomicAdd(d + 0, (rgba.x + 0.0f) * w);
        atomicAdd(d + 1, (rgba.y + 0.0f) * w);
        atomicAdd(d + 2, (rgba.z + 0.0f) * w);
        atomicAdd(d + 3, (rgba.w + 0.0f) * w);
        atomicAdd(a.F.wt(lin), w);
        if (a.F.zkey) atomicMin(a.F.zkey + lin, closest_key_of(a.ctr, V.pos_z[v].w, visit_gid(V, (uint32_t)v)));
        if (a.F.zkey_dbg && visit_feeds_debug_directly(a.P, pz, vi, tr, invd, [&]() { return V.raydir_time[v]; }, V.cam))
          atomicMin(a.F.zkey_dbg + lin, closest_key_of(a.ctr, pz.w, visit_gid(V, (uint32_t)v)));
        for (uint32_t k = 0; k < V.n_extra; ++k) {
          if (a.F.closest_mask & (2u << k)) continue;
          const float4 c = V.extra[k][v];
          float *dk = reinterpret_cast<float *>(a.F.aov(lin, k + 1));
          atomicAdd(dk + 0, (c.x + 0.0f) * w);
          atomicAdd(dk + 1, (c.y + 0.0f) * w);
          atomicAdd(dk + 2, (c.z + 0.0f) * w);
          atomicAdd(dk + 3, (c.w + 0.0f) * w);
        }
      }
    }
    wq.push(flagged, (uint32_t)v, (uint32_t)samples, a);
    wq.end_tile(a);
  }
  wq.finish(a);
  flush_row_range(a.ctr, rmin, rmax_p1);
  scan_block_done(a);
}
// K1+K2+K6 for ragged streams, ordered.  A capturing filter_pixel appends the samples of ONE pixel, in iterator order, as a
// run of consecutive visits (src/lentil_filter.cpp:105 walks them; include/lentil_bridge.h, lentil_stage_append); the
// runs of different pixels arrive in whatever order the render threads got to them.  scan_ragged_kernel above adds
// every visit with atomics of its own: the own-pixel sums then take their roundings in arbitrary order (2e-5 of the
// reference's sequential sum).  Here a run is added up in its order -- the reference's (src/lentil.h:938-955) -- by the
// lane of its first visit, from values the wave staged in LDS with coalesced loads, and lands in the pixel record as one
// atomic add per float: on a cleared record that is the sequential sum, bit for bit (a pixel is filtered once per
// frame, so its record meets one run).  A run that leaves the wave's 64 visits is followed into the next ones (their
// own lanes see that they continue a run and add nothing) -- for kRunFollow more groups of 64: a single lane walking a
// run of thousands of visits (a degenerate pixel column, very high AA) would serialise the scan, so the part of a run
// further in than that is summed per group of 64, in order, by the group's first lane and added as one term per group
// (a run of up to 320 visits, AA 17, is the reference's sequential sum bit for bit; a longer one is a blocked sum of the
// same terms, closer to the exact sum than the sequential one and inside the 1e-5 bar).  A stream without runs
// degenerates to a visit per run: what scan_ragged_kernel does.
constexpr uint64_t kRunFollow = 4;
__global__ __launch_bounds__(256) void scan_runs_kernel(ScanArgs a) {
  const VisitsDev &V = a.V;
  __shared__ uint2 s_queue[4 * kWaveQueueLds];
  __shared__ float4 s_val[4][64];
  __shared__ float s_w[4][64];
  __shared__ uint32_t s_pix[4][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  WaveQueue wq;
  wq.init(s_queue + wave * kWaveQueueLds);
  const uint64_t n_groups = (a.v_end - a.v_begin + 63ull) / 64ull;
  const uint64_t wave_global = (uint64_t)blockIdx.x * 4u + wave, wave_stride = (uint64_t)gridDim.x * 4u;
  uint32_t rmin = 0x7FFFFFFFu, rmax_p1 = 0u;
  constexpr uint32_t kNoPixel = 0xFFFFFFFFu;
  // what visit u adds to its own pixel: weight (0: redistributed) and weighted value of column `col`
  auto own = [&](uint64_t u, const float4 *col, float &w) -> float4 {
    const float invd = V.inv_density ? V.inv_density[u] : a.P.inverse_sample_density;
    const bool red = visit_redistributes(a.P, a.lens_length, V.pos_z[u], V.volume_ignore[u], V.transmission[u], invd,
                                         [&]() { return V.raydir_time[u]; }, V.cam);
    w = red ? 0.0f : 1.0f * invd;
    if (red) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 c = col[u];
    return make_float4((c.x + 0.0f) * w, (c.y + 0.0f) * w, (c.z + 0.0f) * w, (c.w + 0.0f) * w);
  };
  for (uint64_t g = wave_global; g < n_groups; g += wave_stride) {
    const uint64_t v = a.v_begin + g * 64ull + lane;
    const bool valid = v < a.v_end;
    const uint32_t pix = valid ? V.pixel[v] : kNoPixel;
    uint32_t prev = __shfl_up(pix, 1);
    if (lane == 0) prev = v > 0 ? V.pixel[v - 1] : kNoPixel;      // (a run may have begun in the chunk before)
    if (v == 0) prev = kNoPixel;
    const bool start = valid && pix != prev;
    // (the run's first lane covers its own group of 64 and the kRunFollow after it: a visit whose run began before that)
    bool beyond = false;
    if (valid && !start && (v >> 6) > kRunFollow) {
      const uint64_t edge = ((v >> 6) - kRunFollow) << 6;          // first visit of the earliest group an owner could sit in
      beyond = V.pixel[edge - 1] == pix && V.pixel[edge] == pix;
      // (two probes first: all but the visits of a long run stop here.  A stream may bring a pixel back in several runs --
      // captured samples in arbitrary order --, so what decides is that every visit in between is the pixel's as well)
      if (beyond)
        for (uint64_t u = v - 1; u > edge; --u)
          if (V.pixel[u] != pix) { beyond = false; break; }
    }
    bool flagged = false;
    int samples = 0;
    float w = 0.f;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      const float4 rgba = V.rgba[v];
      const float invd = V.inv_density ? V.inv_density[v] : a.P.inverse_sample_density;
      const float4 pz = V.pos_z[v], vi = V.volume_ignore[v], tr = V.transmission[v];
      if (visit_redistributes(a.P, a.lens_length, pz, vi, tr, invd, [&]() { return V.raydir_time[v]; }, V.cam)) {
        flagged = true;
        samples = visit_prologue(a.P, a.lens_length, rgba, pz, V.raydir_time[v], vi, tr, invd, V.cam).samples;
      } else {
        w = 1.0f * invd;
        val = make_float4((rgba.x + 0.0f) * w, (rgba.y + 0.0f) * w, (rgba.z + 0.0f) * w, (rgba.w + 0.0f) * w);
        if (a.F.zkey) {
          const uint32_t px = pix & 0xFFFFu, py = pix >> 16;
          atomicMin(a.F.zkey + ((uint64_t)px + (uint64_t)py * a.P.xres), closest_key_of(a.ctr, pz.w, visit_gid(V, (uint32_t)v)));
        }
        if (a.F.zkey_dbg && visit_feeds_debug_directly(a.P, pz, vi, tr, invd, [&]() { return V.raydir_time[v]; }, V.cam)) {
          const uint32_t px = pix & 0xFFFFu, py = pix >> 16;
          atomicMin(a.F.zkey_dbg + ((uint64_t)px + (uint64_t)py * a.P.xres), closest_key_of(a.ctr, pz.w, visit_gid(V, (uint32_t)v)));
        }
      }
    }
    wq.push(flagged, (uint32_t)v, (uint32_t)samples, a);
    wq.end_tile(a);
    s_pix[wave][lane] = pix;
    s_w[wave][lane] = w;
    const uint32_t px = pix & 0xFFFFu, py = pix >> 16;
    const uint64_t lin = (uint64_t)px + (uint64_t)py * a.P.xres;
    if (start) {
      rmin = py < rmin ? py : rmin;
      rmax_p1 = py + 1u > rmax_p1 ? py + 1u : rmax_p1;
    }
    // the beauty, then every gaussian AOV column: staged by all lanes, summed per run in order by the run's first lane
    for (uint32_t k = 0; k <= V.n_extra; ++k) {
      if (k && (a.F.closest_mask & (1u << k))) continue;          // closest AOVs are gathered from the winners later
      if (k) {
        val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && w != 0.0f) {
          const float4 c = V.extra[k - 1][v];
          val = make_float4((c.x + 0.0f) * w, (c.y + 0.0f) * w, (c.z + 0.0f) * w, (c.w + 0.0f) * w);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      s_val[wave][lane] = val;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      if (start) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        float ws = 0.f;
        uint32_t j = lane;
        for (; j < 64u && s_pix[wave][j] == pix; ++j) {
          const float4 c = s_val[wave][j];
          sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
          ws += s_w[wave][j];
        }
        if (j == 64u) {
          // the run goes on beyond this wave's visits
          const float4 *col = k ? V.extra[k - 1] : V.rgba;
          const uint64_t u0 = a.v_begin + g * 64ull + 64ull;
          const uint64_t u1 = (u0 + kRunFollow * 64ull) < V.n ? (u0 + kRunFollow * 64ull) : V.n;
          for (uint64_t u = u0; u < u1 && V.pixel[u] == pix; ++u) {
            float wu;
            const float4 c = own(u, col, wu);
            sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
            ws += wu;
          }
        }
        float *d = reinterpret_cast<float *>(a.F.aov(lin, k));
        // (a visit that adds nothing left +0; an accumulator is never -0: adding the sum to a cleared record stores it)
        atomicAdd(d + 0, sum.x); atomicAdd(d + 1, sum.y); atomicAdd(d + 2, sum.z); atomicAdd(d + 3, sum.w);
        if (k == 0) atomicAdd(a.F.wt(lin), ws);
      } else if (beyond && lane == 0u) {
        // (what an owner covers ends on a group boundary, so the visits of a run beyond it begin at lane 0 of their group
        // and are contiguous: lane 0 sums this group's part in order and adds it as one term)
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        float ws = 0.f;
        for (uint32_t j = 0; j < 64u && s_pix[wave][j] == pix; ++j) {
          const float4 c = s_val[wave][j];
          sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
          ws += s_w[wave][j];
        }
        float *d = reinterpret_cast<float *>(a.F.aov(lin, k));
        atomicAdd(d + 0, sum.x); atomicAdd(d + 1, sum.y); atomicAdd(d + 2, sum.z); atomicAdd(d + 3, sum.w);
        if (k == 0) atomicAdd(a.F.wt(lin), ws);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
  wq.finish(a);
  flush_row_range(a.ctr, rmin, rmax_p1);
  scan_block_done(a);
}

// ---------------------------------------------------------------------------------------
// Streamed pass, stage two: publish_kernel.  A handful of one-wave blocks that follow the scan's range queue and
// turn every announced work-list entry into an item (publish_item): header, progress record, result space, the
// first batch's solve tasks.  In a kernel of its own so that the scan's hot loop carries none of it (the scan
// kernels sit at the SGPR limit; six more VGPRs would cost them a wave per SIMD beside the solve kernel).
// A wave draws a ticket on the range queue and polls its slot; the last scan block to sign off puts an end marker
// behind the last range for every publisher, the last publisher one behind the last task for every solve wave.
// ---------------------------------------------------------------------------------------
struct PublishArgs {
  lentil_params P;
  VisitsDev V;
  StreamPub S;
  DevCounters *ctr;
  const uint2 *work;
  uint64_t work_cap;
  const uint64_t *ranges;
  uint32_t range_cap;
  uint32_t end_tasks;      // solve waves of the first round (both launches)
  uint64_t stuck_ticks;    // how long a wave waits for a queue slot before it declares the pass stuck (100 MHz ticks; 0: kStuckTicks)
};

// (At most 96 VGPRs: a publisher wave must fit into what a CU has left beside two resident solve blocks and a scanning block --
// 512 - 2 x 168 - 80 = 96 registers per SIMD lane -- whichever of the pass's kernels the dispatcher places first; with the
// first-batch model's blend inlined the kernel took 108, i.e. 112, and a pass whose publishers found no room anywhere waited
// for them until its time-out: the odd stalled pass of round 5's first sessions.  Capped at 64 it spilt 45 registers into
// the blend's inner loop and the publishers fell 0.3 ms behind the scan: the first round then ended 0.14 ms later.)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) void publish_kernel(PublishArgs a) {
  LENTIL_TL_SPAN(SPAN_PUBLISH);
  const uint32_t lane = threadIdx.x;
  while (true) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&a.ctr->range_head, 1u);
    const uint32_t ticket = __builtin_amdgcn_readfirstlane(t);
    if (ticket >= a.range_cap) break;
    uint64_t rec = 0;
    bool over = false;
    uint32_t naps = 1u;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (true) {
      if (lane == 0) rec = ld_coherent64(a.ranges + ticket);
      rec = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(rec >> 32)) << 32) |
            (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)rec);
      if (((uint32_t)(rec >> 32) >> kTaskTagShift) == a.S.epoch) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > (a.stuck_ticks ? a.stuck_ticks : kStuckTicks)) {
        if (lane == 0) atomicCAS(&a.ctr->stuck, 0u, 1u | (ticket << 2));      // (the first to give up is the one worth knowing)
        over = true;
        break;
      }
      for (uint32_t i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(64);      // 2 us, doubling to 15 us
      if (naps < 8u) naps <<= 1;
    }
    if (over) break;
    const uint32_t base = (uint32_t)rec, n = (uint32_t)(rec >> 32) & ((1u << kTaskTagShift) - 1u);
    if (n == kEndRange) break;                                  // behind the last range
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
      const uint32_t item = base + i0 + lane;
      const bool mine = i0 + lane < n && (uint64_t)item < a.work_cap;
      uint64_t wi = 0;
      if (mine) wi = ld_coherent64(a.work + item);              // written through by the scan before the range record
      float cs[3] = {0.0f, 0.0f, 0.0f};
      if (mine) {
        const uint32_t v = (uint32_t)wi;
        visit_camera_space(a.P, a.V.pos_z[v], [&]() { return a.V.raydir_time[v]; }, cs, a.V.cam);
      }
      const uint32_t samples = (uint32_t)(wi >> 32);
      uint32_t count = first_batch_hi(samples, (uint32_t)a.S.retries, a.S.extra_num, a.S.extra_const);
      if (a.S.model.land) {
        // the first batch from the lens and the frame: the wave looks at its items one after the other, all lanes on one item
        unsigned long long todo = __ballot(mine);
        while (todo) {
          const int j = __builtin_ctzll(todo);
          todo &= todo - 1ull;
          // (read as scalars: the cell, its eight node indices and weights then live in SGPRs, not in sixteen VGPRs)
          const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs[0]), j));
          const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs[1]), j));
          const float z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs[2]), j));
          const BatchEstimate e = batch_estimate(a.S.model, x, y, z, lane);
          if ((int)lane == j) count = batch_from_estimate(e, samples, (uint32_t)a.S.retries, count, a.S.model.margin16);
        }
      }
      if (mine) publish_item(a.P, a.V, a.S, a.ctr, item, make_uint2((uint32_t)wi, samples), count, cs);
    }
  }
  // everything this wave published has arrived (publish_item waits for its stores); sign off
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t last = 0;
  if (lane == 0) last = atomicAdd(&a.ctr->publishers_done, 1u) == gridDim.x - 1u ? 1u : 0u;
  if (!__builtin_amdgcn_readfirstlane(last)) return;
  // the last publisher: the task queue is complete -- unless items are still open (extension, ItemLive): then whoever
  // closes the last of them writes the markers (live_close_queue)
  // One end marker for every solve wave that may hold a ticket.
  const uint32_t n = ld_coherent32(&a.ctr->n_tasks[0]);
  for (uint32_t i = lane; i < a.end_tasks; i += 64u)
    if ((uint64_t)n + i < a.S.task_cap)
      st_agent64(reinterpret_cast<uint64_t *>(a.S.tasks0 + n + i) + 1,
                 (uint64_t)(kEndCount | (a.S.epoch << kTaskTagShift)) << 32);
  // extension (ItemLive): its queue ends when no item is open any more -- now, or when a solve wave closes the last one
  if (a.S.live) {
    uint32_t open = 0;
    if (lane == 0) open = ld_coherent32(&a.ctr->items_open);
    if (__builtin_amdgcn_readfirstlane(open) != 0u) return;
    if (lane == 0) st_agent32(&a.ctr->queue_final, 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t ne = ld_coherent32(&a.ctr->ext_n);
    for (uint32_t i = lane; i < a.S.ext_keepers; i += 64u)
      if ((uint64_t)ne + i < a.S.task_cap)
        st_agent64(reinterpret_cast<uint64_t *>(a.S.ext_q + ne + i) + 1, (uint64_t)(kEndCount | (a.S.epoch << kTaskTagShift)) << 32);
  }
}

// ---------------------------------------------------------------------------------------
// K3/K4/K5: draws -- "solve once".
//
// Reference semantics (src/lentil_filter.cpp:248-299, src/lentil.h:592-648): attempt n of an item
// tries aperture draws seeded tea<8>(px*py+px, n+t), t = 0..vignetting_retries, until one traces
// through the lens; then the sensor point must fall inside the frame, else the attempt fails.  The
// accepted draws are the first `samples` successful attempts in attempt order, n < 5*samples.
//
// The seed -- and with it the whole backward trace -- depends on n+t only, so
// try t of attempt n IS try 0 of attempt n+t.  Define for m = 0,1,2,...
//     R(m) = FAIL       the trace with seed (.., m) fails (transmittance <= 0 / pupil clipping)
//            OUT        it succeeds but lands outside the frame (or NaN)
//            pixel p    it succeeds and lands on pixel p
// then attempt n yields the first non-FAIL entry of R(n .. n+retries) (and fails if that is OUT or
// there is none).  Every R(m) is computed exactly once, all of them independently:
//   solve kernel : fills R for a batch of m per item.  Lanes are independent Newton solves advanced
//                  one iteration per round; a lane that finishes is refilled with the next (item, m)
//                  unit, so neither iteration counts (10..100) nor failing draws stall a wave, and the
//                  parallelism is the number of draws, not the number of items.
//   accept kernel: one wave per item walks the attempts in order, applies the rule above, accepts
//                  successes until `samples` are reached, splats them, and -- if draws are still
//                  missing because attempts failed -- schedules the next batch of m (over-provisioned;
//                  the surplus is simply never accepted).
// Thin lens (no retries): R(m) is the closed-form draw of attempt m.
// ---------------------------------------------------------------------------------------
constexpr int kMaxBokehRows = 2048;
constexpr uint32_t kCodeFail = 0xFFFFFFFFu;
constexpr uint32_t kCodeOut = 0xFFFFFFFEu;
// R(m) of a solve that was parked for solve_slow_kernel while the first accept of the pass may already run
// (DrawArgs::slow_indirect): 0xFE000000 | slot in the straggler queue; the straggler's result goes to the queue
// record (SlowRec::result), the mark stays.  Pixel indices are below 0xFE000000 (lentil_hip_alloc_frame checks).
constexpr uint32_t kCodePendingBase = 0xFE000000u;
LD_DEV bool code_is_pending(uint32_t c) { return (c & 0xFF000000u) == kCodePendingBase; }

// A solve that is still running after `slow_at` Newton iterations (about one in a thousand) is parked here by
// solve_po_kernel and finished by solve_slow_kernel, a whole wave per solve: the complete loop state, so that
// the iteration sequence continues bit for bit.
struct SlowRec {           // 128 B
  double tx, ty, tz, ap_x, ap_y;
  double x, y, dx, dy, sqr_err, sqr_ap_err;
  int32_t k, error;
  uint32_t res_idx, chan;  // chan: bits 0-1 wavelength channel, bit 8 parity of the result pool, bit 9 result goes to `result` (pending mark in the pool)
  uint32_t result;         // slow_indirect: R(m) once the straggler is through
  uint32_t pad[3];
  uint64_t tag;            // live queue (DrawArgs::slow_live): (pass epoch << 8 | round) << 32 | 1 record, 2 end marker
};
constexpr uint32_t kSlowParBit = 1u << 8, kSlowIndirectBit = 1u << 9;
static_assert(sizeof(SlowRec) == 128, "SlowRec is 128 bytes");
constexpr uint64_t kSlowRecord = 1, kSlowEnd = 2;

struct DrawArgs {
  lentil_params P;
  const DevLens *lens;     // header, global memory (null for the thin lens)
  const DevTerm *terms;    // global memory
  DevBokeh bokeh;
  VisitsDev V;
  FrameDev F;
  const uint2 *work;
  uint64_t n_items;
  DevCounters *ctr;
  ItemHdr *hdr;
  ItemProg *prog;
  Task *tasks[2];
  uint32_t task_cap;
  uint32_t *active[2];
  uint32_t *pool[2];
  uint64_t pool_cap;
  // blind mode: the host enqueues the chunk's draw rounds without waiting for the scan; n_items is then the
  // capacity of the item buffers, the real count is the scan's work_count, and prep_items_kernel checks that
  // everything fits (DevCounters::fallback otherwise: nothing is emitted, the host redoes the chunk)
  uint32_t blind;
  uint64_t work_cap;       // visits in the chunk = capacity of its work list segment
  // first batch: draws beyond samples + retries solved up front (in 1/256 of samples, plus a constant), applied
  // while the chunk's draw sum is below extra_below -- saves the second round where latency matters
  uint32_t extra_num, extra_const;
  uint64_t extra_below;
  SlowRec *slow;           // straggler queue of the current round (null: stragglers stay in their lanes)
  uint32_t slow_cap;
  int32_t slow_at;         // Newton iterations after which a solve counts as a straggler
  int32_t slow_max_lanes;  // a dry wave parks only when at most this many of its lanes are still busy
  int32_t slow_nap_max;    // an idle straggler wave sleeps 1.7 us x 1, 2, 4 ... up to this many between polls of its queue slot (0: 8)
  int32_t slow_prio;       // s_setprio of the straggler kernel's waves (their chains of iterations end the pass; LENTIL_SLOW_PRIO)
  int32_t dispatch_probe;  // LENTIL_DISPATCH_PROBE: DevCounters::probe_*
  int32_t round;           // solve/accept round of the chunk (0 = first batch)
  int32_t slow_from_round; // parking starts with this round: the first round's own ramp-down hides most of its stragglers
  uint64_t slow_below;     // ... in chunks whose draw sum is below this (where the end of a round is what costs;
                           //     a long round absorbs its stragglers and a wave per solve would only cost throughput)
  lentil_draw_record *log;
  uint64_t log_cap;
  unsigned long long *log_count;   // shared by all chunks
  int32_t retries;         // vignetting_retries for PO, 0 for the thin lens
  int32_t parity;
  // streamed pass (solve_po_kernel<.., kStream>): tag of this pass's task slots, publish_kernel waves that will sign off
  uint32_t epoch;
  uint32_t instance;       // 0: beside the scan, 1: after it (debug statistics)
  // chromatic aberration of the polynomial-optics path (src/lentil_filter.cpp:255-268): three traces per
  // attempt, one wavelength each; n_channels is 1 when abb_chromatic == 0
  int32_t n_channels;
  int32_t chroma_weights;  // abb_chromatic > 0: channel c only feeds colour component c, three-fold
  double lambda[3];
  // live straggler queue: solve_slow_kernel runs BESIDE the solve kernel and takes parked solves as they come (every
  // solve that reaches slow_at iterations is parked, not only those of waves running dry); the last solve wave to
  // exit -- solver_waves_total of them, over all launches of the round -- writes one end marker per straggler wave
  int32_t slow_live;
  uint32_t slow_waves;     // waves of the straggler kernel (one end marker each)
  // whoever fills the task queue a kStream solve kernel reads: publish_kernel's waves in the first round, the accept
  // kernel's blocks when a round's solves run beside the accept that schedules them.  Complete when all have signed off.
  const unsigned int *producers_done;
  uint32_t producers_total;
  // accept kernel: the next round's tasks go out tagged and through atomics (its solve kernel is already running and
  // takes them as they come), end_tasks end markers behind them from the last block
  int32_t emit_live;
  uint32_t end_tasks;
  int32_t no_reset;        // solve kernel: leave the other parity's queues alone (the accept beside it is using them)
  // Decoupled streamed pass: ONE straggler queue and one solve_slow_kernel launch serve the first round and the second
  // (which runs beside the first accept): both rounds' solve kernels park into counter set `slow_q` under round tag
  // `slow_round`, only the launch with slow_close set closes the queue, and the first round parks `slow_indirect`:
  // the first accept does not wait for the stragglers (accept_item<1>).  Defaults (-1, -1, 1, 0): a queue per round.
  int32_t slow_q, slow_round, slow_close, slow_indirect;
  uint32_t margin_low_rate;   // ... and twice that below this success rate (sixteenths; 0: never)
  uint32_t batch_margin16;    // a follow-up batch is the shortfall over the item's success rate so far + this many sixteenths + 32
  uint32_t unknown_credit;    // accept_item<1>: eighths of the known attempts' success rate credited to the unknown ones (0: none)
  int32_t slow_after_producers;   // first round of a streamed pass: park only once the scan and its publishers have ended
  int32_t slow_dry_only;      // live queue: park as a plain round does (waves running dry, their last slow_max_lanes lanes)
  int32_t slow_crowd_stays;   // live queue: where more than slow_max_lanes lanes of a wave are past slow_at at once, none is parked
  int32_t accept_narrow;      // accept kernels: 256-attempt steps (accept_item) where accept_item_wide would apply
  // test hook (LENTIL_INJECT_STALL): the first accept of a streamed pass never closes the queue it feeds, so the second round's
  // resident solve waves give up after kStuckTicks -- a pass stalled with draws already accepted, which the host must recover
  int32_t inject_stall;
  ItemLive *live;             // streamed pass with extension (ItemLive): the first round's solve kernel and the first accept
  Task *ext_q;                // ... the queue of the batches it appends, served by its first ext_keeper_blocks blocks
  uint32_t ext_keeper_blocks;
  uint32_t ext_end_tasks;     // ... end markers behind that queue (one per wave of those blocks)
  uint32_t ext_slack;         // ... successes beyond `samples` an item's results must show before it counts as served
  // Lean tail of a streamed pass with extension: no second round's solve kernels are in flight.  The accept behind the first
  // one (accept_kernel<2>) does nothing if the first one had to schedule tasks after all (n_tasks of its parity): the host then
  // runs that round the ordinary way and this accept after it.
  uint64_t stuck_ticks;       // how long a resident wave waits for a queue slot before it declares the pass stuck (0: kStuckTicks)
  int32_t lean_gate;
  // Round 6, the first accept BESIDE the first round's solves (accept_kernel<4>): the solve kernel writes its results through
  // (agent-scope atomics, like every other word another CU reads while a kernel runs), counts them per item
  // (ItemHdr::delivered) and pushes an item whose batch is complete onto `ready_q` (tagged slots, never cleared); the accept
  // -- launched behind the publishers, when the scan's LDS and registers are free -- draws tickets on that queue and walks the
  // items as they come, with the solve kernel still at work on the others.  What is left when the last solve wave exits is
  // the items completed last, not the whole frame's accept.
  int32_t early_accept;
  uint32_t ready_cap;
  uint64_t *ready_q;
  int32_t item_ready;         // streamed pass, lean tail: parked solves are counted per item (ItemHdr::parked / parked_done) for accept_kernel<3>
  int32_t lean_defer;         // ... and the first accept leaves an item that met parked solves to that accept whole: what it still
                              // needs is decided there, from the stragglers' results
};
LD_DEV uint32_t slow_queue(const DrawArgs &a) { return a.slow_q >= 0 ? (uint32_t)a.slow_q : (uint32_t)a.parity; }
LD_DEV uint64_t slow_tag(const DrawArgs &a, uint64_t what) {
  const uint32_t r = a.slow_round >= 0 ? (uint32_t)a.slow_round : (uint32_t)a.round;
  return ((uint64_t)((a.epoch << 8) | (r & 0xFFu)) << 32) | what;
}

// DrawArgs::early_accept: `cnt` results of `item` have been written through (and waited for); the item whose batch is complete with
// them goes onto the ready queue.  Called by one lane.
LD_DEV void ready_deliver(const DrawArgs &a, uint32_t item, uint32_t cnt) {
  const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(&a.hdr[item].issued), (unsigned long long)cnt << 32);
  const uint32_t issued = (uint32_t)old, delivered = (uint32_t)(old >> 32) + cnt;
  if (issued != 0u && delivered == issued) {
    const uint32_t slot = atomicAdd(&a.ctr->n_ready, 1u);
    if (slot < a.ready_cap) st_agent64(a.ready_q + slot, (uint64_t)item | ((uint64_t)((a.epoch << 8) | 1u) << 32));
  }
}

LD_DEV ItemVisit load_item_visit(const DrawArgs &a, uint32_t item, double lens_length) {
  return load_work_visit(a.P, a.V, a.work[item], lens_length);
}

// an item's progress record as an accept kernel takes it: with extension (ItemLive) the first round's batch ends where
// the solve kernel's last appended batch does
LD_DEV ItemProg load_prog(const DrawArgs &a, uint32_t item) {
  ItemProg pg = a.prog[item];
  if (a.live && a.round == 0) {
    const uint32_t hi = (uint32_t)a.live[item].hi_s;       // (low word: the batch's end; flags sit in the high word)
    if (hi > pg.m_hi && pg.m_hi != 0u) pg.m_hi = hi;
  }
  return pg;
}

// emit the solve tasks for m in [m_lo, m_hi) of `item` into the queues of round parity `par`
LD_DEV bool emit_tasks(const DrawArgs &a, uint32_t par, uint32_t item, uint32_t m_lo, uint32_t m_hi, uint32_t &res_off) {
  const uint32_t count = m_hi - m_lo;
  const uint32_t nch = (uint32_t)a.n_channels;
  // channel c's results of the batch live at res_off + c * count
  const unsigned long long off = atomicAdd(&a.ctr->pool_used[par], (unsigned long long)count * nch);
  const uint32_t nt = (count + 63u) / 64u;
  const uint32_t tb = atomicAdd(&a.ctr->n_tasks[par], nt * nch);
  if (off + (unsigned long long)count * nch > a.pool_cap || (unsigned long long)tb + nt * nch > a.task_cap) {
    atomicAdd(&a.ctr->overflow, 1ull);
    return false;
  }
  res_off = (uint32_t)off;
  if (a.emit_live) {
    // the round's solve kernel is running: slot halves through atomics, the half with the tag last (publish_item)
    for (uint32_t i = 0; i < nt * nch; ++i) st_agent64(a.tasks[par] + tb + i, (uint64_t)item | ((uint64_t)(m_lo + (i % nt) * 64u) << 32));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t i = 0; i < nt * nch; ++i) {
      const uint32_t c = i / nt, t = i - c * nt;
      const uint32_t n = (count - t * 64u) < 64u ? (count - t * 64u) : 64u;
      st_agent64(reinterpret_cast<uint64_t *>(a.tasks[par] + tb + i) + 1,
                 (uint64_t)((uint32_t)off + c * count + t * 64u) | ((uint64_t)(n | (c << 8) | (a.epoch << kTaskTagShift)) << 32));
    }
    return true;
  }
  for (uint32_t c = 0; c < nch; ++c) {
    for (uint32_t t = 0; t < nt; ++t) {
      Task k;
      k.item = item;
      k.m_base = m_lo + t * 64u;
      k.res_off = (uint32_t)off + c * count + t * 64u;
      k.count = ((count - t * 64u) < 64u ? (count - t * 64u) : 64u) | (c << 8);
      a.tasks[par][tb + c * nt + t] = k;
    }
  }
  return true;
}

// the queues of one round parity back to empty
LD_DEV void reset_round(DevCounters *c, uint32_t par) {
  c->n_tasks[par] = 0; c->task_head[par] = 0; c->n_active[par] = 0; c->active_head[par] = 0;
  c->pool_used[par] = 0;
  c->n_slow[par] = 0; c->slow_head[par] = 0; c->waves_done[par] = 0; c->waves_started[par] = 0; c->accept_done[par] = 0;
  c->accept_items_done[par] = 0; c->accept_final[par] = 0;
}
// keep_pool: items of the accept that follows still read results out of this parity's pool (ItemProg::p_lo) while that
// accept allocates the next batch's results in it: go on behind them
__global__ void reset_round_kernel(DevCounters *c, uint32_t par, uint32_t keep_pool) {
  LENTIL_TL_SPAN(SPAN_RESET);
  const unsigned long long used = c->pool_used[par];
  reset_round(c, par);
  if (keep_pool) c->pool_used[par] = used;
}

// one thread per item: header + first batch R(0 .. samples-1+retries)
__global__ __launch_bounds__(256) void prep_items_kernel(DrawArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t n_items = a.n_items;
  const unsigned long long sum_samples = a.ctr->sum_samples;
  if (a.blind) {
    const unsigned long long wc = a.ctr->work_count;
    const uint64_t n = wc < a.work_cap ? wc : a.work_cap;
    // same bounds as the host's sizing (enqueue_chunk_draws): any round's results, tasks
    const uint64_t nch = (uint64_t)a.n_channels;
    const uint64_t units = nch * (4ull * sum_samples + (uint64_t)(3 * a.retries + 32) * n);
    const uint64_t tasks = units / 64 + 2 * n + 64;
    if (n > a.n_items || units > a.pool_cap || tasks > a.task_cap) {
      if (i == 0) { a.ctr->fallback = 32ull; a.ctr->n_active[0] = 0u; }
      return;
    }
    n_items = n;
  }
  if (i == 0) a.ctr->n_active[0] = (unsigned int)n_items;
  if (i >= n_items) return;
  const uint32_t item = (uint32_t)i;
  const ItemVisit h = load_item_visit(a, item, a.lens ? a.lens->length : 0.0);
  const ItemHdr hd = make_item_hdr(a.P, h.I.cs, h.px, h.py);
  a.hdr[item] = hd;
  const uint32_t samples = h.samples, max_total = samples * 5u;
  const uint32_t m_limit = max_total + (uint32_t)a.retries;
  uint32_t m_hi = samples + (uint32_t)a.retries;
  if (sum_samples < a.extra_below) m_hi += (uint32_t)(((unsigned long long)samples * a.extra_num) >> 8) + a.extra_const;
  if (m_hi > m_limit) m_hi = m_limit;
  ItemProg pg{};
  pg.m_lo = 0;
  pg.m_hi = m_hi;
  uint32_t off = 0;
  if (!emit_tasks(a, 0u, item, 0u, m_hi, off)) pg.m_hi = 0;
  pg.res_off = off;
  a.prog[item] = pg;
  a.active[0][item] = item;
}

// What a finished backward trace leaves in the result pool: the tail of trace_ray_bw_po (transmittance and
// inner-pupil tests, src/lentil.h:633-645, sensor shift :654-655) and the sensor -> pixel step of
// filter_pixel (src/lentil_filter.cpp:276-290).
template <class LensT>
LD_DEV uint32_t solve_result(const lentil_params &P, const LensT &L, const NewtonState &s) {
  const DevLens &k = L.consts();
  double out4;
  const float transmittance = (float)newton_finish(L, s, out4);
  if (transmittance <= 0) return kCodeFail;
  const double ipx = s.x + s.dx * k.back_focal_length;
  const double ipy = s.y + s.dy * k.back_focal_length;
  if (ipx * ipx + ipy * ipy > k.inner_pupil_radius * k.inner_pupil_radius) return kCodeFail;
  const double sx = s.x + s.dx * -P.sensor_shift;
  const double sy = s.y + s.dy * -P.sensor_shift;
  uint32_t pix;
  return po_sensor_to_pixel(P, sx, sy, pix) ? pix : kCodeOut;
}

// ---- extension (ItemLive): results delivered, batches closed, queues ended ----------------------------------------
// One lane reports `cnt` results of `item`, `okc` of them pixels.  Returns 0, or 1 if that closed the LAST open item while
// the publishers have all signed off: the caller's wave then writes the queue's end markers (live_close_queue).
constexpr uint64_t kLiveEarly = 1ull << 62, kLiveClosed = 1ull << 63;     // ItemLive::hi_s flags
constexpr uint32_t kLiveEarlyAt = 256u;                                    // results after which an item is first looked at
LD_DEV bool cas64(unsigned long long *p, unsigned long long expect, unsigned long long desired) {
  return atomicCAS(p, expect, desired) == expect;
}
// append R(m_hi .. new_hi) of `item` to the extension's queue (the caller has moved ItemLive::hi_s to new_hi)
LD_DEV void live_emit(const DrawArgs &a, ItemLive *L, uint32_t item, uint32_t m_hi, uint32_t new_hi) {
  const uint32_t count = new_hi - m_hi, nt = (count + 63u) / 64u;
  const uint32_t tb = atomicAdd(&a.ctr->ext_n, nt);
  const bool fits = (unsigned long long)tb + nt + a.ext_end_tasks <= a.task_cap;
  if (!fits) atomicAdd(&a.ctr->overflow, 1ull);        // (the queue is as long as the main one: not reached; the pass would be void)
  const uint32_t res0 = ld_coherent32(&L->res_off);
  for (uint32_t t = 0; t < nt && fits; ++t) st_agent64(a.ext_q + tb + t, (uint64_t)item | ((uint64_t)(m_hi + t * 64u) << 32));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (uint32_t t = 0; t < nt && fits; ++t) {
    const uint32_t c = (count - t * 64u) < 64u ? (count - t * 64u) : 64u;
    st_agent64(reinterpret_cast<uint64_t *>(a.ext_q + tb + t) + 1,
               (uint64_t)(res0 + m_hi + t * 64u) | ((uint64_t)(c | (a.epoch << kTaskTagShift)) << 32));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  atomicAdd(&a.ctr->ext_tasks, nt);
  atomicAdd(&a.ctr->ext_items, 1u);
}
LD_DEV uint32_t live_deliver(const DrawArgs &a, uint32_t item, uint32_t cnt, uint32_t okc, uint32_t outc) {
  ItemLive *L = a.live + item;
  const unsigned long long old = atomicAdd(&L->cnt, (unsigned long long)cnt | ((unsigned long long)okc << 21) | ((unsigned long long)outc << 42));
  const uint32_t done0 = (uint32_t)old & 0x1FFFFFu;
  const uint32_t done = done0 + cnt, ok = ((uint32_t)(old >> 21) & 0x1FFFFFu) + okc, out = ((uint32_t)(old >> 42) & 0x1FFFFFu) + outc;
  const unsigned long long hs = ld_coherent64(&L->hi_s);
  const uint32_t m_hi = (uint32_t)hs, S = (uint32_t)(hs >> 32) & 0xFFFFu;
  const uint32_t retries = (uint32_t)a.retries, m_limit = S * 5u + retries;
  const uint32_t n = m_hi - retries;          // attempts the issued results cover
  if (done != m_hi) {
    // Not the batch's last result.  The first look at an item, after kLiveEarlyAt of its results: where those say that the
    // batch will fall short -- an attempt is decided by the first of its tries that is not vignetted, a pixel or a point
    // outside the frame, so about n * pixels / (pixels + outside) of n attempts succeed -- the rest is appended NOW, beside
    // the batch, not behind it (behind it the items found last ended the pass 0.2 ms later).  Sized from the lower end of what
    // the sample allows (two standard deviations), + 10 % + 32.
    if (done0 < kLiveEarlyAt && done >= kLiveEarlyAt && !(hs & (kLiveEarly | kLiveClosed)) && ok + out >= 32u && m_hi < m_limit) {
      const float k = (float)(ok + out), f = (float)ok / k;
      float f_lo = f - 2.0f * sqrtf(f * (1.0f - f) / k);
      if (f_lo < 0.02f) f_lo = 0.02f;
      if ((float)n * f_lo < (float)(S + a.ext_slack)) {
        float want = (float)(S + a.ext_slack) / f_lo;
        want = want * 1.1f + 32.0f + (float)retries;
        const uint32_t new_hi = want >= (float)m_limit ? m_limit : (uint32_t)want;
        if (new_hi > m_hi && cas64(&L->hi_s, hs, (hs & 0xFFFFFFFF00000000ull) | (unsigned long long)new_hi | kLiveEarly))
          live_emit(a, L, item, m_hi, new_hi);
      }
    }
    return 0u;
  }
  // The batch is complete (parked solves are not known yet and count as neither pixel nor outside).  Served when the
  // estimate is ext_slack beyond `samples`; else the next batch, sized like accept_next_batch does it.
  uint32_t est = (ok + out) ? (uint32_t)(((unsigned long long)n * ok) / (ok + out)) : 0u;
  {
    // ... less the attempts all of whose tries are vignetted (a share f of the results fails in the lens or is still parked: f to
    // the power of retries + 1 of the attempts), and 1 % for what the counts cannot know
    const float f = done ? (float)(done - ok - out) / (float)done : 0.0f;
    float lost = 1.0f;
    for (uint32_t t = 0; t <= retries && t < 32u; ++t) lost *= f;
    const float e2 = (float)est * (1.0f - lost) - 0.01f * (float)S;
    est = e2 > 0.0f ? (uint32_t)e2 : 0u;
  }
  if (est < S + a.ext_slack && m_hi < m_limit) {
    const uint32_t remaining = S + a.ext_slack - est;
    unsigned long long need = est ? ((unsigned long long)remaining * n + est - 1u) / est : (unsigned long long)(m_limit - m_hi);
    need += need / 4u + 32u;
    unsigned long long hi2 = (unsigned long long)m_hi + need;
    if (hi2 > m_limit) hi2 = m_limit;
    if (cas64(&L->hi_s, hs, (hs & 0xFFFFFFFF00000000ull) | hi2)) live_emit(a, L, item, m_hi, (uint32_t)hi2);
    return 0u;                              // the item stays open (or somebody else has just moved its end)
  }
  // served (or out of attempts): the item is closed -- unless its end has just been moved
  if (!cas64(&L->hi_s, hs, hs | kLiveClosed)) return 0u;
  const uint32_t open_before = atomicSub(&a.ctr->items_open, 1u);
  if (open_before != 1u) return 0u;
  return ld_coherent32(a.producers_done) >= a.producers_total ? 1u : 0u;
}
// all lanes of a wave: the end markers behind the extension's task queue
LD_DEV void live_close_queue(const DrawArgs &a, uint32_t lane) {
  if (lane == 0) st_agent32(&a.ctr->queue_final, 1u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t n = 0;
  if (lane == 0) n = ld_coherent32(&a.ctr->ext_n);
  n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
  for (uint32_t i = lane; i < a.ext_end_tasks; i += 64u)
    if ((uint64_t)n + i < a.task_cap)
      st_agent64(reinterpret_cast<uint64_t *>(a.ext_q + n + i) + 1, (uint64_t)(kEndCount | (a.epoch << kTaskTagShift)) << 32);
}

// ---- solve, polynomial optics ------------------------------------------------------------------
// kChroma: tasks carry a wavelength channel (src/lentil_filter.cpp:255-268); every lane then reads the lens
// header through its own pointer into three LDS copies that differ in the lambda powers only.
// kStream (first round of a streamed pass): the task queue is still being filled by the scan kernels while this
// kernel runs.  A wave that needs work draws a ticket and polls its slot between Newton iterations; the queue is
// complete once every wave of publish_kernel has signed off (DevCounters::publishers_done).
// Three waves per SIMD (at most 168 VGPRs): the generated petzval code would take 174-179 and leave the SIMD with two --
// a handful of values spilt outside the Newton loop costs less than the third wave brings.
#ifndef LENTIL_SOLVE_ATTR
#define LENTIL_SOLVE_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#endif
template <class LensT, bool kTables, bool kChroma = false, bool kStream = false>
__global__ __launch_bounds__(256) LENTIL_SOLVE_ATTR void solve_po_kernel(DrawArgs a) {
  LENTIL_TL_SPAN(a.round == 0 ? SPAN_SOLVE_R0 : (a.round == 1 ? SPAN_SOLVE_R1 : SPAN_SOLVE_R2));
  __shared__ DevTerm s_terms[kTables ? kMaxTerms : 1];
  __shared__ DevLens s_k;
  __shared__ DevLens s_kc[kChroma ? 3 : 1];
  __shared__ float s_cdfRow[kMaxBokehRows];
  __shared__ uint64_t s_hdr[4][4];                    // per wave: the header of the task being handed out
  // Batched start and end of the solves (round 4).  A solve's first step (tea<8> seed, aperture draw: ~250 instructions)
  // and its last (transmittance polynomial, pupil tests, sensor -> pixel: ~250) used to run inside the round in which a
  // lane needed them -- in steady state four lanes of 64 in nearly every round, i.e. 300 wave instructions per round
  // for 6 % of the lanes (PMC, profiles/r04_pmc_solve_*: 1 952 vector instructions per round at 54.8 active lanes,
  // the SIMDs' vector units 97 % busy: the kernel is bound by its instruction count).  Now the aperture draws of a
  // task's 64 units are computed by all lanes at once when the wave takes the task (s_ap), and a finished solve leaves
  // its state in the wave's queue (s_fin*), which all lanes empty together when it is full.
  __shared__ double s_ap[4][64][2];
  __shared__ double s_fin[4][6][64];                  // x, y, dx, dy, out[0], out[1]
  __shared__ uint32_t s_fin_err[4][64], s_fin_res[4][64];      // error bits | channel << 8; result slot
  __shared__ uint32_t s_fin_item[4][64];                       // extension (ItemLive): whose result it is
  if (kTables) {
    const uint32_t nt = a.lens->n_terms;
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) s_terms[i] = a.terms[i];
  }
  if (threadIdx.x == 0) s_k = *a.lens;
  if (kChroma && threadIdx.x < 3) {
    DevLens h = *a.lens;
    const double lam = a.lambda[threadIdx.x];
    h.lambda_pow[0] = 1.0; h.lambda_pow[1] = lam;
    for (uint32_t e = 2; e <= kMaxExp; ++e) h.lambda_pow[e] = ipow_u(lam, e);     // lens_ipow, like the host
    s_kc[threadIdx.x] = h;
  }
  const bool row_in_lds = a.P.bokeh_enable_image && a.bokeh.y <= kMaxBokehRows;
  if (row_in_lds)
    for (int i = threadIdx.x; i < a.bokeh.y; i += blockDim.x) s_cdfRow[i] = a.bokeh.cdfRow[i];
  __syncthreads();
  const float *cdfRow = row_in_lds ? s_cdfRow : a.bokeh.cdfRow;
  LensT L;
  if constexpr (kTables) { L.terms = s_terms; L.k = &s_k; } else { L.k = &s_k; }
  const DevLens &k = s_k;
  const lentil_params &P = a.P;

  // round r+1's queues are filled by this round's accept kernel: reset them here (nothing else touches them now)
  if (blockIdx.x == 0 && threadIdx.x == 0 && !a.no_reset) reset_round(a.ctr, (uint32_t)a.parity ^ 1u);
  const uint32_t par = (uint32_t)a.parity;
  const Task *tasks = a.tasks[par];
  uint32_t *res = a.pool[par];
  uint32_t n_tasks = 0;
  if constexpr (!kStream) n_tasks = a.ctr->n_tasks[par] < a.task_cap ? a.ctr->n_tasks[par] : a.task_cap;
  const uint32_t lane = lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  if (a.slow_live && a.slow && lane == 0) atomicAdd(&a.ctr->waves_started[par], 1u);
#ifdef LENTIL_PROBE_BUILD
  if (kStream && a.dispatch_probe && a.round == 1 && lane == 0) atomicAdd(&a.ctr->probe_res_xcc[0][xcc_id()], 1u);
#endif

  // wave-uniform cursor into the current task
  uint32_t cur_item = 0, cur_m = 0, cur_res = 0, cur_left = 0, cur_chan = 0;
  (void)cur_chan;
  bool no_more = false;
  constexpr uint32_t kNoTicket = 0xFFFFFFFFu;
  uint32_t ticket = kNoTicket, polls = 0;     // kStream: the queue slot this wave is waiting for
  uint64_t ticket_t0 = 0;
  uint32_t idle_naps = 1u;
  (void)ticket; (void)polls; (void)ticket_t0; (void)idle_naps;
  // per-lane solve
  bool busy = false;
  uint32_t res_idx = 0, cur_chan_lane = 0, my_item = 0;
  (void)cur_chan_lane;
  const uint32_t wv = threadIdx.x >> 6;
  // extension (ItemLive): the first round of a streamed pass counts what it delivers per item and appends batches itself
  const bool extend = kStream && !kChroma && a.live != nullptr && a.round == 0;
  const bool early = kStream && !kChroma && a.early_accept != 0 && a.round == 0;       // accept_kernel<4> beside this kernel (DrawArgs::early_accept)
  bool close_queue = false;      // this wave closed the last open item: it writes the queue's end markers
  const bool keeper = extend && blockIdx.x < a.ext_keeper_blocks;
  bool on_ext = false;
  const Task *tq = a.tasks[(uint32_t)a.parity];
  unsigned int *tq_head = &a.ctr->task_head[(uint32_t)a.parity];
  (void)keeper; (void)on_ext; (void)tq; (void)tq_head;
  uint32_t cur_pos = 0;        // wave-uniform: units of the current task handed out so far (index into s_ap)
  uint32_t fin_n = 0;          // wave-uniform: finished solves waiting in s_fin
  // the finished solves of the queue, one per lane: the tail of trace_ray_bw_po + sensor -> pixel (solve_result)
  auto flush_finished = [&]() {
    if (fin_n == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    bool fin_ok = false, fin_out = false;
    uint32_t fin_item = 0;
    if (lane < fin_n) {
      NewtonState t;
      t.x = s_fin[wv][0][lane]; t.y = s_fin[wv][1][lane]; t.dx = s_fin[wv][2][lane]; t.dy = s_fin[wv][3][lane];
      t.out[0] = s_fin[wv][4][lane]; t.out[1] = s_fin[wv][5][lane]; t.out[2] = 0.0; t.out[3] = 0.0;
      t.sqr_err = 0.0; t.sqr_ap_err = 0.0; t.k = 0;
      const uint32_t ew = s_fin_err[wv][lane];
      t.error = (int)(ew & 0xFFu);
      LensT Lf = L;
      if constexpr (kChroma) Lf.k = &s_kc[(ew >> 8) & 3u];
      const uint32_t code = solve_result(P, Lf, t);
      if (early) st_agent32(res + s_fin_res[wv][lane], code);      // (read by accept_kernel<4> on another CU while this kernel runs)
      else res[s_fin_res[wv][lane]] = code;
      fin_ok = code < kCodePendingBase;
      fin_out = code == kCodeOut;
      fin_item = s_fin_item[wv][lane];
    }
    if (early) {
      // every result of this flush is out before any of them is counted; then per item of the queue (usually one or two)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long todo = __ballot(lane < fin_n);
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)fin_item, leader);
        const unsigned long long same = __ballot(lane < fin_n && fin_item == it) & todo;
        if ((int)lane == leader) ready_deliver(a, it, (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
      }
    }
    if (extend) {
      // per item of the queue: how many results, how many of them pixels / outside the frame (usually one or two items)
      unsigned long long todo = __ballot(lane < fin_n);
      const unsigned long long okm = __ballot(lane < fin_n && fin_ok), outm = __ballot(lane < fin_n && fin_out);
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)fin_item, leader);
        const unsigned long long same = __ballot(lane < fin_n && fin_item == it) & todo;
        uint32_t r = 0;
        if ((int)lane == leader)
          r = live_deliver(a, it, (uint32_t)__builtin_popcountll(same), (uint32_t)__builtin_popcountll(same & okm), (uint32_t)__builtin_popcountll(same & outm));
        if (__ballot(r != 0u)) close_queue = true;
        todo &= ~same;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    fin_n = 0u;
    if (close_queue) { live_close_queue(a, lane); close_queue = false; }       // (wherever the flush was called from)
  };
  // a task has been taken (cur_* set, its header in s_hdr[wv]): the aperture draws of all its units at once --
  // the reference's try with seed (seed_a, m), src/lentil.h:596-609
  auto start_task = [&]() {
    cur_pos = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    if (lane < cur_left) {
      double ax = 0.0, ay = 0.0;
      po_aperture_sample(P, a.bokeh, cdfRow, (uint32_t)s_hdr[wv][3], cur_m + lane, ax, ay);
      s_ap[wv][lane][0] = ax; s_ap[wv][lane][1] = ay;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  double target[3] = {0, 0, 1};
  double ap_x = 0.0, ap_y = 0.0;
  NewtonState s;
  newton_init(s);
  uint32_t st_iters = 0, st_tries = 0, st_rounds = 0;
  // (streamed pass: the draw sum is not known yet; the host decides from the previous pass and passes no queue otherwise)
  const bool parking = a.slow != nullptr && a.round >= a.slow_from_round && (kStream || a.ctr->sum_samples < a.slow_below);
  bool producers_gone = false;
  (void)producers_gone;

  while (true) {
    const unsigned long long busy_mask = __ballot(busy);
    const uint32_t inflight = (uint32_t)__builtin_popcountll(busy_mask);
    const uint32_t n_idle = 64u - inflight;
    const uint32_t my_rank = (uint32_t)__builtin_popcountll(~busy_mask & lt_mask);
    uint32_t filled = 0;
    while (filled < n_idle) {
      if (cur_left == 0) {
        if (no_more) break;
        if constexpr (kStream) {
          if (ticket == kNoTicket) {
            uint32_t q = 0;
            if (lane == 0) q = atomicAdd(tq_head, 1u);
            ticket = __builtin_amdgcn_readfirstlane(q);
            polls = 0;
            ticket_t0 = __builtin_amdgcn_s_memrealtime();
          }
          if (ticket >= a.task_cap) { no_more = true; break; }
          uint64_t w1 = 0;
          if (lane == 0) w1 = ld_coherent64(reinterpret_cast<const uint64_t *>(tq + ticket) + 1);
          const uint32_t w1_hi = __builtin_amdgcn_readfirstlane((uint32_t)(w1 >> 32));
          if ((w1_hi >> kTaskTagShift) != a.epoch) {
            if (lane == 0) tl_add(TL_POLLS_EMPTY, 1u);
            // nothing there (yet): back to the solves in flight (an idle wave naps, below)
            if (__builtin_amdgcn_s_memrealtime() - ticket_t0 > (a.stuck_ticks ? a.stuck_ticks : kStuckTicks)) {
              if (lane == 0 && atomicCAS(&a.ctr->stuck, 0u, 2u | (ticket << 2)) == 0u) {      // the host redoes the pass chunk by chunk
                unsigned int *si = a.ctr->stuck_info;
                si[0] = (uint32_t)a.round; si[1] = par; si[2] = ld_coherent32(&a.ctr->n_tasks[par]);
                si[3] = ld_coherent32(&a.ctr->accept_done[0]); si[4] = ld_coherent32(&a.ctr->accept_started[0]);
                si[5] = w1_hi; si[6] = blockIdx.x; si[7] = ld_coherent32(&a.ctr->task_head[par]);
              }
              no_more = true;
            }
            break;
          }
          if ((w1_hi & 0xFFu) == kEndCount) {      // behind the last task
            // (extension: the first blocks of the launch go on with the queue of the batches the solve waves append)
            if (keeper && !on_ext) { on_ext = true; tq = a.ext_q; tq_head = &a.ctr->ext_head; ticket = kNoTicket; continue; }
            no_more = true;
            break;
          }
          uint64_t w0 = 0;
          if (lane == 0) w0 = ld_coherent64(reinterpret_cast<const uint64_t *>(tq + ticket));
          cur_item = __builtin_amdgcn_readfirstlane((uint32_t)w0);
          cur_m = __builtin_amdgcn_readfirstlane((uint32_t)(w0 >> 32));
          cur_res = __builtin_amdgcn_readfirstlane((uint32_t)w1);
          cur_left = w1_hi & 0xFFu;
          if (kChroma) cur_chan = (w1_hi >> 8) & 3u;
          ticket = kNoTicket;
          if (lane == 0) tl_add(TL_TASKS_TAKEN, 1u);
          if (cur_left) {
            // the item's header, written by a publisher on another CU while this kernel runs: lane 0 fetches it with
            // atomics and parks it in the wave's LDS slot, where the lanes that take solves of this task pick it up
            if (lane < 4u) {
              const uint64_t *hp = reinterpret_cast<const uint64_t *>(a.hdr + cur_item);
              s_hdr[wv][lane] = ld_coherent64(hp + lane);       // (four lanes, one round trip)
            }
            start_task();
          }
          if (cur_left == 0) continue;        // an item that did not fit left empty tasks
        } else {
          uint32_t q = 0;
          if (lane == 0) q = atomicAdd(&a.ctr->task_head[par], 1u);
          q = __builtin_amdgcn_readfirstlane(q);
          if (q >= n_tasks) { no_more = true; break; }
          const Task t = tasks[q];
          cur_item = __builtin_amdgcn_readfirstlane(t.item);
          cur_m = __builtin_amdgcn_readfirstlane(t.m_base);
          cur_res = __builtin_amdgcn_readfirstlane(t.res_off);
          cur_left = __builtin_amdgcn_readfirstlane(t.count) & 0xFFu;
          if (kChroma) cur_chan = (__builtin_amdgcn_readfirstlane(t.count) >> 8) & 3u;
          if (cur_left) {
            if (lane < 4u) s_hdr[wv][lane] = reinterpret_cast<const uint64_t *>(a.hdr + cur_item)[lane];
            start_task();
          }
          if (cur_left == 0) continue;
        }
      }
      uint32_t take = n_idle - filled;
      if (take > cur_left) take = cur_left;
      if (!busy && my_rank >= filled && my_rank < filled + take) {
        const uint32_t j = my_rank - filled;
        res_idx = cur_res + j;
        // (kStream: written by a publisher on another CU during this launch; fetched with the task)
        const uint64_t *hs = s_hdr[wv];
        target[0] = __longlong_as_double((long long)hs[0]);
        target[1] = __longlong_as_double((long long)hs[1]);
        target[2] = __longlong_as_double((long long)hs[2]);
        ap_x = s_ap[wv][cur_pos + j][0];
        ap_y = s_ap[wv][cur_pos + j][1];
        newton_init(s);
        ++st_tries;
        busy = true;
        my_item = cur_item;
        if constexpr (kChroma) { L.k = &s_kc[cur_chan]; cur_chan_lane = cur_chan; }
      }
      cur_pos += take; cur_res += take; cur_left -= take; filled += take;
    }
    // (extension: an item's batch is judged when its last result is DELIVERED -- a wave that found no work for its idle lanes
    // delivers what it holds at once, however little: the SIMD has nothing better to do then)
    if (extend && fin_n && filled < n_idle) flush_finished();
    if (inflight + filled == 0u) {
      flush_finished();
      if (!kStream || no_more) break;
      // An idle wave polls its slot, nothing else -- and rarely: a thousand waves asking every microsecond keep the
      // L2 channel that holds the queue busy enough to hold up every DMA group of the scan that touches it
      // (measured: scan 0.95 -> 2.6 ms beside 1024 idle waves polling every ~2 us).  4 us, doubling to 30 us.
      for (uint32_t i = 0; i < idle_naps; ++i) __builtin_amdgcn_s_sleep(127);
      if (idle_naps < (on_ext ? 2u : 8u)) idle_naps <<= 1;       // (the extension's queue: a few hundred tasks, latency is all)
      continue;
    }
    idle_naps = 1u;

    if (busy) { newton_iter(L, target, ap_x, ap_y, s); ++st_iters; }
    ++st_rounds;
#ifdef LENTIL_TIMELINE
    { const uint32_t nb_ = (uint32_t)__builtin_popcountll(__ballot(busy)); if (lane == 0) tl_add(!kStream || a.round ? TL_ITERS_LATER : (a.instance ? TL_ITERS_B : TL_ITERS_A), nb_); }
#endif

    {
      // finished solves go to the wave's queue; the queue is emptied by all lanes together when the next ones do not fit
      const bool fin = busy && !newton_continue(s);
      const unsigned long long fmask = __ballot(fin);
      if (fmask) {
        const uint32_t nfin = (uint32_t)__builtin_popcountll(fmask);
        if (fin_n + nfin > 64u) flush_finished();
        if (fin) {
          const uint32_t q = fin_n + (uint32_t)__builtin_popcountll(fmask & lt_mask);
          s_fin[wv][0][q] = s.x; s_fin[wv][1][q] = s.y; s_fin[wv][2][q] = s.dx; s_fin[wv][3][q] = s.dy;
          s_fin[wv][4][q] = s.out[0]; s_fin[wv][5][q] = s.out[1];
          s_fin_err[wv][q] = ((uint32_t)s.error & 0xFFu) | ((kChroma ? cur_chan_lane : 0u) << 8);
          s_fin_res[wv][q] = res_idx;
          s_fin_item[wv][q] = my_item;
          busy = false;
        }
        fin_n += nfin;
      }
    }
    // Stragglers: about one solve in a thousand is still running after slow_at iterations and may need all 100.
    // Park its loop state for solve_slow_kernel (a whole wave per solve, ~4x less time per iteration) instead of
    // holding this wave -- and the end of the round -- for it.
    // Only while this wave is running dry (no task left to refill its lanes from): as long as there is work, a slow
    // solve costs one lane; once there is none, it holds the wave and the end of the round.
    // ... and only its last few lanes: a lens whose solves routinely take more than slow_at iterations (the petzval
    // table: heavy vignetting, thousands of such solves per round) would otherwise send them all to a kernel that
    // spends a wave on each (config 4: 17.5 ms per frame with that, 3 ms of it per solve_slow_kernel launch).
    // (slow_after_producers: nothing is parked while the scan still feeds the queue -- the straggler kernel's waves only
    // find room on the chip when solve waves begin to leave, which is when the scan has ended; a solve parked before that
    // just waits, while a slow solve in a lane of a wave that keeps refilling its other lanes costs that lane alone and
    // is usually through by then.  Lane 0 looks every 16 rounds.)
    if (kStream && a.slow_after_producers && !producers_gone && (st_rounds & 15u) == 0u) {
      uint32_t pd = 0;
      if (lane == 0) pd = ld_coherent32(a.producers_done);
      producers_gone = (uint32_t)__builtin_amdgcn_readfirstlane(pd) >= a.producers_total;
    }
    if (parking && (!kStream || !a.slow_after_producers || producers_gone || no_more) &&
        ((a.slow_live && !a.slow_dry_only) || (no_more && __builtin_popcountll(__ballot(busy)) <= a.slow_max_lanes))) {
      const bool park = busy && s.k >= a.slow_at;
      unsigned long long pmask = __ballot(park);
      // Live queue: outliers only.  Where many lanes of a wave are past slow_at at once it is not a straggler but the item:
      // near the frame's edge, where the lens vignettes, a third of an item's solves run 20-40 iterations before they
      // raise an error bit (2 700 of a headline frame's solves -- in ten of its 1 100 items; 83 % of them fail).  Parked,
      // they swamp the 256 straggler waves and leave half of such an item's attempts unknown to the first accept; in
      // their lanes they cost the wave a few iterations more.  The crowd thins out by itself, what stays is parked.
      if (a.slow_live && a.slow_crowd_stays && __builtin_popcountll(pmask) > a.slow_max_lanes) pmask = 0ull;
      if (pmask) {
        bool parked_closed = false;
        const uint32_t sq = slow_queue(a);
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(pmask)) base = atomicAdd(&a.ctr->n_slow[sq], (uint32_t)__builtin_popcountll(pmask));
        base = __shfl(base, __builtin_ctzll(pmask));
        const uint32_t slot = base + (uint32_t)__builtin_popcountll(pmask & lt_mask);
        if (park && pmask && slot < a.slow_cap) {       // a full queue leaves the solve where it is
          tl_add(TL_PARKED, 1u);
          double *d = reinterpret_cast<double *>(a.slow + slot);
          const uint32_t chan_word = (kChroma ? cur_chan_lane : 0u) | (par ? kSlowParBit : 0u) |
                                     ((a.slow_live && a.slow_indirect) ? kSlowIndirectBit : 0u);
          if (a.slow_live) {
            // read by a wave of solve_slow_kernel on another CU while both kernels run: atomics on both sides, the
            // payload waited for before the word that publishes it
            const double pay[11] = {target[0], target[1], target[2], ap_x, ap_y, s.x, s.y, s.dx, s.dy, s.sqr_err, s.sqr_ap_err};
#pragma unroll
            for (int i = 0; i < 11; ++i) st_agent64(d + i, (uint64_t)__double_as_longlong(pay[i]));
            st_agent64(d + 11, (uint64_t)(uint32_t)s.k | ((uint64_t)(uint32_t)s.error << 32));
            st_agent64(d + 12, (uint64_t)res_idx | ((uint64_t)chan_word << 32));
            if (a.item_ready && a.slow_indirect) {
              // whose solve it is, for the straggler wave that finishes it; counted with the item before the record is published
              st_agent64(d + 13, (uint64_t)my_item << 32);
              atomicAdd(&a.hdr[my_item].parked, 1u);
            }
            // the first accept may look at this result before the straggler is through: it finds the mark (and, behind
            // it, where the result will be); the straggler's result never overwrites it
            if (a.slow_indirect) st_agent32(res + res_idx, kCodePendingBase | slot);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_agent64(d + 15, slow_tag(a, kSlowRecord));
            // (early accept: a parked solve is delivered -- its mark is in the pool; whether its result is in by the time the
            // accept comes to the item is ItemHdr::parked / parked_done's business)
            if (early) ready_deliver(a, my_item, 1u);
            // (extension: a parked solve is delivered -- as a failure, for what the item's batch is judged by)
            if (extend && live_deliver(a, my_item, 1u, 0u, 0u)) parked_closed = true;
          } else {
            d[0] = target[0]; d[1] = target[1]; d[2] = target[2]; d[3] = ap_x; d[4] = ap_y;
            d[5] = s.x; d[6] = s.y; d[7] = s.dx; d[8] = s.dy; d[9] = s.sqr_err; d[10] = s.sqr_ap_err;
            uint32_t *u = reinterpret_cast<uint32_t *>(d + 11);
            u[0] = (uint32_t)s.k; u[1] = (uint32_t)s.error; u[2] = res_idx; u[3] = chan_word;
          }
          busy = false;
        }
        if (__ballot(parked_closed)) close_queue = true;
      }
    }
    if (close_queue) { live_close_queue(a, lane); close_queue = false; }
  }
  flush_finished();
  if (close_queue) { live_close_queue(a, lane); close_queue = false; }
#ifdef LENTIL_PROBE_BUILD
  if (kStream && a.dispatch_probe && a.round == 1 && lane == 0) atomicSub(&a.ctr->probe_res_xcc[0][xcc_id()], 1u);
#endif
  if (a.slow_live && a.slow && a.slow_close) {
    // Everything this wave parked has arrived.  The straggler queue is closed by whichever wave finds, on leaving, that
    // every wave that has begun has left and every task has been taken (and, streamed, published): no solve can be
    // parked any more.  Waves of a launch that begin later -- blocks that had to wait for room, perhaps for the very
    // room the straggler kernel's waves take up -- find the task queue at its end, park nothing, and come to the same
    // conclusion: the end markers are simply written again.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t close = 0;
    if (lane == 0) {
      const uint32_t done = atomicAdd(&a.ctr->waves_done[par], 1u) + 1u;
      const uint32_t begun = ld_coherent32(&a.ctr->waves_started[par]);
      const uint32_t taken = ld_coherent32(&a.ctr->task_head[par]);
      const uint32_t published = kStream ? ld_coherent32(&a.ctr->n_tasks[par]) : n_tasks;
      const bool complete = !kStream || (ld_coherent32(a.producers_done) >= a.producers_total &&
                                         (!extend || ld_coherent32(&a.ctr->queue_final) != 0u));
      close = (done == begun && complete && taken >= published) ? 1u : 0u;
    }
    if (__builtin_amdgcn_readfirstlane(close)) {
      uint32_t n = ld_coherent32(&a.ctr->n_slow[slow_queue(a)]);
      if (n > a.slow_cap) n = a.slow_cap;
      for (uint32_t i = lane; i < a.slow_waves; i += 64u)
        st_agent64(reinterpret_cast<double *>(a.slow + n + i) + 15, slow_tag(a, kSlowEnd));      // (the queue holds slow_cap + slow_waves records)
    }
  }
  unsigned long long it64 = st_iters, tr64 = st_tries, rd64 = st_rounds;
  for (int off = 32; off > 0; off >>= 1) {
    it64 += __shfl_down(it64, off);
    tr64 += __shfl_down(tr64, off);
    rd64 += __shfl_down(rd64, off);
  }
  if (lane == 0) {
    if (it64) atomicAdd(&a.ctr->newton_iters, it64);
    if (tr64) atomicAdd(&a.ctr->tries, tr64);
    if (rd64) atomicAdd(&a.ctr->lane_rounds, rd64);
  }
}

// ---- stragglers: one wave per solve ----------------------------------------------------------------
// CoopLens evaluates the 14 polynomials of an iteration with all 64 lanes: (1) lane (var, e) computes
// lens_ipow(v[var], e) with the reference's recursion (src/lens.h:226-233) into a power table, (2) every lane
// multiplies out its share of the terms, c * f(x) * f(y) * f(dx) * f(dy) * f(lambda) left to right -- a factor
// with exponent 0 is 1.0 there, which changes no bit -- (3) lane p adds polynomial p's products in term order.
// Same operations in the same order as LdsLens / the generated code, so the iteration sequence of a parked
// solve continues unchanged; the sequential pupil transforms that follow are computed redundantly by all lanes.
constexpr int kCoopPolys = 14;
// (Round 5: the term table, the products and the index list are sized by the lens at launch -- 306 / 408 terms for the two
// benchmark tables, 9-12 KB a wave where kMaxTerms = 1536 made it 41 KB: a CU has room for several straggler waves beside
// two resident solve blocks and the accept's, coop_lds_bytes.)
struct CoopShared {
  double pw[64];                // [var][e]
  double sum[16];
  double lambda_pow[3][kMaxExp + 1];
  uint32_t first[16], count[16], n_needed;
  DevLens k;
};

LD_DEV double ipow_lane(double x, uint32_t e) {     // lens_ipow for a per-lane exponent 0..15, branch free
  double p = 1.0;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    const uint32_t n = e >> j;
    const double sq = p * p, xp = (x * p) * p;     // n = 1: (x * 1) * 1 = x;  n = 2: x * x
    p = n == 0u ? p : ((n & 1u) ? xp : sq);
  }
  return p;
}

inline __host__ __device__ size_t coop_lds_bytes(uint32_t n_terms) {
  const size_t nt = ((size_t)n_terms + 7u) & ~(size_t)7u;
  return nt * (sizeof(DevTerm) + sizeof(double) + sizeof(uint16_t));
}

struct CoopLens {
  CoopShared *sh;
  DevTerm *terms;        // [n_terms] the lens table (dynamic LDS)
  double *prod;          // [n_terms] the iteration's products, in evaluation order
  uint16_t *idx;         // [n_terms] term ids of the polynomials an iteration needs, in evaluation order
  const double *lp;      // lambda powers of this solve's wavelength channel (LDS)

  LD_DEV void eval_bw(const double v[4], double pred_ap[2], double Jap[4], double out[4], double Jout[4]) const {
    const uint32_t lane = threadIdx.x;
    {
      const uint32_t var = lane >> 4, e = lane & 15u;
      const double x = var == 0u ? v[0] : (var == 1u ? v[1] : (var == 2u ? v[2] : v[3]));
      sh->pw[lane] = ipow_lane(x, e);
    }
    __syncthreads();
    const uint32_t nn = sh->n_needed;
    for (uint32_t j = lane; j < nn; j += 64u) {
      const DevTerm t = terms[idx[j]];
      double term = t.c;
      term = term * sh->pw[t.e & 15u];
      term = term * sh->pw[16u + ((t.e >> 4) & 15u)];
      term = term * sh->pw[32u + ((t.e >> 8) & 15u)];
      term = term * sh->pw[48u + ((t.e >> 12) & 15u)];
      term = term * lp[(t.e >> 16) & 15u];
      prod[j] = term;
    }
    __syncthreads();
    if (lane < (uint32_t)kCoopPolys) {
      const uint32_t f = sh->first[lane], c = sh->count[lane];
      double sum = 0.0;
      if (c) {
        sum = prod[f];
        uint32_t i = 1;
        for (; i + 4u <= c; i += 4u) {
          const double p0 = prod[f + i], p1 = prod[f + i + 1], p2 = prod[f + i + 2], p3 = prod[f + i + 3];
          sum = sum + p0; sum = sum + p1; sum = sum + p2; sum = sum + p3;
        }
        for (; i < c; ++i) sum = sum + prod[f + i];
      }
      sh->sum[lane] = sum;
    }
    __syncthreads();
    pred_ap[0] = sh->sum[0]; pred_ap[1] = sh->sum[1];
    Jap[0] = sh->sum[2]; Jap[1] = sh->sum[3]; Jap[2] = sh->sum[4]; Jap[3] = sh->sum[5];
    out[0] = sh->sum[6]; out[1] = sh->sum[7]; out[2] = sh->sum[8]; out[3] = sh->sum[9];
    Jout[0] = sh->sum[10]; Jout[1] = sh->sum[11]; Jout[2] = sh->sum[12]; Jout[3] = sh->sum[13];
  }
  // once per solve: plain table walk (the state is wave-uniform)
  LD_DEV double transmittance(const double v[4]) const {
    const uint32_t first = sh->k.first[P_OUT_T], count = sh->k.count[P_OUT_T];
    double sum = 0.0;
    for (uint32_t i = 0; i < count; ++i) {
      const DevTerm t = terms[first + i];
      const uint32_t e = __builtin_amdgcn_readfirstlane(t.e);
      double term = t.c;
#pragma unroll
      for (int var = 0; var < 4; ++var) {
        const uint32_t ev = (e >> (4 * var)) & 15u;
        if (ev == 1) term = term * v[var];
        else if (ev > 1) term = term * ipow_u(v[var], ev);
      }
      const uint32_t el = (e >> 16) & 15u;
      if (el) term = term * lp[el];
      sum = (i == 0) ? term : sum + term;
    }
    return sum;
  }
  LD_DEV const DevLens &consts() const { return sh->k; }
};

// One 64-thread block = one wave; blocks pull parked solves until the queue is empty.  Launched blind after every
// solve_po_kernel: with an empty queue a block returns before it stages anything.
#ifndef LENTIL_SLOW_ATTR
#define LENTIL_SLOW_ATTR
#endif
__global__ __launch_bounds__(64) LENTIL_SLOW_ATTR void solve_slow_kernel(DrawArgs a) {
  LENTIL_TL_SPAN(a.round == 0 ? SPAN_SLOW_R0 : (a.round == 1 ? SPAN_SLOW_R1 : SPAN_SLOW_R2));
#ifdef LENTIL_PROBE_BUILD
  const uint32_t probe_xcc = a.dispatch_probe ? xcc_id() : 0u;
  const uint32_t probe_kind = a.round == 0 ? 2u : 1u;      // (the first round's stragglers / the second's)
  if (a.dispatch_probe && threadIdx.x == 0) atomicAdd(&a.ctr->probe_res_xcc[probe_kind][probe_xcc], 1u);
#endif
  if (a.slow_prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (a.slow_prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (a.slow_prio >= 3) __builtin_amdgcn_s_setprio(3);
  __shared__ CoopShared sh;
  __shared__ uint32_t s_q;
  extern __shared__ __align__(16) unsigned char s_coop[];      // coop_lds_bytes(n_terms)
  const uint32_t par = (uint32_t)a.parity, sq = slow_queue(a);
  uint32_t n_slow = a.slow_live ? 0xFFFFFFFFu : a.ctr->n_slow[sq];
  if (!a.slow_live) {
    if (n_slow > a.slow_cap) n_slow = a.slow_cap;
    if (n_slow == 0u || blockIdx.x >= n_slow) return;
  }
  const uint32_t lane = threadIdx.x;
  const uint32_t nt = a.lens->n_terms;
  const uint32_t nt_pad = (nt + 7u) & ~7u;
  CoopLens L;
  L.sh = &sh;
  L.terms = reinterpret_cast<DevTerm *>(s_coop);
  L.prod = reinterpret_cast<double *>(L.terms + nt_pad);
  L.idx = reinterpret_cast<uint16_t *>(L.prod + nt_pad);
  for (uint32_t i = lane; i < nt; i += 64u) L.terms[i] = a.terms[i];
  if (lane == 0) sh.k = *a.lens;
  {
    // the polynomials an iteration needs, in the order eval_bw hands them out: lane p looks after polynomial p
    // (one parallel read of the header, a prefix sum over the 14 counts, then every lane lists its own terms)
    const int order[kCoopPolys] = {P_AP_X, P_AP_Y, P_DAP_00, P_DAP_01, P_DAP_10, P_DAP_11, P_OUT_X, P_OUT_Y,
                                   P_OUT_DX, P_OUT_DY, P_DOUT_00, P_DOUT_01, P_DOUT_10, P_DOUT_11};
    uint32_t f = 0, c = 0;
    if (lane < (uint32_t)kCoopPolys) {
      int pid = order[0];
#pragma unroll
      for (int p = 1; p < kCoopPolys; ++p) pid = lane == (uint32_t)p ? order[p] : pid;
      f = a.lens->first[pid]; c = a.lens->count[pid];
    }
    uint32_t incl = c;
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      if ((int)lane >= off) incl += up;
    }
    if (lane < (uint32_t)kCoopPolys) {
      const uint32_t start = incl - c;
      sh.first[lane] = start; sh.count[lane] = c;
      for (uint32_t i = 0; i < c; ++i) L.idx[start + i] = (uint16_t)(f + i);
      if (lane == (uint32_t)kCoopPolys - 1u) sh.n_needed = incl;
    }
  }
  if (lane < 3u) {
    // lambda powers per wavelength channel, as solve_po_kernel prepares them
    const double lam = a.n_channels == 3 ? a.lambda[lane] : 0.0;
    for (uint32_t e = 0; e <= (uint32_t)kMaxExp; ++e)
      sh.lambda_pow[lane][e] = a.n_channels == 3 ? (e == 0 ? 1.0 : (e == 1 ? lam : ipow_u(lam, e))) : a.lens->lambda_pow[e];
  }
  __syncthreads();
  unsigned long long iters = 0, solves = 0;
  while (true) {
    __syncthreads();
    if (lane == 0) s_q = atomicAdd(&a.ctr->slow_head[sq], 1u);
    __syncthreads();
    const uint32_t q = s_q;
    if (q >= n_slow) break;
    SlowRec r;
    if (a.slow_live) {
      // the solve kernel is still running: wait for this slot's record, or for the end marker the last solve wave leaves
      if (q >= a.slow_cap + a.slow_waves) break;
      const uint64_t *src = reinterpret_cast<const uint64_t *>(a.slow + q);
      uint64_t tag = 0;
      uint32_t naps = 1u;
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      bool over = false;
      while (true) {
        if (lane == 0) tag = ld_coherent64(src + 15);
        tag = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(tag >> 32)) << 32) | (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)tag);
        if ((tag >> 32) == (slow_tag(a, 0) >> 32) && (uint32_t)tag != 0u) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > (a.stuck_ticks ? a.stuck_ticks : kStuckTicks)) { if (lane == 0) atomicCAS(&a.ctr->stuck, 0u, 3u | ((q & 0xFFFFFu) << 2) | ((uint32_t)a.round << 24)); over = true; break; }
        for (uint32_t i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(64);
        // (every idle wave's poll is a returning atomic at the memory side: a thousand of them every few microseconds are felt
        // by the solve kernel's own queue traffic -- DrawArgs::slow_nap_max)
        if (naps < (a.slow_nap_max ? (uint32_t)a.slow_nap_max : 8u)) naps <<= 1;
      }
      if (over || (uint32_t)tag == (uint32_t)kSlowEnd) break;
      uint64_t wv = 0;
      if (lane < 14u) wv = ld_coherent64(src + lane);
      auto word = [&](int i) { return ((uint64_t)(uint32_t)__shfl((int)(wv >> 32), i) << 32) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)wv, i); };
      r.tx = __longlong_as_double((long long)word(0)); r.ty = __longlong_as_double((long long)word(1)); r.tz = __longlong_as_double((long long)word(2));
      r.ap_x = __longlong_as_double((long long)word(3)); r.ap_y = __longlong_as_double((long long)word(4));
      r.x = __longlong_as_double((long long)word(5)); r.y = __longlong_as_double((long long)word(6));
      r.dx = __longlong_as_double((long long)word(7)); r.dy = __longlong_as_double((long long)word(8));
      r.sqr_err = __longlong_as_double((long long)word(9)); r.sqr_ap_err = __longlong_as_double((long long)word(10));
      const uint64_t w11 = word(11), w12 = word(12);
      r.k = (int32_t)(uint32_t)w11; r.error = (int32_t)(uint32_t)(w11 >> 32);
      r.res_idx = (uint32_t)w12; r.chan = (uint32_t)(w12 >> 32);
      r.pad[0] = (uint32_t)(word(13) >> 32);        // the item (DrawArgs::item_ready)
    } else {
      r = a.slow[q];
    }
    L.lp = sh.lambda_pow[(r.chan & 3u) < 3u ? (r.chan & 3u) : 0u];
    const double target[3] = {r.tx, r.ty, r.tz};
    NewtonState s;
    s.x = r.x; s.y = r.y; s.dx = r.dx; s.dy = r.dy;
    s.sqr_err = r.sqr_err; s.sqr_ap_err = r.sqr_ap_err;
    s.out[0] = s.out[1] = s.out[2] = s.out[3] = 0.0;     // a parked solve runs at least one more iteration
    s.k = r.k; s.error = r.error;
    // the loop condition is wave-uniform in value; make it uniform for the barriers inside eval_bw as well
    while (__builtin_amdgcn_readfirstlane((int)newton_continue(s))) {
      newton_iter(L, target, r.ap_x, r.ap_y, s);
      ++iters;
      if (lane == 0) tl_add(TL_ITERS_SLOW, 1u);
    }
    const uint32_t code = solve_result(a.P, L, s);
    if (lane == 0) dbg_add(code < kCodePendingBase ? 0 : (code == kCodeOut ? 1 : 2), 1);        // stragglers: pixel / out / fail
    if (lane == 0) dbg_add(3 + (s.k >= 100 ? 1 : 0), 1);                                          // ... ended before / at 100 iterations
    if (lane == 0) {
      // (the record says which pool: one queue may serve two rounds, DrawArgs::slow_q)
      if (r.chan & kSlowIndirectBit) {
        st_agent32(&a.slow[q].result, code);
        if (a.item_ready && a.slow_live) {
          // the result first, then the count that tells accept_kernel<3> that the item's parked solves are all through
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          atomicAdd(&a.hdr[r.pad[0]].parked_done, 1u);
        }
      } else a.pool[a.slow_live ? ((r.chan & kSlowParBit) ? 1u : 0u) : par][r.res_idx] = code;
    }
    ++solves;
  }
  if (lane == 0) {
    if (iters) atomicAdd(&a.ctr->newton_iters, iters);
    if (solves) atomicAdd(&a.ctr->slow_solves, solves);
#ifdef LENTIL_PROBE_BUILD
    if (a.dispatch_probe) atomicSub(&a.ctr->probe_res_xcc[probe_kind][probe_xcc], 1u);
#endif
  }
}

// ---- solve, thin lens (K4): closed form, one wave per task ---------------------------------------
__global__ __launch_bounds__(256) void solve_thinlens_kernel(DrawArgs a) {
  __shared__ float s_cdfRow[kMaxBokehRows];
  const bool row_in_lds = a.P.bokeh_enable_image && a.bokeh.y <= kMaxBokehRows;
  if (row_in_lds)
    for (int i = threadIdx.x; i < a.bokeh.y; i += blockDim.x) s_cdfRow[i] = a.bokeh.cdfRow[i];
  __syncthreads();
  const float *cdfRow = row_in_lds ? s_cdfRow : a.bokeh.cdfRow;
  // round r+1's queues are filled by this round's accept kernel: reset them here (nothing else touches them now)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const uint32_t nx = (uint32_t)a.parity ^ 1u;
    a.ctr->n_tasks[nx] = 0; a.ctr->task_head[nx] = 0; a.ctr->n_active[nx] = 0; a.ctr->active_head[nx] = 0;
    a.ctr->pool_used[nx] = 0;
  }
  const uint32_t par = (uint32_t)a.parity;
  const uint32_t n_tasks = a.ctr->n_tasks[par] < a.task_cap ? a.ctr->n_tasks[par] : a.task_cap;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q = wave; q < n_tasks; q += n_waves) {
    const Task t = a.tasks[par][q];
    if (lane < t.count) {
      const ItemHdr hd = a.hdr[t.item];
      const float cs[3] = {(float)hd.tx, (float)hd.ty, (float)hd.tz};
      const int px = hd.px_py & 0xFFFF, py = hd.px_py >> 16;
      uint32_t pix;
      const bool ok = thinlens_draw(a.P, a.bokeh, cdfRow, cs, px, py, t.m_base + lane, pix);
      a.pool[par][t.res_off + lane] = ok ? pix : kCodeOut;     // no retries: a failed draw ends the attempt
    }
  }
}

// ---- ordered acceptance + splat -------------------------------------------------------------------
// One 256-thread block per item; each step resolves 256 consecutive attempts (wave w: attempts
// n + 64w .. n + 64w + 63) and ranks the successes across the four waves through LDS, so that exactly
// the first `samples` successes in attempt order are accepted.
// Draw-log slots for the lanes of `mask` (the calling lanes: all of them, no others): the first takes popcount(mask)
// slots with one atomic, every lane gets its own by rank.
LD_DEV unsigned long long wave_log_slots(unsigned long long *log_count, unsigned long long mask, uint32_t lane) {
  const int first = __builtin_ctzll(mask);
  unsigned long long base = 0;
  if ((int)lane == first) base = atomicAdd(log_count, (unsigned long long)__builtin_popcountll(mask));
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)base, first), hi = (uint32_t)__shfl((int)(uint32_t)(base >> 32), first);
  return (((unsigned long long)hi << 32) | lo) + (unsigned long long)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
}

struct AcceptShared {
  uint32_t first_u[4];
  uint32_t nsucc[4];
  uint32_t top[4];
  uint32_t pix[4][64];
  uint32_t cnt[4][64];                      // draws of this wave step on the same pixel, at the first of them (0 at the others)
  float val[4 * LENTIL_MAX_AOVS + 1];      // what one accepted draw of the item adds, float by float
  uint32_t off[4 * LENTIL_MAX_AOVS + 1];   // ... and where inside the pixel record
  // R(n .. n + 255 + retries) of the step, per wavelength channel (stage_results): accept_item / accept_item_chroma only --
  // it lies in AcceptWideShared::win, which those forms do not use (3.8 KB less LDS per block: five accept blocks fit a CU
  // beside a straggler wave's 41 KB, and a headline frame's ~1 200 items are all resident at once)
  uint32_t (*rwin)[256 + 64];
  uint32_t nunk[4], nsucc1[4];             // accept_item<1, 2>: unknown attempts / mode 1's known successes per wave
};
// accept_item<1>: an attempt that met a pending mark is no success (yet); modes 0 and 2 have resolved it
template <int kMode> LD_DEV bool unk_blocks(bool unk) { return kMode == 1 && unk; }

// The results a 256-attempt step looks at, R(n .. n + 255 + retries), fetched by the block with one round of
// coalesced loads into LDS.  Every attempt then walks its tries there.  (Walking them in global memory is a chain of
// dependent loads per attempt -- up to retries + 1 of them where the lens vignettes -- and the block waits for its
// slowest lane: 25 us per step for the items near the frame's edge, 130 of the accept kernel's 175 us.)
constexpr uint32_t kAcceptWinRetries = 64;        // more retries than this: the attempts read global memory
constexpr uint32_t kCodeBeyond = 0xFFFFFFFDu;     // R(m) with m >= m_hi: not part of this batch
LD_DEV uint32_t result_at(const uint32_t *res, const ItemProg &pg, uint32_t res_base, uint32_t m) {
  return m < pg.m_lo ? kCodeFail : (m >= pg.m_hi ? kCodeBeyond : res[res_base + (m - pg.m_lo)]);
}
LD_DEV void stage_results(uint32_t *win, const uint32_t *res, const ItemProg &pg, uint32_t res_base, uint32_t n, uint32_t count) {
  for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) win[i] = result_at(res, pg, res_base, n + i);
}
// A block barrier that orders LDS only.  __syncthreads() also waits for the thread's outstanding global memory
// operations (s_waitcnt vmcnt(0)) -- here the splat atomics of the step before, 5-7 us each time, for nothing: no
// thread of the block reads what they wrote.
LD_DEV void block_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct AcceptResult {
  bool more;                // draws are still missing and more attempts exist: [new_lo, new_hi) must be solved
  uint32_t new_lo, new_hi;
  ItemProg prog;            // updated progress (n_done, accepted, last_ok; m_lo/m_hi/res_off to be filled by the caller)
  uint32_t samples;
};

// The accepted draws of one wave step that landed on the same pixel are added as ONE atomic of count x value:
// every draw of an item adds the same amounts, so where the draws of an in-focus highlight pile up on a pixel
// (thin lens: all 64 of a step) the sum no longer takes a rounding per draw -- it stays within 1e-5 of the exact
// sum where 64 x more fp32 atomics in arbitrary order did not -- and the memory side sees one request, not 64.
// Lane d < T looks at the step's pixel list (LDS, broadcast reads): count of its pixel, and whether it is the first.
LD_DEV void count_same_pixel(const uint32_t *pix, uint32_t *cnt, uint32_t T, uint32_t lane) {
  if (lane < T) {
    const uint32_t mine = pix[lane];
    uint32_t n = 0;
    bool first = true;
    for (uint32_t e = 0; e < T; ++e) {
      const bool same = pix[e] == mine;
      n += same ? 1u : 0u;
      first = first && !(same && e < lane);
    }
    cnt[lane] = first ? n : 0u;
  }
}

// The same from registers (accept_item_wide): lane d < T holds entry d of the wave's compacted pixel list; every entry is
// broadcast with v_readlane -- three instructions per entry and no LDS round trip in the loop (the LDS version above waits
// ~70 cycles per entry: 2 us per 64-draw slab).  Returns the count at the first lane of every pixel, 0 at the others.
LD_DEV uint32_t count_same_pixel_reg(uint32_t mine, uint32_t T, uint32_t lane) {
  uint32_t n = 0;
  bool first = true;
  for (uint32_t e = 0; e < T; ++e) {
    const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)e);
    const bool same = p == mine;
    n += same ? 1u : 0u;
    first = first && !(same && e < lane);
  }
  return (lane < T && first) ? n : 0u;
}

// R(m) as the accept of an item finds it: the current batch in `res`, the batch before -- if the item carries one
// (ItemProg::p_lo) -- in the other parity's pool, FAIL below that, "beyond" above.
LD_DEV uint32_t result_at2(const uint32_t *res, const uint32_t *res_prev, const ItemProg &pg, uint32_t m) {
  if (m >= pg.m_hi) return kCodeBeyond;
  if (m >= pg.m_lo) return res[pg.res_off + (m - pg.m_lo)];
  if (m >= pg.p_lo && m < pg.p_hi) return res_prev[pg.p_off + (m - pg.p_lo)];
  return kCodeFail;
}

// What an item's walk leaves for the next round: whether draws are still missing and attempts exist, and which R(m) the
// next batch must cover (shared by accept_item and accept_item_wide).
template <int kMode>
LD_DEV void accept_next_batch(const DrawArgs &a, const ItemProg &pg, AcceptResult &r, uint32_t n, uint32_t acc, uint32_t uacc,
                              uint32_t S, uint32_t max_total, uint32_t m_limit, uint32_t retries) {
  if (kMode == 1 && uacc && a.unknown_credit) {
    // What the next batch is sized from.  Counting every unknown attempt as a failure is safe and costly: unknowns come
    // in clusters -- in a headline frame ten items near the frame's edge, where the lens vignettes, hold all of them,
    // ~500 each (their solves run 20-40 iterations, are parked, and 83 % of them fail -- after which the attempt simply
    // moves on to its next try and mostly succeeds) -- and the batches that pessimism sends those items are as slow as
    // their first ones: thousands of parked solves for 256 straggler waves.  So the unknown attempts are credited with
    // unknown_credit / 8 of the success rate of the item's known ones.  Where that is too generous by a few attempts
    // the first batch's spare ones (DrawArgs::extra_const) make up for it; beyond that the item costs a third round.
    const uint32_t known = n > uacc ? n - uacc : 1u;
    const unsigned long long est = (unsigned long long)acc + ((unsigned long long)uacc * acc * a.unknown_credit) / (8ull * known);
    acc = est < S ? (uint32_t)est : S;
  }
  r.more = acc < S && n < max_total && pg.m_hi < m_limit && pg.m_hi > 0;
  r.new_lo = pg.m_hi;
  r.new_hi = pg.m_hi;
  if (r.more) {
    // draws are still missing: schedule the next batch, sized from the item's own success rate so far
    // (+25 % + 32; a surplus is simply never accepted, a shortfall costs another round)
    const uint32_t remaining = S - acc;
    unsigned long long need = acc ? ((unsigned long long)remaining * n + acc - 1) / acc : (unsigned long long)(max_total - n);
    // (twice the margin where fewer than margin_low_rate / 16 of the item's attempts got through so far: with the petzval
    // table, where a third of all attempts is vignetted, +25 % left some item short in every pass -- a third round, 0.4 ms
    // of latency for a few thousand lane-iterations; config 4 8.65-8.8 against 9.2 ms, the other configurations unchanged)
    const uint32_t m16 = (unsigned long long)acc * 16ull < (unsigned long long)n * a.margin_low_rate ? 2u * a.batch_margin16 : a.batch_margin16;
    need += (need * m16) / 16 + 32;
    unsigned long long n_target = (unsigned long long)n + need;
    if (n_target > max_total) n_target = max_total;
    uint32_t new_hi = (uint32_t)n_target + retries;
    if (new_hi > m_limit) new_hi = m_limit;
    r.new_hi = new_hi;
  }
}

// Processes the current result batch of `item` (block-cooperative; must be called by all 256 threads).
// dry: the walk without its effects -- nothing is splatted, logged or flagged; the result (how far the batch got, whether
// and how many more attempts are needed) is what the real walk will find.
//
// kMode 0: every R(m) the walk meets is final.
// kMode 1: the first accept of a streamed pass that does not wait for the parked solves of its round (the slowest of them
//   takes another 0.3 ms, and nearly every item has one).  A parked solve's R(m) is a *pending mark*; an attempt whose
//   tries reach one before a result that ends them is UNKNOWN.  With s known successes and u unknown attempts before it,
//   a known success is among the first S successes whatever the unknowns turn out to be iff s + u < S (S = samples):
//   those are splatted now -- all but a handful per item.  Draws still missing are counted as if every unknown attempt
//   failed, and the next batch is scheduled from that (a surplus is never accepted).  An item that met an unknown is
//   handed to the next accept whole.
// kMode 2: that next accept (the stragglers are through; their results sit in the queue records behind the marks).  It
//   walks such an item again from attempt 0 with the final results, replays beside it what mode 1 saw (the marks are
//   still there, ItemProg::n_end1 says how far it looked), and splats the accepted draws mode 1 did not.  What mode 1
//   splatted is a subset of the final set: a known success with s + u < S has final rank <= s + u.
template <int kMode>
LD_DEV AcceptResult accept_item(const DrawArgs &a, AcceptShared &sh, uint32_t item, const ItemProg pg, const uint32_t *res,
                                const uint32_t *res_prev, uint32_t &rmin, uint32_t &rmax_p1, bool dry = false) {
  uint32_t *s_first_u = sh.first_u, *s_nsucc = sh.nsucc, *s_top = sh.top;
  uint32_t(*s_pix)[64] = sh.pix;
  float *s_val = sh.val;
  uint32_t *s_off = sh.off;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const uint32_t retries = (uint32_t)a.retries;
  const double lens_length = a.lens ? a.lens->length : 0.0;
    const ItemVisit h = load_item_visit(a, item, lens_length);
    const uint32_t S = h.samples, max_total = S * 5u, m_limit = max_total + retries;
    const float ae = h.I.add_energy, w = h.w;
    const unsigned long long zk = (a.F.zkey || a.F.zkey_dbg) ? closest_key_of(a.ctr, h.I.depth, visit_gid(a.V, h.visit)) : 0ull;
    // per-item add table: gaussian AOVs' (value + add_energy) * w, then the weight itself
    uint32_t U = 1;
    for (uint32_t k = 0; k < a.F.n_aovs; ++k) if (!(a.F.closest_mask & (1u << k))) U += 4;
    if (threadIdx.x < a.F.n_aovs * 4u) {
      const uint32_t k = threadIdx.x >> 2, c = threadIdx.x & 3u;
      if (!(a.F.closest_mask & (1u << k))) {
        uint32_t slot = 0;
        for (uint32_t j = 0; j < k; ++j) if (!(a.F.closest_mask & (1u << j))) slot += 4;
        const float4 v = k == 0 ? h.rgba : a.V.extra[k - 1][h.visit];
        const float vc = c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
        s_val[slot + c] = (vc + ae) * w;
        s_off[slot + c] = 4u * k + c;
      }
    }
    if (threadIdx.x == 0) { s_val[U - 1] = w; s_off[U - 1] = 4u * a.F.n_aovs; }
    __syncthreads();
    // mode 2 starts an item that carries unknowns over: its walk is the whole walk
    const bool replay = kMode == 2 && pg.uacc != 0u;
    uint32_t n = pg.n_done, acc = pg.accepted, last_ok = pg.last_ok;
    uint32_t uacc = 0;               // mode 1: unknown attempts so far; mode 2: the same, replayed
    uint32_t acc1 = 0;               // mode 2: known successes mode 1 counted so far
    bool stalled = false;
    // the window of a step is requested a step ahead (entries threadIdx.x and 256 + threadIdx.x), before the splat
    // atomics of the step in between: loads return in order with them, so a load issued behind 1 280 atomics waits
    // for all of them
    const bool win = retries <= kAcceptWinRetries;
    uint32_t pre0 = kCodeBeyond, pre1 = kCodeBeyond;
    if (win) {
      pre0 = result_at2(res, res_prev, pg, n + threadIdx.x);
      if (threadIdx.x < retries) pre1 = result_at2(res, res_prev, pg, n + 256u + threadIdx.x);
    }
    while (!stalled && acc < S && n < max_total) {
      const uint32_t my_i = wave * 64u + lane;              // position inside this 256-attempt step
      const uint32_t my_n = n + my_i;
      const bool valid = my_n < max_total;
      uint32_t code = kCodeFail;       // what ends the attempt (mode 1: as far as known)
      uint32_t code1 = kCodeFail;      // mode 2: what mode 1 knew
      bool unresolved = false, unk = false, seen1 = false;
      (void)code1; (void)seen1;
      if (win) {
        sh.rwin[0][threadIdx.x] = pre0;
        if (threadIdx.x < retries) sh.rwin[0][256u + threadIdx.x] = pre1;
        block_sync_lds();
      }
      if (valid) {
        for (uint32_t t = 0; t <= retries; ++t) {
          uint32_t c = win ? sh.rwin[0][my_i + t] : result_at2(res, res_prev, pg, my_n + t);
          if (c == kCodeBeyond) { unresolved = (pg.m_hi < m_limit); break; }
          if (code_is_pending(c)) {
            if (kMode == 1) { unk = true; break; }
            if (kMode == 2) {
              if (!seen1) { unk = true; seen1 = true; }
              c = ld_coherent32(&a.slow[c & 0x00FFFFFFu].result);      // the straggler is through (its kernel may still run: accept_kernel<3>)
            } else {
              atomicAdd(&a.ctr->overflow, 1ull);               // a mark nobody is going to resolve: the pass is void
              c = kCodeFail;
            }
          } else if (kMode == 2 && !seen1 && c != kCodeFail) {
            code1 = c; seen1 = true;
          }
          if (c != kCodeFail) { code = c; break; }
        }
      }
      // first unresolved attempt of the step (all later ones are unresolved too)
      const unsigned long long umask = __ballot(valid && unresolved);
      if (lane == 0) s_first_u[wave] = umask ? wave * 64u + (uint32_t)__builtin_ctzll(umask) : 256u;
      block_sync_lds();
      uint32_t limit = max_total - n < 256u ? max_total - n : 256u;
      {
        uint32_t fu = s_first_u[0];
        if (s_first_u[1] < fu) fu = s_first_u[1];
        if (s_first_u[2] < fu) fu = s_first_u[2];
        if (s_first_u[3] < fu) fu = s_first_u[3];
        if (fu < limit) { limit = fu; stalled = true; }
      }
      if (win && !stalled) {
        pre0 = result_at2(res, res_prev, pg, n + limit + threadIdx.x);
        if (threadIdx.x < retries) pre1 = result_at2(res, res_prev, pg, n + limit + 256u + threadIdx.x);
      }
      const bool succ = my_i < limit && !unk_blocks<kMode>(unk) && code < kCodePendingBase;
      const unsigned long long smask = __ballot(succ);
      // mode 1: the unknown attempts; mode 2: the same and mode 1's known successes, as far as mode 1 looked
      const bool was1 = kMode == 2 && replay && my_n < pg.n_end1 && my_i < limit;
      const bool unk1 = kMode == 1 ? (my_i < limit && unk) : (was1 && unk);
      const bool succ1 = was1 && !unk && code1 < kCodePendingBase;
      const unsigned long long u1mask = kMode != 0 ? __ballot(unk1) : 0ull;
      const unsigned long long s1mask = kMode == 2 ? __ballot(succ1) : 0ull;
      if (lane == 0) {
        s_nsucc[wave] = (uint32_t)__builtin_popcountll(smask);
        if (kMode != 0) sh.nunk[wave] = (uint32_t)__builtin_popcountll(u1mask);
        if (kMode == 2) sh.nsucc1[wave] = (uint32_t)__builtin_popcountll(s1mask);
      }
      block_sync_lds();
      uint32_t before = 0, total = 0, ubefore = 0, utotal = 0, before1 = 0, total1 = 0;
      for (uint32_t k = 0; k < 4; ++k) {
        if (k < wave) before += s_nsucc[k];
        total += s_nsucc[k];
        if (kMode != 0) { if (k < wave) ubefore += sh.nunk[k]; utotal += sh.nunk[k]; }
        if (kMode == 2) { if (k < wave) before1 += sh.nsucc1[k]; total1 += sh.nsucc1[k]; }
      }
      const uint32_t rank = acc + before + (uint32_t)__builtin_popcountll(smask & lt_mask);
      const bool take = succ && rank < S;              // mode 1: among the first S *known* successes
      // what is splatted now
      bool splat = take;
      if (kMode == 1) {
        const uint32_t ub = uacc + ubefore + (uint32_t)__builtin_popcountll(u1mask & lt_mask);
        splat = take && rank + ub < S;
      } else if (kMode == 2) {
        const uint32_t ub = uacc + ubefore + (uint32_t)__builtin_popcountll(u1mask & lt_mask);
        const uint32_t rank1 = acc1 + before1 + (uint32_t)__builtin_popcountll(s1mask & lt_mask);
        splat = take && !(succ1 && rank1 + ub < S);    // mode 1 has added that one
      }
      // Camera::add_to_buffer, src/lentil.h:827-830 -- transposed: the wave's accepted pixels go through
      // LDS, then lane q adds float (q % U) of accepted draw (q / U): consecutive lanes hit consecutive
      // floats of one pixel record.
      const unsigned long long tmask0 = __ballot(take);
      const unsigned long long pmask0 = __ballot(splat);
      const uint32_t T = (uint32_t)__builtin_popcountll(pmask0);
      if (splat && !dry) {
        const uint32_t pix = code;
        const uint32_t row = pix / a.P.xres;
        rmin = row < rmin ? row : rmin;
        rmax_p1 = row + 1u > rmax_p1 ? row + 1u : rmax_p1;
        s_pix[wave][(uint32_t)__builtin_popcountll(pmask0 & lt_mask)] = pix;
        if (a.F.touched) { a.F.touched[pix >> 6] = a.round ? 2 : 1; a.F.touched_px[pix] = a.round ? 2 : 1; }      // (2: by a round after the first, see resolve_touched_kernel)
        if (a.F.zkey) atomicMin(a.F.zkey + pix, zk);
        if (a.F.zkey_dbg) atomicMin(a.F.zkey_dbg + pix, zk);      // value.r = samples != 0 for every draw
        if (a.log_cap) {
          // one returning atomic per wave step, not per draw: a million of them on ONE counter cost the pass 1.1 ms
          const unsigned long long li = wave_log_slots(a.log_count, pmask0, lane);
          if (li < a.log_cap) { a.log[li].visit = h.visit; a.log[li].attempt = my_n; a.log[li].pixel = pix; }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      if (!dry) count_same_pixel(s_pix[wave], sh.cnt[wave], T, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      for (uint32_t q = lane; !dry && q < T * U; q += 64u) {
        const uint32_t d = q / U, ch = q - d * U;
        const uint32_t c = sh.cnt[wave][d];
        if (c) atomicAdd(a.F.acc + (size_t)s_pix[wave][d] * a.F.stride + s_off[ch], (float)c * s_val[ch]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      // highest accepted attempt of the step
      const unsigned long long tmask = tmask0;
      if (lane == 0) s_top[wave] = tmask ? n + wave * 64u + (63u - (uint32_t)__builtin_clzll(tmask)) : 0u;
      block_sync_lds();
      uint32_t taken = total < S - acc ? total : S - acc;
      if (taken) {
        uint32_t top = s_top[0];
        if (s_top[1] > top) top = s_top[1];
        if (s_top[2] > top) top = s_top[2];
        if (s_top[3] > top) top = s_top[3];
        last_ok = top;
        acc += taken;
      }
      uacc += utotal;
      if (kMode == 2) acc1 += total1;       // (mode 1 capped its count at S; beyond that rank1 >= S either way)
      n += limit;
    }
    AcceptResult r;
    r.samples = S;
    r.prog = pg;
    r.prog.n_done = n; r.prog.accepted = acc; r.prog.last_ok = last_ok;
    r.prog.uacc = kMode == 1 ? uacc : 0u;
    accept_next_batch<kMode>(a, pg, r, n, acc, uacc, S, max_total, m_limit, retries);
    return r;
}

// ---- accept_item, wide (round 4) ------------------------------------------------------------------
// The same walk with kWide x 256 attempts per step instead of 256: a first batch (samples + retries + spare <= 2 048
// attempts) is ONE step.  accept_item above takes 13 us per 256-attempt step -- four block barriers, a round trip to the
// result pool, the splat atomics of the step before in the way of the next step's loads -- and a dry walk before the real
// one where the next round is waiting for its tasks: 85 us for an item of 1 024 draws, 190 us for the headline frame's
// first accept, which stands between the first round's last solve and the second round's first (profiles/r04_*).  Here
// the whole batch is fetched into LDS at once, every attempt is resolved there (three times over, LDS reads being what they
// cost: once for the step's limit, once for the ranks, once to splat), ranks come from one prefix over the step's
// 4 x kWide ballots, and the splats of a wave's kWide slabs follow one another without a block barrier in between.
// Same decisions as accept_item, attempt for attempt (modes 0, 1, 2; gaussian, closest and debug AOVs; the draw log).
constexpr int kWide = 8;
struct AcceptWideShared {
  uint32_t win[256 * kWide + kAcceptWinRetries];      // R(n .. n + 256 kWide - 1 + retries) of the step
  uint32_t code[256 * kWide];                          // what ends every attempt of the step (wide_resolve), for the splats
  unsigned long long m_succ[kWide * 4], m_unk[kWide * 4], m_succ1[kWide * 4];      // ballots per slab of 64 attempts
  uint32_t before[3][kWide * 4];                       // exclusive prefix of the three masks' popcounts
  uint32_t first_u[kWide * 4];
  uint32_t limit, stalled, total[3], top;              // the step's outcome, from wave 0
};

template <int kMode>
struct WideAttempt {
  uint32_t code, code1;
  bool unresolved, unk, seen1;
};

// one attempt's tries in the step's window (the loop of accept_item, src/lentil.h:592-648 as "first non-FAIL of R(n .. n + retries)")
template <int kMode>
LD_DEV WideAttempt<kMode> wide_resolve(const DrawArgs &a, const AcceptWideShared &ws, const ItemProg &pg, uint32_t my_i, uint32_t retries,
                                       uint32_t m_limit) {
  WideAttempt<kMode> r;
  r.code = kCodeFail; r.code1 = kCodeFail; r.unresolved = false; r.unk = false; r.seen1 = false;
  for (uint32_t t = 0; t <= retries; ++t) {
    uint32_t c = ws.win[my_i + t];
    if (c == kCodeBeyond) { r.unresolved = (pg.m_hi < m_limit); break; }
    if (code_is_pending(c)) {
      if (kMode == 1) { r.unk = true; break; }
      if (kMode == 2) {
        if (!r.seen1) { r.unk = true; r.seen1 = true; }
        c = ld_coherent32(&a.slow[c & 0x00FFFFFFu].result);      // the straggler is through (its kernel may still run: accept_kernel<3>)
      } else {
        atomicAdd(&a.ctr->overflow, 1ull);               // a mark nobody is going to resolve: the pass is void
        c = kCodeFail;
      }
    } else if (kMode == 2 && !r.seen1 && c != kCodeFail) {
      r.code1 = c; r.seen1 = true;
    }
    if (c != kCodeFail) { r.code = c; break; }
  }
  return r;
}

// The lanes of the wave that hold the same value as this one (all 64 lanes call; `bits` = width of the values):
// one ballot per bit.  What count_same_pixel finds with a 64-step loop per lane.
LD_DEV unsigned long long match_any_bits(uint32_t v, uint32_t bits) {
  unsigned long long peers = ~0ull;
  for (uint32_t b = 0; b < bits; ++b) {
    const bool set = (v >> b) & 1u;
    const unsigned long long m = __ballot(set);
    peers &= set ? m : ~m;
  }
  return peers;
}

// decided(r): called by all threads once the walk's outcome is known -- before the last step's draws are splatted, so that
// the next round's tasks leave first (what accept_item needs a dry walk for).
template <int kMode, class Decided>
LD_DEV AcceptResult accept_item_wide(const DrawArgs &a, AcceptShared &sh, AcceptWideShared &ws, uint32_t item, const ItemProg pg,
                                     const ItemVisit &h, const uint32_t *res, const uint32_t *res_prev, uint32_t &rmin, uint32_t &rmax_p1,
                                     Decided &&decided) {
  (void)item;
  uint32_t(*s_pix)[64] = sh.pix;
  float *s_val = sh.val;
  uint32_t *s_off = sh.off;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const uint32_t retries = (uint32_t)a.retries;
  const uint32_t S = h.samples, max_total = S * 5u, m_limit = max_total + retries;
  const float ae = h.I.add_energy, w = h.w;
  const unsigned long long zk = (a.F.zkey || a.F.zkey_dbg) ? closest_key_of(a.ctr, h.I.depth, visit_gid(a.V, h.visit)) : 0ull;
  // per-item add table: gaussian AOVs' (value + add_energy) * w, then the weight itself
  uint32_t U = 1;
  for (uint32_t k = 0; k < a.F.n_aovs; ++k) if (!(a.F.closest_mask & (1u << k))) U += 4;
#ifdef LENTIL_TIMELINE
  unsigned long long tp_ = __builtin_amdgcn_s_memrealtime();
#define LENTIL_PHASE(I) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0) dbg_add(20 + (I), t_ - tp_); tp_ = t_; } while (0)
#else
#define LENTIL_PHASE(I) do {} while (0)
#endif
  block_sync_lds();          // (the add table and the masks of the item before)
  LENTIL_PHASE(0);
  if (threadIdx.x < a.F.n_aovs * 4u) {
    const uint32_t k = threadIdx.x >> 2, c = threadIdx.x & 3u;
    if (!(a.F.closest_mask & (1u << k))) {
      uint32_t slot = 0;
      for (uint32_t j = 0; j < k; ++j) if (!(a.F.closest_mask & (1u << j))) slot += 4;
      const float4 v = k == 0 ? h.rgba : a.V.extra[k - 1][h.visit];
      const float vc = c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
      s_val[slot + c] = (vc + ae) * w;
      s_off[slot + c] = 4u * k + c;
    }
  }
  if (threadIdx.x == 0) { s_val[U - 1] = w; s_off[U - 1] = 4u * a.F.n_aovs; }
  const bool replay = kMode == 2 && pg.uacc != 0u;
  uint32_t n = pg.n_done, acc = pg.accepted, last_ok = pg.last_ok;
  uint32_t uacc = 0;               // mode 1: unknown attempts so far; mode 2: the same, replayed
  uint32_t acc1 = 0;               // mode 2: known successes mode 1 counted so far
  bool stalled = false;
  // Lane (dd, ch) = (lane / U, lane % U) adds float ch of every (64 / U)-th draw of the wave's list: the division and the
  // table entries once per item, not once per atomic
  const uint32_t dpi = 64u / U;                                   // draws per atomic instruction
  const uint32_t my_dd = lane / U, my_ch = lane - my_dd * U;
  const bool adds = my_dd < dpi;
  uint32_t pix_bits = 1;
  while ((a.F.np - 1u) >> pix_bits) ++pix_bits;                  // width of a pixel index
  constexpr uint32_t kStep = 256u * (uint32_t)kWide;
  constexpr uint32_t kSlabs = (uint32_t)kWide * 4u;
  static_assert(kSlabs <= 32, "one lane of wave 0 per slab");
  while (!stalled && acc < S && n < max_total) {
    // ---- the step's window
    block_sync_lds();              // (the window and the masks of the step before, the add table)
    const uint32_t span = max_total - n < kStep ? max_total - n : kStep;      // attempts of this step that exist at all
    // slabs of 256 attempts that hold anything: what the batch covers (beyond m_hi every R(m) is "beyond", i.e. unresolved)
    uint32_t jn = (span + 255u) / 256u;
    {
      const uint32_t covered = pg.m_hi > n ? pg.m_hi - n : 0u;      // attempts with at least their first try inside the batch
      const uint32_t jc = covered / 256u + 1u;
      if (jc < jn) jn = jc;
    }
    if (a.early_accept) {
      // accept_kernel<4>: the solve kernel that wrote these results is still running on other CUs -- they were written through
      // with agent-scope atomics and are read with returning atomics (ld_coherent*: an L2 of this XCD may hold the line from
      // before), two results at a time where the batch lies in one piece (an item's first batch always does)
      const uint32_t count = jn * 256u + retries;
      if (pg.p_hi == pg.p_lo && n >= pg.m_lo) {
        const uint32_t e0 = pg.res_off + (n - pg.m_lo), odd = e0 & 1u;
        for (uint32_t j = threadIdx.x; 2u * j < count + odd; j += 256u) {
          const uint32_t e = e0 - odd + 2u * j;                      // even: an 8-byte word of the pool
          uint32_t lo, hi;
          if ((uint64_t)e + 1u < a.pool_cap) { const uint64_t w = ld_coherent64(res + e); lo = (uint32_t)w; hi = (uint32_t)(w >> 32); }
          else { lo = ld_coherent32(res + e); hi = kCodeBeyond; }
          const uint32_t i1 = 2u * j + 1u - odd;                    // window index of the word's upper half (the lower: i1 - 1)
          if (i1 >= 1u && i1 - 1u < count) ws.win[i1 - 1u] = n + (i1 - 1u) >= pg.m_hi ? kCodeBeyond : lo;
          if (i1 < count) ws.win[i1] = n + i1 >= pg.m_hi ? kCodeBeyond : hi;
        }
      } else {
        for (uint32_t i = threadIdx.x; i < count; i += 256u) {
          const uint32_t m = n + i;
          uint32_t c = kCodeFail;
          if (m >= pg.m_hi) c = kCodeBeyond;
          else if (m >= pg.m_lo) c = ld_coherent32(res + pg.res_off + (m - pg.m_lo));
          else if (m >= pg.p_lo && m < pg.p_hi) c = ld_coherent32(res_prev + pg.p_off + (m - pg.p_lo));
          ws.win[i] = c;
        }
      }
    } else
    for (uint32_t i = threadIdx.x; i < jn * 256u + retries; i += 256u) ws.win[i] = result_at2(res, res_prev, pg, n + i);
    if (threadIdx.x < kSlabs) { ws.first_u[threadIdx.x] = kStep; ws.m_succ[threadIdx.x] = 0ull; ws.m_unk[threadIdx.x] = 0ull; ws.m_succ1[threadIdx.x] = 0ull; }
    block_sync_lds();
    LENTIL_PHASE(1);
    // ---- pass A: every attempt resolved once; per slab of 64 the ballots of the unresolved, the successes (modes 0, 2:
    // final; mode 1: as far as known), the unknown attempts and mode 1's successes (mode 2's replay)
#pragma unroll 1
    for (uint32_t j = 0; j < jn; ++j) {
      const uint32_t my_i = j * 256u + wave * 64u + lane;
      const uint32_t my_n = n + my_i;
      bool unres = false, succ = false, unk1 = false, succ1 = false;
      if (my_i < span) {
        const WideAttempt<kMode> r = wide_resolve<kMode>(a, ws, pg, my_i, retries, m_limit);
        unres = r.unresolved;
        succ = !unk_blocks<kMode>(r.unk) && r.code < kCodePendingBase;
        const bool was1 = kMode == 2 && replay && my_n < pg.n_end1;
        unk1 = kMode == 1 ? r.unk : (was1 && r.unk);
        succ1 = was1 && !r.unk && r.code1 < kCodePendingBase;
        ws.code[my_i] = r.code;
      }
      const unsigned long long umask = __ballot(unres);
      const unsigned long long smask = __ballot(succ);
      const unsigned long long u1mask = kMode != 0 ? __ballot(unk1) : 0ull;
      const unsigned long long s1mask = kMode == 2 ? __ballot(succ1) : 0ull;
      if (lane == 0) {
        const uint32_t q = j * 4u + wave;
        ws.first_u[q] = umask ? my_i + (uint32_t)__builtin_ctzll(umask) : kStep;
        ws.m_succ[q] = smask; ws.m_unk[q] = u1mask; ws.m_succ1[q] = s1mask;
      }
    }
    block_sync_lds();
    // ---- wave 0: the step's limit (first unresolved attempt; all later ones are unresolved too), the ballots cut there,
    // their prefix, the highest accepted attempt
    if (wave == 0) {
      const uint32_t q = lane < kSlabs ? lane : kSlabs - 1u;
      uint32_t fu = lane < kSlabs ? ws.first_u[q] : kStep;
      for (int off = 16; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(fu, off); fu = o < fu ? o : fu; }
      fu = (uint32_t)__builtin_amdgcn_readfirstlane((int)fu);
      const uint32_t lim = fu < span ? fu : span;
      const uint32_t start = q * 64u;
      unsigned long long keep = 0ull;
      if (lane < kSlabs && start < lim) keep = lim - start >= 64u ? ~0ull : ((1ull << (lim - start)) - 1ull);
      const unsigned long long m0 = ws.m_succ[q] & keep, m1 = ws.m_unk[q] & keep, m2 = ws.m_succ1[q] & keep;
      uint32_t tot[3];
      uint32_t bef0 = 0;
#pragma unroll
      for (int kind = 0; kind < 3; ++kind) {
        const uint32_t c = (uint32_t)__builtin_popcountll(kind == 0 ? m0 : (kind == 1 ? m1 : m2));
        uint32_t incl = c;
        for (int off = 1; off < 32; off <<= 1) { const uint32_t up = __shfl_up(incl, off); if ((int)lane >= off) incl += up; }
        if (lane < kSlabs) ws.before[kind][lane] = incl - c;
        if (kind == 0) bef0 = incl - c;
        tot[kind] = (uint32_t)__shfl((int)incl, (int)kSlabs - 1);
      }
      // highest accepted attempt: the slab's successes with rank < S are its first (S - acc - before) ones
      uint32_t top = 0u;
      {
        const uint32_t cnt_s = (uint32_t)__builtin_popcountll(m0);
        const uint32_t room = acc + bef0 < S ? S - acc - bef0 : 0u;
        if (lane < kSlabs && room && cnt_s) {
          uint32_t pos = 63u - (uint32_t)__builtin_clzll(m0);
          if (room < cnt_s) {
            unsigned long long m = m0;
            for (uint32_t b = 0; b < room; ++b) { pos = (uint32_t)__builtin_ctzll(m); m &= m - 1ull; }
          }
          top = n + start + pos;
        }
        for (int off = 16; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(top, off); top = o > top ? o : top; }
      }
      if (lane < kSlabs) { ws.m_succ[q] = m0; ws.m_unk[q] = m1; ws.m_succ1[q] = m2; }
      if (lane == 0) { ws.limit = lim; ws.stalled = fu < span ? 1u : 0u; ws.total[0] = tot[0]; ws.total[1] = tot[1]; ws.total[2] = tot[2]; ws.top = top; }
    }
    block_sync_lds();
    LENTIL_PHASE(2);
    const uint32_t limit = ws.limit;
    if (ws.stalled) stalled = true;
    const uint32_t total = ws.total[0], utotal = kMode != 0 ? ws.total[1] : 0u, total1 = kMode == 2 ? ws.total[2] : 0u;
    const uint32_t taken = total < S - acc ? total : S - acc;
    const uint32_t new_last_ok = taken ? ws.top : last_ok;
    // ---- if this is the walk's last step: the decision about the item's next batch, which goes out before anything is splatted
    {
      const uint32_t n2 = n + limit, acc2 = acc + taken, uacc2 = uacc + utotal;
      if (stalled || !(acc2 < S && n2 < max_total)) {
        AcceptResult r;
        r.samples = S;
        r.prog = pg;
        r.prog.n_done = n2; r.prog.accepted = acc2; r.prog.last_ok = new_last_ok;
        r.prog.uacc = kMode == 1 ? uacc2 : 0u;
        accept_next_batch<kMode>(a, pg, r, n2, acc2, uacc2, S, max_total, m_limit, retries);
        LENTIL_PHASE(3);
        decided(r);
        LENTIL_PHASE(4);
      }
    }
    // ---- pass B: what is splatted now
    const uint32_t my_off = adds ? s_off[my_ch] : 0u;
    const float my_val = adds ? s_val[my_ch] : 0.f;
#pragma unroll 1
    for (uint32_t j = 0; j < jn; ++j) {
      const uint32_t my_i = j * 256u + wave * 64u + lane;
      const uint32_t my_n = n + my_i;
      const uint32_t q = j * 4u + wave;
      const unsigned long long smask = ws.m_succ[q];
      if (smask == 0ull) continue;      // (no success in this slab: nothing taken, nothing splatted)
      const bool succ = (smask >> lane) & 1ull;
      const uint32_t rank = acc + ws.before[0][q] + (uint32_t)__builtin_popcountll(smask & lt_mask);
      const bool take = succ && rank < S;              // mode 1: among the first S *known* successes
      bool splat = take;
      if (kMode == 1) {
        const unsigned long long u1mask = ws.m_unk[q];
        const uint32_t ub = uacc + ws.before[1][q] + (uint32_t)__builtin_popcountll(u1mask & lt_mask);
        splat = take && rank + ub < S;
      } else if (kMode == 2) {
        const unsigned long long u1mask = ws.m_unk[q], s1mask = ws.m_succ1[q];
        const uint32_t ub = uacc + ws.before[1][q] + (uint32_t)__builtin_popcountll(u1mask & lt_mask);
        const uint32_t rank1 = acc1 + ws.before[2][q] + (uint32_t)__builtin_popcountll(s1mask & lt_mask);
        const bool succ1 = (s1mask >> lane) & 1ull;
        splat = take && !(succ1 && rank1 + ub < S);    // mode 1 has added that one
      }
      const unsigned long long pmask0 = __ballot(splat);
      if (pmask0 == 0ull) continue;
      // Camera::add_to_buffer, src/lentil.h:827-830 -- transposed: the wave's accepted pixels go through
      // LDS, then lane (dd, ch) adds float ch of accepted draw dd, dd + 64 / U, ...: consecutive lanes hit consecutive
      // floats of one pixel record.  Draws of the slab on the same pixel are added as ONE atomic of count x value
      // (count_same_pixel in accept_item; here from a match over the pixel index's bits).
      const uint32_t pix = splat ? ws.code[my_i] : ((1u << pix_bits) | lane);      // (lanes without a draw: 64 values no pixel has)
      const unsigned long long peers = match_any_bits(pix, pix_bits + 1u) & pmask0;
      uint32_t slot = 0, cnt = 0;
      const bool leader = splat && (uint32_t)__builtin_ctzll(peers) == lane;      // the first lane of every pixel
      const unsigned long long lmask = __ballot(leader);
      if (splat) {
        const uint32_t row = pix / a.P.xres;
        rmin = row < rmin ? row : rmin;
        rmax_p1 = row + 1u > rmax_p1 ? row + 1u : rmax_p1;
        if (a.F.touched) { a.F.touched[pix >> 6] = a.round ? 2 : 1; a.F.touched_px[pix] = a.round ? 2 : 1; }      // (2: by a round after the first, see resolve_touched_kernel)
        if (a.F.zkey) atomicMin(a.F.zkey + pix, zk);
        if (a.F.zkey_dbg) atomicMin(a.F.zkey_dbg + pix, zk);      // value.r = samples != 0 for every draw
        if (a.log_cap) {
          const unsigned long long li = wave_log_slots(a.log_count, pmask0, lane);
          if (li < a.log_cap) { a.log[li].visit = h.visit; a.log[li].attempt = my_n; a.log[li].pixel = pix; }
        }
      }
      if (leader) {
        slot = (uint32_t)__builtin_popcountll(lmask & lt_mask);
        cnt = (uint32_t)__builtin_popcountll(peers);
        s_pix[wave][slot] = pix;
        sh.cnt[wave][slot] = cnt;
      }
      const uint32_t T = (uint32_t)__builtin_popcountll(lmask);       // distinct pixels of the slab
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      for (uint32_t d = my_dd; adds && d < T; d += dpi) {
        const uint32_t c = sh.cnt[wave][d];
        const uint32_t px_ = s_pix[wave][d];
        atomicAdd(a.F.acc + (size_t)px_ * a.F.stride + my_off, (float)c * my_val);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
    }
    LENTIL_PHASE(5);
    if (taken) { last_ok = new_last_ok; acc += taken; }
    uacc += utotal;
    if (kMode == 2) acc1 += total1;       // (mode 1 capped its count at S; beyond that rank1 >= S either way)
    n += limit;
  }
  AcceptResult r;
  r.samples = S;
  r.prog = pg;
  r.prog.n_done = n; r.prog.accepted = acc; r.prog.last_ok = last_ok;
  r.prog.uacc = kMode == 1 ? uacc : 0u;
  accept_next_batch<kMode>(a, pg, r, n, acc, uacc, S, max_total, m_limit, retries);
  if (!(pg.accepted < S && pg.n_done < max_total)) decided(r);      // (an item whose walk has no step: nothing to look at)
  return r;
}

// Chromatic mode (abb_chromatic != 0), src/lentil_filter.cpp:248-299: every attempt traces three
// wavelength channels with the same aperture draws; each channel that gets through and lands inside the
// frame is splatted (into its own colour component, three-fold, when abb_chromatic > 0), each one that
// does not takes one off `count`, and the attempt itself adds one: count += successes - 2.  The loop
// runs while count < samples and attempts < 5 * samples, so count can go down and an attempt is
// executed iff the running count was below `samples` before it -- a block-wide prefix sum and the
// first index where it reaches `samples` (steps are at most +1, so it is hit exactly).
LD_DEV AcceptResult accept_item_chroma(const DrawArgs &a, AcceptShared &sh, uint32_t item, const ItemProg pg,
                                       const uint32_t *res, uint32_t &rmin, uint32_t &rmax_p1) {
  uint32_t(*s_pix)[64] = sh.pix;
  float *s_val = sh.val;
  uint32_t *s_off = sh.off;
  int *s_sum = reinterpret_cast<int *>(sh.nsucc);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const uint32_t retries = (uint32_t)a.retries;
  const double lens_length = a.lens ? a.lens->length : 0.0;
  const ItemVisit h = load_item_visit(a, item, lens_length);
  const uint32_t S = h.samples, max_total = S * 5u, m_limit = max_total + retries;
  const float ae = h.I.add_energy, w = h.w;
  const unsigned long long zk = (a.F.zkey || a.F.zkey_dbg) ? closest_key_of(a.ctr, h.I.depth, visit_gid(a.V, h.visit)) : 0ull;
  uint32_t U = 1;
  for (uint32_t k = 0; k < a.F.n_aovs; ++k) if (!(a.F.closest_mask & (1u << k))) U += 4;
  if (threadIdx.x < a.F.n_aovs * 4u) {
    const uint32_t k = threadIdx.x >> 2, c = threadIdx.x & 3u;
    if (!(a.F.closest_mask & (1u << k))) {
      uint32_t slot = 0;
      for (uint32_t j = 0; j < k; ++j) if (!(a.F.closest_mask & (1u << j))) slot += 4;
      const float4 v = k == 0 ? h.rgba : a.V.extra[k - 1][h.visit];
      const float vc = c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
      s_val[slot + c] = (vc + ae) * w;
      s_off[slot + c] = 4u * k + c;
    }
  }
  if (threadIdx.x == 0) { s_val[U - 1] = w; s_off[U - 1] = 4u * a.F.n_aovs; }
  __syncthreads();
  const uint32_t cnt = pg.m_hi - pg.m_lo;          // results per channel in this batch
  uint32_t n = pg.n_done, last_ok = pg.last_ok, splats = pg.splats;
  int count = (int)pg.accepted;
  bool stalled = false;
  while (!stalled && count < (int)S && n < max_total) {
    const uint32_t my_i = wave * 64u + lane;
    const uint32_t my_n = n + my_i;
    const bool valid = my_n < max_total;
    uint32_t code[3] = {kCodeFail, kCodeFail, kCodeFail};
    bool unresolved = false;
    if (retries <= kAcceptWinRetries) {
      for (uint32_t c = 0; c < 3; ++c) stage_results(sh.rwin[c], res, pg, pg.res_off + c * cnt, n, 256u + retries);
      __syncthreads();
      if (valid) {
#pragma unroll
        for (uint32_t c = 0; c < 3; ++c) {
          for (uint32_t t = 0; t <= retries; ++t) {
            const uint32_t r = sh.rwin[c][my_i + t];
            if (r == kCodeBeyond) { unresolved = unresolved || (pg.m_hi < m_limit); break; }
            if (r != kCodeFail) { code[c] = r; break; }
          }
        }
      }
    } else if (valid) {
#pragma unroll
      for (uint32_t c = 0; c < 3; ++c) {
        for (uint32_t t = 0; t <= retries; ++t) {
          const uint32_t m = my_n + t;                 // m >= n_done == m_lo: a stalled batch is re-solved from n_done
          if (m >= pg.m_hi) { unresolved = unresolved || (pg.m_hi < m_limit); break; }
          const uint32_t r = (m < pg.m_lo) ? kCodeFail : res[pg.res_off + c * cnt + (m - pg.m_lo)];
          if (r != kCodeFail) { code[c] = r; break; }
        }
      }
    }
    const unsigned long long umask = __ballot(valid && unresolved);
    if (lane == 0) sh.first_u[wave] = umask ? wave * 64u + (uint32_t)__builtin_ctzll(umask) : 256u;
    __syncthreads();
    uint32_t limit = max_total - n < 256u ? max_total - n : 256u;
    {
      uint32_t fu = sh.first_u[0];
      if (sh.first_u[1] < fu) fu = sh.first_u[1];
      if (sh.first_u[2] < fu) fu = sh.first_u[2];
      if (sh.first_u[3] < fu) fu = sh.first_u[3];
      if (fu < limit) { limit = fu; stalled = true; }
    }
    const bool in_step = my_i < limit;
    bool ok[3];
    int delta = 0;
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) { ok[c] = in_step && code[c] < kCodeOut; delta += ok[c] ? 1 : 0; }
    delta = in_step ? delta - 2 : 0;
    // block-wide inclusive prefix sum of delta
    int incl = delta;
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if ((int)lane >= off) incl += up;
    }
    if (lane == 63) s_sum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (uint32_t k = 0; k < wave; ++k) before += s_sum[k];
    const int running = count + before + incl;                  // count after attempt my_i
    const unsigned long long rmask = __ballot(in_step && running >= (int)S);
    if (lane == 0) sh.top[wave] = rmask ? wave * 64u + (uint32_t)__builtin_ctzll(rmask) : 256u;
    __syncthreads();
    uint32_t istar = sh.top[0];
    if (sh.top[1] < istar) istar = sh.top[1];
    if (sh.top[2] < istar) istar = sh.top[2];
    if (sh.top[3] < istar) istar = sh.top[3];
    const bool reached = istar < 256u;
    const bool executed = in_step && (!reached || my_i <= istar);
    uint32_t my_splats = 0;
#pragma unroll
    for (uint32_t c = 0; c < 3; ++c) {
      const bool take = executed && ok[c];
      my_splats += take ? 1u : 0u;
      const unsigned long long tmask = __ballot(take);
      const uint32_t T = (uint32_t)__builtin_popcountll(tmask);
      if (take) {
        const uint32_t pix = code[c];
        const uint32_t row = pix / a.P.xres;
        rmin = row < rmin ? row : rmin;
        rmax_p1 = row + 1u > rmax_p1 ? row + 1u : rmax_p1;
        s_pix[wave][(uint32_t)__builtin_popcountll(tmask & lt_mask)] = pix;
        if (a.F.touched) { a.F.touched[pix >> 6] = a.round ? 2 : 1; a.F.touched_px[pix] = a.round ? 2 : 1; }      // (as accept_item does: a later round's splats are resolved again)
        if (a.F.zkey) atomicMin(a.F.zkey + pix, zk);
        if (a.F.zkey_dbg) atomicMin(a.F.zkey_dbg + pix, zk);      // value.r = samples != 0 for every draw
        if (a.log_cap) {
          const unsigned long long li = wave_log_slots(a.log_count, tmask, lane);
          if (li < a.log_cap) { a.log[li].visit = h.visit; a.log[li].attempt = my_n | (c << 30); a.log[li].pixel = pix; }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      count_same_pixel(s_pix[wave], sh.cnt[wave], T, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      for (uint32_t q = lane; q < T * U; q += 64u) {
        const uint32_t d = q / U, ch = q - d * U;
        const uint32_t same = sh.cnt[wave][d];
        if (!same) continue;
        float val = s_val[ch];
        if (a.chroma_weights && ch != U - 1u) {
          const uint32_t comp = s_off[ch] & 3u;                // r, g, b, a of the AOV
          if (comp < 3u) {
            if (comp != c) continue;                            // rgb_weight is 0 there: nothing to add
            val = val * 3.0f;                                   // (value + add_energy) * w * rgb_weight
          }
        }
        atomicAdd(a.F.acc + (size_t)s_pix[wave][d] * a.F.stride + s_off[ch], (float)same * val);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
    }
    // block totals: splats, and the count after the last executed attempt
    int sp = (int)my_splats;
    int dsum = executed ? delta : 0;
    for (int off = 32; off > 0; off >>= 1) { sp += __shfl_down(sp, off); dsum += __shfl_down(dsum, off); }
    __syncthreads();
    if (lane == 0) { s_sum[wave] = sp; reinterpret_cast<int *>(sh.first_u)[wave] = dsum; }
    __syncthreads();
    splats += (uint32_t)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
    const int *s_d = reinterpret_cast<const int *>(sh.first_u);
    count += s_d[0] + s_d[1] + s_d[2] + s_d[3];
    if (reached) { n += istar + 1u; last_ok = n - 1u; }
    else n += limit;
    __syncthreads();
  }
  AcceptResult r;
  r.samples = S;
  r.prog = pg;
  r.prog.n_done = n; r.prog.accepted = (uint32_t)count; r.prog.last_ok = last_ok; r.prog.splats = splats;
  r.more = count < (int)S && n < max_total && pg.m_hi < m_limit && pg.m_hi > 0;
  r.new_lo = n;            // re-solve from the stalled attempt: a channel resolved in the old batch must stay readable
  r.new_hi = pg.m_hi;
  if (r.more) {
    const uint32_t remaining = S - (uint32_t)(count > 0 ? count : 0);
    unsigned long long need = count > 0 ? ((unsigned long long)remaining * n + (uint32_t)count - 1) / (uint32_t)count
                                        : (unsigned long long)(max_total - n);
    const uint32_t m16 = (unsigned long long)(count > 0 ? count : 0) * 16ull < (unsigned long long)n * a.margin_low_rate ? 2u * a.batch_margin16 : a.batch_margin16;
    need += (need * m16) / 16 + 32;       // (as accept_item)
    unsigned long long n_target = (unsigned long long)n + need;
    if (n_target > max_total) n_target = max_total;
    uint32_t new_hi = (uint32_t)n_target + retries;
    if (new_hi > m_limit) new_hi = m_limit;
    r.new_hi = new_hi;
  }
  return r;
}

// (six waves per SIMD: beside four accept blocks per CU a SIMD then has room for a solve wave -- the next round's solves
// run beside the accept that schedules them, DrawArgs::emit_live)
// kMode: accept_item's -- 1 for the first accept of a decoupled streamed pass, 2 for the one behind it, 0 otherwise.
#ifndef LENTIL_ACCEPT_EU
#define LENTIL_ACCEPT_EU 5
#endif
template <int kMode>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LENTIL_ACCEPT_EU, LENTIL_ACCEPT_EU))) void accept_kernel(DrawArgs a) {
  LENTIL_TL_SPAN(kMode == 0 ? SPAN_ACCEPT0 : (kMode == 2 ? SPAN_ACCEPT2 : SPAN_ACCEPT1));
  // kMode 3 (round 6): the first accept of a lean streamed pass that neither waits for the round's parked solves nor splats
  // around them.  It runs behind the solve kernel and BESIDE the straggler kernel, takes every item whose parked solves are all
  // through (ItemHdr::parked == parked_done) and walks it once, with final results -- mode 2's walk for an item that carries
  // nothing over -- and leaves the others untouched on the next accept's list (mode 2, behind the stragglers).  Nothing is
  // walked twice (mode 1 splatted what was certain of nearly every item and mode 2 then replayed nearly every item: 134 us
  // behind the stragglers for a headline frame); what is left for the accept behind the stragglers is the items whose
  // stragglers were still at work when this kernel came to them.  It never waits for anything.
  // kMode 4: mode 3 BESIDE the solve kernel (DrawArgs::early_accept): launched behind the publishers, it draws tickets on the queue
  // of items whose first batch is complete -- pushed by the solve wave that delivered the item's last result -- and walks each
  // as it comes; its last block leaves when the pass's last item is through, a walk or two behind the solve kernel's last wave
  // instead of a whole accept (160-190 us for a 4K headline frame).  It waits for the solve kernel (resident since the pass
  // began, holding everything it will ever need), never for the stragglers: an item with a parked solve still out goes to the
  // accept behind them, as in mode 3.
  constexpr int kWalk = (kMode == 3 || kMode == 4) ? 2 : kMode;
  __shared__ uint32_t s_item, s_ready;
  __shared__ AcceptShared sh;
  __shared__ AcceptWideShared ws;
  static_assert(sizeof(ws.win) >= 3 * (256 + 64) * sizeof(uint32_t), "AcceptShared::rwin lies in AcceptWideShared::win");
  if (threadIdx.x == 0) sh.rwin = reinterpret_cast<uint32_t (*)[256 + 64]>(ws.win);      // (the loop below begins with a barrier)
#ifdef LENTIL_ACCEPT_PRIO
  // the accept stands between two rounds of solves: beside the next round's solve and straggler waves its instructions go first
  __builtin_amdgcn_s_setprio(LENTIL_ACCEPT_PRIO);
#endif
  const uint32_t par = (uint32_t)a.parity, nxt = par ^ 1u;
  if (threadIdx.x == 0) atomicAdd(&a.ctr->accept_started[par], 1u);
#ifdef LENTIL_PROBE_BUILD
  if (a.dispatch_probe && kMode == 1 && threadIdx.x == 0) atomicAdd(&a.ctr->probe_accept_xcc[xcc_id()], 1u);
#endif
  if (a.lean_gate && a.ctr->n_tasks[par] != 0u) return;       // (tasks nobody has solved yet: DrawArgs::lean_gate)
  // (a streamed or blind pass whose buffers were too small: nothing is accepted, the host redoes the draws)
  const uint32_t n_active = (a.ctr->fallback || a.ctr->stuck) ? 0u : a.ctr->n_active[par];
  // (every block says so, not block 0 alone: a pass that stalls with a part of this grid never begun -- block 0 perhaps among it --
  // must still know that draws were added, or its recovery adds them a second time)
  if (threadIdx.x == 0 && n_active) atomicMax(&a.ctr->rounds_used, (unsigned long long)(a.round + 1));
  const uint32_t *res = a.pool[par], *res_prev = a.pool[nxt];
  unsigned long long tot_attempted = 0, tot_accepted = 0;
  uint32_t rmin = 0x7FFFFFFFu, rmax_p1 = 0u;       // rows this thread's accepted draws went to
  // The next round's solve kernel is waiting for tasks (emit_live): a block takes its share of the items at once, finds
  // out what each of them still needs with a dry walk (a third of the real one's time: no splats) and hands those tasks
  // out before it splats anything -- the last tasks leave ~0.06 ms into the kernel instead of ~0.15 ms.
  // (accept_item_wide needs no dry walk: it knows what an item needs before it splats the item's last step.)
  constexpr uint32_t kGroup = 8;
  __shared__ uint32_t s_emit_off[kGroup], s_emit_hi[kGroup], s_emitted[kGroup];
  const bool chroma = a.n_channels == 3;
  // (LENTIL_ACCEPT_WIDE=0 -> DrawArgs::accept_narrow: the 256-attempt steps of accept_item for everything)
  // (... and for records wider than one wave: accept_item_wide's lane (draw, float) layout serves 64 / U draws per atomic
  // instruction, which is none at U = 65 -- sixteen gaussian AOVs, LENTIL_MAX_AOVS; accept_item loops over draw x float)
  uint32_t add_floats = 1;
  for (uint32_t k = 0; k < a.F.n_aovs; ++k) if (!(a.F.closest_mask & (1u << k))) add_floats += 4;
  const bool wide = !chroma && (uint32_t)a.retries <= kAcceptWinRetries && !a.accept_narrow && add_floats <= 64u;
  const bool dry_first = a.emit_live && !chroma && !wide;
  uint32_t per = 1;
  if (dry_first || wide) { per = (n_active + gridDim.x - 1u) / gridDim.x; per = per < 1u ? 1u : (per > kGroup ? kGroup : per); }
  const double lens_length = a.lens ? a.lens->length : 0.0;
  // what thread 0 does with an item's outcome: progress record, the next round's list and tasks, the pass's totals
  auto finish_item = [&](uint32_t item, const ItemProg &pg, const AcceptResult &r, bool emitted, uint32_t off, uint32_t emitted_hi) {
    tl_add(TL_ITEMS_ACCEPTED, 1u);
    if (kMode == 1) {
      const uint32_t u = r.prog.uacc;
      dbg_add(8 + (u == 0 ? 0 : (u <= 4 ? 1 : (u <= 16 ? 2 : (u <= 64 ? 3 : (u <= 256 ? 4 : 5))))), 1);
      dbg_add(14, u);
      if (u) dbg_add(15, r.more ? 1 : 0);
    }
    if (kMode == 2 && pg.uacc) { dbg_add(16, 1); dbg_add(17, r.more ? 1 : 0); dbg_add(18, r.prog.accepted); dbg_add(19, r.samples); }
    if (kMode == 1 && r.prog.uacc != 0u) {
      // met attempts whose solves are still parked: the next accept walks the item again, this batch beside the next
      ItemProg np_ = pg;
      np_.uacc = r.prog.uacc;
      np_.n_end1 = r.prog.n_done;
      np_.p_lo = pg.m_lo; np_.p_hi = pg.m_hi; np_.p_off = pg.res_off;
      np_.m_lo = np_.m_hi = pg.m_hi; np_.res_off = 0;
      if (!emitted && r.more && !a.lean_defer) { emitted = emit_tasks(a, nxt, item, r.new_lo, r.new_hi, off); emitted_hi = r.new_hi; }
      if (emitted) { np_.m_hi = emitted_hi; np_.res_off = off; }
      a.prog[item] = np_;
      const uint32_t slot = atomicAdd(&a.ctr->n_active[nxt], 1u);
      a.active[nxt][slot] = item;
    } else if (r.more) {
      ItemProg np_ = r.prog;
      np_.m_lo = r.new_lo; np_.m_hi = r.new_hi;
      np_.uacc = 0; np_.p_lo = np_.p_hi = np_.p_off = 0; np_.n_end1 = 0;
      if (emitted) np_.m_hi = emitted_hi;         // (the dry walk's word: the same, its view of the results being the same)
      if (emitted || emit_tasks(a, nxt, item, r.new_lo, r.new_hi, off)) {
        np_.res_off = off;
        a.prog[item] = np_;
        const uint32_t slot = atomicAdd(&a.ctr->n_active[nxt], 1u);
        a.active[nxt][slot] = item;
      }
    } else {
      // total_samples_taken when the reference's loop ends, src/lentil_filter.cpp:248
      tot_attempted += ((int)r.prog.accepted >= (int)r.samples) ? (unsigned long long)r.prog.last_ok + 1ull
                                                                 : (unsigned long long)r.samples * 5ull;
      tot_accepted += chroma ? r.prog.splats : r.prog.accepted;
    }
  };
  // emit_live: the end markers behind the next round's task queue, by the first 64 threads of whichever block finds the
  // queue complete -- the block that finishes the pass's last item, not the grid's last block: the solve kernel that waits
  // for them must not depend on blocks of this grid that have not been dispatched ("a stall found and removed")
  __shared__ uint32_t s_write_end;
  auto write_end_markers = [&]() {
    if (threadIdx.x < 64u) {
      const uint32_t n = ld_coherent32(&a.ctr->n_tasks[nxt]);
      for (uint32_t i = threadIdx.x; i < a.end_tasks; i += 64u)
        if ((uint64_t)n + i < a.task_cap)
          st_agent64(reinterpret_cast<uint64_t *>(a.tasks[nxt] + n + i) + 1, (uint64_t)(kEndCount | (a.epoch << kTaskTagShift)) << 32);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) st_agent32(&a.ctr->accept_final[par], 1u);
    }
  };
  auto ticket_done = [&](uint32_t cnt) {
    if constexpr (kMode == 2) { (void)cnt; return; }       // (the accept behind the stragglers feeds nobody beside it)
    if (!a.emit_live || a.inject_stall) return;
    __syncthreads();                         // (thread 0 has finished the ticket's items: their tasks are counted in n_tasks)
    if (threadIdx.x == 0) s_write_end = (atomicAdd(&a.ctr->accept_items_done[par], cnt) + cnt == n_active) ? 1u : 0u;
    __syncthreads();
    if (s_write_end) write_end_markers();
#ifdef LENTIL_PROBE_BUILD
    if (s_write_end && a.dispatch_probe && kMode == 1 && threadIdx.x == 0) {
      // the pass's last item is done: has every block of this grid begun?  If not, who is resident where
      const uint32_t begun = ld_coherent32(&a.ctr->accept_started[par]);
      if (begun < gridDim.x && ld_coherent32(&a.ctr->probe_snap[0]) == 0u) {
        unsigned int *sn = a.ctr->probe_snap;
        for (int x = 0; x < 8; ++x) sn[1 + x] = ld_coherent32(&a.ctr->probe_accept_xcc[x]);
        for (int k = 0; k < 3; ++k) for (int x = 0; x < 8; ++x) sn[9 + 8 * k + x] = ld_coherent32(&a.ctr->probe_res_xcc[k][x]);
        sn[33] = begun; sn[34] = gridDim.x;
        sn[0] = 1u;
      }
    }
#endif
  };
  if (kMode != 2 && a.emit_live && !a.inject_stall && n_active == 0u) {
    // nothing to accept: whichever block comes first closes the queue
    if (threadIdx.x == 0) s_write_end = atomicAdd(&a.ctr->accept_items_done[par], 1u) == 0u ? 1u : 0u;
    __syncthreads();
    if (s_write_end) write_end_markers();
  }
  while (true) {
    __syncthreads();
    uint32_t ai0 = 0, ready_item = 0;
    if constexpr (kMode == 4) {
      // a ticket on the ready queue; its slot is filled when the item it will hold is complete
      if (threadIdx.x == 0) {
        uint32_t item = 0xFFFFFFFFu;
        const uint32_t ticket = atomicAdd(&a.ctr->ready_head, 1u);
        if (ticket < n_active && ticket < a.ready_cap) {
          const uint32_t tag = (a.epoch << 8) | 1u;
          const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
          uint32_t naps = 1u, polls = 0u;
          while (true) {
            const uint64_t w = ld_coherent64(a.ready_q + ticket);
            if ((uint32_t)(w >> 32) == tag) { item = (uint32_t)w; break; }
            if ((++polls & 15u) == 0u && (ld_coherent32(&a.ctr->stuck) != 0u || ld_coherent64(&a.ctr->fallback) != 0ull)) break;      // the pass is void
            if (__builtin_amdgcn_s_memrealtime() - t0 > (a.stuck_ticks ? a.stuck_ticks : kStuckTicks)) {
              if (atomicCAS(&a.ctr->stuck, 0u, 2u | (ticket << 2)) == 0u) {
                unsigned int *si = a.ctr->stuck_info;
                si[0] = 0xACCu; si[1] = par; si[2] = ld_coherent32(&a.ctr->n_ready); si[3] = n_active; si[4] = ld_coherent32(&a.ctr->ready_head);
                si[5] = (uint32_t)(w >> 32); si[6] = blockIdx.x; si[7] = ld_coherent32(&a.ctr->task_head[par]);
              }
              break;
            }
            for (uint32_t i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(64);
            if (naps < 8u) naps <<= 1;
          }
        }
        s_item = item;
      }
      __syncthreads();
      if (s_item == 0xFFFFFFFFu) break;
      ready_item = s_item;
    } else {
      if (threadIdx.x == 0) s_item = atomicAdd(&a.ctr->active_head[par], per);
      __syncthreads();
      ai0 = s_item;
      if (ai0 >= n_active) break;
    }
    const uint32_t cnt = kMode == 4 ? 1u : (n_active - ai0 < per ? n_active - ai0 : per);
    if (wide) {
      for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t item = kMode == 4 ? ready_item : a.active[par][ai0 + j];
        const ItemProg pg = load_prog(a, item);
#ifdef LENTIL_TIMELINE
        const unsigned long long tm0_ = __builtin_amdgcn_s_memrealtime();
#endif
        const ItemVisit h = load_item_visit(a, item, lens_length);
#ifdef LENTIL_TIMELINE
        if (threadIdx.x == 0 && h.samples) { dbg_add(26, __builtin_amdgcn_s_memrealtime() - tm0_); dbg_add(27, 1); }
#endif
        if constexpr (kMode == 3 || kMode == 4) {
          if (threadIdx.x == 0) {
            const uint64_t w = ld_coherent64(reinterpret_cast<const uint64_t *>(a.hdr + item) + 4);
            const bool ready = (uint32_t)w == (uint32_t)(w >> 32);
            s_ready = ready ? 1u : 0u;
            if (!ready) {
              // to the accept behind the stragglers, whole: this round's batch becomes "the batch before" -- that accept runs with the
              // other parity, its own pool empty for this item (accept_item<1> hands an item over the same way, minus the replay)
              ItemProg np_ = pg;
              np_.uacc = 0; np_.n_end1 = 0;
              np_.p_lo = pg.m_lo; np_.p_hi = pg.m_hi; np_.p_off = pg.res_off;
              np_.m_lo = np_.m_hi = pg.m_hi; np_.res_off = 0;
              a.prog[item] = np_;
              const uint32_t slot = atomicAdd(&a.ctr->n_active[nxt], 1u);
              a.active[nxt][slot] = item;
            }
          }
          __syncthreads();
          const bool ready = s_ready != 0u;
          __syncthreads();
          if (!ready) continue;
        }
        (void)accept_item_wide<kWalk>(a, sh, ws, item, pg, h, res, res_prev, rmin, rmax_p1,
                                      [&](const AcceptResult &r) { if (threadIdx.x == 0) finish_item(item, pg, r, false, 0u, 0u); });
      }
      ticket_done(cnt);
      continue;
    }
    if constexpr (kMode == 3 || kMode == 4) {
      // (the host launches these modes for frames the wide walk serves; anything else goes to the accept behind the stragglers whole)
      if (threadIdx.x == 0)
        for (uint32_t j = 0; j < cnt; ++j) {
          const uint32_t item = kMode == 4 ? ready_item : a.active[par][ai0 + j];
          const ItemProg pg = load_prog(a, item);
          ItemProg np_ = pg;
          np_.uacc = 0; np_.n_end1 = 0;
          np_.p_lo = pg.m_lo; np_.p_hi = pg.m_hi; np_.p_off = pg.res_off;
          np_.m_lo = np_.m_hi = pg.m_hi; np_.res_off = 0;
          a.prog[item] = np_;
          const uint32_t slot = atomicAdd(&a.ctr->n_active[nxt], 1u);
          a.active[nxt][slot] = item;
        }
      continue;
    }
    if (dry_first) {
      for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t item = a.active[par][ai0 + j];
        const ItemProg pg = load_prog(a, item);
        const AcceptResult rd = accept_item<kWalk>(a, sh, item, pg, res, res_prev, rmin, rmax_p1, true);
        if (threadIdx.x == 0) {
          uint32_t off = 0;
          s_emitted[j] = (rd.more && emit_tasks(a, nxt, item, rd.new_lo, rd.new_hi, off)) ? 1u : 0u;
          s_emit_off[j] = off; s_emit_hi[j] = rd.new_hi;
        }
        __syncthreads();
      }
    }
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t item = a.active[par][ai0 + j];
      const ItemProg pg = load_prog(a, item);
      const AcceptResult r = chroma ? accept_item_chroma(a, sh, item, pg, res, rmin, rmax_p1)
                                    : accept_item<kWalk>(a, sh, item, pg, res, res_prev, rmin, rmax_p1);
      if (threadIdx.x == 0)
        finish_item(item, pg, r, dry_first && s_emitted[j] != 0u, dry_first ? s_emit_off[j] : 0u, dry_first ? s_emit_hi[j] : 0u);
      __syncthreads();       // (s_emit_*[j] and the add table in `sh` are the block's)
    }
    ticket_done(cnt);
  }
  if (threadIdx.x == 0) {
    if (tot_attempted) atomicAdd(&a.ctr->attempted, tot_attempted);
    if (tot_accepted) atomicAdd(&a.ctr->accepted, tot_accepted);
  }
  flush_row_range(a.ctr, rmin, rmax_p1);
  if (kMode != 2 && a.emit_live && threadIdx.x < 64u && !a.inject_stall) {
    // everything this block emitted has arrived; the last block puts the end markers behind the queue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t last = 0;
    if (threadIdx.x == 0) last = atomicAdd(&a.ctr->accept_done[par], 1u) == gridDim.x - 1u ? 1u : 0u;
    // (the grid's last block writes them once more -- same slots, same words: n_tasks has not moved since the last item)
    if (__builtin_amdgcn_readfirstlane(last)) write_end_markers();
  }
}

// ---------------------------------------------------------------------------------------
// Occlusion probes (include/lentil_hip.h, lentil_hip_set_occlusion_probe; src/lentil.h:613-629, src/lentil_filter.cpp:356-375).
// Between a round's solves and its accept: probe_list_kernel walks the round's tasks and, for every try whose result is not FAIL
// (it got through the lens; whether it landed in the frame or not), writes the segment the reference would probe -- from the
// sample's world position to cam_to_world * (lens point / unit) -- and where the try's result lives; the host answers a byte
// per segment; probe_apply_kernel turns the occluded tries' results into FAIL, which is what an occluded try is to the walk
// (polynomial optics: tries++, the attempt's next try is looked at; thin lens: the attempt is lost).  Samples the skydome
// supplied are exempt (sample_is_from_skydome, src/lentil_filter.cpp:119-133) and never listed.
// ---------------------------------------------------------------------------------------
struct ProbeArgs {
  lentil_probe_segment *seg;
  uint32_t *idx;                 // index into the round's result pool
  uint32_t cap;
  unsigned int *count;
  const uint8_t *occluded;
  float c2w[4][4];               // AiCameraToWorldMatrix (static camera)
  const float *c2w_keys;         // a moving camera: one per motion key, blended like the world-to-camera keys (CamMotion)
};

LD_DEV void probe_target(const lentil_params &P, const float c2w[4][4], float lx, float ly, float lz, float out[3]) {
  float div = 1.0f;
  if (P.unitModel == LENTIL_UNIT_MM) div = 0.1f;
  else if (P.unitModel == LENTIL_UNIT_DM) div = 10.0f;
  else if (P.unitModel == LENTIL_UNIT_M) div = 100.0f;
  const float c = 1.0f / div;                  // AtVector /= float multiplies by 1.0f / f (SDK, recalled)
  const float vx = lx * c, vy = ly * c, vz = lz * c;
  out[0] = vx * c2w[0][0] + vy * c2w[1][0] + vz * c2w[2][0] + c2w[3][0];      // AiM4PointByMatrixMult
  out[1] = vx * c2w[0][1] + vy * c2w[1][1] + vz * c2w[2][1] + c2w[3][1];
  out[2] = vx * c2w[0][2] + vy * c2w[1][2] + vz * c2w[2][2] + c2w[3][2];
}

__global__ __launch_bounds__(256) void probe_list_kernel(DrawArgs a, ProbeArgs pr) {
  const uint32_t par = (uint32_t)a.parity;
  const uint32_t n_tasks = a.ctr->n_tasks[par] < a.task_cap ? a.ctr->n_tasks[par] : a.task_cap;
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  const bool po = a.P.cameraType == LENTIL_POLYNOMIAL_OPTICS;
  for (uint32_t q = wave; q < n_tasks; q += n_waves) {
    const Task t = a.tasks[par][q];
    const uint32_t cnt = t.count & 0xFFu;
    const uint32_t code = lane < cnt ? a.pool[par][t.res_off + lane] : kCodeFail;
    bool want = code != kCodeFail && !code_is_pending(code);
    float org[3] = {0.f, 0.f, 0.f}, tgt[3] = {0.f, 0.f, 0.f};
    if (__ballot(want)) {
      const ItemHdr hd = a.hdr[t.item];
      const uint32_t v = a.work[t.item].x;
      const float4 pos_z = a.V.pos_z[v];
      const float4 raydir_time = a.V.raydir_time[v];
      const bool small = fabsf(pos_z.x) < kAiEpsilon && fabsf(pos_z.y) < kAiEpsilon && fabsf(pos_z.z) < kAiEpsilon;
      const bool from_skydome = ((double)pos_z.w == (double)kAiInfinite) || small;      // (an item at infinity exists only with enable_skydome)
      if (from_skydome) want = false;
      if (want) {
        float c2w[4][4];
        const CamMotion &cm = a.V.cam;
        if (cm.n >= 2u && pr.c2w_keys) {
          float tt = (raydir_time.w - cm.t0) * cm.inv_dt;
          tt = tt < 0.0f ? 0.0f : (tt > 1.0f ? 1.0f : tt);
          const float sc = tt * (float)(cm.n - 1u);
          uint32_t i0 = (uint32_t)sc;
          if (i0 > cm.n - 2u) i0 = cm.n - 2u;
          const float f = sc - (float)i0;
          const float *ka = pr.c2w_keys + (size_t)i0 * 16u, *kb = ka + 16;
          for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) c2w[r][c] = ((kb[r * 4 + c] - ka[r * 4 + c]) * f) + ka[r * 4 + c];
        } else {
          for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) c2w[r][c] = pr.c2w[r][c];
        }
        org[0] = pos_z.x; org[1] = pos_z.y; org[2] = pos_z.z;
        const uint32_t m = t.m_base + lane;
        if (po) {
          double ax = 0.0, ay = 0.0;
          po_aperture_sample(a.P, a.bokeh, a.bokeh.cdfRow, hd.seed_a, m, ax, ay);
          probe_target(a.P, c2w, (float)(-ax * 0.1), (float)(-ay * 0.1), 0.0f, tgt);       // src/lentil.h:614
        } else {
          const float cs[3] = {(float)hd.tx, (float)hd.ty, (float)hd.tz};
          TlRay ray;
          if (thinlens_ray(a.P, a.bokeh, a.bokeh.cdfRow, cs, hd.px_py & 0xFFFF, hd.px_py >> 16, m, ray)) probe_target(a.P, c2w, ray.lx, ray.ly, 0.0f, tgt);
          else want = false;       // (the optical vignetting test failed it: its result is not a pixel anyway)
        }
      }
    }
    const unsigned long long wmask = __ballot(want);
    if (wmask) {
      uint32_t base = 0;
      if (lane == (uint32_t)__builtin_ctzll(wmask)) base = atomicAdd(pr.count, (unsigned int)__builtin_popcountll(wmask));
      base = __shfl(base, __builtin_ctzll(wmask));
      const uint32_t slot = base + (uint32_t)__builtin_popcountll(wmask & lt_mask);
      if (want && slot < pr.cap) {
        lentil_probe_segment sg;
        sg.origin[0] = org[0]; sg.origin[1] = org[1]; sg.origin[2] = org[2];
        sg.target[0] = tgt[0]; sg.target[1] = tgt[1]; sg.target[2] = tgt[2];
        pr.seg[slot] = sg;
        pr.idx[slot] = t.res_off + lane;
      }
    }
  }
}

__global__ __launch_bounds__(256) void probe_apply_kernel(DrawArgs a, ProbeArgs pr, uint32_t n) {
  uint32_t *res = a.pool[(uint32_t)a.parity];
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (pr.occluded[i]) res[pr.idx[i]] = kCodeFail;
}

// ---------------------------------------------------------------------------------------
// Thin lens with abb_chromatic > 0 (src/lentil_filter.cpp:393-406): every attempt that survives the optical
// vignetting test draws its colour channel from xor128 (src/global.h:22-27) -- ONE generator state for the whole
// process upstream, so the channel of an attempt depends on how many such attempts every earlier visit made, and
// those numbers depend on the channels drawn (a channel moves the focus plane, hence whether the draw lands inside the
// frame and counts).  The order is defined here as the single-threaded one: visits in iterator order, the generator
// state handed in and out (lentil_hip_set_xor128_state / _get_xor128_state).
//   tl_chroma_solve_kernel : everything that does not depend on the channel drawn, for every attempt an item could
//                            make (n < 5 * samples): vignetted?, and the outcome for each of the three channels.
//   tl_chroma_walk_kernel  : ONE block walks the items in visit order, 256 attempts per step: prefix count of the
//                            survivors = each attempt's index into the generator's output, channel, outcome, ordered
//                            acceptance, splat.  Sequential in the items by construction (~15 us per item).
// ---------------------------------------------------------------------------------------
struct TlChromaArgs {
  lentil_params P;
  DevBokeh bokeh;
  VisitsDev V;
  FrameDev F;
  const uint2 *work;             // items (visit, samples), sorted by visit
  uint32_t n_items;
  const uint64_t *att_off;       // [n_items + 1]: first attempt slot of every item (an item has 5 * samples slots)
  uint32_t *res;                 // [slots][3]: outcome per channel; all three kCodeFail = vignetted
  const uint2 *tasks;            // (item, first attempt) per block of tl_chroma_solve_kernel
  uint32_t n_tasks;
  uint32_t *xor_state;           // [4] x, y, z, w: in and out
  DevCounters *ctr;
  lentil_draw_record *log;
  uint64_t log_cap;
  unsigned long long *log_count;
};

__global__ __launch_bounds__(256) void tl_chroma_solve_kernel(TlChromaArgs a) {
  __shared__ float s_cdfRow[kMaxBokehRows];
  const bool row_in_lds = a.P.bokeh_enable_image && a.bokeh.y <= kMaxBokehRows;
  if (row_in_lds)
    for (int i = threadIdx.x; i < a.bokeh.y; i += blockDim.x) s_cdfRow[i] = a.bokeh.cdfRow[i];
  __syncthreads();
  const float *cdfRow = row_in_lds ? s_cdfRow : a.bokeh.cdfRow;
  for (uint32_t q = blockIdx.x; q < a.n_tasks; q += gridDim.x) {
    const uint2 t = a.tasks[q];
    const ItemVisit h = load_work_visit(a.P, a.V, a.work[t.x], 0.0);
    const uint32_t n = t.y + threadIdx.x;
    if (n >= h.samples * 5u) continue;
    uint32_t code[3] = {kCodeFail, kCodeFail, kCodeFail};
    TlRay ray;
    if (thinlens_ray(a.P, a.bokeh, cdfRow, h.I.cs, h.px, h.py, n, ray)) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        uint32_t pix;
        code[c] = thinlens_project(a.P, ray, thinlens_chroma_image_dist(a.P, ray, c - 1), pix) ? pix : kCodeOut;
      }
    }
    uint32_t *dst = a.res + (a.att_off[t.x] + n) * 3ull;
    dst[0] = code[0]; dst[1] = code[1]; dst[2] = code[2];
  }
}

LD_DEV uint32_t xor128_next(uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &w) {      // src/global.h:22-27
  const uint32_t t = x ^ (x << 11);
  x = y; y = z; z = w;
  return w = (w ^ (w >> 19) ^ t ^ (t >> 8));
}

__global__ __launch_bounds__(256) void tl_chroma_walk_kernel(TlChromaArgs a) {
  __shared__ uint32_t s_x[256];            // the step's generator outputs, in order
  __shared__ uint32_t s_state[256][4];     // generator state after each of them
  __shared__ uint32_t s_cnt[4], s_cnt2[4], s_pix[4][64], s_same[4][64], s_ch[4][64], s_key[4][64];
  __shared__ uint32_t s_cur[4];            // current generator state
  __shared__ float s_val[4 * LENTIL_MAX_AOVS + 1];
  __shared__ uint32_t s_off[4 * LENTIL_MAX_AOVS + 1];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  if (threadIdx.x < 4) s_cur[threadIdx.x] = a.xor_state[threadIdx.x];
  unsigned long long tot_attempted = 0, tot_accepted = 0;
  uint32_t rmin = 0x7FFFFFFFu, rmax_p1 = 0u;
  __syncthreads();
  for (uint32_t item = 0; item < a.n_items; ++item) {
    const ItemVisit h = load_work_visit(a.P, a.V, a.work[item], 0.0);
    const uint32_t S = h.samples, max_total = S * 5u;
    const float ae = h.I.add_energy, w = h.w;
    const unsigned long long zk = (a.F.zkey || a.F.zkey_dbg) ? closest_key_of(a.ctr, h.I.depth, visit_gid(a.V, h.visit)) : 0ull;
    const uint32_t *res = a.res + a.att_off[item] * 3ull;
    // what an accepted draw adds (gaussian AOVs: (value + add_energy) * w per component, then the weight itself)
    uint32_t U = 1;
    for (uint32_t k = 0; k < a.F.n_aovs; ++k) if (!(a.F.closest_mask & (1u << k))) U += 4;
    __syncthreads();
    if (threadIdx.x < a.F.n_aovs * 4u) {
      const uint32_t k = threadIdx.x >> 2, c = threadIdx.x & 3u;
      if (!(a.F.closest_mask & (1u << k))) {
        uint32_t slot = 0;
        for (uint32_t j = 0; j < k; ++j) if (!(a.F.closest_mask & (1u << j))) slot += 4;
        const float4 v = k == 0 ? h.rgba : a.V.extra[k - 1][h.visit];
        const float vc = c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
        s_val[slot + c] = (vc + ae) * w;
        s_off[slot + c] = 4u * k + c;
      }
    }
    if (threadIdx.x == 0) { s_val[U - 1] = w; s_off[U - 1] = 4u * a.F.n_aovs; }
    __syncthreads();
    uint32_t n = 0, acc = 0, last_n = 0;
    bool done = false;
    while (!done && acc < S && n < max_total) {
      const uint32_t my_n = n + threadIdx.x;
      const bool valid = my_n < max_total;
      uint32_t code[3] = {kCodeFail, kCodeFail, kCodeFail};
      if (valid) { code[0] = res[my_n * 3ull]; code[1] = res[my_n * 3ull + 1]; code[2] = res[my_n * 3ull + 2]; }
      const bool survives = valid && !(code[0] == kCodeFail && code[1] == kCodeFail && code[2] == kCodeFail);
      // index of this attempt's generator output within the step: survivors before it
      const unsigned long long vm = __ballot(survives);
      if (lane == 0) s_cnt[wave] = (uint32_t)__builtin_popcountll(vm);
      __syncthreads();
      uint32_t vbefore = 0, vtotal = 0;
      for (uint32_t k = 0; k < 4; ++k) { if (k < wave) vbefore += s_cnt[k]; vtotal += s_cnt[k]; }
      const uint32_t xi = vbefore + (uint32_t)__builtin_popcountll(vm & lt_mask);
      if (threadIdx.x == 0) {
        uint32_t x = s_cur[0], y = s_cur[1], z = s_cur[2], ww = s_cur[3];
        for (uint32_t i = 0; i < vtotal; ++i) {
          s_x[i] = xor128_next(x, y, z, ww);
          s_state[i][0] = x; s_state[i][1] = y; s_state[i][2] = z; s_state[i][3] = ww;
        }
      }
      __syncthreads();
      int channel = 0;
      uint32_t my_code = kCodeFail;
      if (survives) {
        channel = (int)floor(((double)s_x[xi] / 4294967296.0) * 3.0) - 1;      // :397
        my_code = code[channel + 1];
      }
      const bool succ = survives && my_code < kCodeOut;
      // the attempt at which the count reaches `samples` ends the loop: later attempts are not made (and draw nothing)
      const unsigned long long sm = __ballot(succ);
      if (lane == 0) s_cnt2[wave] = (uint32_t)__builtin_popcountll(sm);
      __syncthreads();
      uint32_t sbefore = 0, stotal = 0;
      for (uint32_t k = 0; k < 4; ++k) { if (k < wave) sbefore += s_cnt2[k]; stotal += s_cnt2[k]; }
      const uint32_t rank = acc + sbefore + (uint32_t)__builtin_popcountll(sm & lt_mask);      // successes before this attempt
      const bool take = succ && rank < S;
      const bool is_last = take && rank + 1u == S;        // the S-th success: the last attempt the reference makes
      const unsigned long long lm = __ballot(is_last);
      __syncthreads();
      if (lane == 0) s_cnt[wave] = lm ? wave * 64u + (uint32_t)__builtin_ctzll(lm) : 0xFFFFFFFFu;
      __syncthreads();
      uint32_t end_i = 0xFFFFFFFFu;                        // position of that attempt within the step, if it is here
      for (uint32_t k = 0; k < 4; ++k) if (s_cnt[k] != 0xFFFFFFFFu) end_i = s_cnt[k];
      const uint32_t step_n = (max_total - n) < 256u ? (max_total - n) : 256u;
      const uint32_t executed = end_i != 0xFFFFFFFFu ? end_i + 1u : step_n;      // attempts of this step that are made
      // generator outputs consumed: survivors among the executed attempts
      const unsigned long long cm = __ballot(survives && threadIdx.x < executed);
      __syncthreads();
      if (lane == 0) s_cnt2[wave] = (uint32_t)__builtin_popcountll(cm);
      __syncthreads();
      const uint32_t consumed = s_cnt2[0] + s_cnt2[1] + s_cnt2[2] + s_cnt2[3];
      // splat the accepted draws: channel c feeds colour component c only, three-fold; alpha and weight as they are
      const unsigned long long tm = __ballot(take);
      const uint32_t T = (uint32_t)__builtin_popcountll(tm);
      if (take) {
        const uint32_t pix = my_code;
        const uint32_t row = pix / a.P.xres;
        rmin = row < rmin ? row : rmin;
        rmax_p1 = row + 1u > rmax_p1 ? row + 1u : rmax_p1;
        const uint32_t slot = (uint32_t)__builtin_popcountll(tm & lt_mask);
        s_pix[wave][slot] = pix;
        s_ch[wave][slot] = (uint32_t)(channel + 1);
        s_key[wave][slot] = (pix << 2) | (uint32_t)(channel + 1);       // (frames of up to 2^30 pixels)
        if (a.F.touched) { a.F.touched[pix >> 6] = 1; a.F.touched_px[pix] = 1; }
        if (a.F.zkey) atomicMin(a.F.zkey + pix, zk);
        if (a.F.zkey_dbg) atomicMin(a.F.zkey_dbg + pix, zk);
        if (a.log_cap) {
          const unsigned long long li = wave_log_slots(a.log_count, tm, lane);
          if (li < a.log_cap) { a.log[li].visit = h.visit; a.log[li].attempt = my_n | ((uint32_t)(channel + 1) << 30); a.log[li].pixel = pix; }
        }
      }
      __syncthreads();
      // draws of the step with the same pixel and channel: one atomic of count x value (accept_item does the same)
      count_same_pixel(s_key[wave], s_same[wave], T, lane);
      __syncthreads();
      for (uint32_t q = lane; q < T * U; q += 64u) {
        const uint32_t d = q / U, ch = q - d * U;
        const uint32_t same = s_same[wave][d];
        if (!same) continue;
        float val = s_val[ch];
        if (ch != U - 1u) {
          const uint32_t comp = s_off[ch] & 3u;
          if (comp < 3u) {
            if (comp != s_ch[wave][d]) continue;          // rgb_weight is 0 there
            val = val * 3.0f;
          }
        }
        atomicAdd(a.F.acc + (size_t)s_pix[wave][d] * a.F.stride + s_off[ch], (float)same * val);
      }
      uint32_t taken = stotal < S - acc ? stotal : S - acc;
      acc += taken;
      n += executed;
      if (end_i != 0xFFFFFFFFu) { done = true; last_n = n; }
      __syncthreads();
      if (threadIdx.x == 0 && consumed) {
        s_cur[0] = s_state[consumed - 1][0]; s_cur[1] = s_state[consumed - 1][1];
        s_cur[2] = s_state[consumed - 1][2]; s_cur[3] = s_state[consumed - 1][3];
      }
      __syncthreads();
    }
    (void)last_n;
    tot_attempted += n;            // total_samples_taken when the loop ends
    tot_accepted += acc;
  }
  if (threadIdx.x == 0) {
    for (int i = 0; i < 4; ++i) a.xor_state[i] = s_cur[i];
    if (tot_attempted) atomicAdd(&a.ctr->attempted, tot_attempted);
    if (tot_accepted) atomicAdd(&a.ctr->accepted, tot_accepted);
  }
  flush_row_range(a.ctr, rmin, rmax_p1);
}

// closest-filter AOVs: copy the winning visit's value into AOVData::buffer (src/lentil.h:835)
__global__ __launch_bounds__(256) void closest_gather_kernel(FrameDev F, VisitsDev V) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < F.np; p += stride) {
    const unsigned long long key = F.zkey[p];
    if (key == ~0ull) continue;
    uint32_t visit;
    if (!visit_from_gid(V, 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull), visit)) continue;   // another GPU's visit won
    for (uint32_t k = 1; k < F.n_aovs; ++k)
      if ((F.closest_mask & (1u << k)) && !(F.debug_mask & (1u << k)))
        *F.aov(p, k) = V.extra[k - 1][visit];
  }
}

// lentil_debug (src/lentil_filter.cpp:209-211, src/lentil.h:838-845): the value is the draw count of the visit
// (samples * redistribute, an int assigned to an AtRGBA: all four components), written only by redistributed
// draws, closest by its own z-buffer.  The count is recomputed from the winner's columns.
__global__ __launch_bounds__(256) void debug_gather_kernel(FrameDev F, VisitsDev V, lentil_params P, double lens_length) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < F.np; p += stride) {
    const unsigned long long key = F.zkey_dbg[p];
    if (key == ~0ull) continue;
    uint32_t v;
    if (!visit_from_gid(V, 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull), v)) continue;
    const float invd = V.inv_density ? V.inv_density[v] : P.inverse_sample_density;
    const VisitInfo I = visit_prologue(P, lens_length, V.rgba[v], V.pos_z[v], V.raydir_time[v], V.volume_ignore[v],
                                       V.transmission[v], invd, V.cam);
    const float s = (float)I.samples;
    for (uint32_t k = 1; k < F.n_aovs; ++k)
      if (F.debug_mask & (1u << k)) *F.aov(p, k) = make_float4(s, s, s, s);
  }
}

// Multi-GPU tiles: rows of another GPU's accumulators (same record layout) are merged into
// this GPU's: gaussian slots and the weight add up; closest-filtered slots follow the smaller winner key.
// The keys themselves are merged by a second launch (every float of a pixel reads both keys first).
// (lentil_debug slots follow their own key plane, zkey_dbg / src_keys_dbg.)
__global__ __launch_bounds__(256) void merge_rows_kernel(FrameDev F, uint64_t p_begin, uint64_t n_pix,
                                                         const float *src, const unsigned long long *src_keys,
                                                         const unsigned long long *src_keys_dbg) {
  const uint64_t total = n_pix * F.stride;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t used = 4u * F.n_aovs + 1u;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint64_t i = t / F.stride;
    const uint32_t j = (uint32_t)(t - i * F.stride);
    if (j >= used) continue;
    float *dst = F.acc + (p_begin + i) * F.stride + j;
    const uint32_t aov = j >> 2;
    if (j < 4u * F.n_aovs && (F.closest_mask & (1u << aov))) {
      const bool dbg = (F.debug_mask >> aov) & 1u;
      const unsigned long long *sk = dbg ? src_keys_dbg : src_keys;
      const unsigned long long *mine = dbg ? F.zkey_dbg : F.zkey;
      if (sk && sk[i] < mine[p_begin + i]) *dst = src[t];
    } else {
      *dst += src[t];
    }
  }
}

// Rows for another GPU without the padding of the pixel records: 4 n_aovs + 1 floats per pixel instead of `stride`
// (5 instead of 8 for a beauty-only frame) -- the exchange over xGMI is what a tiled step adds to a pass.
__global__ __launch_bounds__(256) void pack_rows_kernel(FrameDev F, uint64_t p_begin, uint64_t n_pix, float *dst) {
  const uint32_t used = 4u * F.n_aovs + 1u;
  const uint64_t total = n_pix * used;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint64_t i = t / used;
    const uint32_t j = (uint32_t)(t - i * used);
    dst[t] = F.acc[(p_begin + i) * F.stride + j];
  }
}

// ... and without the pixels nothing was added to: in the scan-dominated regime a band's draws reach ~100 rows into
// its neighbours but fill only ~2 % of them.  Every pixel of the rows with a non-zero weight (or, in frames with
// closest-filtered AOVs, a winner key) becomes one entry: its frame-wide index, its 4 n_aovs + 1 floats, its key.
// Entries beyond `cap` are counted but not written (the caller then sends the rows whole).
__global__ __launch_bounds__(256) void compact_rows_kernel(FrameDev F, uint64_t p_begin, uint64_t n_pix, uint32_t *idx,
                                                           float *vals, unsigned long long *keys, unsigned long long *keys_dbg,
                                                           uint32_t cap, unsigned int *count) {
  const uint32_t used = 4u * F.n_aovs + 1u;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t n_round = (n_pix + 63ull) & ~63ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    const uint64_t p = p_begin + i;
    bool live = false;
    if (i < n_pix) live = F.acc[p * F.stride + 4u * F.n_aovs] != 0.0f || (F.zkey && F.zkey[p] != ~0ull) ||
                          (F.zkey_dbg && F.zkey_dbg[p] != ~0ull);
    const unsigned long long m = __ballot(live);
    if (m == 0ull) continue;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned int)__builtin_popcountll(m));
    base = __shfl(base, 0);
    const uint32_t slot = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (live && slot < cap) {
      idx[slot] = (uint32_t)p;
      for (uint32_t j = 0; j < used; ++j) vals[(uint64_t)slot * used + j] = F.acc[p * F.stride + j];
      if (keys) keys[slot] = F.zkey[p];
      if (keys_dbg) keys_dbg[slot] = F.zkey_dbg[p];
    }
  }
}

// the receiving side: entries of ONE sender (every pixel at most once), launches of different senders are ordered
// (n_dev: the entry count is the message's own header word, read here -- the host never learns it before the merge; a
// count above `cap` means the sender's entries did not fit and the rows follow whole, lentil_comm.h)
__global__ __launch_bounds__(256) void merge_sparse_kernel(FrameDev F, uint32_t n, const uint32_t *idx, const float *vals,
                                                           const unsigned long long *keys, const unsigned long long *keys_dbg,
                                                           const uint32_t *n_dev, uint32_t cap) {
  if (n_dev) { const uint32_t c = *n_dev; n = c <= cap ? c : 0u; }
  const uint32_t used = 4u * F.n_aovs + 1u;
  const uint64_t total = (uint64_t)n * used;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint64_t e = t / used;
    const uint32_t j = (uint32_t)(t - e * used);
    const uint64_t p = idx[e];
    if (p >= F.np) continue;
    float *dst = F.acc + p * F.stride + j;
    const uint32_t aov = j >> 2;
    if (j < 4u * F.n_aovs && (F.closest_mask & (1u << aov))) {
      const bool dbg = (F.debug_mask >> aov) & 1u;
      const unsigned long long *sk = dbg ? keys_dbg : keys;
      const unsigned long long *mine = dbg ? F.zkey_dbg : F.zkey;
      if (sk && sk[e] < mine[p]) *dst = vals[t];
    } else {
      *dst += vals[t];
    }
  }
}

__global__ __launch_bounds__(256) void merge_sparse_keys_kernel(unsigned long long *mine, uint64_t np, uint32_t n,
                                                                const uint32_t *idx, const unsigned long long *keys,
                                                                const uint32_t *n_dev, uint32_t cap) {
  if (n_dev) { const uint32_t c = *n_dev; n = c <= cap ? c : 0u; }
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
    const uint64_t p = idx[e];
    if (p < np && keys[e] < mine[p]) mine[p] = keys[e];
  }
}

// merge_rows_kernel for rows that arrived packed
__global__ __launch_bounds__(256) void merge_packed_rows_kernel(FrameDev F, uint64_t p_begin, uint64_t n_pix,
                                                                const float *src, const unsigned long long *src_keys,
                                                                const unsigned long long *src_keys_dbg) {
  const uint32_t used = 4u * F.n_aovs + 1u;
  const uint64_t total = n_pix * used;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint64_t i = t / used;
    const uint32_t j = (uint32_t)(t - i * used);
    float *dst = F.acc + (p_begin + i) * F.stride + j;
    const uint32_t aov = j >> 2;
    if (j < 4u * F.n_aovs && (F.closest_mask & (1u << aov))) {
      const bool dbg = (F.debug_mask >> aov) & 1u;
      const unsigned long long *sk = dbg ? src_keys_dbg : src_keys;
      const unsigned long long *mine = dbg ? F.zkey_dbg : F.zkey;
      if (sk && sk[i] < mine[p_begin + i]) *dst = src[t];
    } else {
      *dst += src[t];
    }
  }
}

__global__ __launch_bounds__(256) void merge_keys_kernel(unsigned long long *mine, uint64_t p_begin, uint64_t n_pix,
                                                         const unsigned long long *src_keys) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    const unsigned long long k = src_keys[i];
    if (k < mine[p_begin + i]) mine[p_begin + i] = k;
  }
}

// clear_frame while FrameDev::touched is trusted: wipe the groups of 64 records that received splats, nothing else
__global__ __launch_bounds__(256) void clear_touched_kernel(FrameDev F, uint64_t n_groups) {
  LENTIL_TL_SPAN(SPAN_CLEAR);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t q = F.stride >> 2;
  float4 *acc4 = reinterpret_cast<float4 *>(F.acc);
  const uint64_t wave_global = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint64_t wave_stride = (uint64_t)gridDim.x * (blockDim.x >> 6);
  for (uint64_t g0 = wave_global * 64u; g0 < n_groups; g0 += wave_stride * 64u) {
    // 64 flags per wave step; then every flagged group by the whole wave
    const uint64_t g = g0 + lane;
    const bool set = g < n_groups && F.touched[g] != 0;
    unsigned long long m = __ballot(set);
    if (set) F.touched[g] = 0;
    while (m) {
      const uint32_t b = (uint32_t)__builtin_ctzll(m);
      m &= m - 1ull;
      const uint64_t p0 = (g0 + b) * 64u;
      const uint64_t n_pix = (F.np - p0) < 64ull ? (F.np - p0) : 64ull;
      // (a lane per pixel: the records a draw reached -- their flags -- and nothing else of the group)
      if (lane < n_pix && F.touched_px[p0 + lane] != 0) {
        F.touched_px[p0 + lane] = 0;
        for (uint32_t j = 0; j < q; ++j) acc4[(p0 + lane) * q + j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}

// FrameDev::dir -> FrameDev::acc for a range of pixels (and zero there): for everything but the resolve that looks at
// the accumulators (downloads, the exchange between GPUs, a second pass into the same frame)
__global__ __launch_bounds__(256) void fold_direct_kernel(FrameDev F, float *dir, uint64_t p_begin, uint64_t p_end) {
  const uint32_t q = F.stride >> 2;
  float4 *acc4 = reinterpret_cast<float4 *>(F.acc);
  float4 *dir4 = reinterpret_cast<float4 *>(dir);
  const uint64_t i0 = p_begin * q, i1 = p_end * q;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += stride) {
    const float4 d = dir4[i];
    if (d.x != 0.0f || d.y != 0.0f || d.z != 0.0f || d.w != 0.0f) {
      float4 a = acc4[i];
      a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
      acc4[i] = a;
      dir4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// K7 -- driver_process_bucket's normalisation, src/lentil_imager.cpp:169-186.  Reads the pixel records,
// writes one planar RGBA image per AOV (what the imager copies into Arnold's buckets).
// One 64-pixel group: the records, read with fully coalesced float4 loads into the wave's LDS tile; lane p then
// normalises pixel p and every AOV plane receives 64 adjacent RGBA values (1 KiB per store instruction).
LD_DEV void resolve_group(const FrameDev &F, float *resolved, float4 *tile, uint64_t p0, uint32_t n_pix, uint32_t lane) {
  const uint32_t q = F.stride >> 2;                       // float4 per record (stride is a multiple of 8 floats)
  const float4 *acc4 = reinterpret_cast<const float4 *>(F.acc);
  const float4 *dir4 = reinterpret_cast<const float4 *>(F.dir);
  const uint32_t n4 = n_pix * q;
  if (dir4 && F.touched && !(F.touched[p0 >> 6] | F.touched[(p0 + n_pix - 1u) >> 6])) {
    // nothing was splatted into these records: they are all zero, the direct sums are the whole story
    for (uint32_t i = lane; i < n4; i += 64u) tile[i] = dir4[p0 * q + i];
  } else if (dir4) {
    // what the pixels' own visits added (scan_dma_kernel) + what was splatted: a pixel that received no draw has a
    // zero record in `acc`, and 0 + x is x bit for bit (the scan's sums are never -0)
    for (uint32_t i = lane; i < n4; i += 64u) {
      const float4 a = acc4[p0 * q + i], d = dir4[p0 * q + i];
      tile[i] = make_float4(a.x + d.x, a.y + d.y, a.z + d.z, a.w + d.w);
    }
  } else {
    for (uint32_t i = lane; i < n4; i += 64u) tile[i] = acc4[p0 * q + i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  if (lane < n_pix) {
    const float *rec = reinterpret_cast<const float *>(tile + (size_t)lane * q);
    const float wt = rec[4u * F.n_aovs];
    const float inv = wt != 0.0f ? 1.0f / wt : 1.0f;      // AtRGBA /= float multiplies by 1.0f/f
    for (uint32_t a = 0; a < F.n_aovs; ++a) {
      float4 c = tile[(size_t)lane * q + a];
      if (F.closest_mask & (1u << a)) {
        c.w = 1.0f;                                      // closest: (r, g, b, 1), src/lentil_imager.cpp:181-186
      } else if (wt != 0.0f) {
        c.x *= inv; c.y *= inv; c.z *= inv; c.w *= inv;
      }
      reinterpret_cast<float4 *>(resolved)[(uint64_t)a * F.np + p0 + lane] = c;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
}

// K7 -- driver_process_bucket's normalisation, src/lentil_imager.cpp:169-186.  Reads the pixel records,
// writes one planar RGBA image per AOV (what the imager copies into Arnold's buckets).  A wave takes 64 consecutive
// pixels at a time.
__global__ __launch_bounds__(256) void resolve_kernel(FrameDev F, float *resolved, uint64_t p_begin, uint64_t p_end) {
  LENTIL_TL_SPAN(SPAN_RESOLVE);
  extern __shared__ float4 s_rec[];                       // [waves per block][64 * stride / 4]
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float4 *tile = s_rec + (size_t)wave * 64u * (F.stride >> 2);
  const uint64_t n_tiles = (p_end - p_begin + 63ull) / 64ull;
  const uint64_t wave_global = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  const uint64_t wave_stride = (uint64_t)gridDim.x * (blockDim.x >> 6);
  for (uint64_t t = wave_global; t < n_tiles; t += wave_stride) {
    const uint64_t p0 = p_begin + t * 64ull;
    resolve_group(F, resolved, tile, p0, (uint32_t)((p_end - p0) < 64ull ? (p_end - p0) : 64ull), lane);
  }
}

// The second half of a resolve that ran early -- behind the pass's first accept, while its later rounds were still
// adding draws (see lentil_hip_redistribute): just the 64-pixel groups a later round's draw was splatted into (flag 2;
// the headline frame's first round touches about half of the groups, its second a few hundred).  A wave looks at 64 groups'
// flags with one load and walks the set bits.  (min_flag 1: every group that received a draw -- the pass whose whole-frame
// resolve ran behind the scan, before any accept, see redistribute_streamed.)
// (Round 6: within a flagged group only the pixels a draw reached are written again -- resolve_touched_group.)
// One flagged 64-pixel group of a frame whose untouched state is resolved already (the whole-frame resolve ran earlier in the
// pass, from the pixels' own sums alone or with an earlier round's draws): the group's 64 per-pixel flags are read (one coalesced
// 64-byte access), and only the pixels a draw reached -- FrameDev::touched_px at least `min_flag`: 1 for everything the pass has
// added, 2 for what the accepts behind the first added -- read their splat records and their own sums, normalise and store.  A 4K
// headline frame's draws reach about half of the groups and an eighth of the pixels; a petzval frame's every group and an eighth
// of the pixels (reading whole groups to find them was the frame's accumulator block once over: 386 us for nine AOVs, now 185).
LD_DEV void resolve_touched_group(const FrameDev &F, float *resolved, uint64_t p0, uint32_t n_pix, uint32_t lane, uint32_t min_flag) {
  const uint32_t q = F.stride >> 2;
  const float4 *acc4 = reinterpret_cast<const float4 *>(F.acc);
  const float4 *dir4 = reinterpret_cast<const float4 *>(F.dir);
  // (a lane per pixel: its flag -- one coalesced 64-byte read per group --, and only a flagged pixel reads its records)
  if (lane < n_pix && F.touched_px[p0 + lane] >= min_flag) {
    const uint64_t p = p0 + lane;
    float wt = reinterpret_cast<const float *>(acc4 + p * q)[4u * F.n_aovs];
    if (dir4) wt += reinterpret_cast<const float *>(dir4 + p * q)[4u * F.n_aovs];
    const float inv = wt != 0.0f ? 1.0f / wt : 1.0f;      // AtRGBA /= float multiplies by 1.0f/f
    for (uint32_t a = 0; a < F.n_aovs; ++a) {
      float4 c = acc4[p * q + a];
      if (dir4) { const float4 d = dir4[p * q + a]; c.x += d.x; c.y += d.y; c.z += d.z; c.w += d.w; }
      if (F.closest_mask & (1u << a)) {
        c.w = 1.0f;                                      // closest: (r, g, b, 1), src/lentil_imager.cpp:181-186
      } else if (wt != 0.0f) {
        c.x *= inv; c.y *= inv; c.z *= inv; c.w *= inv;
      }
      reinterpret_cast<float4 *>(resolved)[(uint64_t)a * F.np + p] = c;
    }
  }
}

__global__ __launch_bounds__(256) void resolve_touched_kernel(FrameDev F, float *resolved, uint32_t min_flag) {
  LENTIL_TL_SPAN(SPAN_RESOLVE_TOUCHED);
  extern __shared__ float4 s_rec[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float4 *tile = s_rec + (size_t)wave * 64u * (F.stride >> 2);
  // (per pixel only where the record's weight tells: gaussian AOVs alone -- a closest-filtered AOV's value is not a sum)
  const bool per_pixel = F.closest_mask == 0u && F.dir != nullptr;
  const uint64_t n_groups = (F.np + 63ull) / 64ull;
  const uint64_t n_chunks = (n_groups + 63ull) / 64ull;
  // (Round 6: a chunk of 64 flags per BLOCK, its flagged groups dealt round the block's four waves -- a wave used to walk all of
  // a chunk's flagged groups one after the other, a load round trip each: 32 of them for a headline frame, 90 us of latency
  // on a quarter of the waves the launch has.)
  const uint32_t wpb = blockDim.x >> 6;
  for (uint64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    const uint64_t g = ch * 64ull + lane;
    unsigned long long mask = __ballot(g < n_groups && F.touched[g] >= min_flag);
    uint32_t k = 0;
    while (mask) {
      const uint32_t b = (uint32_t)__builtin_ctzll(mask);
      mask &= mask - 1ull;
      if ((k++ % wpb) != wave) continue;
      const uint64_t p0 = (ch * 64ull + b) * 64ull;
      const uint32_t n_pix = (uint32_t)((F.np - p0) < 64ull ? (F.np - p0) : 64ull);
      if (per_pixel) resolve_touched_group(F, resolved, p0, n_pix, lane, min_flag);
      else resolve_group(F, resolved, tile, p0, n_pix, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Focus search, Camera::logarithmic_focus_search (src/lentil.h:1445-1460) over logarithmic_values()
// (src/lens.h:395-407): 20 001 candidate sensor shifts, for each the distance at which the ray through a quarter of
// the aperture housing radius crosses the optical axis (camera_get_y0_intersection_distance), the winner the
// candidate whose miss = focal_distance - distance is the smallest positive one -- the first such in candidate order,
// as the reference's sequential `new_distance < closest_distance` keeps it.  One candidate per lane; the candidates
// (an fp64 running sum and std::pow on the host side of the reference) come from the host.
// ---------------------------------------------------------------------------------------
struct FocusArgs {
  const DevLens *lens;
  const DevTerm *terms;
  double lambda, housing_radius, focal_distance;
  uint32_t n;
  const double *shift;     // [n] candidates
  double *miss;            // [n] focal_distance - intersection distance
  double *sensor, *out;    // optional [n][5] each: what lens_pt_sample_aperture / lens_evaluate left (parity tests)
  double *best;            // [2]: winning shift, its miss (focus_argmin_kernel)
};

__global__ __launch_bounds__(256) void focus_miss_kernel(FocusArgs f) {
  __shared__ DevTerm s_terms[kMaxTerms];
  __shared__ DevLens s_k;
  const uint32_t nt = f.lens->n_terms;
  for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) s_terms[i] = f.terms[i];
  if (threadIdx.x == 0) {
    s_k = *f.lens;
    s_k.lambda_pow[0] = 1.0; s_k.lambda_pow[1] = f.lambda;
    for (uint32_t e = 2; e <= kMaxExp; ++e) s_k.lambda_pow[e] = ipow_u(f.lambda, e);     // lens_ipow, like the host
  }
  __syncthreads();
  const LdsLens L{s_terms, &s_k};
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ii = i < f.n ? i : f.n - 1u;      // whole waves stay converged: table reads go through readfirstlane
  double sensor[4], out[4], T;
  const double dist = camera_get_y0_intersection_distance(L, f.shift[ii], f.housing_radius, sensor, out, T);
  if (i < f.n) {
    f.miss[i] = f.focal_distance - dist;
    if (f.sensor) { for (int c = 0; c < 4; ++c) f.sensor[(size_t)i * 5 + c] = sensor[c]; f.sensor[(size_t)i * 5 + 4] = f.lambda; }
    if (f.out) { for (int c = 0; c < 4; ++c) f.out[(size_t)i * 5 + c] = out[c]; f.out[(size_t)i * 5 + 4] = T; }
  }
}

// one block: smallest positive miss, ties to the earlier candidate; no positive miss: shift 0 (the reference's initial value)
__global__ __launch_bounds__(1024) void focus_argmin_kernel(FocusArgs f) {
  __shared__ double s_val[1024];
  __shared__ uint32_t s_idx[1024];
  double bv = 999999999.0;           // closest_distance starts here: a miss must be below it to count
  uint32_t bi = 0xFFFFFFFFu;
  for (uint32_t i = threadIdx.x; i < f.n; i += blockDim.x) {
    const double m = f.miss[i];
    if (m > 0.0 && m < bv) { bv = m; bi = i; }      // ascending i per thread: strict < keeps the earlier one
  }
  s_val[threadIdx.x] = bv; s_idx[threadIdx.x] = bi;
  __syncthreads();
  for (uint32_t off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const double ov = s_val[threadIdx.x + off];
      const uint32_t oi = s_idx[threadIdx.x + off];
      if (oi != 0xFFFFFFFFu && (ov < s_val[threadIdx.x] || (ov == s_val[threadIdx.x] && oi < s_idx[threadIdx.x]))) {
        s_val[threadIdx.x] = ov; s_idx[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t w = s_idx[0];
    f.best[0] = w != 0xFFFFFFFFu ? f.shift[w] : 0.0;
    f.best[1] = w != 0xFFFFFFFFu ? s_val[0] : 999999999.0;
  }
}

// ---------------------------------------------------------------------------------------
// single-function test kernels
// ---------------------------------------------------------------------------------------
struct TestArgs {
  lentil_params P;
  const DevLens *lens;
  const DevTerm *terms;
  DevBokeh bokeh;
  uint64_t n;
  const double *in0;   // scene / target
  const double *in1;   // ap
  const int32_t *i0, *i1, *i2;
  const uint32_t *u0, *u1;
  double lambda;
  double *o0, *o1, *o2;
  int32_t *oi;
};

template <int WHAT>
__global__ __launch_bounds__(256) void test_kernel(TestArgs t) {
  __shared__ DevTerm s_terms[kMaxTerms];
  __shared__ DevLens s_k;
  if (t.lens) {
    const uint32_t nt = t.lens->n_terms;
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) s_terms[i] = t.terms[i];
    if (threadIdx.x == 0) {
      s_k = *t.lens;
      if (WHAT == 0) {   // explicit lambda: recompute lens_ipow(lambda, e) exactly like the host does
        s_k.lambda_pow[0] = 1.0; s_k.lambda_pow[1] = t.lambda;
        for (uint32_t e = 2; e <= kMaxExp; ++e) s_k.lambda_pow[e] = ipow_u(t.lambda, e);
      }
    }
  }
  __syncthreads();
  const LdsLens L{s_terms, &s_k};
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // keep whole waves converged: table reads go through readfirstlane
  const uint64_t ii = i < t.n ? i : t.n - 1;
  if (WHAT == 0) {
    double sensor[4], out[5] = {0, 0, 0, 0, t.lambda};
    const double scene[3] = {t.in0[ii * 3], t.in0[ii * 3 + 1], t.in0[ii * 3 + 2]};
    int iters = 0;
    const double T = lt_sample_aperture(L, scene, t.in1[ii * 2], t.in1[ii * 2 + 1], sensor, out, &iters);
    if (i < t.n) {
      for (int c = 0; c < 4; ++c) t.o0[i * 5 + c] = sensor[c];
      t.o0[i * 5 + 4] = t.lambda;
      for (int c = 0; c < 5; ++c) t.o1[i * 5 + c] = out[c];
      t.o2[i] = T;
      if (t.oi) t.oi[i] = iters;
    }
  } else if (WHAT == 1) {
    const double target[3] = {t.in0[ii * 3], t.in0[ii * 3 + 1], t.in0[ii * 3 + 2]};
    double sx = 0, sy = 0;
    const bool ok = trace_ray_bw_po(t.P, L, t.bokeh, t.bokeh.cdfRow, target, t.i0[ii], t.i1[ii], t.i2[ii], sx, sy);
    if (i < t.n) { t.o0[i * 2] = sx; t.o0[i * 2 + 1] = sy; t.oi[i] = ok ? 1 : 0; }
  } else {
    double ax, ay;
    po_aperture_sample(t.P, t.bokeh, t.bokeh.cdfRow, t.u0[ii], t.u1[ii], ax, ay);
    if (i < t.n) { t.o0[i * 2] = ax; t.o0[i * 2 + 1] = ay; }
  }
}

