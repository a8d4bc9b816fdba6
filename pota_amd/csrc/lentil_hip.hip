// lentil_hip.hip -- C-ABI of liblentil_hip.so (gfx950 only): host side.  Kernels: lentil_kernels.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "lentil_kernels.h"
#include "lentil_closest_replay.h"
#include "lentil_lens_jit.h"
#include "generated/embedded_sources.inc"

#define LENTIL_API extern "C" __attribute__((visibility("default")))

// =======================================================================================
// host side
// =======================================================================================
// LENTIL_HOST_TRACE=n: host time stamps of the first n passes' calls, printed by lentil_hip_sync (development aid: where a
// step's time goes between the device's last kernel and its next first one)
struct HostTrace {
  std::vector<std::pair<const char *, int64_t>> marks;
  int passes_left = -1;
};
static int host_trace_passes() { static const int n = getenv("LENTIL_HOST_TRACE") ? atoi(getenv("LENTIL_HOST_TRACE")) : 0; return n; }

// What the end of a streamed pass needs to know of its beginning (redistribute_streamed -> streamed_finish; kept in
// lentil_hip_ctx::Inflight while the pass's end is pending).
struct StreamTail {
  hipStream_t tail = nullptr;         // the stream the pass's last kernels and its counter read-back are on
  std::chrono::steady_clock::time_point pass_t0;
  bool calibrates_now = false, predicted = false, lean = false, extend = false, live = false, inject = false;
  int blind_rounds = 2;
  DrawArgs da{};
  SlowRec *slow_base = nullptr;
  uint32_t slow_cap_all = 0;
  unsigned accept_blocks = 0;
  uint32_t item_cap = 0, task_cap = 0, range_cap = 0;
  uint64_t pool_cap = 0, stuck_ticks = 0;
  bool deferred = false;              // the pass's end was left to whoever observes it next
  bool did_more = false;              // streamed_finish had to enqueue more work (rounds, a redo): the frame moved on
};

struct lentil_hip_ctx {
  HostTrace trace;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::string err;
  int num_cu = 256;

  lentil_params P{};
  bool have_params = false;

  DevLens hlens{};
  DevLens *d_lens = nullptr;
  DevTerm *d_terms = nullptr;
  bool have_lens = false;
  unsigned long long lens_hash = 0;   // FNV-1a of the base table, matches gen::Lens_*::kTableHash
  // Run-time specialisation for a table without a compiled-in kernel (lentil_lens_jit.h): the shared entry of the table's hash,
  // and this device's module once the code object has arrived
  std::vector<DevTerm> h_terms;       // the packed table as lentil_hip_set_lens built it (the emitter's input)
  bool jit_enabled = true;            // LENTIL_LENS_JIT=0: the interpreter for every table without a compiled-in kernel
  std::shared_ptr<lentil_jit::Entry> jit;
  hipModule_t jit_module = nullptr;
  hipFunction_t jit_fn[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  bool jit_loaded = false, jit_load_failed = false;
  bool use_generated = true;          // LENTIL_FORCE_TABLES=1 forces the table interpreter

  DevBokeh bokeh{};
  bool have_bokeh = false;
  double *d_blade_sc = nullptr;       // DevBokeh::blade_sc (see there)
  int blade_count = 0;

  FrameDev F{};
  uint8_t kind[LENTIL_MAX_AOVS] = {0};
  float *d_resolved = nullptr;
  bool have_frame = false;
  // FrameDev::dir (see there): the allocation, whether its content counts (F.dir is set exactly then), whether
  // it may hold non-zeros at all, and the pixels of the stream that wrote them
  float *d_dir = nullptr;
  bool dir_dirty = false;
  // FrameDev::touched (see there): the allocation; F.touched is set while it is trusted
  uint8_t *d_touched = nullptr;
  bool cleared_since_pass = false;     // clear_frame / alloc_frame, and no redistribute since
  // closest-filtered AOVs with candidates at depth 0 / NaN (lentil_closest_replay.h): the pass raised
  // DevCounters::degenerate_depth; a draw log the library set up for the replay itself (no caller asked for one)
  bool degenerate_seen = false;
  bool closest_auto_log = false;
  uint64_t closest_auto_log_cap = 0;
  uint64_t n_degenerate_replays = 0;
  struct DirRegion {
    int32_t x0 = 0, y0 = 0; uint32_t row_stride = 0, ppr = 0; uint64_t npix = 0;
    bool operator==(const DirRegion &o) const { return x0 == o.x0 && y0 == o.y0 && row_stride == o.row_stride && ppr == o.ppr && npix == o.npix; }
  } dir_region;
  bool scan_dma = true;               // LENTIL_SCAN_DMA=0: register-staged scan kernels only

  VisitsDev V{};
  bool have_visits = false;
  uint64_t visits_gen = 0;            // counts the visit streams bound so far (cryptomatte columns belong to one of them)
  std::vector<void *> owned_visit_mem;

  uint2 *d_work = nullptr;
  uint64_t work_cap = 0;
  // The visit stream is processed in n_chunks contiguous chunks.  All scans run back to back on the
  // main stream; each chunk's draw pipeline (prep, solve/accept rounds) runs on its own stream as soon
  // as its scan has finished, so the HBM-bound scan of chunk i+1 overlaps the fp64-bound solves of
  // chunk i, and the latency-bound tails of different chunks overlap each other.
  struct Chunk {
    hipStream_t stream = nullptr;
    hipEvent_t scanned = nullptr, done = nullptr;
    uint64_t v_begin = 0, v_end = 0, tile_begin = 0, tile_end = 0;
    uint64_t n_items = 0;
    ItemHdr *hdr = nullptr;
    ItemProg *prog = nullptr;
    uint32_t *active[2] = {nullptr, nullptr};
    uint64_t item_cap = 0;
    Task *tasks[2] = {nullptr, nullptr};
    uint64_t task_cap = 0;
    uint32_t *pool[2] = {nullptr, nullptr};
    uint64_t pool_cap = 0;
    SlowRec *slow = nullptr;     // straggler queue (solve_slow_kernel)
    uint64_t slow_cap = 0;
    // what the last pass found in this chunk: sizes the buffers and grids of the next pass, whose draw rounds are
    // then enqueued without waiting for the scan ("blind"; prep_items_kernel checks that everything fits)
    bool have_est = false, was_blind = false;
    bool heavy_pending = false;    // this pass: the chunk's first solve round fills the chip; `done` marks its end
    uint64_t est_items = 0, est_sum = 0;
    int est_rounds = 3;            // rounds the chunk's items needed (DevCounters::rounds_used)
  };
  std::vector<Chunk> chunks;
  int n_chunks = 2;
  double first_chunk_frac = 0.5;             // LENTIL_FIRST_CHUNK_FRAC
  uint64_t early_cap_samples = 4ull << 20;   // LENTIL_EARLY_CAP_SAMPLES
  uint64_t max_pool_units = 1ull << 32;    // 16 GiB per pool at most (LENTIL_MAX_POOL_UNITS overrides)
  double mean_iters = 0.0;                   // Newton iterations per solve in the last pass (0: unknown)
  double parked_frac = 0.0;                  // share of the last pass's solves that were parked for solve_slow_kernel
  bool park_dry_seen = false;                // a live queue has been flooded with this lens (set_lens forgets it)
  int park_dry_only = -1;                    // LENTIL_PARK_DRY_ONLY: 1 live queues too take only the last lanes of waves running dry, 0 never, -1 by parked_frac
  int slow_at = 20;                          // LENTIL_SLOW_AT: iterations after which a solve is parked (0: never)
 uint64_t slow_below = 8ull << 20;          // LENTIL_SLOW_BELOW: ... in chunks whose draw sum is below this
 int slow_from_round = 0;                   // LENTIL_SLOW_FROM_ROUND
  int slow_max_lanes = 4;                    // LENTIL_SLOW_MAX_LANES
  int slow_prio = 0;                         // LENTIL_SLOW_PRIO: instruction priority of the straggler kernel's waves (0-3)
  int slow_waves_per_cu = 0;                 // LENTIL_SLOW_WAVES_PER_CU: straggler waves per CU of a streamed pass (0: one)
  int slow_nap_max = 0;                      // LENTIL_SLOW_NAP_MAX: DrawArgs::slow_nap_max
  bool blind = true;                         // LENTIL_BLIND=0: always wait for a chunk's scan before sizing its draw rounds
  uint32_t extra_num = 0, extra_const = 16;  // LENTIL_EXTRA_256THS / LENTIL_EXTRA_CONST: first-batch over-provisioning (16 spare attempts: what a decoupled first accept's guess about its unknown attempts may be off by, accept_item<1>)
  uint64_t extra_below = 8ull << 20;         // LENTIL_EXTRA_BELOW: ... while a chunk's draw sum is below this
  int solve_cap_blocks = 1;                  // LENTIL_EARLY_CAP_BLOCKS: solve blocks per CU while later chunks are scanned
  int accept_max_blocks = 4;                 // LENTIL_ACCEPT_BLOCKS: accept blocks per CU at most
  int accept_stream_blocks = 2;              // ... in a streamed pass, whose accepts share the CUs with the next round's solves
  bool extend = false;                       // LENTIL_EXTEND=1: a streamed pass's first round appends the batches its items still need itself
                                             // (ItemLive, lentil_kernels.h).  Built, correct, and measured slower: 2.30 against 2.00 ms, see there
  bool lean_tail = true;                     // LENTIL_LEAN_TAIL=0: ... and its second round's kernels are always in flight
  bool lean_ok = true;                       // the last streamed pass with extension left its first accept nothing to schedule
  uint64_t n_lean_lost = 0;                  // passes whose lean tail had to run the second round after all
  // First batches from the lens and the frame (lentil_batch_model.h): the calibration table, what it was built for, and how
  // the bets on it went.  LENTIL_PREDICT=0: every item starts with samples + retries + spare, as before round 5.
  // Streamed pass, beauty-only frames (scan_dma2_kernel): per cent of the CUs that get a scan block.  The others hold three
  // resident solve blocks instead of two: the pass is bound by the solves' fp64 issue slots and the scan's HBM bytes at the
  // same time, and a CU that does both does the solves at two waves per SIMD with the scan's waves in between.
  // LENTIL_SCAN_CUS_PCT.
  double longest_pass_ms = 0.0;              // the longest streamed pass of this context so far (host time; sizes the stuck time-out)
  uint64_t longest_pass_visits = 0, longest_pass_sum = 0;      // ... and how large that pass was (visits, draws expected)
  int scan_cus_pct = 84;                     // (100 / 92 / 84 / 76: 2.10 / 2.12 / 2.02 / 2.04 ms, means of four runs of 60 steps on one box)
  int scan_cus_pct_multi = 100;              // ... for frames with extra AOV columns (scan_dma_multi_kernel): LENTIL_SCAN_CUS_PCT_MULTI
  unsigned last_scan_skipped = 0;             // blocks of the last scan launch that left at once (scan_dma2_kernel, ScanArgs::skip_blocks)
  int crypto_tile_blocks = 16;               // blocks per CU of crypto_direct_tile_kernel (LENTIL_CRYPTO_TILE_BLOCKS)
  bool predict = true;
  uint64_t predict_max_draws = 1536;         // LENTIL_PREDICT_MAX_DRAWS: mean draws per item up to which a pass sizes first batches from the model
  float4 *d_bm_land = nullptr;
  float4 *d_bm_box = nullptr;
  uint32_t *d_bm_npass = nullptr;
  bool bm_valid = false;
  uint64_t bm_lens_sig = 0, bm_bokeh_sig = 0; // what the calibration was traced through (FNV-1a of lens header + terms / of the aperture tables)
  uint32_t bm_nx = 17, bm_ny = 11, bm_nz = 9; // LENTIL_PREDICT_GRID=nx,ny,nz
  float bm_fx0 = 0, bm_fx_step = 0, bm_fy0 = 0, bm_fy_step = 0, bm_u0 = 0, bm_u_step = 0;
  uint32_t bm_margin16 = 0;                  // sixteenths the model's margin has been widened by (a lost bet adds one, at most 4)
  uint32_t bm_since_loss = 0;
  uint64_t n_bm_built = 0, n_lean = 0;       // calibrations run / passes that ran with the lean tail
  ItemLive *d_live = nullptr;                // ... one record per item of chunk 0
  uint64_t live_cap = 0;
  Task *d_ext_q = nullptr;                   // ... the queue of the batches it appends
  uint64_t ext_q_cap = 0;
  int solve_max_blocks = 4;                  // LENTIL_SOLVE_BLOCKS: solve blocks per CU at most
  // Streamed pass (polynomial optics, from the second pass of a context on): one scan launch that publishes its items
  // and their first-batch tasks itself, persistent solve waves that follow the task queue while the scan runs.
  bool stream_mode = true;                   // LENTIL_STREAM=0: chunked passes only
  bool stream_below_set = false;             // LENTIL_STREAM_BELOW given: that number alone decides
  uint64_t stream_below = 5ull << 19;        // LENTIL_STREAM_BELOW: ... and for passes with at least this many draws (previous pass's count; 2.5 Mi: measured crossover between 1.2 M draws, streamed 2.6 vs 2.9 ms, and 3.1 M, 5.1 vs 4.9 ms)
  int stream_blocks = 2;                     // LENTIL_STREAM_BLOCKS: solve blocks per CU beside the scan (1 or 2)
  uint32_t epoch = 0;                        // tag of the current pass's task slots
  uint64_t *d_ranges = nullptr;              // range queue scan -> publish_kernel
  uint64_t range_cap = 0;
  int publish_waves = 64;                    // LENTIL_PUBLISH_WAVES
  bool have_total_est = false;               // what the last pass found in the whole stream
  uint64_t est_items_total = 0, est_sum_total = 0;
  int est_rounds_total = 3;
  uint32_t last_streamed = 0;
  uint64_t n_stuck = 0;                      // streamed passes whose waves gave up waiting (redone the chunked way)
  // ... of which had accepted draws by then: the frame is wiped and the whole pass run again, chunked (lentil_hip_redistribute)
  uint64_t n_stall_redone = 0;
  bool stall_redo = false;                   // redistribute_streamed -> lentil_hip_redistribute
  int inject_stall_at = 0;                   // LENTIL_INJECT_STALL=k: the context's k-th streamed pass stalls after its first accept (tests)
  uint64_t n_streamed = 0;
  int last_rounds = 0;
  uint32_t last_blind = 0, last_fallback = 0;
  std::string redo_note;                     // why the last streamed pass that was redone gave up (lentil_hip_last_redo_note)
  uint32_t last_scan_launches = 0;
  // rows that may be non-zero since the last clear_frame (only trusted when dirty_known)
  int32_t dirty_lo = 0, dirty_hi = 0;
  bool dirty_known = false;
  hipEvent_t scans_done = nullptr;   // after the last chunk's scan of a pass
  bool overlap_rounds = true;        // LENTIL_OVERLAP_ROUNDS=0: a streamed pass's second round starts after its first accept has ended
  bool slow_live = true;             // LENTIL_SLOW_LIVE=0: stragglers of a streamed pass wait for their round's solve kernel to end
  int crowd_stays_first = 0, crowd_stays_later = 0;    // LENTIL_CROWD_STAYS=ab (two digits): DrawArgs::slow_crowd_stays of a streamed pass's first / later rounds
  bool solve_b = false;              // LENTIL_SOLVE_B=1: the second solve launch behind the scan (its blocks only find room when the first launch's leave: measured idle)
  uint32_t margin_low_rate = 10;      // LENTIL_BATCH_MARGIN_LOW_RATE (sixteenths, 0..16): DrawArgs::margin_low_rate
  uint32_t batch_margin16 = 4;        // LENTIL_BATCH_MARGIN (sixteenths, 0..16): DrawArgs::batch_margin16
  uint32_t unknown_credit = 7;        // LENTIL_UNKNOWN_CREDIT (0..8): DrawArgs::unknown_credit
  bool chain_streams = true;          // LENTIL_CHAIN_STREAMS=0: a decoupled pass keeps its accepts on the main stream
  hipEvent_t ev_solve = nullptr, ev_slow1 = nullptr;
  hipStream_t aux_stream = nullptr;     // a fifth stream that runs beside the four (more than four hardware queues to be had), or null
  hipEvent_t ev_crypto = nullptr;       // the cryptomatte own-pixel adds a streamed pass has put on aux_stream
  bool crypto_direct_enqueued = false;
  bool streams_concurrent = false;      // pick_concurrent_streams: the streamed pass's four streams run their kernels side by side
  hipStream_t slow1_stream = nullptr;   // a decoupled pass's second-round straggler kernel (beside the first round's, which is still at work)
  bool decouple = true;              // LENTIL_DECOUPLE=0: a streamed pass's first accept waits for the first round's stragglers
  hipEvent_t ev_slow = nullptr, ev_round = nullptr;
  // A streamed pass resolves the frame while its second round is still solving (the chip's HBM is idle then) and, at the
  // end, once more the 64-pixel groups that received draws: lentil_hip_resolve then finds its work done.
  bool early_resolve = true;         // LENTIL_EARLY_RESOLVE=0: lentil_hip_resolve does all of it
  bool early_resolve_pending = false;    // the early half of this pass is enqueued (ev_res marks its end)
  bool late_resolve_done = false;    // ... and so is the second half, behind the last accept
  bool resolved_valid = false;       // d_resolved holds the frame as it is
  hipEvent_t ev_acc1 = nullptr, ev_res = nullptr, ev_b = nullptr;
  hipStream_t pub_stream = nullptr;  // streamed pass: publish_kernel, then the live straggler kernel
  hipEvent_t pub_done = nullptr;
  bool pass_pending = false;         // a redistribute ran whose rows have not been asked for yet
  bool closest_deferred = false;     // multi-GPU: the caller min-reduces the keys before the gather
  uint32_t visit_id_base = 0;
  DevCounters *d_ctr = nullptr;
  std::vector<DevCounters> h_ctr;    // the chunks' counters as read back at the end of the last (blind) pass
  DevCounters *d_ctr_host = nullptr;     // the pinned staging block as the device addresses it (report_counters_kernel)
  uint32_t *h_seq = nullptr;             // ... the sequence number behind its records: the pass whose counters they are
  uint32_t seq = 0;
  bool spin_readback = false;            // LENTIL_SPIN_READBACK=1: the counters by a kernel into pinned memory, the host polls a sequence
                                         // number (measured: 1.992 against 1.993 ms, four runs each on one box -- hipStreamSynchronize
                                         // already spins; off)
  DevCounters *h_ctr_pinned = nullptr;   // staging for that read-back (pinned: an asynchronous copy at the end of each chunk's stream)
  bool h_ctr_valid = false;
  lentil_draw_record *d_log = nullptr;
  uint64_t log_cap = 0;
  bool timed_draw = false, timed_resolve = false;
  // A streamed pass's one scan launch carries its own start / stop events (hipExtLaunchKernelGGL: the timestamps of the
  // kernel's dispatch itself).  Events recorded around the launch read ~90 us more than rocprofv3's kernel trace of the same
  // launch -- the marker behind the scan is only processed once the command processor gets round to that queue again,
  // with four other queues of the pass busy.
  hipEvent_t ev_scan_k[2] = {nullptr, nullptr};
  bool scan_kernel_timed = false;
  // thin lens with abb_chromatic > 0: the xor128 state the colour channels are drawn from (src/global.h:22-27), handed
  // from pass to pass, and the pass's buffers
  uint32_t xor_state[4] = {123456789u, 362436069u, 521288629u, 88675123u};
  uint32_t *d_xor = nullptr, *d_tlc_res = nullptr;
  uint64_t *d_tlc_off = nullptr;
  uint2 *d_tlc_tasks = nullptr;
  uint64_t tlc_res_cap = 0, tlc_off_cap = 0, tlc_tasks_cap = 0;
  double lens_housing_radius = 0.0;  // lens_aperture_housing_radius of the current table (focus search)
  float4 *d_dummy = nullptr;          // ScanArgs::dummy
  float *d_cam_keys = nullptr;        // lentil_hip_set_camera_motion
  float shutter_t0 = 0.0f, shutter_inv_dt = 1.0f;     // lentil_hip_set_camera_shutter: the keys' first time, 1 / (last - first)
  uint32_t n_cam_keys = 0;
  // ---- the asynchronous end of a streamed pass (round 6; lentil_hip_set_async, LENTIL_ASYNC_END=0 switches it off) ---------------
  // lentil_hip_redistribute used to end with the host waiting for the pass's counters (did everything fit?  did the lean tail's
  // bet hold?  did a wave give up waiting?) -- and the GPU then idled until the host had come round to the next frame's clear and
  // scan: 55-133 us of a 2 ms step, depending on the box's host.  Now a streamed pass with the lean tail returns once its kernels
  // and the copy of its counters are enqueued; the verdict is read by whichever call next OBSERVES the frame or the counters
  // (sync, downloads, get_counters, last_timing, every set-up call: CHECK_CTX) -- that call waits for the pass, and if the pass needs
  // more work (a lost bet, buffers that were too small, a stall) does that work then, so the observer sees exactly what it saw
  // before.  clear_frame / bind_visits / redistribute / resolve do not observe: a caller that pipelines frames (clear, pass,
  // resolve, next clear ...) keeps the GPU fed.  A pass whose frame is cleared away before anybody looked at it is ABANDONED: its
  // counters are still read (estimates for the next pass, lentil_hip_pass_totals), but work it still needed is not done -- nobody can
  // see that frame any more; pass_totals.abandoned_incomplete counts such passes.  At most two passes are in flight unobserved.
  bool async_end = true;
  // clear_frame's wipe of the splatted groups off the main stream (round 6; LENTIL_CLEAR_OFFSTREAM=0: on it, as before): a streamed
  // pass's scan neither reads nor writes the splat accumulators -- the pixels' own sums go to FrameDev::dir --, so the wipe (20-25 us
  // of HBM writes for a 4K frame) runs on the stream of the pass's solve and accept kernels, beside the scan; the first accept
  // follows it in stream order, the early resolve waits for ev_clear, and every other consumer of the frame joins it first
  // (join_clear: CHECK_CTX, lentil_hip_resolve, every pass that is not the streamed one).
  bool clear_offstream = true;
  bool clear_pending = false;
  hipEvent_t ev_pre_clear = nullptr, ev_clear = nullptr;
  // accept_kernel<3> (lentil_kernels.h) as the lean tail's first accept: LENTIL_READY_ACCEPT=0 restores round 5's pair, LENTIL_READY_ACCEPT_BLOCKS
  // its blocks per CU; LENTIL_RESOLVE_AFTER_SCAN=0 / 1 decides where the whole-frame resolve runs whatever the accept (-1: with accept_kernel<3>)
  bool ready_accept = true;
  int ready_blocks = 2;              // (2 / 3 / 4 per CU: 2.011 / 2.013 / 2.013 ms, eight interleaved 60-step runs each on one box: gpurun_out/r06s05)
  // ... and that accept BESIDE the first round's solves (accept_kernel<4>, DrawArgs::early_accept): LENTIL_EARLY_ACCEPT=0 keeps it behind
  // them (accept_kernel<3>); LENTIL_EARLY_ACCEPT_BLOCKS its blocks per CU (1: a wave per SIMD beside two solve waves)
  // MEASURED AND OFF BY DEFAULT (LENTIL_EARLY_ACCEPT=1 switches it on): frames bit-identical (tests pass with it on), and the pass
  // 2.33 ms against 2.01 on one box (gpurun_out/r06s07: six interleaved 60-step runs each).  The accept's 160 us do move under the
  // solve kernel -- and cost more than they were: the solve kernel runs 1.62-1.65 ms where it took 1.47-1.51 (its results go out as
  // write-through atomics and every flush waits for them and for a returning atomic per item; the accept's waves take issue slots and
  // memory-side atomic bandwidth from it), the straggler waves -- 152 registers -- find no room on a SIMD that holds two solve waves
  // AND an accept wave, so the parked solves start when the solve waves leave and end 210-270 us behind them (170 without), and the
  // accept behind the stragglers gets the items that met them: 95 us where it took 33.
  bool early_accept = false;
  int early_blocks = 1;
  uint64_t *d_ready = nullptr;       // the queue of completed items (tagged slots, one per item)
  uint64_t ready_cap = 0;
  int resolve_after_scan = -1;
  struct Slot {                       // what a pass needs of its own while another pass is being enqueued: events, the counters' landing block
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_scan_k[2] = {nullptr, nullptr};
    hipEvent_t ev_tail = nullptr;     // behind the counters' copy on the pass's tail stream
    DevCounters *h_ctr_pinned = nullptr;
    bool timing_open = false;         // a pass used these events and its times are not in the totals yet
    bool timed_draw = false, timed_resolve = false, scan_kernel_timed = false;
  } slots[3];
  int slot = 0;
  struct Inflight {                   // a streamed pass whose end nobody has looked at
    int slot = 0;
    bool abandoned = false;           // its frame has been cleared since
    bool resolve_requested = false;   // lentil_hip_resolve was called behind it
    StreamTail t;
    VisitsDev V{};                    // the stream it ran on (bind_visits may have moved on)
    bool have_visits = false;
  };
  std::deque<Inflight> inflight;
  bool holds_turn = false;            // this context owns g_stream_owner[device]
  lentil_pass_totals totals{};
  std::vector<std::string> notes;     // redo notes since the totals were last reset (at most 8 kept)
  // occlusion probes (lentil_hip_set_occlusion_probe): the renderer's callback, AiCameraToWorldMatrix (given, or the fp64 inverse of the
  // world-to-camera matrix / of every motion key), the list buffers (device, and pinned host copies the callback reads and fills)
  lentil_probe_fn probe_fn = nullptr;
  void *probe_user = nullptr;
  bool probe_c2w_given = false;
  float probe_c2w[16];
  std::vector<float> h_cam_keys;              // lentil_hip_set_camera_motion's keys, kept for their inverses
  float *d_c2w_keys = nullptr;
  lentil_probe_segment *d_probe_seg = nullptr, *h_probe_seg = nullptr;
  uint32_t *d_probe_idx = nullptr;
  uint8_t *d_probe_occ = nullptr, *h_probe_occ = nullptr;
  unsigned int *d_probe_count = nullptr;
  uint64_t probe_cap = 0;
  uint64_t n_probes = 0, n_probes_occluded = 0, n_probe_calls = 0;     // since the context was created (lentil_hip_probe_stats)
  struct LentilUpload *upload = nullptr;   // lentil_upload.h: the visit stream handed over piece by piece
  struct LentilComm *comm = nullptr; // lentil_comm.h: this context's RCCL communicator, if one was asked for
  struct LentilCrypto *crypto = nullptr;   // lentil_crypto.h: cryptomatte AOVs, if any were allocated
  uint64_t crypto_auto_log = 0;            // capacity of the draw log the cryptomatte replay allocated itself (0: the log is the caller's, or none)
  uint64_t crypto_auto_log_hint = 0;       // what the last pass needed, for the next log it allocates
};

static thread_local std::string g_err;
static std::atomic<lentil_hip_ctx *> g_stream_owner[64];
// (an owner, not a mutex: a pass whose end is left to its next observer keeps the device's turn beyond the call that started
// it -- possibly to be given back from another thread --, and a later pass of the SAME context, which follows it on the same
// streams, needs no turn of its own)
static void release_turn(lentil_hip_ctx *ctx);
struct DeviceTurn {
  lentil_hip_ctx *ctx;
  bool taken = false, kept = false;
  explicit DeviceTurn(lentil_hip_ctx *c) : ctx(c) {}
  bool take(bool wait);
  void keep() { kept = true; }
  ~DeviceTurn();
};
bool DeviceTurn::take(bool wait) {
  if (ctx->holds_turn) return true;
  std::atomic<lentil_hip_ctx *> &o = g_stream_owner[ctx->device & 63];
  lentil_hip_ctx *none = nullptr;
  while (!o.compare_exchange_strong(none, ctx)) {
    if (!wait) return false;
    none = nullptr;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  taken = true;
  ctx->holds_turn = true;
  return true;
}
static void release_turn(lentil_hip_ctx *ctx) {
  if (!ctx->holds_turn || !ctx->inflight.empty()) return;
  g_stream_owner[ctx->device & 63].store(nullptr);
  ctx->holds_turn = false;
}
DeviceTurn::~DeviceTurn() { if (taken && !kept) release_turn(ctx); }

static void apply_camera_motion(lentil_hip_ctx *ctx);

// One streamed pass at a time per device and process: its solve waves are resident while they wait for the scan's
// output, and the waves of two such passes can fill the CUs' register files between them before either scan is
// placed (contexts driven from several threads; across processes the bounded wait and the chunked redo catch it).

// What every context of this process has seen (lentil_hip_process_stats): streamed passes begun, passes whose resident waves hit
// the stuck time-out, how many of those were asked for (LENTIL_INJECT_STALL), passes redone after draws had been accepted.  A
// stalled pass is redone and its result is correct -- which is exactly why it needs a number somebody can assert on: the
// GPU suite's session ends with "no stall that was not injected" (tests/conftest.py).
static std::atomic<uint64_t> g_stat_streamed{0}, g_stat_stuck{0}, g_stat_stuck_injected{0}, g_stat_redone{0};
// ... and what the waves that gave up left behind (lentil_hip_process_stall_notes): the redo notes of the passes counted in
// g_stat_stuck that nobody asked for, the first eight of them
static std::mutex g_stall_notes_mutex;
static std::vector<std::string> g_stall_notes;

static int fail(lentil_hip_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg; else g_err = msg;
  return code;
}
#define HIP_TRY(ctx, call)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(ctx, LENTIL_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
  } while (0)
static int settle(lentil_hip_ctx *ctx);
static int probe_round(lentil_hip_ctx *ctx, const DrawArgs &da, hipStream_t st);
static int join_clear(lentil_hip_ctx *ctx);
static void harvest_ready(lentil_hip_ctx *ctx);
static int resolve_range(lentil_hip_ctx *ctx, uint64_t p_begin, uint64_t p_end);
// Every entry point observes the context's last pass -- waits for it, finishes what it left open (settle) -- except the four a
// caller pipelines frames with (clear_frame, bind_visits, redistribute, resolve: CHECK_CTX_PIPELINED) and the pure getters.
#define CHECK_CTX_PIPELINED(ctx) \
  if (!(ctx)) return fail(nullptr, LENTIL_ERR_INVALID, "null context")
#define CHECK_CTX(ctx)                                                     \
  do {                                                                     \
    if (!(ctx)) return fail(nullptr, LENTIL_ERR_INVALID, "null context"); \
    if (!(ctx)->inflight.empty()) {                                        \
      const int rc_settle_ = settle(ctx);                                  \
      if (rc_settle_) return rc_settle_;                                   \
    }                                                                      \
    if ((ctx)->clear_pending) {                                            \
      const int rc_join_ = join_clear(ctx);                                \
      if (rc_join_) return rc_join_;                                       \
    }                                                                      \
  } while (0)

// the context's stream behind a wipe that clear_frame put on another stream (lentil_hip_ctx::clear_offstream)
static int join_clear(lentil_hip_ctx *ctx) {
  if (!ctx->clear_pending) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_clear, 0));
  ctx->clear_pending = false;
  return LENTIL_OK;
}

static inline void ht_mark(lentil_hip_ctx *ctx, const char *what) {
  if (!host_trace_passes()) return;
  if (ctx->trace.passes_left < 0) ctx->trace.passes_left = host_trace_passes();
  if (ctx->trace.passes_left == 0) return;
  ctx->trace.marks.emplace_back(what, (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                                          std::chrono::steady_clock::now().time_since_epoch()).count());
}
static void ht_dump(lentil_hip_ctx *ctx) {
  if (!host_trace_passes() || ctx->trace.marks.empty()) return;
  const int64_t t0 = ctx->trace.marks.front().second;
  fprintf(stderr, "[host trace]");
  for (const auto &m : ctx->trace.marks) fprintf(stderr, " %s %.1f", m.first, (double)(m.second - t0) * 1.0e-3);
  fprintf(stderr, " (us; epoch %.1f)\n", (double)t0 * 1.0e-3);
  ctx->trace.marks.clear();
  if (ctx->trace.passes_left > 0) --ctx->trace.passes_left;
}

static double host_ipow(double x, int e) {   // lens_ipow, src/lens.h:226-233
  if (e == 0) return 1.0;
  if (e == 1) return x;
  if (e == 2) return x * x;
  const double p2 = host_ipow(x, e / 2);
  if (e & 1) return x * p2 * p2;
  return p2 * p2;
}

LENTIL_API int lentil_hip_abi_version(void) { return LENTIL_ABI_VERSION; }

// ---------------------------------------------------------------------------------------
// A streamed pass keeps four kernels resident that hand work to one another -- scan (main stream), publish_kernel and
// the straggler kernel (pub_stream), the persistent solve kernel (the first chunk's stream), the second round's
// stragglers (slow1_stream).  The runtime multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES,
// 4 by default), and kernels of streams that share one run one after the other: the pass still ends (nothing in it waits
// for a kernel submitted later), but publish_kernel behind the scan means the solves start when the scan has ended --
// 3.4 ms for the headline frame instead of 2.3, seen for every context but the first one a process creates.  So the
// four streams are chosen: a candidate is kept if a kernel on it runs while a kernel on each stream kept so far is
// waiting for it (probe_wait_kernel / probe_set_kernel), else another stream is created -- the rejected ones stay
// alive until the choice is made, so that the runtime hands out a different queue next.  LENTIL_STREAM_PROBE=0: take
// the streams as they come.
// ---------------------------------------------------------------------------------------
static bool streams_run_together(lentil_hip_ctx *ctx, hipStream_t x, hipStream_t y, uint32_t *d_flag) {
  uint32_t *d_seen = d_flag + 1;
  if (hipMemsetAsync(d_flag, 0, 2 * sizeof(uint32_t), x) != hipSuccess) return false;
  if (hipStreamSynchronize(x) != hipSuccess) return false;
  hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, x, d_flag, d_seen);
  hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, y, d_flag);
  (void)hipStreamSynchronize(x);
  (void)hipStreamSynchronize(y);
  uint32_t seen = 0;
  if (hipMemcpy(&seen, d_seen, sizeof(seen), hipMemcpyDeviceToHost) != hipSuccess) return false;
  (void)ctx;
  return seen != 0u;
}

static int pick_concurrent_streams(lentil_hip_ctx *ctx) {
  ctx->streams_concurrent = false;
  if (const char *e = getenv("LENTIL_STREAM_PROBE")) if (e[0] == '0') return LENTIL_OK;
  uint32_t *d_flag = nullptr;
  HIP_TRY(ctx, hipMalloc(&d_flag, 2 * sizeof(uint32_t)));
  hipStream_t *role[4] = {&ctx->stream, &ctx->chunks[0].stream, &ctx->pub_stream, &ctx->slow1_stream};
  std::vector<hipStream_t> rejected;
  bool all = true;
  int created = 0;
  for (int r = 1; r < 4; ++r) {
    while (true) {
      bool ok = true;
      for (int q = 0; q < r && ok; ++q) ok = streams_run_together(ctx, *role[q], *role[r], d_flag);
      if (ok) break;
      if (created >= 12) { all = false; break; }       // (fewer than four hardware queues to be had: as they come)
      rejected.push_back(*role[r]);
      hipStream_t s = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { *role[r] = rejected.back(); rejected.pop_back(); all = false; break; }
      *role[r] = s;
      ++created;
    }
  }
  // a spare one beside the four (the cryptomatte replay's first half runs there while the draws go on): only where the
  // runtime has a fifth hardware queue to give (GPU_MAX_HW_QUEUES > 4)
  if (all) {
    // (lowest priority: what runs there -- the cryptomatte replay's own-pixel kernel, thousands of 41 KB blocks -- must not
    // stand between the pass's latency-bound tail kernels and the CU slots its blocks give back; LENTIL_AUX_PRIO=0: default priority)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const bool low = !(getenv("LENTIL_AUX_PRIO") && getenv("LENTIL_AUX_PRIO")[0] == '0');
    for (int attempt = 0; attempt < 6 && !ctx->aux_stream; ++attempt) {
      hipStream_t s = nullptr;
      if ((low ? hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_least) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) break;
      bool ok = true;
      for (int q = 0; q < 4 && ok; ++q) ok = streams_run_together(ctx, *role[q], s, d_flag) && streams_run_together(ctx, s, *role[q], d_flag);
      if (ok) ctx->aux_stream = s;
      else rejected.push_back(s);
    }
  }
  for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
  (void)hipFree(d_flag);
  (void)hipGetLastError();
  ctx->streams_concurrent = all;
  if (getenv("LENTIL_STREAM_DEBUG")) fprintf(stderr, "[stream] the pass's four streams run together: %d (%d streams tried beyond the first four), a fifth beside them: %d\n", (int)all, created, ctx->aux_stream ? 1 : 0);
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_streams_concurrent(lentil_hip_ctx *ctx, int *concurrent) {
  CHECK_CTX(ctx);
  if (!concurrent) return fail(ctx, LENTIL_ERR_INVALID, "concurrent is null");
  *concurrent = ctx->streams_concurrent ? (ctx->aux_stream ? 2 : 1) : 0;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_create(int device, lentil_hip_ctx **out_ctx) {
  if (!out_ctx) return fail(nullptr, LENTIL_ERR_INVALID, "out_ctx is null");
  *out_ctx = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, LENTIL_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(nullptr, LENTIL_ERR_INVALID, "device index out of range");
  lentil_hip_ctx *ctx = new (std::nothrow) lentil_hip_ctx();
  if (!ctx) return fail(nullptr, LENTIL_ERR_NOMEM, "out of host memory");
  ctx->device = device;
  HIP_TRY(ctx, hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (auto &ev : ctx->ev) HIP_TRY(ctx, hipEventCreate(&ev));
  if (const char *nc = getenv("LENTIL_CHUNKS")) { ctx->n_chunks = atoi(nc); }
  if (ctx->n_chunks < 1) ctx->n_chunks = 1;
  if (ctx->n_chunks > 16) ctx->n_chunks = 16;
  // one DevCounters per chunk + one shared (draw-log cursor)
  HIP_TRY(ctx, hipMalloc(&ctx->d_ctr, sizeof(DevCounters) * (ctx->n_chunks + 1)));
  // (+ one record's room behind them for the sequence number of report_counters_kernel)
  HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_ctr_pinned, sizeof(DevCounters) * (ctx->n_chunks + 1), hipHostMallocDefault));
  memset(ctx->h_ctr_pinned, 0, sizeof(DevCounters) * (ctx->n_chunks + 1));
  ctx->h_seq = reinterpret_cast<uint32_t *>(ctx->h_ctr_pinned + ctx->n_chunks);
  if (hipHostGetDevicePointer((void **)&ctx->d_ctr_host, ctx->h_ctr_pinned, 0) != hipSuccess) { ctx->d_ctr_host = nullptr; (void)hipGetLastError(); }
  if (const char *e = getenv("LENTIL_SPIN_READBACK")) ctx->spin_readback = atoi(e) != 0;
  if (const char *e = getenv("LENTIL_ASYNC_END")) ctx->async_end = e[0] != '0';
  if (const char *e = getenv("LENTIL_CLEAR_OFFSTREAM")) ctx->clear_offstream = e[0] != '0';
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_pre_clear, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_clear, hipEventDisableTiming));
  if (const char *e = getenv("LENTIL_READY_ACCEPT")) ctx->ready_accept = e[0] != '0';
  if (const char *e = getenv("LENTIL_READY_ACCEPT_BLOCKS")) { ctx->ready_blocks = atoi(e); if (ctx->ready_blocks < 1) ctx->ready_blocks = 1; if (ctx->ready_blocks > 6) ctx->ready_blocks = 6; }
  if (const char *e = getenv("LENTIL_RESOLVE_AFTER_SCAN")) ctx->resolve_after_scan = e[0] == '1' ? 1 : 0;
  if (const char *e = getenv("LENTIL_EARLY_ACCEPT")) ctx->early_accept = e[0] != '0';
  if (const char *e = getenv("LENTIL_EARLY_ACCEPT_BLOCKS")) { ctx->early_blocks = atoi(e); if (ctx->early_blocks < 1) ctx->early_blocks = 1; if (ctx->early_blocks > 4) ctx->early_blocks = 4; }
  HIP_TRY(ctx, hipEventCreate(&ctx->ev_scan_k[0]));
  HIP_TRY(ctx, hipEventCreate(&ctx->ev_scan_k[1]));
  // the slots of the asynchronous end (lentil_hip_ctx::Slot): slot 0 holds what has just been created, 1 and 2 their own
  for (int si = 0; si < 3; ++si) {
    lentil_hip_ctx::Slot &sl = ctx->slots[si];
    if (si == 0) {
      for (int i = 0; i < 5; ++i) sl.ev[i] = ctx->ev[i];
      sl.ev_scan_k[0] = ctx->ev_scan_k[0]; sl.ev_scan_k[1] = ctx->ev_scan_k[1];
      sl.h_ctr_pinned = ctx->h_ctr_pinned;
    } else {
      for (int i = 0; i < 5; ++i) HIP_TRY(ctx, hipEventCreate(&sl.ev[i]));
      HIP_TRY(ctx, hipEventCreate(&sl.ev_scan_k[0]));
      HIP_TRY(ctx, hipEventCreate(&sl.ev_scan_k[1]));
      HIP_TRY(ctx, hipHostMalloc((void **)&sl.h_ctr_pinned, sizeof(DevCounters) * (ctx->n_chunks + 1), hipHostMallocDefault));
      memset(sl.h_ctr_pinned, 0, sizeof(DevCounters) * (ctx->n_chunks + 1));
    }
    HIP_TRY(ctx, hipEventCreateWithFlags(&sl.ev_tail, hipEventDisableTiming));
  }
  HIP_TRY(ctx, hipMalloc(&ctx->d_dummy, 64 * sizeof(float4)));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctr, 0, sizeof(DevCounters) * (ctx->n_chunks + 1), ctx->stream));
  ctx->chunks.resize(ctx->n_chunks);
  for (auto &ch : ctx->chunks) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ch.stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ch.scanned, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ch.done, hipEventDisableTiming));
  }
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->scans_done, hipEventDisableTiming));
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pub_stream, hipStreamNonBlocking));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_slow, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_round, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pub_done, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_acc1, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_res, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_solve, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_slow1, hipEventDisableTiming));
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->slow1_stream, hipStreamNonBlocking));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_crypto, hipEventDisableTiming));
  { const int rc = pick_concurrent_streams(ctx); if (rc) return rc; }
  if (const char *e = getenv("LENTIL_EARLY_RESOLVE")) ctx->early_resolve = atoi(e) != 0;
  if (const char *e = getenv("LENTIL_INJECT_STALL")) ctx->inject_stall_at = atoi(e);
  if (const char *ft = getenv("LENTIL_FORCE_TABLES")) ctx->use_generated = !(ft[0] == '1');
  if (const char *fc = getenv("LENTIL_FIRST_CHUNK_FRAC")) {
    const double f = atof(fc);
    if (f > 0.0 && f < 1.0) ctx->first_chunk_frac = f;
  }
  if (const char *ec = getenv("LENTIL_EARLY_CAP_SAMPLES")) ctx->early_cap_samples = strtoull(ec, nullptr, 10);
  if (const char *mp = getenv("LENTIL_MAX_POOL_UNITS")) ctx->max_pool_units = strtoull(mp, nullptr, 10);
  if (const char *sa = getenv("LENTIL_SLOW_AT")) ctx->slow_at = atoi(sa);
  if (ctx->slow_at < 0 || ctx->slow_at >= 100) ctx->slow_at = 0;
  if (const char *e = getenv("LENTIL_SLOW_BELOW")) ctx->slow_below = strtoull(e, nullptr, 10);
  if (const char *e = getenv("LENTIL_SLOW_FROM_ROUND")) ctx->slow_from_round = atoi(e);
  if (const char *e = getenv("LENTIL_SLOW_MAX_LANES")) ctx->slow_max_lanes = atoi(e);
  if (const char *e = getenv("LENTIL_SLOW_PRIO")) ctx->slow_prio = atoi(e);
  if (const char *e = getenv("LENTIL_SLOW_WAVES_PER_CU")) ctx->slow_waves_per_cu = atoi(e);
  if (const char *e = getenv("LENTIL_SLOW_NAP_MAX")) ctx->slow_nap_max = atoi(e);
  if (const char *e = getenv("LENTIL_BLIND")) ctx->blind = !(e[0] == '0');
  if (const char *e = getenv("LENTIL_EXTRA_256THS")) ctx->extra_num = (uint32_t)strtoul(e, nullptr, 10);
  if (const char *e = getenv("LENTIL_EXTRA_CONST")) ctx->extra_const = (uint32_t)strtoul(e, nullptr, 10);
  if (const char *e = getenv("LENTIL_EXTRA_BELOW")) ctx->extra_below = strtoull(e, nullptr, 10);
  if (ctx->extra_num > 512) ctx->extra_num = 512;       // the result pool holds 4 x the draw sum
  if (ctx->extra_const > 64) ctx->extra_const = 64;
  if (const char *e = getenv("LENTIL_EARLY_CAP_BLOCKS")) ctx->solve_cap_blocks = atoi(e);
  if (ctx->solve_cap_blocks < 1) ctx->solve_cap_blocks = 1;
  if (const char *e = getenv("LENTIL_ACCEPT_BLOCKS")) ctx->accept_max_blocks = ctx->accept_stream_blocks = atoi(e);
  if (const char *e = getenv("LENTIL_EXTEND")) ctx->extend = e[0] != '0';
  if (const char *e = getenv("LENTIL_LEAN_TAIL")) ctx->lean_tail = e[0] != '0';
  if (const char *e = getenv("LENTIL_PREDICT")) ctx->predict = e[0] != '0';
  if (const char *e = getenv("LENTIL_PREDICT_MAX_DRAWS")) ctx->predict_max_draws = strtoull(e, nullptr, 10);
  if (const char *e = getenv("LENTIL_LENS_JIT")) ctx->jit_enabled = e[0] != '0';
  if (const char *e = getenv("LENTIL_SCAN_CUS_PCT_MULTI")) { ctx->scan_cus_pct_multi = atoi(e); if (ctx->scan_cus_pct_multi < 25) ctx->scan_cus_pct_multi = 25; if (ctx->scan_cus_pct_multi > 100) ctx->scan_cus_pct_multi = 100; }
  if (const char *e = getenv("LENTIL_SCAN_CUS_PCT")) { ctx->scan_cus_pct = atoi(e); if (ctx->scan_cus_pct < 25) ctx->scan_cus_pct = 25; if (ctx->scan_cus_pct > 100) ctx->scan_cus_pct = 100; }
  if (const char *e = getenv("LENTIL_PREDICT_GRID")) {
    unsigned a = 0, b = 0, c = 0;
    if (sscanf(e, "%u,%u,%u", &a, &b, &c) == 3 && a >= 2 && b >= 2 && c >= 2 && a <= 64 && b <= 64 && c <= 32) { ctx->bm_nx = a; ctx->bm_ny = b; ctx->bm_nz = c; }
  }
  if (ctx->accept_max_blocks < 1) ctx->accept_max_blocks = 1;
  if (const char *e = getenv("LENTIL_SOLVE_BLOCKS")) ctx->solve_max_blocks = atoi(e);
  if (ctx->solve_max_blocks < 1) ctx->solve_max_blocks = 1;
  if (const char *e = getenv("LENTIL_STREAM")) ctx->stream_mode = !(e[0] == '0');
  if (const char *e = getenv("LENTIL_SCAN_DMA")) ctx->scan_dma = !(e[0] == '0');
  if (const char *e = getenv("LENTIL_SLOW_LIVE")) ctx->slow_live = !(e[0] == '0');
  if (const char *e = getenv("LENTIL_PARK_DRY_ONLY")) ctx->park_dry_only = atoi(e);
  if (const char *e = getenv("LENTIL_OVERLAP_ROUNDS")) ctx->overlap_rounds = !(e[0] == '0');
  if (const char *e = getenv("LENTIL_DECOUPLE")) ctx->decouple = !(e[0] == '0');
  if (const char *e = getenv("LENTIL_CHAIN_STREAMS")) ctx->chain_streams = !(e[0] == '0');
  if (const char *e = getenv("LENTIL_BATCH_MARGIN_LOW_RATE")) { const int v = atoi(e); ctx->margin_low_rate = (uint32_t)(v < 0 ? 0 : (v > 16 ? 16 : v)); }
  if (const char *e = getenv("LENTIL_BATCH_MARGIN")) { const int v = atoi(e); ctx->batch_margin16 = (uint32_t)(v < 0 ? 0 : (v > 16 ? 16 : v)); }
  if (const char *e = getenv("LENTIL_UNKNOWN_CREDIT")) { const int v = atoi(e); ctx->unknown_credit = (uint32_t)(v < 0 ? 0 : (v > 8 ? 8 : v)); }
  if (const char *e = getenv("LENTIL_CROWD_STAYS")) { ctx->crowd_stays_first = e[0] == '1'; ctx->crowd_stays_later = e[0] && e[1] == '1'; }
  if (const char *e = getenv("LENTIL_SOLVE_B")) ctx->solve_b = e[0] == '1';
  if (const char *e = getenv("LENTIL_STREAM_BELOW")) { ctx->stream_below = strtoull(e, nullptr, 10); ctx->stream_below_set = true; }
  if (const char *e = getenv("LENTIL_STREAM_BLOCKS")) ctx->stream_blocks = atoi(e);
  if (ctx->stream_blocks < 1) ctx->stream_blocks = 1;
  // Never three: three solve blocks per CU fill the register file (3 x 168 of 512 VGPRs per lane), and should they be
  // placed before the scan's blocks -- the two kernels sit in different hardware queues -- the scan they wait for
  // never starts (seen on the first streamed pass of a context, when the scan's launch trails a memset: 250 ms until
  // the waves give up, then the chunked redo).  Two leave room for a scan block (72) and a publisher (40) everywhere.
#ifndef LENTIL_STREAM_BLOCKS_MAX
#define LENTIL_STREAM_BLOCKS_MAX 2
#endif
  if (ctx->stream_blocks > LENTIL_STREAM_BLOCKS_MAX) ctx->stream_blocks = LENTIL_STREAM_BLOCKS_MAX;
  if (const char *e = getenv("LENTIL_PUBLISH_WAVES")) ctx->publish_waves = atoi(e);
  if (ctx->publish_waves < 1) ctx->publish_waves = 1;
  {
    // The accept kernels spill a few registers: the first launch of one on a stream makes the runtime size that
    // queue's scratch memory, and it does so behind everything already in flight -- in a streamed pass that is the
    // resident solve waves which are waiting for this very accept's tasks (seen as a 250 ms stall of the first
    // streamed pass of a context, then the chunked redo).  One empty launch of each form here, on the stream the
    // streamed pass launches them on, settles that before anything waits for anything.
    DrawArgs w{};
    w.ctr = ctx->d_ctr;        // zeroed above: no active items, the kernels return at once
    const unsigned wg = (unsigned)ctx->num_cu * 8u;       // (the runtime sizes the scratch by the grid: the largest any pass launches)
    hipStream_t on[3] = {ctx->stream, ctx->chunks[0].stream, ctx->pub_stream};      // (where a streamed pass launches them)
    for (hipStream_t st : on) {
      hipLaunchKernelGGL(accept_kernel<0>, dim3(wg), dim3(256), 0, st, w);
      hipLaunchKernelGGL(accept_kernel<1>, dim3(wg), dim3(256), 0, st, w);
      hipLaunchKernelGGL(accept_kernel<2>, dim3(wg), dim3(256), 0, st, w);
      hipLaunchKernelGGL(accept_kernel<3>, dim3(wg), dim3(256), 0, st, w);
      hipLaunchKernelGGL(accept_kernel<4>, dim3(wg), dim3(256), 0, st, w);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipStreamSynchronize(st));
    }
  }
  // (the multi-AOV scan's ring can take all of a CU's LDS)
  (void)hipFuncSetAttribute((const void *)scan_dma_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void *)scan_dma2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipGetLastError();
  *out_ctx = ctx;
  return LENTIL_OK;
}

static void free_visits(lentil_hip_ctx *ctx) {
  for (void *p : ctx->owned_visit_mem) (void)hipFree(p);
  ctx->owned_visit_mem.clear();
  ctx->have_visits = false;
}
static void free_bokeh(lentil_hip_ctx *ctx) {
  if (ctx->have_bokeh) {
    (void)hipFree((void *)ctx->bokeh.cdfRow);
    (void)hipFree((void *)ctx->bokeh.rowIndices);
    (void)hipFree((void *)ctx->bokeh.cdfColumn);
    (void)hipFree((void *)ctx->bokeh.columnIndices);
  }
  ctx->bokeh = DevBokeh{};
  ctx->bokeh.blade_sc = ctx->blade_count ? ctx->d_blade_sc : nullptr;
  ctx->bokeh.blade_count = ctx->blade_count;
  ctx->have_bokeh = false;
}

LENTIL_API int lentil_hip_comm_destroy(lentil_hip_ctx *ctx);
static void upload_destroy(lentil_hip_ctx *ctx);
static void upload_release(lentil_hip_ctx *ctx, bool free_columns);
static int launch_resolve_half(lentil_hip_ctx *ctx, hipStream_t st, uint32_t only_touched);
static void crypto_destroy(lentil_hip_ctx *ctx);
static void crypto_columns_gone(lentil_hip_ctx *ctx);
static uint32_t crypto_count(const lentil_hip_ctx *ctx);
static void crypto_bind_uploaded(lentil_hip_ctx *ctx, uint32_t n_crypto, uint32_t entries, uint64_t n, float *const *hash, float *const *weight);
static int crypto_clear(lentil_hip_ctx *ctx);
static int crypto_before_pass(lentil_hip_ctx *ctx);
static int crypto_after_pass(lentil_hip_ctx *ctx);
static int crypto_exchange_bands(lentil_hip_ctx *ctx, const int32_t *bands, int32_t lo, int32_t hi);
static int crypto_enqueue_direct(lentil_hip_ctx *ctx, hipStream_t st);

LENTIL_API int lentil_hip_destroy(lentil_hip_ctx *ctx) {
  if (!ctx) return LENTIL_OK;
  (void)hipSetDevice(ctx->device);
  // passes nobody has looked at: the device must be through with them before their buffers go; their verdicts no longer matter
  for (const lentil_hip_ctx::Inflight &in : ctx->inflight) (void)hipEventSynchronize(ctx->slots[in.slot].ev[2]);
  ctx->inflight.clear();
  release_turn(ctx);
  (void)lentil_hip_comm_destroy(ctx);
  (void)hipSetDevice(ctx->device);
  upload_destroy(ctx);
  (void)hipSetDevice(ctx->device);
  crypto_destroy(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_visits(ctx);
  free_bokeh(ctx);
  (void)hipFree(ctx->d_lens);
  (void)hipFree(ctx->d_terms);
  (void)hipFree(ctx->d_blade_sc);
  (void)hipFree(ctx->d_cam_keys);
  (void)hipFree(ctx->d_dummy);
  (void)hipFree(ctx->F.acc);
  (void)hipFree(ctx->F.zkey);
  (void)hipFree(ctx->F.zkey_dbg);
  (void)hipFree(ctx->d_resolved);
  (void)hipFree(ctx->d_work);
  for (auto &ch : ctx->chunks) {
    if (ch.stream) (void)hipStreamSynchronize(ch.stream);
    (void)hipFree(ch.hdr); (void)hipFree(ch.prog);
    for (int i = 0; i < 2; ++i) { (void)hipFree(ch.active[i]); (void)hipFree(ch.tasks[i]); (void)hipFree(ch.pool[i]); }
    (void)hipFree(ch.slow);
    if (ch.scanned) (void)hipEventDestroy(ch.scanned);
    if (ch.done) (void)hipEventDestroy(ch.done);
    if (ch.stream) (void)hipStreamDestroy(ch.stream);
  }
  if (ctx->scans_done) (void)hipEventDestroy(ctx->scans_done);
  if (ctx->pub_stream) { (void)hipStreamSynchronize(ctx->pub_stream); (void)hipStreamDestroy(ctx->pub_stream); }
  if (ctx->pub_done) (void)hipEventDestroy(ctx->pub_done);
  if (ctx->ev_slow) (void)hipEventDestroy(ctx->ev_slow);
  if (ctx->ev_round) (void)hipEventDestroy(ctx->ev_round);
  if (ctx->ev_pre_clear) (void)hipEventDestroy(ctx->ev_pre_clear);
  if (ctx->ev_clear) (void)hipEventDestroy(ctx->ev_clear);
  for (lentil_hip_ctx::Slot &sl : ctx->slots) {
    for (hipEvent_t e : sl.ev) if (e) (void)hipEventDestroy(e);
    if (sl.ev_scan_k[0]) (void)hipEventDestroy(sl.ev_scan_k[0]);
    if (sl.ev_scan_k[1]) (void)hipEventDestroy(sl.ev_scan_k[1]);
    if (sl.ev_tail) (void)hipEventDestroy(sl.ev_tail);
    if (sl.h_ctr_pinned) (void)hipHostFree(sl.h_ctr_pinned);
  }
  if (ctx->ev_acc1) (void)hipEventDestroy(ctx->ev_acc1);
  if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
  if (ctx->ev_res) (void)hipEventDestroy(ctx->ev_res);
  if (ctx->ev_solve) (void)hipEventDestroy(ctx->ev_solve);
  if (ctx->ev_slow1) (void)hipEventDestroy(ctx->ev_slow1);
  if (ctx->slow1_stream) { (void)hipStreamSynchronize(ctx->slow1_stream); (void)hipStreamDestroy(ctx->slow1_stream); }
  if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
  if (ctx->ev_crypto) (void)hipEventDestroy(ctx->ev_crypto);
  (void)hipFree(ctx->d_ctr);
  (void)hipFree(ctx->d_ranges);
  (void)hipFree(ctx->d_ready);
  (void)hipFree(ctx->d_c2w_keys); (void)hipFree(ctx->d_probe_seg); (void)hipFree(ctx->d_probe_idx); (void)hipFree(ctx->d_probe_occ);
  (void)hipFree(ctx->d_probe_count);
  if (ctx->h_probe_seg) (void)hipHostFree(ctx->h_probe_seg);
  if (ctx->h_probe_occ) (void)hipHostFree(ctx->h_probe_occ);
  (void)hipFree(ctx->d_live);
  if (ctx->jit_module) (void)hipModuleUnload(ctx->jit_module);
  (void)hipFree(ctx->d_bm_land); (void)hipFree(ctx->d_bm_box); (void)hipFree(ctx->d_bm_npass);
  (void)hipFree(ctx->d_ext_q);
  (void)hipFree(ctx->d_xor); (void)hipFree(ctx->d_tlc_res); (void)hipFree(ctx->d_tlc_off); (void)hipFree(ctx->d_tlc_tasks);
  (void)hipFree(ctx->d_log);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);       // (ctx->ev[]: one of the slots' sets, destroyed with them above)
  delete ctx;
  return LENTIL_OK;
}

LENTIL_API const char *lentil_hip_last_error(const lentil_hip_ctx *ctx) {
  return ctx ? ctx->err.c_str() : g_err.c_str();
}

LENTIL_API const char *lentil_hip_last_redo_note(const lentil_hip_ctx *ctx) {
  return ctx ? ctx->redo_note.c_str() : "";
}

static uint64_t fnv1a(const void *data, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
  const unsigned char *b = static_cast<const unsigned char *>(data);
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
  return h;
}

LENTIL_API int lentil_hip_set_params(lentil_hip_ctx *ctx, const lentil_params *p) {
  CHECK_CTX(ctx);
  if (!p) return fail(ctx, LENTIL_ERR_INVALID, "params is null");
  if (p->cameraType != LENTIL_THINLENS && p->cameraType != LENTIL_POLYNOMIAL_OPTICS)
    return fail(ctx, LENTIL_ERR_INVALID, "cameraType must be ThinLens or PolynomialOptics");
  if (p->xres == 0 || p->yres == 0) return fail(ctx, LENTIL_ERR_INVALID, "xres/yres must be non-zero");
  if (p->samples_override < 0 || p->samples_override > (1 << 24))
    return fail(ctx, LENTIL_ERR_INVALID, "samples_override out of range");
  if (ctx->have_frame && (p->xres != ctx->P.xres || p->yres != ctx->P.yres))
    return fail(ctx, LENTIL_ERR_INVALID, "xres/yres changed after alloc_frame");
  // (the calibration of the first-batch model holds for the parameters it was traced with)
  if (!ctx->have_params || memcmp(&ctx->P, p, sizeof(lentil_params)) != 0) { ctx->bm_valid = false; ctx->bm_margin16 = 0; ctx->lean_ok = true; }
  ctx->P = *p;
  ctx->have_params = true;
  // polygonal apertures: the corner angles' sin / cos from this host's libm (DevBokeh::blade_sc)
  {
    constexpr int kMaxBlades = 255;
    const int blades = p->bokeh_aperture_blades;
    if (blades >= 2 && blades <= kMaxBlades) {
      if (blades != ctx->blade_count) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (!ctx->d_blade_sc) HIP_TRY(ctx, hipMalloc(&ctx->d_blade_sc, sizeof(double) * 2 * (kMaxBlades + 1)));
        std::vector<double> sc(2 * (size_t)(blades + 1));
        for (int k = 0; k <= blades; ++k) {
          const double ph = (double)(2.0f * 3.14159265358979f / (float)blades * (float)k);     // AI_PI is a float constant
          sc[2 * (size_t)k] = std::sin(ph);
          sc[2 * (size_t)k + 1] = std::cos(ph);
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // (no pass is reading the old table)
        HIP_TRY(ctx, hipMemcpy(ctx->d_blade_sc, sc.data(), sc.size() * sizeof(double), hipMemcpyHostToDevice));
        ctx->blade_count = blades;
      }
    } else {
      ctx->blade_count = 0;
    }
    ctx->bokeh.blade_sc = ctx->blade_count ? ctx->d_blade_sc : nullptr;
    ctx->bokeh.blade_count = ctx->blade_count;
  }
  // lambda powers depend on the params' wavelength
  if (ctx->have_lens) {
    const double lam = (double)p->lambda_bw;
    for (int e = 0; e <= kMaxExp; ++e) ctx->hlens.lambda_pow[e] = host_ipow(lam, e);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lens, &ctx->hlens, sizeof(DevLens), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return LENTIL_OK;
}

static bool pack_terms(const lentil_lens_table *t, const lentil_poly &p, int derive_var, std::vector<DevTerm> &out,
                       uint16_t &first, uint16_t &count) {
  first = (uint16_t)out.size();
  uint32_t c = 0;
  for (uint32_t i = 0; i < p.count; ++i) {
    const lentil_term &s = t->terms[p.first + i];
    uint8_t e[5] = {s.e[0], s.e[1], s.e[2], s.e[3], s.e[4]};
    double coef = s.c;
    if (derive_var >= 0) {
      if (e[derive_var] == 0) continue;
      coef = s.c * (double)e[derive_var];          // derivative: c*e, exponent-1
      e[derive_var] = (uint8_t)(e[derive_var] - 1);
    }
    for (int v = 0; v < 5; ++v) if (e[v] > kMaxExp) return false;
    DevTerm d;
    d.c = coef;
    d.e = (uint32_t)e[0] | ((uint32_t)e[1] << 4) | ((uint32_t)e[2] << 8) | ((uint32_t)e[3] << 12) | ((uint32_t)e[4] << 16);
    d.pad = 0;
    out.push_back(d);
    ++c;
  }
  count = (uint16_t)c;
  return true;
}

// ---- run-time lens specialisation (lentil_lens_jit.h) -----------------------------------------------------------------
// The flags the library itself was built with (kBuildFlags, written into generated/embedded_sources.inc by
// __graft_entry__.build() from its own HIP_FLAGS: same arithmetic, same code generation, same -D set), for the architecture
// of the context's device.
static std::vector<std::string> jit_flags(const std::string &arch) {
  std::vector<std::string> f = {"--offload-arch=" + arch};
  for (const char *b : kBuildFlags) f.push_back(b);
  return f;
}
// the layouts the run-time kernels share with this library, checked inside their translation unit
static std::string jit_layout_checks() {
  char b[512];
  snprintf(b, sizeof b,
           "static_assert(sizeof(DrawArgs) == %zu, \"DrawArgs differs from the library's\");\n"
           "static_assert(sizeof(DevCounters) == %zu, \"DevCounters differs from the library's\");\n"
           "static_assert(sizeof(SlowRec) == %zu && sizeof(Task) == %zu && sizeof(ItemProg) == %zu, \"queue records differ from the library's\");\n",
           sizeof(DrawArgs), sizeof(DevCounters), sizeof(SlowRec), sizeof(Task), sizeof(ItemProg));
  return b;
}
static std::string device_arch(int device) {
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, device) != hipSuccess) { (void)hipGetLastError(); return "gfx950"; }
  std::string a = pr.gcnArchName;
  const size_t c = a.find(':');        // (target features -- sramecc, xnack -- stay the compiler's defaults, as in the library's build)
  if (c != std::string::npos) a.resize(c);
  return a.empty() ? std::string("gfx950") : a;
}
// what a cached code object must have been built from: the kernel sources, the flags, the architecture, the compiler
static uint64_t jit_source_hash(const std::string &arch) {
  static const uint64_t base = [] {
    uint64_t v = lentil_jit::fnv("lentil-jit-2", 12);
    for (const lentil_jit::Source &s : kEmbeddedSources) v = lentil_jit::fnv(s.text, strlen(s.text), v);
    for (const char *f : kBuildFlags) v = lentil_jit::fnv(f, strlen(f), v);
    const std::string chk = jit_layout_checks();
    v = lentil_jit::fnv(chk.data(), chk.size(), v);
    int ver[2] = {0, 0};
    lentil_jit::Rtc &r = lentil_jit::rtc();
    if (r.Version) (void)r.Version(&ver[0], &ver[1]);
    return lentil_jit::fnv(ver, sizeof ver, v);
  }();
  return lentil_jit::fnv(arch.data(), arch.size(), base);
}
// (the library going away -- process exit, dlclose of the plugin: no compilation may still be running inside it)
__attribute__((destructor)) static void lentil_jit_teardown() { lentil_jit::join_all(); }
static bool lens_is_compiled_in(unsigned long long hash) {
#define LENTIL_HASH_MATCH(NAME) if (hash == gen::Lens_##NAME::kTableHash) return true;
  LENTIL_GENERATED_LENSES(LENTIL_HASH_MATCH)
#undef LENTIL_HASH_MATCH
  return false;
}
// the table's entry: looked up, loaded from the cache on disk, or handed to a compiling thread
static void jit_request(lentil_hip_ctx *ctx) {
  if (ctx->jit_module) { (void)hipModuleUnload(ctx->jit_module); ctx->jit_module = nullptr; }
  ctx->jit.reset();
  ctx->jit_loaded = ctx->jit_load_failed = false;
  if (!ctx->jit_enabled || lens_is_compiled_in(ctx->lens_hash)) return;
  // everything that reads the environment or the device happens here, on the caller's thread (the host may call setenv
  // while a compilation runs): the architecture, hiprtc's binding, the cache directory, the debug switch
  const std::string arch = device_arch(ctx->device);
  (void)lentil_jit::rtc();
  const bool debug = getenv("LENTIL_STREAM_DEBUG") != nullptr;
  const uint64_t key = lentil_jit::fnv(arch.data(), arch.size(), ctx->lens_hash);       // (one entry per table and architecture)
  std::lock_guard<std::mutex> lock(lentil_jit::registry_mutex());
  auto &reg = lentil_jit::registry();
  auto it = reg.find(key);
  if (it != reg.end()) { ctx->jit = it->second; return; }
  auto e = std::make_shared<lentil_jit::Entry>();
  reg[key] = e;
  ctx->jit = e;
  const std::string path = lentil_jit::cache_file(lentil_jit::cache_dir(true), ctx->lens_hash, jit_source_hash(arch));
  if (lentil_jit::cache_load(path, e->co)) { e->from_cache = true; e->state.store(lentil_jit::Entry::kReady); return; }
  const std::string src = lentil_jit::lens_jit_emit(ctx->hlens, ctx->h_terms, ctx->lens_hash);
  const std::vector<std::string> flags = jit_flags(arch);
  const std::string checks = jit_layout_checks();
  lentil_jit::Entry *ep = e.get();       // (the registry keeps the entry alive for as long as the library is mapped)
  try {
    e->worker = std::thread([ep, src, path, flags, checks, debug]() {
      bool ok = false;
      try {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<lentil_jit::Source> sources(std::begin(kEmbeddedSources), std::end(kEmbeddedSources));
        ok = lentil_jit::compile(sources, src, flags, ep->co, ep->log, checks);
        ep->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (ok) lentil_jit::cache_store(path, ep->co);
        if (!ok && debug) fprintf(stderr, "[lens jit] compilation failed:\n%s\n", ep->log.c_str());
      } catch (...) {       // (bad_alloc in a string or vector: the interpreter keeps serving the lens)
        ok = false;
        try { ep->log = "run-time compilation threw an exception"; } catch (...) {}
      }
      ep->state.store(ok ? lentil_jit::Entry::kReady : lentil_jit::Entry::kFailed);
    });
  } catch (...) {           // (no thread to be had)
    e->log = "could not start the compilation thread";
    e->state.store(lentil_jit::Entry::kFailed);
  }
}
// the compiled kernels for this context's device, once the code object is there (null: not yet / not at all)
static hipFunction_t jit_function(lentil_hip_ctx *ctx, bool chroma, bool stream) {
  if (!ctx->jit || !ctx->use_generated || ctx->jit_load_failed) return nullptr;
  if (!ctx->jit_loaded) {
    if (ctx->jit->state.load() != lentil_jit::Entry::kReady) return nullptr;
    (void)hipSetDevice(ctx->device);
    bool ok = hipModuleLoadData(&ctx->jit_module, ctx->jit->co.code.data()) == hipSuccess;
    for (int c = 0; c < 2 && ok; ++c)
      for (int s2 = 0; s2 < 2 && ok; ++s2)
        ok = hipModuleGetFunction(&ctx->jit_fn[c][s2], ctx->jit_module, ctx->jit->co.name[c][s2].c_str()) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); ctx->jit_load_failed = true; return nullptr; }
    ctx->jit_loaded = true;
  }
  return ctx->jit_fn[chroma ? 1 : 0][stream ? 1 : 0];
}

LENTIL_API int lentil_hip_set_lens(lentil_hip_ctx *ctx, const lentil_lens_table *t) {
  CHECK_CTX(ctx);
  if (!t || !t->terms) return fail(ctx, LENTIL_ERR_INVALID, "lens table is null");
  for (int i = 0; i < 5; ++i)
    if ((uint64_t)t->out[i].first + t->out[i].count > t->n_terms) return fail(ctx, LENTIL_ERR_INVALID, "lens poly out of range");
  for (int i = 0; i < 4; ++i)
    if ((uint64_t)t->ap[i].first + t->ap[i].count > t->n_terms) return fail(ctx, LENTIL_ERR_INVALID, "lens poly out of range");
  ctx->park_dry_seen = false; ctx->parked_frac = 0.0; ctx->mean_iters = 0.0;      // what the passes learnt about the previous lens
  std::vector<DevTerm> terms;
  DevLens h{};
  bool ok = true;
  for (int i = 0; i < 5; ++i) ok &= pack_terms(t, t->out[i], -1, terms, h.first[P_OUT_X + i], h.count[P_OUT_X + i]);
  for (int i = 0; i < 4; ++i) ok &= pack_terms(t, t->ap[i], -1, terms, h.first[P_AP_X + i], h.count[P_AP_X + i]);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      ok &= pack_terms(t, t->ap[i], 2 + j, terms, h.first[P_DAP_00 + i * 2 + j], h.count[P_DAP_00 + i * 2 + j]);
      ok &= pack_terms(t, t->out[2 + i], j, terms, h.first[P_DOUT_00 + i * 2 + j], h.count[P_DOUT_00 + i * 2 + j]);
    }
  for (int i = 0; i < 2; ++i)       // behind everything the draw kernels use
    for (int j = 0; j < 2; ++j)
      ok &= pack_terms(t, t->ap[i], j, terms, h.first[P_DAPPOS_00 + i * 2 + j], h.count[P_DAPPOS_00 + i * 2 + j]);
  if (!ok) return fail(ctx, LENTIL_ERR_UNSUPPORTED, "lens table exponent > 15");
  if (terms.size() > (size_t)kMaxTerms)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "lens table has too many terms for the LDS staging area");
  h.outer_pupil_radius = t->lens_outer_pupil_radius;
  h.inner_pupil_radius = t->lens_inner_pupil_radius;
  h.length = t->lens_length;
  h.back_focal_length = t->lens_back_focal_length;
  h.outer_pupil_curvature_radius = t->lens_outer_pupil_curvature_radius;
  h.outer_pupil_geometry = t->lens_outer_pupil_geometry;
  ctx->lens_housing_radius = t->lens_aperture_housing_radius;
  h.n_terms = (uint32_t)terms.size();
  {  // FNV-1a over the base table (tools/gen_lens_code.py: table_hash) -> compiled-in specialisation
    unsigned long long hs = 0xCBF29CE484222325ull;
    auto mix = [&](const void *p, size_t n) {
      const unsigned char *b = (const unsigned char *)p;
      for (size_t i = 0; i < n; ++i) hs = (hs ^ b[i]) * 0x100000001B3ull;
    };
    auto poly = [&](const lentil_poly &pl) {
      for (uint32_t i = 0; i < pl.count; ++i) {
        const lentil_term &tt = t->terms[pl.first + i];
        mix(&tt.c, 8);
        mix(tt.e, 5);
      }
      const unsigned char sep = 0xFF;
      mix(&sep, 1);
    };
    for (int i = 0; i < 5; ++i) poly(t->out[i]);
    for (int i = 0; i < 4; ++i) poly(t->ap[i]);
    ctx->lens_hash = hs;
  }
  const double lam = ctx->have_params ? (double)ctx->P.lambda_bw : (double)0.55f;
  for (int e = 0; e <= kMaxExp; ++e) h.lambda_pow[e] = host_ipow(lam, e);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->d_lens) HIP_TRY(ctx, hipMalloc(&ctx->d_lens, sizeof(DevLens)));
  if (!ctx->d_terms) HIP_TRY(ctx, hipMalloc(&ctx->d_terms, sizeof(DevTerm) * kMaxTerms));
  ctx->hlens = h;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lens, &ctx->hlens, sizeof(DevLens), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_terms, terms.data(), sizeof(DevTerm) * terms.size(), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_lens = true;
  {
    // (the first-batch model was traced through one lens: the same table set again -- every frame, by a caller that sets
    // everything up per frame -- keeps it)
    DevLens hz = h;
    for (int e = 0; e <= kMaxExp; ++e) hz.lambda_pow[e] = 0.0;       // (the wavelength belongs to the parameters)
    const uint64_t sig = fnv1a(&hz, sizeof hz, fnv1a(terms.data(), terms.size() * sizeof(DevTerm)));
    const bool changed = sig != ctx->bm_lens_sig;
    if (changed) { ctx->bm_lens_sig = sig; ctx->bm_valid = false; ctx->bm_margin16 = 0; ctx->lean_ok = true; }
    // (the same table set again keeps its specialised kernel too)
    if (changed || (!ctx->jit && ctx->jit_enabled)) { ctx->h_terms = terms; jit_request(ctx); }
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_lens_jit_status(lentil_hip_ctx *ctx, int *state, double *compile_seconds) {
  CHECK_CTX(ctx);
  int st = 0;
  if (ctx->jit) st = ctx->jit->state.load();
  if (st == lentil_jit::Entry::kReady && ctx->jit_load_failed) st = lentil_jit::Entry::kFailed;
  if (state) *state = st;
  if (compile_seconds) *compile_seconds = (ctx->jit && st != lentil_jit::Entry::kCompiling && !ctx->jit->from_cache) ? ctx->jit->seconds : 0.0;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_lens_jit_wait(lentil_hip_ctx *ctx, double timeout_seconds) {
  CHECK_CTX(ctx);
  if (!ctx->jit) return LENTIL_OK;
  const auto t0 = std::chrono::steady_clock::now();
  while (ctx->jit->state.load() == lentil_jit::Entry::kCompiling) {
    if (timeout_seconds > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds)
      return fail(ctx, LENTIL_ERR_INVALID, "the lens kernel is still being compiled");
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
  if (ctx->jit->state.load() == lentil_jit::Entry::kFailed)
    return fail(ctx, LENTIL_ERR_HIP, "run-time compilation of the lens kernel failed (the table interpreter keeps serving the lens): " +
                                         ctx->jit->log.substr(0, 600));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_debug_lens_jit_source(lentil_hip_ctx *ctx, char *buf, uint64_t capacity, uint64_t *length) {
  CHECK_CTX(ctx);
  if (!ctx->have_lens) return fail(ctx, LENTIL_ERR_INVALID, "no lens table set");
  const std::string src = lentil_jit::lens_jit_emit(ctx->hlens, ctx->h_terms.empty() ? std::vector<DevTerm>() : ctx->h_terms, ctx->lens_hash);
  if (length) *length = src.size();
  if (buf && capacity) { const size_t n = src.size() < capacity - 1 ? src.size() : (size_t)capacity - 1; memcpy(buf, src.data(), n); buf[n] = 0; }
  return LENTIL_OK;
}

// Context-free (no GPU needed: hiprtc cross-compiles): the table packed as lentil_hip_set_lens packs it, the lens code emitted,
// and -- compile != 0 -- the four solve kernels compiled.  The emitted source goes to `source` (capacity bytes, NUL-terminated,
// truncated if longer; *source_length its full length), the compiler's log likewise.  LENTIL_OK, LENTIL_ERR_UNSUPPORTED for a
// table the library cannot hold, LENTIL_ERR_HIP when the compilation fails.
LENTIL_API int lentil_hip_debug_lens_jit_compile(const lentil_lens_table *t, int compile, char *source, uint64_t source_capacity,
                                                 uint64_t *source_length, char *log, uint64_t log_capacity, double *seconds,
                                                 uint64_t *code_bytes) {
  if (!t || !t->terms) return LENTIL_ERR_INVALID;
  std::vector<DevTerm> terms;
  DevLens h{};
  bool ok = true;
  for (int i = 0; i < 5; ++i) ok &= pack_terms(t, t->out[i], -1, terms, h.first[P_OUT_X + i], h.count[P_OUT_X + i]);
  for (int i = 0; i < 4; ++i) ok &= pack_terms(t, t->ap[i], -1, terms, h.first[P_AP_X + i], h.count[P_AP_X + i]);
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      ok &= pack_terms(t, t->ap[i], 2 + j, terms, h.first[P_DAP_00 + i * 2 + j], h.count[P_DAP_00 + i * 2 + j]);
      ok &= pack_terms(t, t->out[2 + i], j, terms, h.first[P_DOUT_00 + i * 2 + j], h.count[P_DOUT_00 + i * 2 + j]);
    }
  if (!ok || terms.size() > (size_t)kMaxTerms) return LENTIL_ERR_UNSUPPORTED;
  const std::string src = lentil_jit::lens_jit_emit(h, terms, 0ull);
  auto put = [](const std::string &text, char *buf, uint64_t cap, uint64_t *len) {
    if (len) *len = text.size();
    if (buf && cap) { const size_t n = text.size() < cap - 1 ? text.size() : (size_t)cap - 1; memcpy(buf, text.data(), n); buf[n] = 0; }
  };
  put(src, source, source_capacity, source_length);
  if (seconds) *seconds = 0.0;
  if (code_bytes) *code_bytes = 0;
  if (!compile) return LENTIL_OK;
  lentil_jit::CodeObject co;
  std::string lg;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<lentil_jit::Source> sources(std::begin(kEmbeddedSources), std::end(kEmbeddedSources));
  const bool cok = lentil_jit::compile(sources, src, jit_flags("gfx950"), co, lg, jit_layout_checks());
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  put(lg, log, log_capacity, nullptr);
  if (code_bytes) *code_bytes = co.code.size();
  return cok ? LENTIL_OK : LENTIL_ERR_HIP;
}

LENTIL_API int lentil_hip_set_lens_mode(lentil_hip_ctx *ctx, int mode) {
  CHECK_CTX(ctx);
  if (mode != 0 && mode != 1) return fail(ctx, LENTIL_ERR_INVALID, "lens mode must be 0 (auto) or 1 (tables)");
  ctx->use_generated = (mode == 0);
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_lens_is_compiled(lentil_hip_ctx *ctx) {
  if (!ctx || !ctx->have_lens) return 0;
#define LENTIL_HASH_MATCH(NAME) if (ctx->lens_hash == gen::Lens_##NAME::kTableHash) return 1;
  LENTIL_GENERATED_LENSES(LENTIL_HASH_MATCH)
#undef LENTIL_HASH_MATCH
  return 0;
}

LENTIL_API int lentil_hip_set_bokeh(lentil_hip_ctx *ctx, const lentil_bokeh_table *b) {
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_bokeh(ctx);
  {
    uint64_t sig = 0;
    if (b && b->x > 0 && b->y > 0 && b->x == b->y && b->cdfRow && b->rowIndices && b->cdfColumn && b->columnIndices) {
      const size_t ny_ = (size_t)b->y, nn_ = (size_t)b->x * b->y;
      sig = fnv1a(b->cdfRow, ny_ * 4, fnv1a(b->rowIndices, ny_ * 4, fnv1a(b->cdfColumn, nn_ * 4, fnv1a(b->columnIndices, nn_ * 4)))) | 1ull;
    }
    if (sig != ctx->bm_bokeh_sig) { ctx->bm_bokeh_sig = sig; ctx->bm_valid = false; ctx->bm_margin16 = 0; ctx->lean_ok = true; }
  }
  if (!b) return LENTIL_OK;
  if (b->x <= 0 || b->y <= 0 || b->x != b->y || !b->cdfRow || !b->rowIndices || !b->cdfColumn || !b->columnIndices)
    return fail(ctx, LENTIL_ERR_INVALID, "bokeh table must be square with all four arrays");
  const size_t ny = (size_t)b->y, nn = (size_t)b->x * b->y;
  float *cr = nullptr, *cc = nullptr;
  int32_t *ri = nullptr, *ci = nullptr;
  HIP_TRY(ctx, hipMalloc(&cr, ny * 4));
  HIP_TRY(ctx, hipMalloc(&ri, ny * 4));
  HIP_TRY(ctx, hipMalloc(&cc, nn * 4));
  HIP_TRY(ctx, hipMalloc(&ci, nn * 4));
  HIP_TRY(ctx, hipMemcpy(cr, b->cdfRow, ny * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ri, b->rowIndices, ny * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(cc, b->cdfColumn, nn * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ci, b->columnIndices, nn * 4, hipMemcpyHostToDevice));
  ctx->bokeh.x = b->x; ctx->bokeh.y = b->y;
  ctx->bokeh.cdfRow = cr; ctx->bokeh.rowIndices = ri; ctx->bokeh.cdfColumn = cc; ctx->bokeh.columnIndices = ci;
  ctx->have_bokeh = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_set_camera_motion(lentil_hip_ctx *ctx, uint32_t n_keys, const float *w2c) {
  CHECK_CTX(ctx);
  if (n_keys > LENTIL_MAX_MOTION_KEYS) return fail(ctx, LENTIL_ERR_INVALID, "more than LENTIL_MAX_MOTION_KEYS camera matrices");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));        // (no pass is reading the old keys)
  if (n_keys < 2 || !w2c) {
    ctx->n_cam_keys = 0;
  } else {
    if (!ctx->d_cam_keys) HIP_TRY(ctx, hipMalloc(&ctx->d_cam_keys, sizeof(float) * 16 * LENTIL_MAX_MOTION_KEYS));
    HIP_TRY(ctx, hipMemcpy(ctx->d_cam_keys, w2c, sizeof(float) * 16 * n_keys, hipMemcpyHostToDevice));
    ctx->n_cam_keys = n_keys;
    ctx->h_cam_keys.assign(w2c, w2c + (size_t)16 * n_keys);       // (the occlusion probe's camera-to-world keys are their inverses)
  }
  apply_camera_motion(ctx);
  return LENTIL_OK;
}

// ---- occlusion probes (include/lentil_hip.h; kernels: probe_list_kernel / probe_apply_kernel) ---------------------------------
// the inverse of a 4x4 matrix by cofactors in fp64, rounded to float at the end: AiCameraToWorldMatrix where the caller gives none
// (the oracle computes the same: the test oracle's invert4x4)
static void invert4x4(const float m_[16], float out[16]) {
  double m[16], inv[16];
  for (int i = 0; i < 16; i++) m[i] = m_[i];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  det = det != 0.0 ? 1.0 / det : 0.0;
  for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] * det);
}

LENTIL_API int lentil_hip_set_occlusion_probe(lentil_hip_ctx *ctx, lentil_probe_fn fn, void *user, const float *camera_to_world) {
  CHECK_CTX(ctx);
  ctx->probe_fn = fn;
  ctx->probe_user = user;
  ctx->probe_c2w_given = fn != nullptr && camera_to_world != nullptr;
  if (ctx->probe_c2w_given) memcpy(ctx->probe_c2w, camera_to_world, sizeof ctx->probe_c2w);
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_probe_stats(lentil_hip_ctx *ctx, uint64_t stats[3]) {
  CHECK_CTX(ctx);
  if (!stats) return fail(ctx, LENTIL_ERR_INVALID, "stats is null");
  stats[0] = ctx->n_probes; stats[1] = ctx->n_probes_occluded; stats[2] = ctx->n_probe_calls;
  return LENTIL_OK;
}

// One round's probes, between its solves (and stragglers) and its accept, on the round's stream: list, callback, apply.  The
// host waits twice -- for the list and, inside the callback, for the renderer.
static int probe_round(lentil_hip_ctx *ctx, const DrawArgs &da, hipStream_t st) {
  if (!ctx->probe_fn) return LENTIL_OK;
  ProbeArgs pr{};
  if (ctx->probe_c2w_given) memcpy(pr.c2w, ctx->probe_c2w, sizeof pr.c2w);
  else invert4x4(&ctx->P.world_to_camera[0][0], &pr.c2w[0][0]);
  if (ctx->n_cam_keys >= 2 && ctx->h_cam_keys.size() == (size_t)16 * ctx->n_cam_keys) {
    float inv[16 * LENTIL_MAX_MOTION_KEYS];
    for (uint32_t k = 0; k < ctx->n_cam_keys; ++k) invert4x4(&ctx->h_cam_keys[(size_t)16 * k], inv + 16 * k);
    if (!ctx->d_c2w_keys) HIP_TRY(ctx, hipMalloc(&ctx->d_c2w_keys, sizeof(float) * 16 * LENTIL_MAX_MOTION_KEYS));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_c2w_keys, inv, sizeof(float) * 16 * ctx->n_cam_keys, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));       // (`inv` is on this function's stack)
    pr.c2w_keys = ctx->d_c2w_keys;
  }
  if (!ctx->d_probe_count) HIP_TRY(ctx, hipMalloc(&ctx->d_probe_count, sizeof(unsigned int)));
  const dim3 grid((unsigned)ctx->num_cu * 4u), block(256);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (ctx->probe_cap == 0) {
      const uint64_t cap = 1ull << 20;
      HIP_TRY(ctx, hipMalloc(&ctx->d_probe_seg, cap * sizeof(lentil_probe_segment)));
      HIP_TRY(ctx, hipMalloc(&ctx->d_probe_idx, cap * sizeof(uint32_t)));
      HIP_TRY(ctx, hipMalloc(&ctx->d_probe_occ, cap));
      HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_probe_seg, cap * sizeof(lentil_probe_segment), hipHostMallocDefault));
      HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_probe_occ, cap, hipHostMallocDefault));
      ctx->probe_cap = cap;
    }
    pr.seg = ctx->d_probe_seg; pr.idx = ctx->d_probe_idx; pr.cap = (uint32_t)(ctx->probe_cap < 0xFFFFFFF0ull ? ctx->probe_cap : 0xFFFFFFF0ull);
    pr.count = ctx->d_probe_count; pr.occluded = ctx->d_probe_occ;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_probe_count, 0, sizeof(unsigned int), st));
    hipLaunchKernelGGL(probe_list_kernel, grid, block, 0, st, da, pr);
    HIP_TRY(ctx, hipGetLastError());
    unsigned int n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, ctx->d_probe_count, sizeof n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if ((uint64_t)n > ctx->probe_cap) {
      // the list did not fit: buffers for what the kernel counted, and once more
      if (attempt) return fail(ctx, LENTIL_ERR_NOMEM, "occlusion probes: the list outgrew its buffers twice");
      (void)hipFree(ctx->d_probe_seg); (void)hipFree(ctx->d_probe_idx); (void)hipFree(ctx->d_probe_occ);
      (void)hipHostFree(ctx->h_probe_seg); (void)hipHostFree(ctx->h_probe_occ);
      ctx->d_probe_seg = nullptr; ctx->d_probe_idx = nullptr; ctx->d_probe_occ = nullptr; ctx->h_probe_seg = nullptr; ctx->h_probe_occ = nullptr;
      const uint64_t cap = (uint64_t)n + (uint64_t)n / 4 + 4096;
      HIP_TRY(ctx, hipMalloc(&ctx->d_probe_seg, cap * sizeof(lentil_probe_segment)));
      HIP_TRY(ctx, hipMalloc(&ctx->d_probe_idx, cap * sizeof(uint32_t)));
      HIP_TRY(ctx, hipMalloc(&ctx->d_probe_occ, cap));
      HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_probe_seg, cap * sizeof(lentil_probe_segment), hipHostMallocDefault));
      HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_probe_occ, cap, hipHostMallocDefault));
      ctx->probe_cap = cap;
      continue;
    }
    if (n == 0) return LENTIL_OK;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_probe_seg, ctx->d_probe_seg, (size_t)n * sizeof(lentil_probe_segment), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memset(ctx->h_probe_occ, 0, n);
    ctx->probe_fn(ctx->probe_user, n, ctx->h_probe_seg, ctx->h_probe_occ);
    uint64_t occ = 0;
    for (unsigned int i = 0; i < n; ++i) occ += ctx->h_probe_occ[i] ? 1u : 0u;
    ctx->n_probes += n; ctx->n_probes_occluded += occ; ++ctx->n_probe_calls;
    if (occ) {
      HIP_TRY(ctx, hipMemcpyAsync(ctx->d_probe_occ, ctx->h_probe_occ, n, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(probe_apply_kernel, grid, block, 0, st, da, pr, (uint32_t)n);
      HIP_TRY(ctx, hipGetLastError());
    }
    return LENTIL_OK;
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_set_camera_shutter(lentil_hip_ctx *ctx, float shutter_start, float shutter_end) {
  CHECK_CTX(ctx);
  if (!(shutter_end > shutter_start)) return fail(ctx, LENTIL_ERR_INVALID, "set_camera_shutter: shutter_end must lie after shutter_start");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));        // (no pass is reading the old range)
  ctx->shutter_t0 = shutter_start;
  ctx->shutter_inv_dt = 1.0f / (shutter_end - shutter_start);       // fp32, as the oracle forms it (orc_frame_set_camera_shutter)
  apply_camera_motion(ctx);
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_alloc_frame(lentil_hip_ctx *ctx, uint32_t n_aovs, const uint8_t *kind) {
  CHECK_CTX(ctx);
  ctx->resolved_valid = false;         // (what an early resolve left in d_resolved no longer describes the frame)
  if (!ctx->have_params) return fail(ctx, LENTIL_ERR_INVALID, "set_params must precede alloc_frame");
  if (n_aovs < 1 || n_aovs > LENTIL_MAX_AOVS) return fail(ctx, LENTIL_ERR_INVALID, "n_aovs out of range");
  for (uint32_t i = 0; i < n_aovs; ++i) {
    const uint8_t k = kind ? kind[i] : LENTIL_FILTER_GAUSSIAN;
    if (k != LENTIL_FILTER_GAUSSIAN && k != LENTIL_FILTER_CLOSEST && k != LENTIL_FILTER_CLOSEST_DEBUG)
      return fail(ctx, LENTIL_ERR_UNSUPPORTED, "variance-original AOVs are a no-op in the reference (src/lentil.h:848-850)");
    if (i == 0 && k != LENTIL_FILTER_GAUSSIAN)
      return fail(ctx, LENTIL_ERR_UNSUPPORTED, "AOV 0 (RGBA, the weight-buffer AOV) must be gaussian-filtered");
    ctx->kind[i] = k;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  crypto_destroy(ctx);          // tables are per pixel of the frame: lentil_hip_alloc_crypto follows alloc_frame
  (void)hipFree(ctx->F.acc);
  (void)hipFree(ctx->d_dir);
  ctx->d_dir = nullptr;
  ctx->dir_dirty = false;
  (void)hipFree(ctx->d_touched);
  ctx->d_touched = nullptr;
  (void)hipFree(ctx->F.zkey);
  (void)hipFree(ctx->F.zkey_dbg);
  (void)hipFree(ctx->d_resolved);
  ctx->F = FrameDev{};
  ctx->d_resolved = nullptr;
  ctx->have_frame = false;
  const uint64_t np = (uint64_t)ctx->P.xres * ctx->P.yres;
  if (np >= (uint64_t)kCodePendingBase) return fail(ctx, LENTIL_ERR_UNSUPPORTED, "frames of 0xFE000000 pixels or more");
  const uint32_t rec = ((4u * n_aovs + 1u) + 7u) & ~7u;      // floats per pixel record
  const uint64_t nfl = np * rec;
  HIP_TRY(ctx, hipMalloc(&ctx->F.acc, nfl * sizeof(float)));
  HIP_TRY(ctx, hipMalloc(&ctx->d_resolved, np * 4 * n_aovs * sizeof(float)));
  ctx->F.stride = rec;
  ctx->F.n_aovs = n_aovs;
  ctx->F.np = np;
  ctx->F.closest_mask = 0;
  for (uint32_t i = 0; i < n_aovs; ++i) {
    if (ctx->kind[i] == LENTIL_FILTER_CLOSEST) ctx->F.closest_mask |= 1u << i;
    if (ctx->kind[i] == LENTIL_FILTER_CLOSEST_DEBUG) { ctx->F.closest_mask |= 1u << i; ctx->F.debug_mask |= 1u << i; }
  }
  if (ctx->F.closest_mask & ~ctx->F.debug_mask) {
    HIP_TRY(ctx, hipMalloc(&ctx->F.zkey, np * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->F.zkey, 0xFF, np * sizeof(unsigned long long), ctx->stream));
  }
  if (ctx->F.debug_mask) {
    HIP_TRY(ctx, hipMalloc(&ctx->F.zkey_dbg, np * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->F.zkey_dbg, 0xFF, np * sizeof(unsigned long long), ctx->stream));
  }
  ctx->have_frame = true;
  HIP_TRY(ctx, hipMemsetAsync(ctx->F.acc, 0, nfl * sizeof(float), ctx->stream));
  ctx->dirty_lo = ctx->dirty_hi = 0; ctx->dirty_known = true; ctx->pass_pending = false;
  ctx->cleared_since_pass = true;
  return LENTIL_OK;
}

static int check_visits(lentil_hip_ctx *ctx, const lentil_visits *v) {
  if (!v) return fail(ctx, LENTIL_ERR_INVALID, "visits is null");
  if (v->n > 0xFFFFFFFFull) return fail(ctx, LENTIL_ERR_UNSUPPORTED, "more than 2^32 visits per call");
  if (v->n_extra > LENTIL_MAX_AOVS - 1) return fail(ctx, LENTIL_ERR_INVALID, "too many extra AOV columns");
  if (v->n && (!v->rgba || !v->pos_z || !v->raydir_time || !v->volume_ignore || !v->transmission))
    return fail(ctx, LENTIL_ERR_INVALID, "a visit column is null");
  if (v->visits_per_pixel == 0 && v->n && !v->pixel)
    return fail(ctx, LENTIL_ERR_INVALID, "visits_per_pixel == 0 needs the per-visit pixel array");
  if (v->visits_per_pixel && (v->pixels_per_row == 0 || v->pixel_row_stride == 0))
    return fail(ctx, LENTIL_ERR_INVALID, "pixels_per_row / pixel_row_stride must be non-zero");
  if (v->visits_per_pixel && v->pixel_row_stride > 1 && v->pixel_y0 < 0)
    return fail(ctx, LENTIL_ERR_INVALID, "pixel_y0 < 0 with a row-interleaved partition");
  if (v->visits_per_pixel && v->pixel_y0 >= 0) {
    // frame-wide visit ids (closest AOV tie-break) are 32 bits: (last row + 1) * visits per row
    const uint64_t row_visits = (uint64_t)v->pixels_per_row * v->visits_per_pixel;
    const uint64_t rows = (v->n + row_visits - 1) / row_visits;
    if (((uint64_t)v->pixel_y0 + rows * v->pixel_row_stride) * row_visits > 0xFFFFFFFFull)
      return fail(ctx, LENTIL_ERR_UNSUPPORTED, "frame-wide visit ids exceed 32 bits");
  } else if ((uint64_t)ctx->visit_id_base + v->n > 0x100000000ull) {
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "visit_id_base + n exceeds 32 bits");
  }
  return LENTIL_OK;
}

static void apply_camera_motion(lentil_hip_ctx *ctx) {
  ctx->V.cam.keys = ctx->n_cam_keys >= 2 ? ctx->d_cam_keys : nullptr;
  ctx->V.cam.n = ctx->n_cam_keys >= 2 ? ctx->n_cam_keys : 0u;
  ctx->V.cam.t0 = ctx->shutter_t0;
  ctx->V.cam.inv_dt = ctx->shutter_inv_dt;
}

static void to_dev(VisitsDev &d, const lentil_visits *v) {
  d.n = v->n;
  d.visits_per_pixel = v->visits_per_pixel;
  d.pixels_per_row = v->pixels_per_row;
  d.pixel_x0 = v->pixel_x0;
  d.pixel_y0 = v->pixel_y0;
  d.pixel_row_stride = v->pixel_row_stride;
  d.n_extra = v->n_extra;
  d.rgba = (const float4 *)v->rgba;
  d.pos_z = (const float4 *)v->pos_z;
  d.raydir_time = (const float4 *)v->raydir_time;
  d.volume_ignore = (const float4 *)v->volume_ignore;
  d.transmission = (const float4 *)v->transmission;
  for (int k = 0; k < LENTIL_MAX_AOVS - 1; ++k) d.extra[k] = (const float4 *)v->extra[k];
  d.pixel = v->pixel;
  d.inv_density = v->inv_density;
  d.id_base = 0;
  d.cam = CamMotion{nullptr, 0u, 0.0f, 1.0f};
}

static int ensure_worklist(lentil_hip_ctx *ctx, uint64_t n) {
  if (n <= ctx->work_cap && ctx->d_work) return LENTIL_OK;
  (void)hipFree(ctx->d_work);
  ctx->d_work = nullptr;
  ctx->work_cap = 0;
  const uint64_t cap = n < 1024 ? 1024 : n;
  HIP_TRY(ctx, hipMalloc(&ctx->d_work, cap * sizeof(uint2)));
  ctx->work_cap = cap;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_bind_visits(lentil_hip_ctx *ctx, const lentil_visits *v) {
  CHECK_CTX_PIPELINED(ctx);
  int rc = check_visits(ctx, v);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // nothing of the previous stream is freed and the work list is large enough: the next pass simply follows the last
  // one on the stream (the caller keeps the columns it bound alive until the pass that reads them is done, as ever)
  const bool frees = !ctx->owned_visit_mem.empty() || ctx->upload != nullptr ||
                     !(v->n <= ctx->work_cap && ctx->d_work);
  // (a pass whose end is still open keeps the stream it ran on -- Inflight::V --; memory of that stream about to be freed: it is looked at first)
  if (frees && !ctx->inflight.empty() && (rc = settle(ctx))) return rc;
  if (frees) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_visits(ctx);
  upload_release(ctx, true);
  to_dev(ctx->V, v);
  apply_camera_motion(ctx);
  ctx->V.id_base = ctx->visit_id_base;
  rc = ensure_worklist(ctx, v->n);
  if (rc) return rc;
  ctx->have_visits = true;
  ++ctx->visits_gen;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_upload_visits(lentil_hip_ctx *ctx, const lentil_visits *v) {
  CHECK_CTX(ctx);
  int rc = check_visits(ctx, v);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_visits(ctx);
  upload_release(ctx, true);
  lentil_visits d = *v;
  auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
    *dst = nullptr;
    if (!src || !bytes) return LENTIL_OK;
    void *p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    ctx->owned_visit_mem.push_back(p);
    HIP_TRY(ctx, hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = p;
    return LENTIL_OK;
  };
  const size_t col = (size_t)v->n * 16;
  if ((rc = up(v->rgba, col, (const void **)&d.rgba))) return rc;
  if ((rc = up(v->pos_z, col, (const void **)&d.pos_z))) return rc;
  if ((rc = up(v->raydir_time, col, (const void **)&d.raydir_time))) return rc;
  if ((rc = up(v->volume_ignore, col, (const void **)&d.volume_ignore))) return rc;
  if ((rc = up(v->transmission, col, (const void **)&d.transmission))) return rc;
  for (uint32_t k = 0; k < v->n_extra; ++k)
    if ((rc = up(v->extra[k], col, (const void **)&d.extra[k]))) return rc;
  if ((rc = up(v->pixel, (size_t)v->n * 4, (const void **)&d.pixel))) return rc;
  if ((rc = up(v->inv_density, (size_t)v->n * 4, (const void **)&d.inv_density))) return rc;
  to_dev(ctx->V, &d);
  apply_camera_motion(ctx);
  ctx->V.id_base = ctx->visit_id_base;
  rc = ensure_worklist(ctx, v->n);
  if (rc) return rc;
  ctx->have_visits = true;
  ++ctx->visits_gen;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_clear_frame(lentil_hip_ctx *ctx) {
  CHECK_CTX_PIPELINED(ctx);
  // a pass nobody has looked at: its frame goes -- abandoned (its counters are still read, what it lacked is no longer done)
  if (!ctx->inflight.empty()) { ctx->inflight.back().abandoned = true; harvest_ready(ctx); }
  if (!ctx->trace.marks.empty()) { ht_mark(ctx, "next_clear"); ht_dump(ctx); }
  ht_mark(ctx, "clear{");
  ctx->resolved_valid = false;         // (what an early resolve left in d_resolved no longer describes the frame)
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // rows known to hold everything added since the last clear (lentil_hip_touched_rows was asked after the
  // pass): wipe only those; otherwise the whole frame
  if (ctx->F.touched) {
    // only splats have been added since the last clear, and their records are flagged: wipe those
    const uint64_t n_groups = (ctx->F.np + 63) / 64;
    uint64_t blocks = (n_groups + 255) / 256;
    const uint64_t max_blocks = (uint64_t)ctx->num_cu * 4;
    if (blocks > max_blocks) blocks = max_blocks;
    hipStream_t cs = ctx->stream;
    const bool off = ctx->clear_offstream && ctx->streams_concurrent && ctx->stream_mode && !ctx->chunks.empty() && ctx->chunks[0].stream && !ctx->clear_pending;
    if (off) {
      // behind everything the main stream holds so far (the last pass's tail is on this very stream already), beside what it gets next
      cs = ctx->chunks[0].stream;
      HIP_TRY(ctx, hipEventRecord(ctx->ev_pre_clear, ctx->stream));
      HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->ev_pre_clear, 0));
    }
    hipLaunchKernelGGL(clear_touched_kernel, dim3((unsigned)blocks), dim3(256), 0, cs, ctx->F, n_groups);
    HIP_TRY(ctx, hipGetLastError());
    if (off) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev_clear, cs));
      ctx->clear_pending = true;
    }
    ctx->F.touched = nullptr;
    ctx->F.touched_px = nullptr;
  } else {
  { const int rcj = join_clear(ctx); if (rcj) return rcj; }
  uint64_t p0 = 0, p1 = ctx->F.np;
  if (ctx->dirty_known && !ctx->pass_pending) {
    p0 = (uint64_t)(ctx->dirty_lo < 0 ? 0 : ctx->dirty_lo) * ctx->P.xres;
    p1 = (uint64_t)(ctx->dirty_hi < ctx->dirty_lo ? ctx->dirty_lo : ctx->dirty_hi) * ctx->P.xres;
    if (p1 > ctx->F.np) p1 = ctx->F.np;
    if (p0 > p1) p0 = p1;
  }
  if (p1 > p0) {
    HIP_TRY(ctx, hipMemsetAsync(ctx->F.acc + p0 * ctx->F.stride, 0, (p1 - p0) * ctx->F.stride * sizeof(float), ctx->stream));
    if (ctx->F.zkey) HIP_TRY(ctx, hipMemsetAsync(ctx->F.zkey + p0, 0xFF, (p1 - p0) * sizeof(unsigned long long), ctx->stream));
    if (ctx->F.zkey_dbg) HIP_TRY(ctx, hipMemsetAsync(ctx->F.zkey_dbg + p0, 0xFF, (p1 - p0) * sizeof(unsigned long long), ctx->stream));
  }
  }
  { const int rc = crypto_clear(ctx); if (rc) return rc; }
  ctx->F.dir = nullptr;         // what the scan stored there no longer counts (wiped or overwritten before it does again)
  ctx->cleared_since_pass = true;
  ctx->dirty_lo = ctx->dirty_hi = 0;
  ctx->dirty_known = true;      // clean frame: nothing is dirty
  ctx->pass_pending = false;
  ht_mark(ctx, "}clear");
  return LENTIL_OK;
}

template <typename T>
static int grow(lentil_hip_ctx *ctx, T **p, uint64_t need) {
  (void)hipFree(*p);
  *p = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)p, need * sizeof(T)));
  return LENTIL_OK;
}

template <bool kStream>
static void launch_solve_po(lentil_hip_ctx *ctx, const DrawArgs &da, hipStream_t st, unsigned blocks, unsigned threads = 256) {
  const bool chroma = da.n_channels == 3;       // chromatic aberration: three wavelength channels per attempt
  bool launched = false;
#define LENTIL_LAUNCH_GEN(NAME)                                                                          \
  if (!launched && ctx->use_generated && ctx->lens_hash == gen::Lens_##NAME::kTableHash) {                \
    if (chroma) hipLaunchKernelGGL((solve_po_kernel<GenLens<gen::Lens_##NAME>, false, true, kStream>), dim3(blocks), dim3(threads), 0, st, da); \
    else hipLaunchKernelGGL((solve_po_kernel<GenLens<gen::Lens_##NAME>, false, false, kStream>), dim3(blocks), dim3(threads), 0, st, da); \
    launched = true;                                                                                     \
  }
  LENTIL_GENERATED_LENSES(LENTIL_LAUNCH_GEN)
#undef LENTIL_LAUNCH_GEN
  if (!launched) {
    // a kernel specialised for this table at run time (lentil_lens_jit.h), once its code object is there
    if (hipFunction_t fn = jit_function(ctx, chroma, kStream)) {
      DrawArgs args = da;
      void *params[] = {&args};
      launched = hipModuleLaunchKernel(fn, blocks, 1, 1, threads, 1, 1, 0, st, params, nullptr) == hipSuccess;
    }
  }
  if (!launched) {
    if (chroma) hipLaunchKernelGGL((solve_po_kernel<LdsLens, true, true, kStream>), dim3(blocks), dim3(threads), 0, st, da);
    else hipLaunchKernelGGL((solve_po_kernel<LdsLens, true, false, kStream>), dim3(blocks), dim3(threads), 0, st, da);
  }
}

// stragglers parked by the solve kernel: one wave each (blind launch; an empty queue costs a few microseconds)
// (not in rounds that do not park: its 41 KB blocks would queue behind another chunk's chip-filling solve kernel
// just to find nothing to do, and hold up this chunk's accept meanwhile)
static void launch_slow(lentil_hip_ctx *ctx, const DrawArgs &da, hipStream_t st) {
  if (da.P.cameraType == LENTIL_POLYNOMIAL_OPTICS && da.slow && da.round >= da.slow_from_round)
    hipLaunchKernelGGL(solve_slow_kernel, dim3((unsigned)ctx->num_cu * 4), dim3(64), coop_lds_bytes(ctx->hlens.n_terms), st, da);
}

static void launch_solve(lentil_hip_ctx *ctx, const DrawArgs &da, hipStream_t st, unsigned blocks) {
  if (da.P.cameraType == LENTIL_POLYNOMIAL_OPTICS) launch_solve_po<false>(ctx, da, st, blocks);
  else hipLaunchKernelGGL(solve_thinlens_kernel, dim3(blocks), dim3(256), 0, st, da);
  launch_slow(ctx, da, st);
}

// Size one chunk's draw-pipeline buffers from its scan result and enqueue prep + `blind_rounds`
// solve/accept rounds on the chunk's stream (no host round trip between the rounds: every kernel
// reads its queue lengths from device memory; a round with empty queues costs a few microseconds).
static int size_chunk_buffers(lentil_hip_ctx *ctx, lentil_hip_ctx::Chunk &ch, uint64_t n_items, uint64_t units) {
  // (+ room for a streamed pass's end markers: one per first-round solve wave)
  const uint64_t tasks = units / 64 + 2 * n_items + 64 + (uint64_t)ctx->num_cu * 64;
  if (tasks > 0xFFFFFFF0ull) return fail(ctx, LENTIL_ERR_NOMEM, "too many solve tasks in one batch");
  int rc;
  if (n_items > ch.item_cap) {
    const uint64_t nc = n_items + n_items / 4 + 1024;
    if ((rc = grow(ctx, &ch.hdr, nc))) return rc;
    if ((rc = grow(ctx, &ch.prog, nc))) return rc;
    if ((rc = grow(ctx, &ch.active[0], nc))) return rc;
    if ((rc = grow(ctx, &ch.active[1], nc))) return rc;
    ch.item_cap = nc;
  }
  if (tasks > ch.task_cap) {
    const uint64_t nc = tasks + tasks / 4;
    if ((rc = grow(ctx, &ch.tasks[0], nc))) return rc;
    if ((rc = grow(ctx, &ch.tasks[1], nc))) return rc;
    ch.task_cap = nc;
    // a streamed pass tells a filled slot by its tag: no stale bits (now: the chunk's own stream may fill it next)
    HIP_TRY(ctx, hipMemsetAsync(ch.tasks[0], 0, nc * sizeof(Task), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ch.tasks[1], 0, nc * sizeof(Task), ctx->stream));      // (second round beside the first accept)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (units > ch.pool_cap) {
    const uint64_t nc = units + units / 4;
    if ((rc = grow(ctx, &ch.pool[0], nc))) return rc;
    if ((rc = grow(ctx, &ch.pool[1], nc))) return rc;
    ch.pool_cap = nc;
  }
  // stragglers are about 0.1 % of the solves; a full queue only means they stay in their lanes
  // (only chunks below slow_below draws park at all, which bounds the queue at a few tens of MB)
  const uint64_t park_units = units < 8 * ctx->slow_below ? units : 8 * ctx->slow_below;
  const uint64_t slow_need = park_units / 128 + 4096 + (uint64_t)ctx->num_cu * 4;     // (+ a live queue's end markers)
  if (ctx->slow_at > 0 && slow_need > ch.slow_cap) {
    const uint64_t nc = slow_need + slow_need / 4;
    if ((rc = grow(ctx, &ch.slow, nc))) return rc;
    ch.slow_cap = nc;
    // a live queue tells a filled slot by its tag: no stale ones from whoever had this memory before
    HIP_TRY(ctx, hipMemsetAsync(ch.slow, 0, nc * sizeof(SlowRec), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return LENTIL_OK;
}

static void bind_chunk_buffers(const lentil_hip_ctx::Chunk &ch, DrawArgs &da) {
  da.hdr = ch.hdr;
  da.prog = ch.prog;
  for (int i = 0; i < 2; ++i) { da.tasks[i] = ch.tasks[i]; da.active[i] = ch.active[i]; da.pool[i] = ch.pool[i]; }
  da.task_cap = (uint32_t)(ch.task_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : ch.task_cap);
  da.pool_cap = ch.pool_cap < 0xFFFFFFFFull ? ch.pool_cap : 0xFFFFFFFFull;
  da.slow = ch.slow;
  // (the last num_cu * 4 records are kept for a live queue's end markers)
  const uint64_t usable = ch.slow_cap > (uint64_t)4096 ? ch.slow_cap - 1024 : 0;
  da.slow_cap = (uint32_t)(usable < 0x00FFFFF0ull ? usable : 0x00FFFFF0ull);      // (a pending mark names its slot in 24 bits)
  if (!usable) da.slow = nullptr;
}

// rounds with a host check after each (used when the blind rounds did not finish a chunk, and for
// the sub-batches of very large chunks); returns the number of rounds run from `first_round` on
static int finish_rounds(lentil_hip_ctx *ctx, int ci, DrawArgs &da, int first_round, int *rounds_out) {
  lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
  int round = first_round;
  for (; round < 64; ++round) {
    unsigned int n_act = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n_act, (char *)(ctx->d_ctr + ci) + offsetof(DevCounters, n_active) +
                                            sizeof(unsigned int) * (round & 1),
                                sizeof(unsigned int), hipMemcpyDeviceToHost, ch.stream));
    HIP_TRY(ctx, hipStreamSynchronize(ch.stream));
    if (n_act == 0) break;
    da.parity = round & 1; da.round = round;
    launch_solve(ctx, da, ch.stream, 256);
    { const int rcp = probe_round(ctx, da, ch.stream); if (rcp) return rcp; }
    hipLaunchKernelGGL(accept_kernel<0>, dim3(n_act < 512u ? n_act : 512u), dim3(256), 0, ch.stream, da);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipStreamSynchronize(ch.stream));
  if (rounds_out) *rounds_out = round;
  return LENTIL_OK;
}

// prep + `blind_rounds` solve/accept rounds of one chunk on the chunk's stream, no host round trip in between:
// every kernel reads its queue lengths from device memory (a round with empty queues costs a few microseconds).
// n_items / sum_samples size the grids only (exact after a scan read-back, last pass's figures in blind mode).
static int launch_chunk_rounds(lentil_hip_ctx *ctx, int ci, DrawArgs &da, uint64_t n_items, uint64_t sum_samples,
                               uint64_t prep_threads, int blind_rounds) {
  lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
  const uint64_t nch = (uint64_t)da.n_channels;
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * (uint64_t)ctx->solve_max_blocks;
  da.parity = 0;
  // Parking is for chunks with few draws (see DrawArgs::slow_below).  Decided here, from the same figure that sizes
  // the grids, and handed to the kernels as "no queue": a solve kernel never parks what no slow kernel will pick up.
  if (sum_samples >= ctx->slow_below) da.slow = nullptr;
  hipLaunchKernelGGL(prep_items_kernel, dim3((unsigned)((prep_threads + 255) / 256)), dim3(256), 0, ch.stream, da);
  HIP_TRY(ctx, hipGetLastError());
  // enough persistent waves for the chunk's first batch, at most 4 blocks per CU
  const uint64_t want = (nch * (sum_samples / 64 + n_items) + 3) / 4;
  unsigned solve_blocks = (unsigned)(want < 1 ? 1 : (want > max_blocks ? max_blocks : want));
  if (ci + 1 < ctx->n_chunks && sum_samples < ctx->early_cap_samples) {
    // Later chunks are still being scanned and this chunk's solves are a fraction of a scan's worth of
    // work: few blocks per CU leave the register file to the scan's waves, which otherwise wait for the
    // persistent solve blocks to drain.
    const unsigned cap = (unsigned)ctx->num_cu * (unsigned)ctx->solve_cap_blocks;
    if (solve_blocks > cap) solve_blocks = cap;
  }
  const uint64_t acc_want = n_items < 1 ? 1 : n_items;
  // one block per item while the items fit the chip at once (85 VGPRs: five blocks per CU)
  const uint64_t acc_max = (uint64_t)ctx->num_cu * (uint64_t)ctx->accept_max_blocks;
  const unsigned accept_blocks = (unsigned)(acc_want > acc_max ? acc_max : acc_want);
  const bool heavy = sum_samples >= ctx->early_cap_samples;
  for (int round = 0; round < blind_rounds; ++round) {
    da.parity = round & 1; da.round = round;
    // Two first-round solve kernels that each fill the chip only get in each other's way (measured: 109 ms side by
    // side against 104 ms one after the other): a heavy chunk's first round waits for the previous heavy chunk's.
    if (round == 0 && heavy && ci > 0 && ctx->chunks[ci - 1].heavy_pending)
      HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->chunks[ci - 1].done, 0));
    // ... and for the scans: a chip-filling solve kernel whose blocks were placed while scan blocks were still
    // resident ran ~10 % slower for its whole life (58 ms against 52 ms, same instruction counts); the overlap it
    // gives up is a fraction of a millisecond
    if (round == 0 && heavy) HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->scans_done, 0));
    launch_solve(ctx, da, ch.stream, round == 0 ? solve_blocks : (solve_blocks > 256 ? 256 : solve_blocks));
    if (round == 0) {
      ch.heavy_pending = heavy;
      if (heavy) HIP_TRY(ctx, hipEventRecord(ch.done, ch.stream));
    }
    // Splats go anywhere in the frame, also into rows a later chunk's scan is still read-modify-writing with
    // plain stores: solves may overlap the remaining scans, the first accept may not.
    if (round == 0) HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->scans_done, 0));
    // A heavy chunk's first accept is released by the same event as the next heavy chunk's first solve.  With its
    // full grid it reaches the CUs first and the solve kernel's blocks are placed around it -- the slow start that
    // costs that kernel 10 % for its whole life (see above).  A quarter block per CU trickles along beside the solve
    // instead and is done before it (heavy regime: 119.2 -> 114.5 ms per frame).
    { const int rcp = probe_round(ctx, da, ch.stream); if (rcp) return rcp; }      // (occlusion probes: the host answers between the solves and the accept)
    unsigned ab = accept_blocks;
    if (round == 0 && heavy && ci + 1 < ctx->n_chunks) {
      const unsigned lim = (unsigned)ctx->num_cu / 4u > 0u ? (unsigned)ctx->num_cu / 4u : 1u;
      if (ab > lim) ab = lim;
    }
    hipLaunchKernelGGL(accept_kernel<0>, dim3(ab), dim3(256), 0, ch.stream, da);
    HIP_TRY(ctx, hipGetLastError());
  }
  return LENTIL_OK;
}

static uint64_t chunk_units(const lentil_params &P, uint64_t nch, uint64_t sum_samples, uint64_t n_items) {
  const bool po = P.cameraType == LENTIL_POLYNOMIAL_OPTICS;
  const uint64_t retries = po ? (uint64_t)(P.vignetting_retries < 0 ? 0 : P.vignetting_retries) : 0u;
  // round 0 needs sum(samples + retries) results per wavelength channel (plus the over-provisioned share); a later
  // round at most the attempts an item has left (chromatic mode re-solves up to `retries` results of a stalled attempt)
  return nch * (4 * sum_samples + (3 * retries + 32) * n_items);
}

// Blind mode: buffers and grids from what the previous pass found in this chunk (with headroom); the stream
// waits for the chunk's scan on the device, the host does not.  prep_items_kernel reads the real item count and
// raises DevCounters::fallback -- emitting nothing -- when it does not fit.
static int enqueue_chunk_draws_blind(lentil_hip_ctx *ctx, int ci, DrawArgs &da, int blind_rounds, bool *done) {
  lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
  *done = false;
  if (!ctx->blind || !ch.have_est) return LENTIL_OK;
  const uint64_t cap = ch.v_end - ch.v_begin;
  uint64_t items = ch.est_items + ch.est_items / 4 + 1024;
  if (items > cap) items = cap;
  const uint64_t sum = ch.est_sum + ch.est_sum / 4 + 65536;
  const uint64_t nch = (uint64_t)da.n_channels;
  const uint64_t units = chunk_units(ctx->P, nch, sum, items);
  if (units > ctx->max_pool_units || items == 0) return LENTIL_OK;
  int rc;
  if ((rc = size_chunk_buffers(ctx, ch, items, units))) return rc;
  bind_chunk_buffers(ch, da);
  const bool po = ctx->P.cameraType == LENTIL_POLYNOMIAL_OPTICS;
  da.ctr = ctx->d_ctr + ci;
  da.retries = po ? (int32_t)(ctx->P.vignetting_retries < 0 ? 0 : ctx->P.vignetting_retries) : 0;
  da.work = ctx->d_work + ch.v_begin;
  da.work_cap = cap;
  da.blind = 1u;
  da.n_items = ch.item_cap < cap ? ch.item_cap : cap;      // capacity; the count comes from the scan's counter
  if ((rc = launch_chunk_rounds(ctx, ci, da, ch.est_items, ch.est_sum, da.n_items, blind_rounds))) return rc;
  ch.n_items = 1;          // unknown until the counters are read back: have the continuation loop look
  ch.was_blind = true;
  *done = true;
  return LENTIL_OK;
}

// Sizes one chunk's draw-pipeline buffers from its scan result (host waits for the chunk's scan) and enqueues
// its rounds.  A chunk whose result pool would exceed max_pool_units is processed in sub-batches of items, each
// run to completion before the next reuses the buffers.
static int enqueue_chunk_draws(lentil_hip_ctx *ctx, int ci, DrawArgs &da, int blind_rounds) {
  lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
  const lentil_params &P = ctx->P;
  DevCounters *dctr = ctx->d_ctr + ci;
  // wait for this chunk's scan only (later chunks keep scanning meanwhile), fetch its item count / draw sum
  HIP_TRY(ctx, hipEventSynchronize(ch.scanned));
  DevCounters c;
  HIP_TRY(ctx, hipMemcpyAsync(&c, dctr, sizeof(c), hipMemcpyDeviceToHost, ch.stream));
  HIP_TRY(ctx, hipStreamSynchronize(ch.stream));
  const uint64_t cap = ch.v_end - ch.v_begin;
  const uint64_t n_items = c.work_count < cap ? c.work_count : cap;
  ch.n_items = n_items;
  ch.was_blind = false;
  ch.have_est = true; ch.est_items = n_items; ch.est_sum = c.sum_samples;
  if (n_items == 0) return LENTIL_OK;
  const bool po = P.cameraType == LENTIL_POLYNOMIAL_OPTICS;
  const uint32_t retries = po ? (uint32_t)(P.vignetting_retries < 0 ? 0 : P.vignetting_retries) : 0u;
  da.ctr = dctr;
  da.retries = (int32_t)retries;
  da.blind = 0u;
  da.work_cap = cap;
  const uint64_t nch = (uint64_t)da.n_channels;
  const uint64_t units = chunk_units(P, nch, c.sum_samples, n_items);
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 4;
  int rc;
  if (units <= ctx->max_pool_units) {
    // with the headroom the next (blind) pass will ask for, so that it does not have to grow the buffers again
    uint64_t items_h = n_items + n_items / 4 + 1024;
    if (items_h > cap) items_h = cap;
    uint64_t units_h = chunk_units(P, nch, c.sum_samples + c.sum_samples / 4 + 65536, items_h);
    if (units_h > ctx->max_pool_units) { units_h = units; items_h = n_items; }
    if ((rc = size_chunk_buffers(ctx, ch, items_h, units_h))) return rc;
    bind_chunk_buffers(ch, da);
    da.work = ctx->d_work + ch.v_begin;
    da.n_items = n_items;
    return launch_chunk_rounds(ctx, ci, da, n_items, c.sum_samples, n_items, blind_rounds);
  }
  // ---- too many draws for one result pool: sub-batches of items, bounded by the per-item worst case
  const uint64_t max_samples = P.samples_override > 0 ? (uint64_t)P.samples_override : 2000ull;
  const uint64_t per_item = nch * (4 * max_samples + 3 * retries + 32);
  uint64_t batch_items = ctx->max_pool_units / per_item;
  if (batch_items < 1) return fail(ctx, LENTIL_ERR_NOMEM, "LENTIL_MAX_POOL_UNITS is too small for a single item");
  for (uint64_t i0 = 0; i0 < n_items; i0 += batch_items) {
    const uint64_t ni = n_items - i0 < batch_items ? n_items - i0 : batch_items;
    if ((rc = size_chunk_buffers(ctx, ch, ni, ni * per_item))) return rc;
    bind_chunk_buffers(ch, da);
    // fresh queues for this batch
    HIP_TRY(ctx, hipMemsetAsync((char *)dctr + offsetof(DevCounters, n_tasks), 0,
                                offsetof(DevCounters, inv_row_min) - offsetof(DevCounters, n_tasks), ch.stream));
    da.work = ctx->d_work + ch.v_begin + i0;
    da.n_items = ni;
    da.parity = 0; da.round = 0;
    hipLaunchKernelGGL(prep_items_kernel, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, ch.stream, da);
    launch_solve(ctx, da, ch.stream, (unsigned)max_blocks);
    if (i0 == 0) HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->scans_done, 0));   // see launch_chunk_rounds
    if ((rc = probe_round(ctx, da, ch.stream))) return rc;
    hipLaunchKernelGGL(accept_kernel<0>, dim3((unsigned)ctx->num_cu * 2), dim3(256), 0, ch.stream, da);
    HIP_TRY(ctx, hipGetLastError());
    int rounds = 0;
    if ((rc = finish_rounds(ctx, ci, da, 1, &rounds))) return rc;
    if (rounds > ctx->last_rounds) ctx->last_rounds = rounds;
  }
  ch.n_items = 0;       // complete: nothing left for the continuation loop
  return LENTIL_OK;
}

static void init_draw_args(lentil_hip_ctx *ctx, DrawArgs &da) {
  const lentil_params &P = ctx->P;
  const int C = ctx->n_chunks;
  da.P = P;
  // wavelength channels of the polynomial-optics draw loop, src/lentil_filter.cpp:254-268 (float arithmetic of
  // linear_interpolate, src/global.h:3-5); abb_chromatic < 0 runs three white channels at 0.55, like upstream
  da.n_channels = (P.cameraType == LENTIL_POLYNOMIAL_OPTICS && P.abb_chromatic != 0.0f) ? 3 : 1;
  da.chroma_weights = P.abb_chromatic > 0.0f ? 1 : 0;
  da.lambda[0] = da.lambda[1] = da.lambda[2] = (double)P.lambda_bw;
  if (da.n_channels == 3 && da.chroma_weights) {
    const float c = P.abb_chromatic;
    const float p0 = (float)(1.0 - (double)c);
    da.lambda[0] = (double)(0.35f + p0 * (0.55f - 0.35f));
    da.lambda[1] = (double)0.55f;
    da.lambda[2] = (double)(0.55f + c * (0.85f - 0.55f));
  }
  da.lens = P.cameraType == LENTIL_POLYNOMIAL_OPTICS ? ctx->d_lens : nullptr;
  da.terms = ctx->d_terms;
  da.bokeh = ctx->bokeh;
  da.V = ctx->V;
  da.F = ctx->F;
  // a straggler is a solve well beyond what this lens usually takes: 1.3 x the mean iteration count of the previous
  // pass where that is above LENTIL_SLOW_AT (petzval table: ~25 iterations on average, 20 would park thousands)
  da.slow_at = ctx->slow_at;
  da.batch_margin16 = ctx->batch_margin16;
  da.margin_low_rate = ctx->margin_low_rate;
  {
    static const bool narrow = getenv("LENTIL_ACCEPT_WIDE") && getenv("LENTIL_ACCEPT_WIDE")[0] == '0';
    da.accept_narrow = narrow ? 1 : 0;
  }
  if (ctx->slow_at > 0 && ctx->mean_iters > 0.0) {
    const int adaptive = (int)(1.3 * ctx->mean_iters + 0.5);
    if (adaptive > da.slow_at) da.slow_at = adaptive < 90 ? adaptive : 90;
  }
  da.slow_below = ctx->slow_below;
  da.slow_from_round = ctx->slow_from_round;
  da.slow_max_lanes = ctx->slow_max_lanes;
  da.slow_prio = ctx->slow_prio;
  da.slow_nap_max = ctx->slow_nap_max;
  { const char *e = getenv("LENTIL_DISPATCH_PROBE"); da.dispatch_probe = (e && e[0] == '1') ? 1 : 0; }
  da.extra_num = ctx->extra_num; da.extra_const = ctx->extra_const; da.extra_below = ctx->extra_below;
  da.log = ctx->d_log;
  da.log_cap = ctx->log_cap;
  da.log_count = &ctx->d_ctr[C].log_count;
  da.slow_q = -1; da.slow_round = -1; da.slow_close = 1; da.slow_indirect = 0;      // a straggler queue per round
}

// FrameDev::touched and ::touched_px share one allocation: a byte per 64-pixel group, then (256-byte aligned) a byte per pixel
static inline uint64_t touched_px_offset(uint64_t np) { return (((np + 63) / 64) + 255) & ~(uint64_t)255; }
static inline uint64_t touched_bytes(uint64_t np) { return touched_px_offset(np) + np; }

// FrameDev::touched stops being the whole truth about `acc` (something other than a splat is written there): the
// flags are reset, the next clear wipes the rows it knows to be dirty -- or everything
static void untrust_touched(lentil_hip_ctx *ctx) {
  // (every caller is about to put something other than a pass's splats into `acc` -- merges, the exchange, a fold: the frame
  // is no longer "cleared and untouched", so a streamed pass into it could not be wiped and redone after a stall, and
  // prepare_direct must not take the splat flags for the whole truth; round-4 ADVICE)
  ctx->cleared_since_pass = false;
  if (!ctx->F.touched) return;
  (void)hipMemsetAsync(ctx->d_touched, 0, touched_bytes(ctx->F.np), ctx->stream);
  ctx->F.touched = nullptr;
  ctx->F.touched_px = nullptr;
}

// ---- FrameDev::dir bookkeeping ---------------------------------------------------------------------------
static int fold_direct(lentil_hip_ctx *ctx, uint64_t p_begin, uint64_t p_end, bool all) {
  if (!ctx->F.dir) return LENTIL_OK;
  untrust_touched(ctx);        // `acc` is about to hold more than splats
  if (p_end > p_begin) {
    uint64_t blocks = ((p_end - p_begin) * (ctx->F.stride / 4) + 255) / 256;
    const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(fold_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->F, ctx->d_dir, p_begin, p_end);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (all) { ctx->F.dir = nullptr; ctx->dir_dirty = false; }
  return LENTIL_OK;
}

// before the scans of a pass: `region` is what scan_dma_kernel is about to store (null: another scan kernel runs)
static int prepare_direct(lentil_hip_ctx *ctx, const lentil_hip_ctx::DirRegion *region) {
  int rc;
  // a second pass into the same frame: what the first one stored becomes part of `acc`
  if (ctx->F.dir && (rc = fold_direct(ctx, 0, ctx->F.np, true))) return rc;
  if (ctx->dir_dirty && !(region && *region == ctx->dir_region)) {
    // left over from a cleared frame and not about to be overwritten pixel for pixel: wipe its rows
    const lentil_hip_ctx::DirRegion &r = ctx->dir_region;
    const uint64_t rows = r.ppr ? (r.npix + r.ppr - 1) / r.ppr : 0;
    uint64_t row_lo = (uint64_t)(r.y0 < 0 ? 0 : r.y0);
    uint64_t row_hi = rows ? row_lo + (rows - 1) * (r.row_stride ? r.row_stride : 1) + 1 : row_lo;
    if (row_hi > ctx->P.yres) row_hi = ctx->P.yres;
    if (row_hi > row_lo)
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_dir + row_lo * ctx->P.xres * ctx->F.stride, 0,
                                  (row_hi - row_lo) * ctx->P.xres * ctx->F.stride * sizeof(float), ctx->stream));
    ctx->dir_dirty = false;
  }
  if (region) {
    if (!ctx->d_dir) {
      const uint64_t nfl = ctx->F.np * ctx->F.stride;
      HIP_TRY(ctx, hipMalloc(&ctx->d_dir, nfl * sizeof(float)));
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_dir, 0, nfl * sizeof(float), ctx->stream));
    }
    ctx->F.dir = ctx->d_dir;
    ctx->dir_dirty = true;
    ctx->dir_region = *region;
    if (ctx->cleared_since_pass) {
      // `acc` is all zeros and this pass only splats into it
      if (!ctx->d_touched) {
        const uint64_t n = touched_bytes(ctx->F.np);
        HIP_TRY(ctx, hipMalloc(&ctx->d_touched, n));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_touched, 0, n, ctx->stream));
      }
      ctx->F.touched = ctx->d_touched;
      ctx->F.touched_px = ctx->d_touched + touched_px_offset(ctx->F.np);
    }
  } else {
    untrust_touched(ctx);
  }
  ctx->cleared_since_pass = false;
  return LENTIL_OK;
}

// Where get_coc_thinlens(P, cz) < 0.4f (lentil_device.h; src/lentil.h:674-692, src/lentil_filter.cpp:185-190) is decided by cz
// alone, for scan_dma2_kernel.  The kernel computes, in fp32, ifd = (-f * -fd) / (-f + -fd), isp = (-f * z) / (-f + z),
// coc = |A (isp - ifd) / isp|.  In exact arithmetic 1 / isp = 1 / z - 1 / f, so coc = |A| |c0 - c1 / z| with c0 = 1 + ifd / f,
// c1 = ifd: in u = 1 / z the set {coc < t} is ONE interval around the focus plane.  The fp32 evaluation differs from that by
// a few 1e-7 relative, times |A| / 0.4 near the threshold (the subtraction isp - ifd carries isp's rounding): with
// eps = 1e-3 + 1e-5 |A| the comparison with 0.4 (1 - eps) / 0.4 (1 + eps) is certain, and the strip in between -- a
// few visits in 10^4 -- is left to the function.  Interval ends are rounded towards the uncertain side; |z| > 1e30 (where
// -f * z overflows and the function returns NaN) is never "certainly below".
static ScanBands scan_bands(const lentil_params &P) {
  ScanBands B;
  for (int i = 0; i < 2; ++i) { B.in_lo[i] = 1.0f; B.in_hi[i] = -1.0f; B.out_lo[i] = 1.0f; B.out_hi[i] = -1.0f; }
  B.out_lo[0] = -INFINITY; B.out_hi[0] = INFINITY;       // nothing usable: every finite depth asks the function
  float fd = (float)P.focus_distance, A = (float)P.aperture_radius;
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS) fd = (float)((double)fd / 10.0);
  else A = (float)((double)A * 10.0);
  const float f = P.focal_length;
  const float ifd = (-f * -fd) / (-f + -fd);
  const double a = std::fabs((double)A), c0 = 1.0 + (double)ifd / (double)f, c1 = (double)ifd;
  if (!(a > 0.0) || !std::isfinite(a) || !std::isfinite(c0) || !std::isfinite(c1) || c1 == 0.0 || !(f > 0.0f)) return B;
  const double eps = 1e-3 + 1e-5 * a;
  if (!(eps < 0.25)) return B;
  auto up = [](double v) { float r = (float)v; if ((double)r < v) r = std::nextafter(r, INFINITY); return r; };       // smallest float >= v
  auto down = [](double v) { float r = (float)v; if ((double)r > v) r = std::nextafter(r, -INFINITY); return r; };    // largest float <= v
  // z-intervals of {coc <= t}: u in [ua, ub], z = 1 / u
  auto pieces = [&](double t, bool inner, float lo[2], float hi[2]) {
    double ua = (c0 - t / a) / c1, ub = (c0 + t / a) / c1;
    if (ua > ub) std::swap(ua, ub);
    lo[0] = lo[1] = 1.0f; hi[0] = hi[1] = -1.0f;
    const double big = inner ? 1e30 : (double)INFINITY;
    if (ua > 0.0 || ub < 0.0) {
      const double zl = 1.0 / ub, zh = 1.0 / ua;
      lo[0] = inner ? up(std::max(zl, -big)) : down(zl);
      hi[0] = inner ? down(std::min(zh, big)) : up(zh);
    } else {
      // the interval holds u = 0: everything beyond 1 / ua on the negative side, beyond 1 / ub on the positive side
      if (ua < 0.0) { lo[0] = inner ? (float)-big : -INFINITY; hi[0] = inner ? down(1.0 / ua) : up(1.0 / ua); }
      if (ub > 0.0) { lo[1] = inner ? up(1.0 / ub) : down(1.0 / ub); hi[1] = inner ? (float)big : INFINITY; }
      if (!inner && (ua == 0.0 || ub == 0.0)) { lo[0] = -INFINITY; hi[0] = INFINITY; }
    }
  };
  ScanBands R = B;
  pieces(0.4 * (1.0 - eps), true, R.in_lo, R.in_hi);
  pieces(0.4 * (1.0 + eps), false, R.out_lo, R.out_hi);
  for (int i = 0; i < 2; ++i)
    if (std::isnan(R.in_lo[i]) || std::isnan(R.in_hi[i]) || std::isnan(R.out_lo[i]) || std::isnan(R.out_hi[i])) return B;
  return R;
}

// test hook (no GPU needed): the intervals scan_dma2_kernel would use for these parameters -- in_lo[2], in_hi[2], out_lo[2], out_hi[2]
LENTIL_API int lentil_hip_debug_scan_bands(const lentil_params *P, float out[8]) {
  if (!P || !out) return LENTIL_ERR_INVALID;
  const ScanBands B = scan_bands(*P);
  for (int i = 0; i < 2; ++i) { out[i] = B.in_lo[i]; out[2 + i] = B.in_hi[i]; out[4 + i] = B.out_lo[i]; out[6 + i] = B.out_hi[i]; }
  return LENTIL_OK;
}

// How the bound visit stream is scanned: kernel, tile size, LDS.
struct ScanPlan {
  ScanArgs sa{};
  size_t lds = 0;
  uint64_t n_tiles = 0;
  bool multi = false;
  bool dma = false;       // scan_dma_kernel (beauty only, uniform weights)
  bool dma2 = false;      // ... its pipelined form, scan_dma2_kernel (one block per CU)
  bool dma_multi = false; // scan_dma_multi_kernel (extra gaussian AOVs, uniform weights)
  uint32_t M = 0;
};

// slots per wave (LENTIL_DMA_MULTI_RING: 2 or 3; no difference measured, the waves are not short of bytes in flight)
static uint32_t dma_multi_ring(const lentil_hip_ctx *) {
  static const int forced = getenv("LENTIL_DMA_MULTI_RING") ? atoi(getenv("LENTIL_DMA_MULTI_RING")) : 0;
  return forced == 3 ? 3u : 2u;
}
// blocks per CU (LENTIL_DMA_MULTI_BLOCKS): two where the CU's LDS holds them and two solve blocks beside them
static uint32_t dma_multi_blocks_per_cu(const lentil_hip_ctx *ctx) {
  static const int forced = getenv("LENTIL_DMA_MULTI_BLOCKS") ? atoi(getenv("LENTIL_DMA_MULTI_BLOCKS")) : 0;
  if (forced >= 1 && forced <= 4) return (uint32_t)forced;
  const size_t one = (size_t)4 * dma_multi_wave_f4(ctx->V.n_extra, dma_multi_ring(ctx)) * 16 + 4 * kWaveQueueLds * sizeof(uint2);
  return 2u * one + 2u * 10u * 1024u <= 160u * 1024u ? 2u : 1u;
}
static size_t dma_multi_lds(const lentil_hip_ctx *ctx) {
  return (size_t)4 * dma_multi_wave_f4(ctx->V.n_extra, dma_multi_ring(ctx)) * 16 + 4 * kWaveQueueLds * sizeof(uint2);
}

// scan_dma_multi_kernel takes the stream: whole pixels of M <= 64 visits, uniform weights, gaussian AOVs only
static bool dma_multi_applies(const lentil_hip_ctx *ctx) {
  static const bool allowed = !(getenv("LENTIL_SCAN_DMA_MULTI") && getenv("LENTIL_SCAN_DMA_MULTI")[0] == '0');
  const uint32_t M = ctx->V.visits_per_pixel;
  return allowed && ctx->scan_dma && M > 0 && M <= 64 && ctx->V.n_extra > 0 && !ctx->V.inv_density && !ctx->F.zkey && !ctx->F.zkey_dbg &&
         ctx->F.closest_mask == 0 && ctx->V.cam.n < 2 && ctx->V.n % M == 0 && dma_multi_lds(ctx) <= 160u * 1024u;
}

// static LDS of one resident solve block of a streamed pass (solve_po_kernel<.., kStream>; tools/kernel_resources.py): 28.2 KB for a
// lens that runs as straight-line code -- built into the library, or specialised at run time once its code object is there --,
// 52.2 KB with the table interpreter
static size_t solve_block_lds(lentil_hip_ctx *ctx) {
  const bool straight = ctx->use_generated && (lentil_hip_lens_is_compiled(ctx) || jit_function(ctx, false, true) != nullptr);
  return straight ? 29184u : 53760u;
}

static int plan_scan(lentil_hip_ctx *ctx, ScanPlan &pl) {
  ScanArgs &sa = pl.sa;
  sa.P = ctx->P;
  sa.lens_length = ctx->have_lens ? ctx->hlens.length : 0.0;
  sa.V = ctx->V;
  sa.F = ctx->F;
  const uint32_t M = ctx->V.visits_per_pixel;
  pl.M = M;
  if (M) {
    // staging: 20 B per visit per wave, 4 waves per block, keep a block under ~48 KiB
    uint32_t ppt = 64;
    // extra AOV columns are streamed one at a time: more, smaller tiles keep enough loads in flight
    // (beauty only: 64-pixel tiles, three blocks per CU.  32-pixel tiles / four blocks per CU are 3 % faster for a
    // scan that has the chip to itself -- 0.946 against 0.972 ms -- and 20 % slower beside the first chunk's solve
    // kernel, which then does not get its wave per SIMD until scan blocks retire)
    uint64_t lds_budget = (ctx->V.n_extra ? 24ull : 48ull) * 1024ull;
    if (const char *e = getenv("LENTIL_SCAN_LDS_KB")) lds_budget = strtoull(e, nullptr, 10) * 1024ull;
    while (ppt > 1 && (uint64_t)ppt * M * 20ull * 4ull > lds_budget) ppt >>= 1;
    if ((uint64_t)ppt * M * 20ull * 4ull > 150ull * 1024ull)
      return fail(ctx, LENTIL_ERR_UNSUPPORTED, "visits_per_pixel too large for the LDS staging area");
    sa.ppt = ppt;
    sa.tv_pad = ppt * M;
    pl.lds = (size_t)sa.tv_pad * 20 * 4 + 4 * kWaveQueueLds * sizeof(uint2);
    // frames with extra AOVs: all columns of a visit in flight at once, one step = 64 / M whole pixels
    pl.multi = ctx->V.n_extra > 0 && M <= 64 && !getenv("LENTIL_SCAN_SINGLE_COLUMN");
    if (pl.multi) {
      ppt = 64 / M;
      sa.ppt = ppt;
      sa.tv_pad = ppt * M;
      const size_t wave_f4 = (size_t)ctx->F.n_aovs * kMultiPlane + 16 + (size_t)ppt * (ctx->F.stride / 4);
      pl.lds = 4 * wave_f4 * 16 + 4 * kWaveQueueLds * sizeof(uint2);
    }
    // ring slots of scan_dma_kernel (LENTIL_DMA_RING; a streamed pass has one scan block per CU and LDS to spare)
    uint32_t dma_ring = kDmaRing;
    if (const char *e = getenv("LENTIL_DMA_RING")) { const int r = atoi(e); if (r >= 2 && r <= 8) dma_ring = (uint32_t)r; }
    const size_t dma_lds = (size_t)4 * dma_wave_f4(M, dma_ring) * 16 + 4 * kWaveQueueLds * sizeof(uint2);
    pl.dma = ctx->scan_dma && ctx->V.n_extra == 0 && !ctx->V.inv_density && !ctx->F.zkey && !ctx->F.zkey_dbg && ctx->V.cam.n < 2 &&
             ctx->V.n % M == 0 && dma_lds <= 80u * 1024u;
    if (pl.dma) {
      ppt = 64;
      sa.ring = dma_ring;
      sa.ppt = ppt;
      sa.tv_pad = ppt * M;
      pl.lds = dma_lds;
      pl.multi = false;
      // scan_dma2_kernel: tiles pipelined into one another (two rgba buffers per wave), one block per CU.  In a streamed pass
      // the block must fit beside the resident solve blocks, or neither it nor they would ever end (LENTIL_SCAN_DMA2=0: never)
      static const bool dma2_allowed = !(getenv("LENTIL_SCAN_DMA2") && getenv("LENTIL_SCAN_DMA2")[0] == '0');
      const size_t dma2_lds = (size_t)4 * dma2_wave_f4(M) * 16 + 4 * kWaveQueueLds * sizeof(uint2);
      // (static LDS of solve_po_kernel<.., kStream>, tools/kernel_resources.py: 28.2 KB compiled, 52.2 KB with the table interpreter)
      const size_t solve_lds = (size_t)ctx->stream_blocks * solve_block_lds(ctx);
      const uint64_t ppr = ctx->V.pixels_per_row;
      pl.dma2 = dma2_allowed && M >= 2 && ppr >= 2 && ppr < (1ull << 31) && ctx->V.n / M < (1ull << 31) &&
                dma2_lds + (ctx->stream_mode ? solve_lds : 0) <= 160u * 1024u;
      if (pl.dma2) {
        pl.lds = dma2_lds;
        sa.bands = scan_bands(ctx->P);
        uint32_t sh = 0;
        while ((2ull << sh) < ppr) ++sh;                 // 2^sh < ppr <= 2^(sh+1)
        sa.ppr_shift = sh;
        sa.ppr_magic = (uint32_t)(((1ull << (32 + sh)) + ppr - 1) / ppr);
      }
    }
    // frames with extra AOVs, all of them gaussian: the LDS-DMA form of the multi-column scan (LENTIL_SCAN_DMA_MULTI=0: never)
    const size_t dmam_lds = dma_multi_lds(ctx);
    pl.dma_multi = dma_multi_applies(ctx);
    if (pl.dma_multi) {
      ppt = 64 / M;
      // the sum lanes come in passes of 64 float4 of the group's records: a pixel fewer per group where that saves the
      // second pass (nine visits, nine AOVs: 7 x 10 float4 = two passes, 6 x 10 = one; 3.19 against 3.39 ms)
      const uint32_t q = ctx->F.stride / 4;
      if (ppt * q > 64 && 64 / q >= 1 && 4 * (64 / q) >= 3 * ppt) ppt = 64 / q;
      if (const char *e = getenv("LENTIL_DMA_MULTI_PPT")) { const uint32_t f = (uint32_t)atoi(e); if (f >= 1 && f <= 64 / M) ppt = f; }
      sa.ppt = ppt;
      sa.tv_pad = ppt * M;
      pl.lds = dmam_lds;
      pl.multi = false;
      pl.dma = false;
      sa.dummy = ctx->d_dummy;
      // (the column loads without the nontemporal hint: the 54-visit groups of nine-visit pixels do not end on 128-byte lines,
      // and the line two groups share is then still in L2 for the second -- 3.05 against 3.16 ms alone; LENTIL_DMA_MULTI_NT=1)
      sa.ring = dma_multi_ring(ctx) | ((getenv("LENTIL_DMA_MULTI_NT") && getenv("LENTIL_DMA_MULTI_NT")[0] == '1') ? 0u : 0x100u);
    }
    const uint64_t n_pixels = (ctx->V.n + M - 1) / M;
    pl.n_tiles = (n_pixels + ppt - 1) / ppt;
  }
  lentil_hip_ctx::DirRegion reg;
  if (pl.dma || pl.dma_multi) {
    reg.x0 = ctx->V.pixel_x0; reg.y0 = ctx->V.pixel_y0; reg.row_stride = ctx->V.pixel_row_stride; reg.ppr = ctx->V.pixels_per_row;
    reg.npix = ctx->V.n / M;
  }
  const int rc = prepare_direct(ctx, (pl.dma || pl.dma_multi) ? &reg : nullptr);
  if (rc) return rc;
  sa.F = ctx->F;
  return LENTIL_OK;
}

// one scan launch over a chunk's range of the stream (ch.tile_begin/_end, ch.v_begin/_end) on the main stream
static int launch_scan(lentil_hip_ctx *ctx, const ScanPlan &pl, const lentil_hip_ctx::Chunk &ch, DevCounters *ctr,
                       unsigned *blocks_out, bool streamed_pass = false) {
  ScanArgs sa = pl.sa;
  // (LENTIL_SCAN_OUTSIDE_IN=1: a streamed pass scans the frame from its top and bottom edge inwards.  Measured neutral,
  // 2.32-2.33 ms either way: the parked solves of the edge items then come early, but the straggler kernel only gets its
  // registers when the scan's waves have left, scan_order in lentil_kernels.h)
  const char *oi = getenv("LENTIL_SCAN_OUTSIDE_IN");
  const bool outside_in = oi && oi[0] == '1';
  sa.outside_in = (outside_in && streamed_pass) ? 1u : 0u;
  sa.work = ctx->d_work + ch.v_begin;
  sa.work_cap = ch.v_end - ch.v_begin;
  sa.ctr = ctr;
  sa.tile_begin = ch.tile_begin; sa.tile_end = ch.tile_end;
  sa.v_begin = ch.v_begin; sa.v_end = ch.v_end;
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
  uint64_t blocks;
  unsigned multi_skipped = 0;
  // (a streamed pass's single launch: timed by its own dispatch, lentil_hip_last_timing; LENTIL_SCAN_EVENTS=0: by the events around it)
  static const bool scan_events = !(getenv("LENTIL_SCAN_EVENTS") && getenv("LENTIL_SCAN_EVENTS")[0] == '0');
  const bool own_events = streamed_pass && scan_events && (pl.dma || pl.dma_multi);
  ctx->scan_kernel_timed = own_events;
  if (pl.dma) {
    // persistent, every wave draws four tiles at a time
    blocks = (ch.tile_end - ch.tile_begin + 15) / 16;
    // one block per CU: 1.07 ms alone against 0.98 with two, but a CU then has room (registers, LDS) for three solve
    // blocks beside it, and a streamed pass ends when its solves do
    uint64_t per_cu = ctx->stream_mode ? 1 : 2;
    if (const char *e = getenv("LENTIL_DMA_BLOCKS")) per_cu = strtoull(e, nullptr, 10);
    if (blocks > (uint64_t)ctx->num_cu * per_cu) blocks = (uint64_t)ctx->num_cu * per_cu;
    if (blocks < 1) blocks = 1;
    if (pl.dma2) {
      blocks = (ch.tile_end - ch.tile_begin + 15) / 16;
      if (blocks > (uint64_t)ctx->num_cu) blocks = (uint64_t)ctx->num_cu;
      if (blocks < 1) blocks = 1;
      // (a streamed pass may leave some CUs without a scanning block: those take a third resident solve block, scan_cus_pct;
      // the blocks that do not scan are launched all the same and leave at once, see the kernel)
      sa.skip_blocks = 0;
      // (only where the pass is bound by its solves: 1080p with 256 draws -- 73 k draws a frame -- is not, 0.74 against 0.71 ms)
      if (streamed_pass && ctx->scan_cus_pct < 100 && blocks == (uint64_t)ctx->num_cu && ctx->est_sum_total >= (1ull << 19))
        sa.skip_blocks = (uint32_t)(blocks - ((uint64_t)ctx->num_cu * (uint64_t)ctx->scan_cus_pct + 99) / 100);
      if (own_events) hipExtLaunchKernelGGL(scan_dma2_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, ctx->ev_scan_k[0], ctx->ev_scan_k[1], 0, sa);
      else hipLaunchKernelGGL(scan_dma2_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, sa);
    } else {
      if (own_events) hipExtLaunchKernelGGL(scan_dma_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, ctx->ev_scan_k[0], ctx->ev_scan_k[1], 0, sa);
      else hipLaunchKernelGGL(scan_dma_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, sa);
    }
  } else if (pl.dma_multi) {
    // persistent: a wave draws runs of 16 groups
    blocks = (ch.tile_end - ch.tile_begin + 4 * kDmaMultiRun - 1) / (4 * kDmaMultiRun);
    if (blocks > (uint64_t)ctx->num_cu * dma_multi_blocks_per_cu(ctx)) blocks = (uint64_t)ctx->num_cu * dma_multi_blocks_per_cu(ctx);
    if (blocks < 1) blocks = 1;
    // (scan_cus_pct: this kernel's blocks draw all their tiles from one counter, so fewer of them is all it takes -- the CUs
    // left alone hold a third resident solve block)
    multi_skipped = 0;
    if (streamed_pass && ctx->scan_cus_pct_multi < 100 && blocks == (uint64_t)ctx->num_cu && dma_multi_blocks_per_cu(ctx) == 1) {
      const uint64_t keep = ((uint64_t)ctx->num_cu * (uint64_t)ctx->scan_cus_pct_multi + 99) / 100;
      multi_skipped = (unsigned)(blocks - keep);
      blocks = keep;
    }
    if (own_events) hipExtLaunchKernelGGL(scan_dma_multi_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, ctx->ev_scan_k[0], ctx->ev_scan_k[1], 0, sa);
    else hipLaunchKernelGGL(scan_dma_multi_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, sa);
  } else if (pl.M) {
    blocks = (ch.tile_end - ch.tile_begin + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    if (pl.multi) hipLaunchKernelGGL(scan_uniform_multi_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, sa);
    else hipLaunchKernelGGL(scan_uniform_kernel, dim3((unsigned)blocks), dim3(256), pl.lds, ctx->stream, sa);
  } else {
    blocks = (ch.v_end - ch.v_begin + 255) / 256;
    if (blocks > max_blocks) blocks = max_blocks;
    // runs of a pixel's visits are summed in their order (LENTIL_SCAN_RUNS=0: an atomic per visit and float, any order)
    const bool runs = !(getenv("LENTIL_SCAN_RUNS") && getenv("LENTIL_SCAN_RUNS")[0] == '0');       // (read per launch: the tests switch it)
    if (runs) hipLaunchKernelGGL(scan_runs_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, sa);
    else hipLaunchKernelGGL(scan_ragged_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, sa);
  }
  HIP_TRY(ctx, hipGetLastError());
  if (blocks_out) *blocks_out = (unsigned)blocks;
  ctx->last_scan_skipped = (pl.dma && pl.dma2) ? sa.skip_blocks : (pl.dma_multi ? multi_skipped : 0u);
  return LENTIL_OK;
}

// ---------------------------------------------------------------------------------------
// Streamed pass.  The chunked pass above cannot start a chunk's solves before the chunk's scan has ended, and
// every chunk's first solve kernel brings its own ramp-down; solve kernels of several chunks resident at once take
// the register file from the scan.  Here the whole stream is ONE scan launch that publishes items and first-batch
// tasks as it finds them (publish_item), and the first round is one task queue followed by persistent solve waves:
//   stream A (chunk 0's): solve_po_kernel<.., kStream>, `stream_blocks` blocks per CU, resident beside the scan
//                         from the start of the pass
//   main stream:          scan, then a second launch of the same kernel on the CUs' remaining room, then -- once
//                         both have run dry -- the parked stragglers, the first accept and the later rounds
//                         (ordinary queues written by the accept kernel).
// Nothing on the device waits for a kernel that has not been submitted before it: under a profiler that runs
// kernels one at a time, A starts after the scan and finds the queue complete.
// Buffers are sized from the previous pass (twice what it found); an item that does not fit raises
// DevCounters::fallback, the accept kernel then does nothing and the host redoes the draws with exact sizes.
// ---------------------------------------------------------------------------------------
// ---- first batches from the lens and the frame (lentil_batch_model.h) ---------------------------------------------
// The calibration: bm_nx x bm_ny x bm_nz targets over the frame's field of view and inverse
// depths from infinity to a quarter of the focus distance, kBmK aperture points each, on the context's stream ahead of the
// pass that first needs it (~0.43 M traces through the table interpreter: ~11 ms, once per camera set-up).
static int ensure_batch_model(lentil_hip_ctx *ctx) {
  if (ctx->bm_valid) return LENTIL_OK;
  const lentil_params &P = ctx->P;
  if (!ctx->have_lens || P.cameraType != LENTIL_POLYNOMIAL_OPTICS) return LENTIL_OK;
  if (!(P.focal_length > 0.0f) || !(P.sensor_width > 0.0) || !(P.focus_distance > 0.0)) return LENTIL_OK;
  const size_t nodes = (size_t)ctx->bm_nx * ctx->bm_ny * ctx->bm_nz;
  if (!ctx->d_bm_land) {
    HIP_TRY(ctx, hipMalloc(&ctx->d_bm_land, nodes * kBmK * sizeof(float4)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_bm_box, nodes * sizeof(float4)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_bm_npass, nodes * sizeof(uint32_t)));
  }
  const float thx = (float)(P.sensor_width * 0.5 / (double)P.focal_length) * 0.97f;       // (the rim beyond is extrapolated, batch_estimate)
  const float thy = thx * (float)P.yres_without_region / (float)P.xres_without_region;
  ctx->bm_fx0 = -thx; ctx->bm_fx_step = 2.0f * thx / (float)(ctx->bm_nx - 1);
  ctx->bm_fy0 = -thy; ctx->bm_fy_step = 2.0f * thy / (float)(ctx->bm_ny - 1);
  const float fd_cm = (float)(P.focus_distance / 10.0);              // (polynomial optics: focus_distance is in mm)
  ctx->bm_u0 = 1.0e-6f; ctx->bm_u_step = (4.0f / fd_cm - ctx->bm_u0) / (float)(ctx->bm_nz - 1);
  BatchModelArgs a{};
  a.P = P; a.lens = ctx->d_lens; a.terms = ctx->d_terms; a.bokeh = ctx->bokeh;
  a.land = ctx->d_bm_land; a.box = ctx->d_bm_box; a.npass = ctx->d_bm_npass;
  a.fx0 = ctx->bm_fx0; a.fx_step = ctx->bm_fx_step; a.fy0 = ctx->bm_fy0; a.fy_step = ctx->bm_fy_step;
  a.u0 = ctx->bm_u0; a.u_step = ctx->bm_u_step;
  a.nx = ctx->bm_nx; a.ny = ctx->bm_ny; a.nz = ctx->bm_nz;
  hipLaunchKernelGGL(batch_model_kernel, dim3((unsigned)nodes), dim3(256), 0, ctx->stream, a);
  HIP_TRY(ctx, hipGetLastError());
  ctx->bm_valid = true;
  ++ctx->n_bm_built;
  return LENTIL_OK;
}
static BatchModelDev batch_model_dev(const lentil_hip_ctx *ctx) {
  BatchModelDev m{};
  m.land = ctx->d_bm_land; m.box = ctx->d_bm_box; m.npass = ctx->d_bm_npass;
  m.fx0 = ctx->bm_fx0; m.fx_inv = 1.0f / ctx->bm_fx_step; m.fy0 = ctx->bm_fy0; m.fy_inv = 1.0f / ctx->bm_fy_step;
  m.u0 = ctx->bm_u0; m.u_inv = 1.0f / ctx->bm_u_step;
  m.nx = ctx->bm_nx; m.ny = ctx->bm_ny; m.nz = ctx->bm_nz;
  m.margin16 = ctx->bm_margin16;
  m.xres = (float)ctx->P.xres; m.yres = (float)ctx->P.yres;
  return m;
}

// The end of a streamed pass: its counters have arrived in `pinned`.  What they say -- everything fitted?  the lean tail's bet
// held?  nobody gave up waiting? -- and, where not, the work that is left (can_fix; false for a pass whose frame has been
// cleared since: its verdict is only counted).  Sizes the next pass from what this one found.
static int streamed_finish(lentil_hip_ctx *ctx, StreamTail &t, const DevCounters *pinned, bool can_fix, bool *streamed) {
  const int C = ctx->n_chunks;
  lentil_hip_ctx::Chunk &ch = ctx->chunks[0];
  DrawArgs &da = t.da;
  const bool predicted = t.predicted, lean = t.lean, extend = t.extend, live = t.live;
  const int blind_rounds = t.blind_rounds;
  SlowRec *const slow_base = t.slow_base;
  const uint32_t slow_cap_all = t.slow_cap_all;
  const unsigned accept_blocks = t.accept_blocks;
  int rc;
  (void)live;
  ctx->h_ctr.assign(pinned, pinned + C);
  ctx->h_ctr_valid = true;
  ctx->last_streamed = 1;
  const DevCounters c = ctx->h_ctr[0];
  ch.was_blind = true;
  if (c.probe_snap[0]) {
    // LENTIL_DISPATCH_PROBE: the first accept's last item was finished while blocks of its grid had not begun
    char buf[640];
    int n = snprintf(buf, sizeof buf, "[probe] epoch %u: accept blocks begun %u of %u when the last item was done; per XCD begun:", ctx->epoch, c.probe_snap[33], c.probe_snap[34]);
    for (int x = 0; x < 8; ++x) n += snprintf(buf + n, sizeof buf - n, " %u", c.probe_snap[1 + x]);
    const char *kinds[3] = {"second round's solve waves resident", "second round's straggler waves resident", "first round's straggler waves resident"};
    for (int k = 0; k < 3; ++k) {
      n += snprintf(buf + n, sizeof buf - n, " | %s:", kinds[k]);
      for (int x = 0; x < 8; ++x) n += snprintf(buf + n, sizeof buf - n, " %u", c.probe_snap[9 + 8 * k + x]);
    }
    fprintf(stderr, "%s\n", buf);
  }
  if (c.fallback || c.stuck) {
    {
      // what made the pass give up, kept for lentil_hip_last_redo_note(): `fallback` bits 1 items, 2 result pool, 4 task queue,
      // 8 a wave's pending flushes, 16 range queue, 32 the blind preparation's bounds; `stuck` = (ticket << 2) | who waited (1 a
      // publisher, 2 a resident solve wave, 3 a straggler wave)
      char note[768];
      snprintf(note, sizeof note,
               "epoch %u: fallback 0x%llx stuck 0x%x (timeout %.0f ms)%s%s | items %llu/%u tasks %u/%u pool %llu/%llu ranges %u/%u | "
               "scan blocks done %u publishers done %u rounds_used %llu | first batches %s, margin16 %u, blind passes before %u | "
               "the wave that gave up: round %u parity %u, its queue's n_tasks %u head %u, accept blocks done %u begun %u, slot word 0x%x (epoch tag 0x%x), block %u",
               ctx->epoch, (unsigned long long)c.fallback, c.stuck, (double)t.stuck_ticks * 1.0e-5,
               c.stuck ? " waited: " : "", c.stuck ? ((c.stuck & 3u) == 1 ? "publisher" : (c.stuck & 3u) == 2 ? "resident solve wave" : "straggler wave") : "",
               (unsigned long long)c.work_count, t.item_cap, c.n_tasks[0], t.task_cap, (unsigned long long)c.pool_used[0],
               (unsigned long long)t.pool_cap, c.n_ranges, t.range_cap, c.scan_blocks_done, c.publishers_done,
               (unsigned long long)c.rounds_used, predicted ? "modelled" : "plain", ctx->bm_margin16, ctx->last_blind - 1u,
               c.stuck_info[0], c.stuck_info[1], c.stuck_info[2], c.stuck_info[7], c.stuck_info[3], c.stuck_info[4], c.stuck_info[5],
               c.stuck_info[5] >> kTaskTagShift, c.stuck_info[6]);
      ctx->redo_note = note;
      if (c.stuck && (c.stuck & 3u) == 1u && ctx->d_ranges) {
        // a publisher gave up on its range slot: what the slot holds now (every kernel of the pass has left), the slots around the
        // cursor and the queue's counters as the host reads them -- a record that IS there was written and not seen
        const uint32_t tk = c.stuck >> 2;
        uint64_t w[4] = {0, 0, 0, 0};
        if ((uint64_t)tk + 2 < ctx->range_cap)
          (void)hipMemcpy(w, ctx->d_ranges + (tk ? tk - 1 : 0), sizeof w, hipMemcpyDeviceToHost);
        char more[256];
        snprintf(more, sizeof more, " | range slots from %u on (host read-back): %016llx %016llx %016llx %016llx, range_head %u",
                 tk ? tk - 1 : 0, (unsigned long long)w[0], (unsigned long long)w[1], (unsigned long long)w[2], (unsigned long long)w[3], c.range_head);
        ctx->redo_note += more;
      }
      if (getenv("LENTIL_STREAM_DEBUG")) fprintf(stderr, "[stream] note: %s\n", ctx->redo_note.c_str());
    }
    if (getenv("LENTIL_STREAM_DEBUG"))
      fprintf(stderr, "[stream] redo: who %u ticket %u epoch %u range_head %u pubs_done %u scan_done %u | fallback %llu stuck %u | items %llu (cap %u) tasks %u (cap %u) pool %llu (cap %llu) ranges %u (cap %u)\n",
              c.stuck & 3u, c.stuck >> 2, ctx->epoch, c.range_head, c.publishers_done, c.scan_blocks_done, c.fallback, c.stuck, c.work_count, t.item_cap, c.n_tasks[0], t.task_cap, c.pool_used[0], (unsigned long long)t.pool_cap,
              c.n_ranges, t.range_cap);
    if (getenv("LENTIL_STREAM_DEBUG")) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
      HIP_TRY(ctx, hipEventSynchronize(ctx->ev[2]));
      float ms_scan = 0.f, ms_all = 0.f;
      (void)hipEventElapsedTime(&ms_scan, ctx->ev[0], ctx->ev[1]);
      (void)hipEventElapsedTime(&ms_all, ctx->ev[0], ctx->ev[2]);
      fprintf(stderr, "[stream] scan %.3f ms, pass until the read-back %.3f ms\n", ms_scan, ms_all);
    }
    if (getenv("LENTIL_STREAM_DEBUG"))
      fprintf(stderr, "[stream] queues: n_tasks %u/%u task_head %u/%u n_active %u/%u active_head %u/%u accept_done %u/%u pool_used %llu/%llu\n",
              c.n_tasks[0], c.n_tasks[1], c.task_head[0], c.task_head[1], c.n_active[0], c.n_active[1], c.active_head[0], c.active_head[1],
              c.accept_done[0], c.accept_done[1], c.pool_used[0], c.pool_used[1]);
    if (getenv("LENTIL_STREAM_DEBUG"))
      fprintf(stderr, "[stream] stragglers: live %d waves_done %u/%u (round 1: %u) parked %u/%u heads %u/%u cap %u waves %u rounds_used %llu\n", (int)live,
              c.waves_done[0], c.waves_started[0], c.waves_done[1], c.n_slow[0], c.n_slow[1], c.slow_head[0], c.slow_head[1],
              da.slow_cap, da.slow_waves, c.rounds_used);
    if (c.stuck) {
      g_stat_stuck.fetch_add(1, std::memory_order_relaxed);
      if (t.inject) g_stat_stuck_injected.fetch_add(1, std::memory_order_relaxed);
      else {
        std::lock_guard<std::mutex> lock(g_stall_notes_mutex);
        if (g_stall_notes.size() < 8) g_stall_notes.push_back(ctx->redo_note + (t.deferred ? " | end deferred" : " | end awaited") + (can_fix ? "" : ", abandoned") +
                                                              " | frame " + std::to_string(ctx->P.xres) + "x" + std::to_string(ctx->P.yres) + ", " + std::to_string(ctx->V.n) + " visits");
      }
    }
    if (ctx->notes.size() < 8) ctx->notes.push_back(ctx->redo_note);
    if (!can_fix) {
      // abandoned: the frame this pass wrote into has been cleared since; what it lacked is counted, not done
      if (c.stuck) ++ctx->n_stuck;
      ++ctx->last_fallback;
      ++ctx->totals.abandoned_incomplete;
      ctx->last_rounds = (int)c.rounds_used;
      if (!c.stuck) {
        // (the scan and the publishers counted everything they met, whether or not it fitted: the next pass is sized from that)
        const uint64_t n_items = c.work_count < ctx->V.n ? c.work_count : ctx->V.n;
        ctx->have_total_est = true;
        ctx->est_items_total = n_items; ctx->est_sum_total = c.sum_samples;
      }
      for (int ci = 0; ci < C; ++ci) ctx->chunks[ci].have_est = false;
      *streamed = true;
      return LENTIL_OK;
    }
    t.did_more = true;
    if (c.stuck && c.rounds_used) {
      // Stalled with draws already accepted: the frame holds a part of the pass.  It held nothing before (the gate at the
      // top), so lentil_hip_redistribute wipes it and runs the whole pass again in the chunked form.
      ++ctx->n_stuck;
      ctx->stall_redo = true;
      ctx->h_ctr_valid = false;
      ctx->late_resolve_done = false;
      *streamed = true;
      return LENTIL_OK;
    }
    if (c.stuck) {
      ++ctx->n_stuck;
      HIP_TRY(ctx, hipMemsetAsync((char *)ctx->d_ctr + offsetof(DevCounters, stuck), 0, sizeof(unsigned int), ch.stream));
    }
    // did not fit: nothing was accepted.  Fresh queues, then the draws again the plain way, sized from the counters
    ctx->h_ctr_valid = false;
    ctx->late_resolve_done = false;
    ++ctx->last_fallback;
    HIP_TRY(ctx, hipMemsetAsync((char *)ctx->d_ctr + offsetof(DevCounters, n_tasks), 0,
                                offsetof(DevCounters, inv_row_min) - offsetof(DevCounters, n_tasks), ch.stream));
    HIP_TRY(ctx, hipMemsetAsync((char *)ctx->d_ctr + offsetof(DevCounters, fallback), 0, sizeof(unsigned long long), ch.stream));
    DrawArgs db{};
    init_draw_args(ctx, db);
    if ((rc = enqueue_chunk_draws(ctx, 0, db, 3))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ch.stream));
    int rounds = 3;
    if (ch.n_items) { if ((rc = finish_rounds(ctx, 0, db, 3, &rounds))) return rc; }
    ch.est_rounds = rounds;
    ctx->last_rounds = rounds;
  } else {
    const uint64_t n_items = c.work_count < ctx->V.n ? c.work_count : ctx->V.n;
    ch.have_est = true; ch.est_items = n_items; ch.est_sum = c.sum_samples; ch.est_rounds = (int)c.rounds_used;
    if (c.tries) { ctx->mean_iters = (double)c.newton_iters / (double)c.tries; ctx->parked_frac = (double)c.slow_solves / (double)c.tries; }
    int rounds = blind_rounds;
    if (extend) ctx->lean_ok = c.n_tasks[1] == 0u;       // (what the next pass may count on)
    if (lean) ++ctx->n_lean;
    // A pass whose first batches came from the model and left an item short all the same: the model's margin widens for the
    // passes that follow (the item is served by further rounds as ever); with the margin at its cap the context stops
    // betting on the lean tail until its camera set-up changes.
    const bool short_after_all = predicted && n_items && (c.n_tasks[1] != 0u || (lean && c.n_active[blind_rounds & 1] != 0u));
    // (an abandoned pass: what it still needed is counted, not done -- nobody can see its frame any more)
    const bool open_end = n_items && ((lean && c.n_tasks[1] != 0u) || c.n_active[blind_rounds & 1] != 0);
    if (open_end && !can_fix) {
      ++ctx->totals.abandoned_incomplete;
      if (ctx->notes.size() < 8) ctx->notes.push_back("abandoned before its end was looked at: the first accept had scheduled another round (lean tail's bet lost)");
    } else if (open_end) {
      t.did_more = true;
    }
    if (short_after_all) {
      // (a pass that loses only now and then keeps betting: the margin comes back down after 16 passes without a loss)
      ctx->bm_since_loss = 0;
      if (ctx->bm_margin16 >= 4u) ctx->lean_ok = false;
      else ctx->bm_margin16 += 1u;
    }
    if (predicted && !short_after_all && ++ctx->bm_since_loss >= 16u && ctx->bm_margin16 > 0u) { --ctx->bm_margin16; ctx->bm_since_loss = 0; }
    if (!can_fix && open_end) {
      if (lean) ++ctx->n_lean_lost;
      rounds = (int)c.rounds_used + 1;
    } else
    if (lean && n_items && c.n_tasks[1] != 0u) {
      // The lean tail's bet was lost: the first accept scheduled tasks, the accept behind it did nothing.  The round the
      // ordinary way -- its solves (the queue is complete), their stragglers, the accept that was held back -- then whatever
      // rounds follow.  (Rare: an item whose estimate in the solve kernel was too kind; ~0.3 ms.)
      ctx->h_ctr_valid = false;
      ctx->late_resolve_done = false;
      da.parity = 1; da.round = 1;
      da.producers_done = nullptr; da.producers_total = 0;
      da.slow_live = 0; da.slow_indirect = 0; da.slow_q = -1; da.slow_round = -1; da.slow_close = 1;
      da.emit_live = 0; da.lean_gate = 0; da.no_reset = 1;
      {
        DrawArgs dr = da;       // (its parked solves go to the upper half of the records: the lower half holds the first round's results)
        if (dr.slow) { dr.slow = slow_base + slow_cap_all / 2u; dr.slow_cap = slow_cap_all - slow_cap_all / 2u; }
        launch_solve(ctx, dr, ch.stream, (unsigned)ctx->num_cu);
      }
      hipLaunchKernelGGL(reset_round_kernel, dim3(1), dim3(1), 0, ch.stream, ctx->d_ctr, 0u, 1u);
      hipLaunchKernelGGL(accept_kernel<2>, dim3(accept_blocks), dim3(256), 0, ch.stream, da);
      HIP_TRY(ctx, hipGetLastError());
      da.no_reset = 0;
      if ((rc = finish_rounds(ctx, 0, da, 2, &rounds))) return rc;
      if (rounds > ch.est_rounds) ch.est_rounds = rounds;
      ++ctx->n_lean_lost;
    } else
    if (n_items && c.n_active[blind_rounds & 1] != 0) {
      ctx->h_ctr_valid = false;
      ctx->late_resolve_done = false;
      if ((rc = finish_rounds(ctx, 0, da, blind_rounds, &rounds))) return rc;
      if (rounds > ch.est_rounds) ch.est_rounds = rounds;
      if (lean) ++ctx->n_lean_lost;
    } else if (lean) {
      rounds = 1;       // one round of solves: the accept behind the first one only waited for that round's parked solves
    }
    ctx->last_rounds = rounds;
  }
  ctx->have_total_est = true;
  ctx->est_items_total = ch.est_items; ctx->est_sum_total = ch.est_sum; ctx->est_rounds_total = ch.est_rounds;
  // (should the next pass run chunked, its chunks look at their scans first: this pass knows nothing about them)
  for (int ci = 0; ci < C; ++ci) ctx->chunks[ci].have_est = false;
  *streamed = true;
  return LENTIL_OK;
}


static int redistribute_streamed(lentil_hip_ctx *ctx, bool *streamed, bool *deferred) {
  *streamed = false;
  *deferred = false;
  const lentil_params &P = ctx->P;
  if (!ctx->stream_mode || P.cameraType != LENTIL_POLYNOMIAL_OPTICS || !ctx->have_total_est || ctx->V.n == 0)
    return LENTIL_OK;
  if (ctx->V.n > 0xFFFFFFF0ull) return LENTIL_OK;
  if (ctx->probe_fn) return LENTIL_OK;       // (occlusion probes: the host answers between a round's solves and its accept -- the round-by-round form)
  // Only into a frame that has been cleared since its last pass (every caller's order: clear, redistribute, resolve): a
  // streamed pass whose waves give up waiting after draws have been accepted is recovered by wiping the frame and running
  // the pass again, which must not cost an earlier pass's sums.  A second pass into the same frame takes the chunked form,
  // whose kernels never wait for one another.
  if (!ctx->cleared_since_pass) return LENTIL_OK;
  // A context whose resident waves have given up waiting before (250 ms each time, then the redo) stops trying: at once
  // where lentil_hip_create found its streams sharing hardware queues (GPU_MAX_HW_QUEUES below 4: a kernel then sits behind
  // the one it waits for, every pass), after the third time anywhere else (a profiler that serialises kernels, a crowded GPU).
  if (ctx->n_stuck >= (ctx->streams_concurrent ? 3u : 1u)) return LENTIL_OK;
  // Streaming pays where the scan is most of the pass.  With many draws the chunked pass is ahead (highlight-heavy
  // frame: 114 ms against 135 ms; 15 M draws: 16.6 against 18.4 ms -- solve waves placed while the scan's are
  // resident keep running slower long after those have left, see launch_chunk_rounds), and so it is with extra
  // AOVs, whose scan kernel leaves the solve waves less room (config 4: 10.8 against 11.5 ms).
  // (LENTIL_STREAM_EXTRA=0: frames with extra AOVs take the chunked pass whatever their scan kernel)
  static const bool stream_extra = !(getenv("LENTIL_STREAM_EXTRA") && getenv("LENTIL_STREAM_EXTRA")[0] == '0');
  // (Round 3: ... or below one draw per 24 visits, whichever is more -- what streaming buys is the scan running beside the
  // solves, and a frame whose scan is long against its draws gains most: BASELINE config 5, 8K with 9.3 M draws, 12.9 ms
  // streamed against 16.3 chunked; 4K with 3.1 M draws 4.3 against 4.8 ms, with 6-48 M draws within +-5 % either way.)
  {
    const uint64_t by_visits = ctx->stream_below_set ? 0ull : ctx->V.n / 24ull;
    if (ctx->est_sum_total >= (ctx->stream_below > by_visits ? ctx->stream_below : by_visits)) return LENTIL_OK;
  }
  if (ctx->V.n_extra && !(stream_extra && dma_multi_applies(ctx))) return LENTIL_OK;
  lentil_hip_ctx::Chunk &ch = ctx->chunks[0];
  // One streamed pass per device at a time: the resident kernels of two of them could keep each other's scan off the chip.
  // A context that finds another one's streamed pass in flight does not wait for it: its pass runs in the chunked form,
  // whose kernels never wait for anything (LENTIL_STREAM_WAIT=1: wait, as rounds 2 did).
  static const bool wait_for_turn = getenv("LENTIL_STREAM_WAIT") && getenv("LENTIL_STREAM_WAIT")[0] == '1';
  DeviceTurn turn(ctx);
  if (!turn.take(wait_for_turn)) return LENTIL_OK;
  DrawArgs da{};
  init_draw_args(ctx, da);
  ++ctx->n_streamed;
  g_stat_streamed.fetch_add(1, std::memory_order_relaxed);
  da.inject_stall = (ctx->inject_stall_at > 0 && ctx->n_streamed == (uint64_t)ctx->inject_stall_at) ? 1 : 0;
  const uint64_t nch = (uint64_t)da.n_channels;
  uint64_t items = 2 * ctx->est_items_total + 4096;
  if (items > ctx->V.n) items = ctx->V.n;
  const uint64_t sum = 2 * ctx->est_sum_total + (1ull << 20);
  const uint64_t units = chunk_units(P, nch, sum, items);
  if (units > ctx->max_pool_units) return LENTIL_OK;       // sub-batches: the chunked pass knows how
  int rc;
  if ((rc = size_chunk_buffers(ctx, ch, items, units))) return rc;
  bind_chunk_buffers(ch, da);
  ScanPlan plan;
  if ((rc = plan_scan(ctx, plan))) return rc;
  da.F = ctx->F;                // (plan_scan decides where the direct sums go and whether splats are flagged)
  // the wipe clear_frame left on the chunk stream (clear_offstream) rides beside the scan only where the scan leaves the splat
  // accumulators alone: own sums to FrameDev::dir, splats flagged
  if (ctx->clear_pending && !(ctx->F.dir && ctx->F.touched) && (rc = join_clear(ctx))) return rc;
  ch.tile_begin = 0; ch.tile_end = plan.n_tiles;
  ch.v_begin = 0; ch.v_end = ctx->V.n;
  for (int ci = 1; ci < ctx->n_chunks; ++ci) {
    lentil_hip_ctx::Chunk &o = ctx->chunks[ci];
    o.tile_begin = o.tile_end = plan.n_tiles; o.v_begin = o.v_end = ctx->V.n; o.n_items = 0;
  }
  ctx->epoch = (ctx->epoch + 1u) & 0x3FFFFFu;
  if (ctx->epoch == 0u) {
    // the 22-bit tag has come round: wipe what older passes left in the queues
    ctx->epoch = 1u;
    HIP_TRY(ctx, hipMemsetAsync(ch.tasks[0], 0, ch.task_cap * sizeof(Task), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ch.tasks[1], 0, ch.task_cap * sizeof(Task), ctx->stream));
    if (ctx->d_ranges) HIP_TRY(ctx, hipMemsetAsync(ctx->d_ranges, 0, ctx->range_cap * sizeof(uint64_t), ctx->stream));
    if (ctx->d_ready) HIP_TRY(ctx, hipMemsetAsync(ctx->d_ready, 0, ctx->ready_cap * sizeof(uint64_t), ctx->stream));
    if (ch.slow) HIP_TRY(ctx, hipMemsetAsync(ch.slow, 0, ch.slow_cap * sizeof(SlowRec), ctx->stream));
    if (ctx->d_ext_q) HIP_TRY(ctx, hipMemsetAsync(ctx->d_ext_q, 0, ctx->ext_q_cap * sizeof(Task), ctx->stream));
  }
  const uint32_t retries = (uint32_t)(P.vignetting_retries < 0 ? 0 : P.vignetting_retries);
  const bool few = ctx->est_sum_total < ctx->slow_below;
  // How long a resident wave waits for its queue slot before it declares the pass stuck (the host then redoes the pass the
  // chunked way).  A wave may rightly wait for a whole scan -- its slot gets its end marker when the scan ends --, so: 250 ms
  // while nothing is known, else sixteen times the longest pass this context has seen, at least 30 ms (round 5: a stall,
  // rare as it is, then costs tens of milliseconds and not a quarter of a second; LENTIL_STUCK_MS overrides).
  uint64_t stuck_ticks = 0;
  {
    static const double forced_ms = getenv("LENTIL_STUCK_MS") ? atof(getenv("LENTIL_STUCK_MS")) : 0.0;
    // (the estimate comes from earlier passes: it holds for a pass no larger than the one that set it -- more visits, or a
    // quarter more draws expected, and nothing is known again: 250 ms -- and every time-out this context has hit doubles it,
    // so that a slow box, a shared GPU or a profiler does not turn into a run of false stalls, each a wipe and a chunked redo)
    double ms = 250.0;
    if (ctx->longest_pass_ms > 0.0 && ctx->V.n <= ctx->longest_pass_visits &&
        ctx->est_sum_total <= ctx->longest_pass_sum + ctx->longest_pass_sum / 4) {
      ms = 16.0 * ctx->longest_pass_ms * (double)(1u << (ctx->n_stuck < 4 ? ctx->n_stuck : 4));
      if (ms < 30.0) ms = 30.0;
      if (ms > 250.0) ms = 250.0;
    }
    if (forced_ms > 0.0) ms = forced_ms;
    stuck_ticks = (uint64_t)(ms * 1.0e5);
  }

  PublishArgs pa{};
  pa.P = P;
  pa.V = ctx->V;
  StreamPub &pub = pa.S;
  pub.epoch = ctx->epoch;
  pub.n_channels = (uint32_t)nch;
  pub.retries = (int32_t)retries;
  pub.extra_num = ctx->est_sum_total < ctx->extra_below ? ctx->extra_num : 0u;
  pub.extra_const = ctx->est_sum_total < ctx->extra_below ? ctx->extra_const : 0u;
  pub.item_cap = (uint32_t)(ch.item_cap < 0xFFFFFFF0ull ? ch.item_cap : 0xFFFFFFF0ull);
  pub.task_cap = da.task_cap;
  pub.pool_cap = da.pool_cap;
  pub.hdr = ch.hdr; pub.prog = ch.prog; pub.active0 = ch.active[0]; pub.tasks0 = ch.tasks[0];
  // range queue: one record per flush of a wave queue (at most one per 64 visits, plus one per wave and tile)
  {
    const uint64_t need = ctx->V.n / 64 + 2 * plan.n_tiles + 65536;
    if (need > ctx->range_cap) {
      if ((rc = grow(ctx, &ctx->d_ranges, need))) return rc;
      ctx->range_cap = need;
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_ranges, 0, need * sizeof(uint64_t), ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  plan.sa.ranges = ctx->d_ranges;
  plan.sa.range_cap = (uint32_t)(ctx->range_cap < 0xFFFFFFF0ull ? ctx->range_cap : 0xFFFFFFF0ull);
  plan.sa.epoch = ctx->epoch;
  plan.sa.end_ranges = (uint32_t)ctx->publish_waves;
  plan.sa.flush_each_tile = ctx->est_items_total < (1u << 16) ? 1u : 0u;
  pa.ctr = ctx->d_ctr;
  pa.stuck_ticks = stuck_ticks;
  da.stuck_ticks = stuck_ticks;
  pa.work = ctx->d_work;
  pa.work_cap = ctx->V.n;
  pa.ranges = ctx->d_ranges;
  pa.range_cap = plan.sa.range_cap;

  da.ctr = ctx->d_ctr;
  da.retries = (int32_t)retries;
  da.work = ctx->d_work;
  da.work_cap = ctx->V.n;
  da.blind = 0u;
  da.n_items = pub.item_cap;
  da.parity = 0; da.round = 0;
  da.epoch = ctx->epoch;
  if (!few) da.slow = nullptr;              // parking is for passes with few draws (DrawArgs::slow_below)

  unsigned scan_blocks = 0;
  const auto pass_t0 = std::chrono::steady_clock::now();
  ht_mark(ctx, "first_launch");
  const bool calibrates_now = ctx->predict && !ctx->bm_valid;       // (this pass's host time holds the calibration kernel's)
  // the first-batch model's calibration, should the camera set-up have changed: on the main stream, ahead of the event the
  // publishers (who read the table) wait for
  if (ctx->predict && !ctx->extend && nch == 1 && (rc = ensure_batch_model(ctx))) return rc;
  // (the scan's start for lentil_hip_last_timing: the host work since the pass began -- sizing, the plan -- is not the kernel's)
  HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  {
    static const bool a_first = getenv("LENTIL_A_FIRST") && getenv("LENTIL_A_FIRST")[0] == '1';
    if (a_first && ctx->streams_concurrent && ctx->slow_live) {
      hipLaunchKernelGGL(wait_waves_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_ctr,
                         (uint32_t)ctx->num_cu * (uint32_t)ctx->stream_blocks * 4u, (uint64_t)30000);      // (0.3 ms at most)
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  if ((rc = launch_scan(ctx, plan, ch, ctx->d_ctr, &scan_blocks, true))) return rc;
  ctx->last_scan_launches = 1;
  HIP_TRY(ctx, hipEventRecord(ch.scanned, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->scans_done, ctx->stream));
  // cryptomatte AOVs: the adds of the visits that stay in their pixel need the scan's work lists and nothing else of the
  // pass -- beside the draws, on the spare stream, where the runtime has one (LENTIL_CRYPTO_OVERLAP=0: after the pass)
  static const bool crypto_overlap = !(getenv("LENTIL_CRYPTO_OVERLAP") && getenv("LENTIL_CRYPTO_OVERLAP")[0] == '0');
  if (ctx->crypto && ctx->aux_stream && crypto_overlap) {
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->scans_done, 0));
    if ((rc = crypto_enqueue_direct(ctx, ctx->aux_stream))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_crypto, ctx->aux_stream));
    ctx->crypto_direct_enqueued = true;
  }

  // The publishers and A, resident beside the scan (the counters they poll were cleared by the memset ahead of
  // ev[0]).  Submitted AFTER what they wait for -- the scan, then the publishers: should the streams share a
  // hardware queue, each finds its producer ahead of it there.
  // As many resident solve blocks per CU as leave a scan block its LDS.  (Round 6: the table interpreter's blocks hold 52 KB each --
  // two of them and an 80 KB scan_dma_kernel block do not fit a CU's 160 KB, and where the dispatcher placed the solve blocks
  // first no scan block ever found room: the resident waves waited for a scan that could not start until a publisher gave up,
  // 250 ms, then the redo.  That was the "odd stalled pass" of small frames run through the interpreter -- every test that runs
  // its second pass with lentil_hip_set_lens_mode(ctx, 1) --, found when the library began to count its stalls.  plan_scan only
  // budgeted for scan_dma2_kernel.)
  unsigned a_per_cu = (unsigned)ctx->stream_blocks;
  {
    const size_t solve_lds = solve_block_lds(ctx);
    while (a_per_cu > 1u && plan.lds + 512u + (size_t)a_per_cu * solve_lds > 160u * 1024u) --a_per_cu;
  }
  unsigned a_blocks = (unsigned)ctx->num_cu * a_per_cu;
  // CUs the scan leaves alone (scan_cus_pct) have registers for one more resident solve block
  if (ctx->last_scan_skipped && a_per_cu == 2u) a_blocks += ctx->last_scan_skipped;
  (void)scan_blocks;
  // Live straggler queue: solve_slow_kernel is launched behind the publishers (who end with the scan) and takes the parked
  // solves as they come, one wave per CU.  (Round 3, from the timeline: its waves are placed as the first solve waves
  // leave -- the idle ones do at once when the publishers' end markers arrive --, not in the registers the scan gives
  // back: a wave's registers are one contiguous range.  The second instance B, three waves per block so that a SIMD per
  // CU kept room for a straggler wave, never got a task and is off: LENTIL_SOLVE_B.)
  const bool live = ctx->slow_live && da.slow != nullptr && P.cameraType == LENTIL_POLYNOMIAL_OPTICS;
  int blind_rounds = ctx->est_rounds_total < 2 ? 2 : (ctx->est_rounds_total > 6 ? 6 : ctx->est_rounds_total);
  if (const char *e = getenv("LENTIL_BLIND_ROUNDS")) { blind_rounds = atoi(e); if (blind_rounds < 1) blind_rounds = 1; if (blind_rounds > 8) blind_rounds = 8; }
  // Second round beside the first accept: the accept kernel hands out the next round's tasks as it goes (tagged slots,
  // end markers from its last block), a kStream solve kernel -- one block per CU, which fits beside four accept blocks
  // -- takes them as they come, the straggler kernel beside both.  Launched AFTER the accept, so that a profiler that
  // serialises kernels runs them in an order that completes.
  // (Round 4: also in passes that park nothing -- more draws than LENTIL_SLOW_BELOW, BASELINE config 5 on one GPU: 4 560 items
  // x 2 048 draws --, whose first accept takes 0.65 ms and whose second round used to wait for all of it: 11.7 -> see DESIGN
  // section 5.  No straggler kernels there, just the accept feeding the resident second-round solves.  LENTIL_OVERLAP_HEAVY=0: off.)
  static const bool overlap_heavy_ok = !(getenv("LENTIL_OVERLAP_HEAVY") && getenv("LENTIL_OVERLAP_HEAVY")[0] == '0');
  const bool overlap_plain = !live && overlap_heavy_ok && da.slow == nullptr && nch == 1 && P.cameraType == LENTIL_POLYNOMIAL_OPTICS;
  const bool overlap = (live || overlap_plain) && ctx->overlap_rounds && blind_rounds >= 2;
  // ... and the first accept does not wait for the first round's stragglers either (accept_item<1> / <2>): one
  // straggler queue and one solve_slow_kernel launch for both rounds, closed by the second round's solve kernel.
  const bool decoupled = live && overlap && ctx->decouple && nch == 1;
  // (A queue and a solve_slow_kernel launch per round, as ever: ONE kernel for both rounds would wait for end markers from
  // kernels submitted after it -- the first accept, the second round's solves -- and where two of the pass's streams share
  // a hardware queue, the default with the runtime's 4, those sit behind it in that queue: 250 ms, then the chunked redo.
  // The second round parks into the upper half of the record buffer, its kernel follows the first round's on their stream.)
  SlowRec *const slow_base = da.slow;
  const uint32_t slow_cap_all = da.slow_cap;
  da.slow_crowd_stays = ctx->crowd_stays_first;
  da.unknown_credit = ctx->unknown_credit;
  // (one straggler wave per CU; LENTIL_SLOW_WAVES_PER_CU, up to 4 -- measured on config 4, whose rounds end in hundreds
  // of parked solves at once: 9.15 / 9.18 / 9.35 / 9.46 ms with 1 / 2 / 3 / 4, the waves take from the solve kernel)
  const int slow_per_cu = ctx->slow_waves_per_cu;
  if (ctx->parked_frac > 1.0 / 256.0) ctx->park_dry_seen = true;
  const bool dry_only = ctx->park_dry_only >= 0 ? ctx->park_dry_only != 0 : ctx->park_dry_seen;
  // (Round 6: four per CU where parked solves are the outliers they are meant to be -- beauty-only frames, the live queue.  The
  // straggler kernel had become what the pass ends on: 2 300 parked solves of a headline frame, 27 000 iterations at ~3 us each,
  // are 320 us of work for 256 waves and half of it was still to do when the solve kernel's last wave left; with 1 024 waves
  // it ends with the solve kernel but for the solves that run all 100 iterations.  2.011 -> 1.989 ms, eight interleaved runs of
  // 60 steps each, gpurun_out/r06s05.  Config 4 parks dry waves' lanes only and keeps one.)
  const uint32_t slow_default = (nch == 1 && ctx->V.n_extra == 0 && !dry_only) ? 4u : 1u;
  const uint32_t slow_waves_all = (uint32_t)ctx->num_cu * (slow_per_cu >= 1 && slow_per_cu <= 4 ? (uint32_t)slow_per_cu : slow_default);
  if (decoupled) { da.slow_indirect = 1; da.slow_cap = slow_cap_all / 2u > slow_waves_all ? slow_cap_all / 2u - slow_waves_all : 0u; }      // (its end markers stay below the upper half)
  static const int b_threads_env = getenv("LENTIL_SOLVE_B_THREADS") ? atoi(getenv("LENTIL_SOLVE_B_THREADS")) : 0;
  const unsigned b_threads = (b_threads_env == 64 || b_threads_env == 128 || b_threads_env == 192 || b_threads_env == 256) ? (unsigned)b_threads_env
                                                                                                                         : (live ? 192u : 256u);
  unsigned b_blocks;
  {
    int b_per_cu = live ? 3 - ctx->stream_blocks : ctx->solve_max_blocks - ctx->stream_blocks;
    if (b_per_cu < 1) b_per_cu = 1;
    const uint64_t want = (nch * (ctx->est_sum_total / 64 + ctx->est_items_total) + 3) / 4;
    uint64_t b = (uint64_t)ctx->num_cu * (uint64_t)b_per_cu;
    if (want < b) b = want < 1 ? 1 : want;
    b_blocks = (unsigned)b;
    // (Measured, round 3: with the straggler kernel's wave on one of a CU's SIMDs the three-wave blocks of this launch are
    // not placed before the first launch's blocks leave -- zero iterations in every pass looked at; the launch then only
    // stands between the first launch's end and the accept.)
    if (live && !ctx->solve_b) b_blocks = 0;
  }
  da.slow_live = live ? 1 : 0;
  // A live queue takes every solve that reaches slow_at iterations -- outliers, by the measure of the previous pass.  With
  // a lens where such solves are not outliers (the petzval table: 3 % of all solves, 62 000 a frame, each a whole wave
  // of the straggler kernel: 12 ms a frame against 10 chunked) only waves running dry park, and only their last lanes.
  // (It stays that way for the lens: a pass that parks dry waves' lanes only says nothing about what a live queue would get.)
  da.slow_dry_only = dry_only ? 1 : 0;
  {
    // (LENTIL_PARK_AFTER_SCAN=0: the first round parks from the start of the pass; 2 486 parked solves per headline pass
    // instead of 1 070, 2.33 against 2.31 ms)
    const char *e = getenv("LENTIL_PARK_AFTER_SCAN");
    da.slow_after_producers = (e ? e[0] != '0' : true) ? 1 : 0;
  }
  da.slow_waves = live ? slow_waves_all : 0u;
  da.producers_done = &ctx->d_ctr->publishers_done;
  da.producers_total = (uint32_t)ctx->publish_waves;
  pa.end_tasks = a_blocks * 4u + b_blocks * (b_threads / 64u);       // every first-round solve wave may hold one ticket past the last task
  // Extension (ItemLive, lentil_kernels.h): the first round's solve kernel appends the batches its items still need
  const bool extend = ctx->extend && decoupled && nch == 1 && ctx->chain_streams;
  // First batches from the lens and the frame (lentil_batch_model.h): every item is published with the traces it is expected
  // to need, so that the first accept finds nothing to schedule and the pass can do without a second round (lean tail, below)
  // (Not for items with very many draws each -- BASELINE config 5's 2 048: their first accept is long, 0.3-0.65 ms, and the
  // second round that runs beside it is all but free, while its traces inside the first round are throughput; the bands of
  // that frame, each alone on one GPU, took 5-19 % longer with the lean tail: profiles/r05_emulated_bands.txt.)
  const bool few_draws_per_item = ctx->est_items_total == 0 || ctx->est_sum_total / ctx->est_items_total <= ctx->predict_max_draws;
  const bool predict = ctx->predict && !extend && decoupled && nch == 1 && ctx->chain_streams && few_draws_per_item;
  if (predict && ctx->bm_valid) pub.model = batch_model_dev(ctx);        // (calibrated ahead of the scan, above)
  const bool predicted = predict && pub.model.land != nullptr;
  if (extend) {
    if (ch.item_cap > ctx->live_cap) {
      if ((rc = grow(ctx, &ctx->d_live, ch.item_cap))) return rc;
      ctx->live_cap = ch.item_cap;
    }
    if (ch.task_cap > ctx->ext_q_cap) {
      if ((rc = grow(ctx, &ctx->d_ext_q, ch.task_cap))) return rc;
      ctx->ext_q_cap = ch.task_cap;
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_ext_q, 0, ch.task_cap * sizeof(Task), ctx->stream));       // (slots are told by their tag)
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    const unsigned keepers = a_blocks < (unsigned)ctx->num_cu ? a_blocks : (unsigned)ctx->num_cu;       // one block per CU stays
    pub.live = ctx->d_live;
    pub.ext_q = ctx->d_ext_q;
    pub.ext_keepers = keepers * 4u;
    da.live = ctx->d_live;
    da.ext_q = ctx->d_ext_q;
    da.ext_keeper_blocks = keepers;
    da.ext_end_tasks = keepers * 4u;
    da.ext_slack = 4u;
  }
  // The lean tail (below): known before anything is launched -- the solve and straggler kernels count parked solves per item for
  // accept_kernel<3> (DrawArgs::item_ready) in such a pass.
  // (LENTIL_INJECT_STALL stalls the second round's resident solve waves: that pass keeps its second round in flight)
  const bool lean_pass = decoupled && ctx->chain_streams && (extend || predicted) && ctx->lean_tail && ctx->lean_ok && blind_rounds <= 2 &&
                         !da.inject_stall;
  // Round 6: the first accept takes the items whose parked solves are through, whole, and leaves the others to the accept behind
  // the stragglers (accept_kernel<3>; LENTIL_READY_ACCEPT=0: round 5's pair of accepts, the first splatting what is certain of
  // every item, the second replaying nearly every item).  For frames the wide walk serves: <= 64 retries, records of <= 64 floats.
  bool ready_accept = false;
  {
    const bool off = !ctx->ready_accept;
    uint32_t add_floats = 1;
    for (uint32_t k = 0; k < ctx->F.n_aovs; ++k) if (!(ctx->F.closest_mask & (1u << k))) add_floats += 4;
    ready_accept = lean_pass && !off && live && retries <= kAcceptWinRetries && !da.accept_narrow && add_floats <= 64u && nch == 1;
  }
  da.item_ready = ready_accept ? 1 : 0;
  // ... and beside the solves (accept_kernel<4>): the queue of completed items, a slot per item
  const bool early = ready_accept && ctx->early_accept && ctx->slow1_stream != nullptr && !b_blocks;
  if (early) {
    if (ch.item_cap > ctx->ready_cap) {
      if ((rc = grow(ctx, &ctx->d_ready, ch.item_cap))) return rc;
      ctx->ready_cap = ch.item_cap;
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_ready, 0, ch.item_cap * sizeof(uint64_t), ctx->stream));       // (slots are told by their tag)
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    da.early_accept = 1;
    da.ready_q = ctx->d_ready;
    da.ready_cap = (uint32_t)(ctx->ready_cap < 0xFFFFFFF0ull ? ctx->ready_cap : 0xFFFFFFF0ull);
  }
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->pub_stream, ctx->ev[0], 0));
  hipLaunchKernelGGL(publish_kernel, dim3((unsigned)ctx->publish_waves), dim3(64), 0, ctx->pub_stream, pa);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->pub_done, ctx->pub_stream));
  if (live && early) {
    // the stragglers on a stream of their own, the first accept behind the publishers on theirs: both start when the scan's
    // registers and LDS are free, the accept takes the completed items as the solve kernel (launched below) delivers them
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->slow1_stream, ctx->pub_done, 0));
    hipLaunchKernelGGL(solve_slow_kernel, dim3(da.slow_waves), dim3(64), coop_lds_bytes(ctx->hlens.n_terms), ctx->slow1_stream, da);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_slow, ctx->slow1_stream));
    if (ctx->clear_pending) HIP_TRY(ctx, hipStreamWaitEvent(ctx->pub_stream, ctx->ev_clear, 0));      // (it splats into what clear_frame is wiping)
    DrawArgs d0 = da;
    d0.emit_live = 0; d0.lean_defer = 1;
    d0.end_tasks = (uint32_t)ctx->num_cu * 4u;
    hipLaunchKernelGGL(accept_kernel<4>, dim3((unsigned)ctx->num_cu * (unsigned)ctx->early_blocks), dim3(256), 0, ctx->pub_stream, d0);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_acc1, ctx->pub_stream));
  } else if (live) {
    // (behind the publishers on their stream: they end with the scan, whose registers this kernel's waves need)
    hipLaunchKernelGGL(solve_slow_kernel, dim3(da.slow_waves), dim3(64), coop_lds_bytes(ctx->hlens.n_terms), ctx->pub_stream, da);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_slow, ctx->pub_stream));
  }
  HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->ev[0], 0));
  da.instance = 0;
  launch_solve_po<true>(ctx, da, ch.stream, a_blocks);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ch.done, ch.stream));

  const uint64_t acc_max = (uint64_t)ctx->num_cu * (uint64_t)(ctx->accept_stream_blocks < 1 ? 1 : ctx->accept_stream_blocks);
  const uint64_t acc_want = ctx->est_items_total + ctx->est_items_total / 4 + 1;
  const unsigned accept_blocks = (unsigned)(acc_want > acc_max ? acc_max : acc_want);
  hipStream_t tail = ctx->stream;       // the stream the pass's last kernels and its counter read-back are on
  bool lean = false;                    // lean tail (below): no second round in flight
  if (decoupled && ctx->chain_streams) {
    // ---- the decoupled pass with its chain of kernels laid along streams: a dependency that crosses streams costs
    // 40-90 us (event, barrier packet, a queue waking up) where a kernel behind its predecessor on ONE stream costs ~2:
    //   chunk stream : A -> first accept                      (the accept starts as the last first-round solve ends)
    //   main stream  : scan -> second round's solves          (released by what the first accept waited for)
    //   straggler st.: publishers -> stragglers of round one -> of round two -> second accept -> later rounds' accepts
    //                  -> the resolve's second half -> counter read-back       (each behind the kernel it ends last)
    // The only cross-stream waits left on the critical path release kernels that then sit waiting for tasks anyway.
    // (Round 4: everything from the second round's stragglers on sits on THEIR stream -- the second accept follows the kernel
    // that ends last, solve_slow_kernel of round two, in-stream; what else it needs -- the second round's solves, the first
    // accept, the first round's stragglers -- has ended before that kernel does, so those waits find their events fired.
    // On the publishers' stream the accept came ~60 us after the stragglers' end: three cross-stream waits in a row.)
    hipStream_t ps = ctx->slow1_stream;
    if (b_blocks) {
      // LENTIL_SOLVE_B=1: a third solve block per CU (three waves: one SIMD stays the straggler wave's) behind the scan on its
      // stream -- it starts when the scan's waves have left, finds the queue complete and never waits; the first accept waits
      // for it as for A
      DrawArgs db = da;
      db.instance = 1;
      launch_solve_po<true>(ctx, db, ctx->stream, b_blocks, b_threads);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipEventRecord(ctx->ev_b, ctx->stream));
      HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->ev_b, 0));
    }
    HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->pub_done, 0));       // (both long past when A ends)
    HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->scans_done, 0));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_round, ch.stream));
    // (lean tail: no next round's solve waves share the CUs with the first accept -- a block per item, as many as fit)
    unsigned accept1_blocks = accept_blocks;
    if (ready_accept) {
      // accept_kernel<3> waits for nothing and is what stands between the solve kernel's end and the touched groups' resolve:
      // a block per item where that many fit (4 per CU: 4 x 20.7 KB of LDS beside a straggler wave's 10, 16 waves of <= 96 registers)
      const uint64_t m = (uint64_t)ctx->num_cu * (uint64_t)ctx->ready_blocks;
      accept1_blocks = (unsigned)(acc_want > m ? m : acc_want);
    } else if (lean_pass) {
      // (measured, same box, headline: 2 / 4 / 5 blocks per CU -> 2.00 / 2.00-2.24 / 2.08-2.30 ms: with more blocks than the
      // stragglers' LDS leaves room for, every third run sits in a mode 0.25 ms slower.  The knob stays; the default is round 4's.)
      static const int lean_blocks = getenv("LENTIL_ACCEPT_LEAN_BLOCKS") ? atoi(getenv("LENTIL_ACCEPT_LEAN_BLOCKS")) : 0;
      if (lean_blocks >= 1) {
        const uint64_t m = (uint64_t)ctx->num_cu * (uint64_t)(lean_blocks > 6 ? 6 : lean_blocks);
        const unsigned b = (unsigned)(acc_want > m ? m : acc_want);
        if (b > accept1_blocks) accept1_blocks = b;
      }
    }
    const bool resolves_early = ctx->early_resolve && ctx->F.dir && ctx->F.touched && ctx->n_chunks >= 2 && !ctx->comm && !ctx->closest_deferred;
    // Lean tail, an option: the frame's resolve does not wait for the first accept.  The whole frame is resolved behind the scan
    // -- the pixels' own sums are complete then, the HBM is idle and the solve waves do not need it --, the groups of pixels
    // the first accept's draws land in are resolved again behind it (about half of a headline frame's groups: 86 us where
    // the whole frame takes 130), the few groups of the last accept once more at the end.
    // (LENTIL_RESOLVE_AFTER_SCAN=1; off by default: measured on the headline, same box, 2.00-2.01 ms with it and 1.98-2.00
    // without -- the resolve is not what the pass ends on, the stragglers and the accept behind them are)
    // (Round 6, with accept_kernel<3>: on -- the accept behind the stragglers now has a few items where it had nearly all of them,
    // so the whole-frame resolve would be what the pass ends on; behind the scan it is under the solves, and behind the first
    // accept only the groups its draws reached are left.  LENTIL_RESOLVE_AFTER_SCAN=0 / 1 decides whatever the accept.)
    const bool resolve_after_scan = lean_pass && resolves_early && (ctx->resolve_after_scan >= 0 ? ctx->resolve_after_scan == 1 : ready_accept);
    if (resolve_after_scan) {
      hipStream_t rs = ctx->chunks[1].stream;
      if (ctx->clear_pending) HIP_TRY(ctx, hipStreamWaitEvent(rs, ctx->ev_clear, 0));      // (it reads the accumulators clear_frame is wiping on the chunk stream)
      HIP_TRY(ctx, hipStreamWaitEvent(rs, ctx->scans_done, 0));
      if ((rc = launch_resolve_half(ctx, rs, 0u))) return rc;
    }
    if (early) {
      // (the first accept is at work beside the solve kernel already -- accept_kernel<4>, above; what follows on this stream follows it)
      HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->ev_acc1, 0));
    } else {
      DrawArgs d0 = da;
      d0.emit_live = lean_pass ? 0 : 1;       // (lean tail: nobody is waiting for tasks)
      d0.lean_defer = lean_pass ? 1 : 0;
      d0.end_tasks = (uint32_t)ctx->num_cu * 4u;
      if (ready_accept) hipLaunchKernelGGL(accept_kernel<3>, dim3(accept1_blocks), dim3(256), 0, ch.stream, d0);
      else hipLaunchKernelGGL(accept_kernel<1>, dim3(accept1_blocks), dim3(256), 0, ch.stream, d0);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipEventRecord(ctx->ev_acc1, ch.stream));
    }
    if (resolves_early) {
      hipStream_t rs = ctx->chunks[1].stream;
      HIP_TRY(ctx, hipStreamWaitEvent(rs, ctx->ev_acc1, 0));
      if ((rc = launch_resolve_half(ctx, rs, resolve_after_scan ? 1u : 0u))) return rc;
      HIP_TRY(ctx, hipEventRecord(ctx->ev_res, rs));
      ctx->early_resolve_pending = true;
    }
    // Lean tail (extension, ItemLive): the first round's solve kernel has appended what its items needed, so the first accept
    // is expected to schedule nothing -- no second round's solve and straggler kernels, no waiting for them: the accept of the
    // items that met parked solves follows the first accept on its stream, behind the first round's stragglers.  Should the
    // first accept have scheduled tasks after all, that accept does nothing (DrawArgs::lean_gate) and the round is run below.
    lean = lean_pass;
    if (lean) {
      hipStream_t ls = ch.stream;
      da.parity = 1; da.round = 1;
      HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->ev_slow, 0));      // the first round's stragglers
      hipLaunchKernelGGL(reset_round_kernel, dim3(1), dim3(1), 0, ls, ctx->d_ctr, 0u, 1u);
      {
        DrawArgs d2 = da;
        d2.lean_gate = 1;
        d2.early_accept = 0;          // (every kernel that wrote its results has ended: plain loads)
        d2.slow_indirect = 0; d2.slow_cap = slow_cap_all; d2.slow_live = 0;
        hipLaunchKernelGGL(accept_kernel<2>, dim3(accept_blocks), dim3(256), 0, ls, d2);
      }
      HIP_TRY(ctx, hipGetLastError());
      da.slow_indirect = 0; da.slow_cap = slow_cap_all; da.slow_live = 0;
      da.early_accept = 0;          // (whatever the host launches behind this pass's kernels reads results they have finished writing)
      if (ctx->early_resolve_pending) {
        HIP_TRY(ctx, hipStreamWaitEvent(ls, ctx->ev_res, 0));
        if ((rc = launch_resolve_half(ctx, ls, 2u))) return rc;
        ctx->late_resolve_done = true;
      }
      tail = ls;
    } else {
    const char *overlap_env = getenv("LENTIL_OVERLAP_ACCEPT");      // (read per pass: the tests switch it)
    const bool overlap_accept = !(overlap_env && overlap_env[0] == '0');
    for (int round = 1; round < blind_rounds; ++round) {
      da.parity = round & 1; da.round = round;
      DrawArgs d1 = da;
      d1.slow_after_producers = 0;      // (its straggler kernel runs beside it from the start)
      d1.slow_crowd_stays = ctx->crowd_stays_later;
      d1.slow_indirect = 0;
      if (round == 1) {
        d1.no_reset = 1;
        d1.producers_done = &ctx->d_ctr->accept_final[0];      // (set behind the queue's end markers)
        d1.producers_total = 1u;
        d1.slow = slow_base + slow_cap_all / 2u;
        d1.slow_cap = slow_cap_all - slow_cap_all / 2u - d1.slow_waves;
        // Round 5: the second round's resident solve and straggler kernels start BEHIND the first accept, not beside it.  Beside
        // it they were waiting -- holding registers and LDS -- for end markers that the accept's LAST block writes, and about one
        // such pass in 25 found only an eighth (or seven eighths) of the accept's blocks ever begun: whole XCDs' shares of the
        // grid stayed undispatched until the waiting waves gave up (250 ms, then the redo; lentil_hip_last_redo_note: "accept
        // blocks done 64 begun 64" of 512).  The blocks that did run had served every item, so nothing was wrong but the wait.
        // A kernel of a pass may spin only on kernels that hold all the resources they will ever need.  What this costs is the
        // head start of the second round's solves (~0.1 ms of a pass that has a second round at all; the lean tail has none).
        // Round 6: beside it again, by default.  Since the end markers of an emitting accept are written by the block that finishes
        // the pass's LAST ITEM (DevCounters::accept_items_done), not by the grid's last block, a share of the grid that is never
        // dispatched keeps nobody waiting; soaked with the dispatch probe's build armed (tools/sessions_r06/r06_session26.sh: 1 600
        // passes with a second round in flight -- config 5's bands, the headline without the first-batch model --, no stall, no
        // probe event; profiles/r06_overlap_accept_soak.txt) and worth 13 % of a config-5 band's pass.  LENTIL_OVERLAP_ACCEPT=0:
        // behind it.
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, overlap_accept ? ctx->ev_round : ctx->ev_acc1, 0));
        launch_solve_po<true>(ctx, d1, ctx->stream, (unsigned)ctx->num_cu);
      } else {
        d1.producers_done = nullptr; d1.producers_total = 0;
        d1.slow_cap = slow_cap_all;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_round, ps));                   // behind the accept that filled this round's queues
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_round, 0));
        launch_solve_po<false>(ctx, d1, ctx->stream, (unsigned)ctx->num_cu);
      }
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipEventRecord(ctx->ev_solve, ctx->stream));
      if (round == 1) {
        // (on a stream of its own: the first round's straggler kernel, ahead of everything on `ps`, is at work for another
        // ~0.25 ms -- its last records come when A ends -- and this round's parked solves need not wait for it)
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->slow1_stream, overlap_accept ? ctx->ev_round : ctx->ev_acc1, 0));
        hipLaunchKernelGGL(solve_slow_kernel, dim3(d1.slow_waves), dim3(64), coop_lds_bytes(ctx->hlens.n_terms), ctx->slow1_stream, d1);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_slow1, ctx->slow1_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->ev_slow, 0));      // the first round's stragglers (publishers' stream)
      } else {
        hipLaunchKernelGGL(solve_slow_kernel, dim3(d1.slow_waves), dim3(64), coop_lds_bytes(ctx->hlens.n_terms), ps, d1);      // beside the round's solves
      }
      HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->ev_solve, 0));
      if (round == 1) {
        HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->ev_acc1, 0));
        // (the first round's queues can go back to empty for what the accept below schedules; its result pool is still read)
        hipLaunchKernelGGL(reset_round_kernel, dim3(1), dim3(1), 0, ps, ctx->d_ctr, 0u, 1u);
        hipLaunchKernelGGL(accept_kernel<2>, dim3(accept_blocks), dim3(256), 0, ps, da);
        da.slow_cap = slow_cap_all;
      } else {
        hipLaunchKernelGGL(accept_kernel<0>, dim3(accept_blocks), dim3(256), 0, ps, da);
      }
      HIP_TRY(ctx, hipGetLastError());
    }
    da.slow_indirect = 0; da.slow_cap = slow_cap_all;
    da.slow_live = 0;
    if (ctx->early_resolve_pending) {
      HIP_TRY(ctx, hipStreamWaitEvent(ps, ctx->ev_res, 0));
      if ((rc = launch_resolve_half(ctx, ps, 2u))) return rc;
      ctx->late_resolve_done = true;
    }
    tail = ps;
    }
  } else {
  if ((rc = join_clear(ctx))) return rc;       // (this form's accepts are on the main stream)
  // B: the rest of the CUs' room, once the scan's waves have left
  da.instance = 1;
  if (b_blocks) launch_solve_po<true>(ctx, da, ctx->stream, b_blocks, b_threads);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ch.done, 0));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->pub_done, 0));
  if (live && !decoupled) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_slow, 0));
  else if (!live) launch_slow(ctx, da, ctx->stream);
  {
    DrawArgs d0 = da;
    d0.emit_live = overlap ? 1 : 0;
    d0.end_tasks = (uint32_t)ctx->num_cu * 4u;
    if (overlap) HIP_TRY(ctx, hipEventRecord(ctx->ev_round, ctx->stream));      // everything the first accept waits for
    if (decoupled) hipLaunchKernelGGL(accept_kernel<1>, dim3(accept_blocks), dim3(256), 0, ctx->stream, d0);
    else hipLaunchKernelGGL(accept_kernel<0>, dim3(accept_blocks), dim3(256), 0, ctx->stream, d0);
    HIP_TRY(ctx, hipGetLastError());
  }
  // The frame's resolve, first half: behind the first accept, beside the second round's solves (one block per CU, no
  // HBM traffic to speak of) on the otherwise idle second chunk stream.  What later accepts add lands in groups of
  // pixels whose `touched` flag is set by then: lentil_hip_redistribute resolves those once more at its end.
  if (ctx->early_resolve && ctx->F.dir && ctx->F.touched && ctx->n_chunks >= 2 && !ctx->comm && !ctx->closest_deferred) {
    hipStream_t rs = ctx->chunks[1].stream;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_acc1, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(rs, ctx->ev_acc1, 0));
    if ((rc = launch_resolve_half(ctx, rs, 0u))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_res, rs));
    ctx->early_resolve_pending = true;
  }
  for (int round = 1; round < blind_rounds; ++round) {
    da.parity = round & 1; da.round = round;
    if (live) {
      // the round's stragglers beside its solves: the straggler kernel on the other stream, released by the accept
      // before it (round 1 with `overlap`: by what that accept itself waited for); this round's accept waits for both
      const bool beside = overlap && round == 1;
      if (beside) {
        DrawArgs d1 = da;
        d1.slow_after_producers = 0;
        d1.no_reset = 1;
        d1.producers_done = &ctx->d_ctr->accept_final[0];      // (set behind the queue's end markers)
        d1.producers_total = 1u;
        d1.slow_crowd_stays = ctx->crowd_stays_later;
        if (decoupled) { d1.slow_indirect = 0; d1.slow = slow_base + slow_cap_all / 2u; d1.slow_cap = slow_cap_all - slow_cap_all / 2u - d1.slow_waves; }
        HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->ev_round, 0));
        launch_solve_po<true>(ctx, d1, ch.stream, (unsigned)ctx->num_cu);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ch.done, ch.stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->pub_stream, ctx->ev_round, 0));
        hipLaunchKernelGGL(solve_slow_kernel, dim3(d1.slow_waves), dim3(64), coop_lds_bytes(ctx->hlens.n_terms), ctx->pub_stream, d1);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_slow, ctx->pub_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ch.done, 0));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_slow, 0));      // (decoupled: behind the first round's straggler kernel too)
        // (the first accept and this round's solves are done: the first round's queues can go back to empty for what
        // the accept below schedules)
        hipLaunchKernelGGL(reset_round_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_ctr, 0u, decoupled ? 1u : 0u);
      } else {
        da.producers_done = nullptr; da.producers_total = 0;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_round, ctx->stream));
        launch_solve_po<false>(ctx, da, ctx->stream, (unsigned)ctx->num_cu);
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->pub_stream, ctx->ev_round, 0));
        hipLaunchKernelGGL(solve_slow_kernel, dim3(da.slow_waves), dim3(64), coop_lds_bytes(ctx->hlens.n_terms), ctx->pub_stream, da);
        HIP_TRY(ctx, hipEventRecord(ctx->ev_slow, ctx->pub_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_slow, 0));
      }
    } else if (overlap && round == 1) {
      // nothing parks in this pass: the second round's resident solves beside the first accept, released by what that
      // accept itself waited for, fed by its tagged task slots and closed by its last block's end markers
      DrawArgs d1 = da;
      d1.slow_after_producers = 0;
      d1.no_reset = 1;
      d1.producers_done = &ctx->d_ctr->accept_final[0];      // (set behind the queue's end markers)
      d1.producers_total = 1u;
      HIP_TRY(ctx, hipStreamWaitEvent(ch.stream, ctx->ev_round, 0));
      launch_solve_po<true>(ctx, d1, ch.stream, (unsigned)ctx->num_cu);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipEventRecord(ch.done, ch.stream));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ch.done, 0));
      hipLaunchKernelGGL(reset_round_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_ctr, 0u, 0u);
    } else {
      launch_solve(ctx, da, ctx->stream, (unsigned)ctx->num_cu);
    }
    if (decoupled && round == 1) hipLaunchKernelGGL(accept_kernel<2>, dim3(accept_blocks), dim3(256), 0, ctx->stream, da);
    else hipLaunchKernelGGL(accept_kernel<0>, dim3(accept_blocks), dim3(256), 0, ctx->stream, da);
    HIP_TRY(ctx, hipGetLastError());
    if (decoupled && round == 1) { da.slow_indirect = 0; da.slow_cap = slow_cap_all; }
  }
  da.slow_indirect = 0; da.slow_cap = slow_cap_all;
  da.slow_live = 0;       // (rounds the host adds one by one, below, park and finish their stragglers the plain way)
  if (ctx->early_resolve_pending) {
    // the resolve's second half behind the last accept enqueued blind (should the host have to add rounds, or redo
    // the draws, lentil_hip_redistribute runs it once more)
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_res, 0));
    if ((rc = launch_resolve_half(ctx, ctx->stream, 2u))) return rc;
    ctx->late_resolve_done = true;
  }
  }
  const int C = ctx->n_chunks;
  ht_mark(ctx, "all_launched");
  ctx->clear_pending = false;       // (every stream of the pass is behind the wipe by now, and the main stream will be behind the pass)
  StreamTail t;
  t.tail = tail; t.pass_t0 = pass_t0; t.calibrates_now = calibrates_now; t.predicted = predicted; t.lean = lean; t.extend = extend;
  t.live = live; t.inject = da.inject_stall != 0; t.blind_rounds = blind_rounds; t.da = da; t.slow_base = slow_base;
  t.slow_cap_all = slow_cap_all; t.accept_blocks = accept_blocks; t.item_cap = pub.item_cap; t.task_cap = pub.task_cap;
  t.range_cap = plan.sa.range_cap; t.pool_cap = pub.pool_cap; t.stuck_ticks = stuck_ticks;
  bool have_counters = false;
  if (ctx->spin_readback && ctx->d_ctr_host) {
    // The counters by a kernel into the host's copy and a sequence number behind them; the host polls that number.  (A copy
    // command's completion reaches the waiting thread ~40 us after the device is through: signal, interrupt, wake-up.)
    const uint32_t seq = ++ctx->seq ? ctx->seq : ++ctx->seq;
    hipLaunchKernelGGL(report_counters_kernel, dim3(1), dim3(256), 0, tail, reinterpret_cast<const uint32_t *>(ctx->d_ctr),
                       reinterpret_cast<uint32_t *>(ctx->d_ctr_host), (uint32_t)(sizeof(DevCounters) * C / sizeof(uint32_t)),
                       reinterpret_cast<uint32_t *>(ctx->d_ctr_host + C), seq);
    HIP_TRY(ctx, hipGetLastError());
    const auto spin_t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (true) {
      if (__atomic_load_n(ctx->h_seq, __ATOMIC_ACQUIRE) == seq) { have_counters = true; break; }
      __builtin_ia32_pause();
      if ((++spins & 0xFFFu) == 0u &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - spin_t0).count() > 2.0) break;      // (then the plain way)
    }
  }
  if (!have_counters)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_ctr_pinned, ctx->d_ctr, sizeof(DevCounters) * C, hipMemcpyDeviceToHost, tail));
  // The asynchronous end (lentil_hip_ctx::async_end): the lean tail is the pass that expects to have nothing left to do, and the
  // frame takes nothing but gaussian splats (closest-filtered AOVs, lentil_debug and cryptomatte have host steps behind the pass).
  // Everything of the pass is behind `tail` by now (the lean tail's accepts wait for the scan, the publishers, the stragglers
  // and the early resolve); the context's own stream waits for it in turn, so whatever the caller enqueues next follows the pass.
  const bool defer = ctx->async_end && lean && !have_counters && !ctx->spin_readback && !ctx->crypto && !ctx->F.zkey && !ctx->F.zkey_dbg &&
                     !ctx->comm && !ctx->closest_deferred && !da.inject_stall && !host_trace_passes() && ctx->inflight.size() < 2;
  if (defer) {
    lentil_hip_ctx::Slot &sl = ctx->slots[ctx->slot];
    HIP_TRY(ctx, hipEventRecord(sl.ev_tail, tail));
    if (tail != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, sl.ev_tail, 0));
    t.deferred = true;
    lentil_hip_ctx::Inflight in;
    in.slot = ctx->slot; in.t = t; in.V = ctx->V; in.have_visits = ctx->have_visits;
    ctx->inflight.push_back(in);
    turn.keep();                  // (the device's turn stays this context's until the pass has been looked at)
    ctx->last_streamed = 1;
    ++ctx->last_blind;
    *deferred = true;
    *streamed = true;
    return LENTIL_OK;
  }
  if (!have_counters) HIP_TRY(ctx, hipStreamSynchronize(tail));
  ht_mark(ctx, "tail_synced");
  if (tail != ctx->stream) {
    // everything the pass enqueued anywhere is behind the read-back that has just arrived; what the caller enqueues on
    // the context's stream next (resolve, downloads, the next pass) follows the main stream's own last kernel
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  ht_mark(ctx, "main_synced");
  {
    // (host time from the pass's first launch to its counters: an upper bound of every wait inside it)
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pass_t0).count();
    if (!calibrates_now && ms > ctx->longest_pass_ms && ms < 200.0) {
      ctx->longest_pass_ms = ms; ctx->longest_pass_visits = ctx->V.n; ctx->longest_pass_sum = ctx->est_sum_total;
    }
  }
  ++ctx->last_blind;
  return streamed_finish(ctx, t, ctx->h_ctr_pinned, true, streamed);
}

// Thin lens with abb_chromatic > 0 (kernels: tl_chroma_*): scan, the work list put into visit order on the host, every
// possible attempt's channel-independent part solved in parallel, then one block walks the items in order.
static int redistribute_tl_chroma(lentil_hip_ctx *ctx) {
  if (ctx->comm || ctx->closest_deferred)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "thin-lens abb_chromatic > 0: the order of the xor128 draws is defined for one GPU only");
  ScanPlan plan;
  int rc = plan_scan(ctx, plan);
  if (rc) return rc;
  lentil_hip_ctx::Chunk &ch = ctx->chunks[0];
  ch.tile_begin = 0; ch.tile_end = plan.n_tiles;
  ch.v_begin = 0; ch.v_end = ctx->V.n;
  if ((rc = launch_scan(ctx, plan, ch, ctx->d_ctr, nullptr))) return rc;
  ++ctx->last_scan_launches;
  HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->scans_done, ctx->stream));
  DevCounters c0;
  HIP_TRY(ctx, hipMemcpyAsync(&c0, ctx->d_ctr, sizeof(c0), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (c0.work_count > ctx->V.n) return fail(ctx, LENTIL_ERR_NOMEM, "work list overflow");
  const uint32_t n_items = (uint32_t)c0.work_count;
  if (!n_items) return LENTIL_OK;
  std::vector<uint2> work(n_items);
  HIP_TRY(ctx, hipMemcpy(work.data(), ctx->d_work, (size_t)n_items * sizeof(uint2), hipMemcpyDeviceToHost));
  std::sort(work.begin(), work.end(), [](const uint2 &a, const uint2 &b) { return a.x < b.x; });      // iterator order
  std::vector<uint64_t> off((size_t)n_items + 1);
  std::vector<uint2> tasks;
  uint64_t slots = 0;
  for (uint32_t i = 0; i < n_items; ++i) {
    off[i] = slots;
    const uint64_t att = (uint64_t)work[i].y * 5ull;
    for (uint64_t n0 = 0; n0 < att; n0 += 256) tasks.push_back(make_uint2(i, (uint32_t)n0));
    slots += att;
  }
  off[n_items] = slots;
  if (slots * 12ull > (64ull << 30)) return fail(ctx, LENTIL_ERR_NOMEM, "thin-lens abb_chromatic > 0: more than 64 GiB of attempt results");
  auto grow_to = [&](auto **p, uint64_t &cap, uint64_t need) -> int {
    if (cap >= need) return LENTIL_OK;
    (void)hipFree(*p);
    *p = nullptr;
    cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)p, need * sizeof(**p)));
    cap = need;
    return LENTIL_OK;
  };
  if ((rc = grow_to(&ctx->d_tlc_res, ctx->tlc_res_cap, slots * 3ull + 4ull))) return rc;
  if ((rc = grow_to(&ctx->d_tlc_off, ctx->tlc_off_cap, (uint64_t)n_items + 1))) return rc;
  if ((rc = grow_to(&ctx->d_tlc_tasks, ctx->tlc_tasks_cap, (uint64_t)tasks.size() + 1))) return rc;
  if (!ctx->d_xor) HIP_TRY(ctx, hipMalloc(&ctx->d_xor, 4 * sizeof(uint32_t)));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_work, work.data(), (size_t)n_items * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_tlc_off, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_tlc_tasks, tasks.data(), tasks.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_xor, ctx->xor_state, sizeof ctx->xor_state, hipMemcpyHostToDevice, ctx->stream));
  TlChromaArgs ta{};
  ta.P = ctx->P; ta.bokeh = ctx->bokeh; ta.V = ctx->V; ta.F = ctx->F;
  ta.work = ctx->d_work; ta.n_items = n_items;
  ta.att_off = ctx->d_tlc_off; ta.res = ctx->d_tlc_res;
  ta.tasks = ctx->d_tlc_tasks; ta.n_tasks = (uint32_t)tasks.size();
  ta.xor_state = ctx->d_xor;
  ta.ctr = ctx->d_ctr;
  ta.log = ctx->d_log; ta.log_cap = ctx->log_cap; ta.log_count = &ctx->d_ctr[ctx->n_chunks].log_count;
  uint64_t blocks = tasks.size();
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(tl_chroma_solve_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ta);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(tl_chroma_walk_kernel, dim3(1), dim3(256), 0, ctx->stream, ta);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(ctx->xor_state, ctx->d_xor, sizeof ctx->xor_state, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // (the host vectors above are the copies' sources)
  ctx->last_rounds = 1;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_set_xor128_state(lentil_hip_ctx *ctx, const uint32_t state[4]) {
  CHECK_CTX(ctx);
  if (!state) return fail(ctx, LENTIL_ERR_INVALID, "state is null");
  memcpy(ctx->xor_state, state, sizeof ctx->xor_state);
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_get_xor128_state(lentil_hip_ctx *ctx, uint32_t state[4]) {
  CHECK_CTX(ctx);
  if (!state) return fail(ctx, LENTIL_ERR_INVALID, "state is null");
  memcpy(state, ctx->xor_state, sizeof ctx->xor_state);
  return LENTIL_OK;
}

static int redistribute_pass(lentil_hip_ctx *ctx);
static int redistribute_impl(lentil_hip_ctx *ctx);
static int pass_finish(lentil_hip_ctx *ctx);
static int redistribute_top(lentil_hip_ctx *ctx);
static void harvest_ready(lentil_hip_ctx *ctx);
static int begin_slot(lentil_hip_ctx *ctx);
static int redo_after_stall(lentil_hip_ctx *ctx);
static void account_pass(lentil_hip_ctx *ctx);

// Closest-filtered AOVs after a pass that met candidates at depth 0 / NaN: lentil_closest_replay.h.  Returns LENTIL_OK with
// *need_log set when the pass kept no (complete) draw log to replay from.
static int closest_degenerate_replay(lentil_hip_ctx *ctx, bool *need_log) {
  *need_log = false;
  const int C = ctx->n_chunks;
  unsigned long long n_log = 0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(&n_log, (char *)(ctx->d_ctr + C) + offsetof(DevCounters, log_count), sizeof n_log, hipMemcpyDeviceToHost));
  unsigned long long accepted = 0;
  for (const DevCounters &k : ctx->h_ctr) accepted += k.accepted;
  if (!ctx->d_log || n_log > ctx->log_cap || (accepted && n_log < accepted)) { *need_log = true; return LENTIL_OK; }
  ReplayArgs a{};
  a.P = ctx->P; a.lens_length = ctx->have_lens ? ctx->hlens.length : 0.0; a.V = ctx->V; a.F = ctx->F;
  a.log = ctx->d_log; a.n_log = n_log;
  uint8_t *flag = nullptr; uint32_t *head = nullptr; unsigned int *count = nullptr; ReplayNode *nodes = nullptr;
  int rc = LENTIL_OK;
  hipError_t e = hipMalloc(&flag, ctx->F.np);
  if (e == hipSuccess) e = hipMalloc(&head, ctx->F.np * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&count, sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemsetAsync(flag, 0, ctx->F.np, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(head, 0xFF, ctx->F.np * sizeof(uint32_t), ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(count, 0, sizeof(unsigned int), ctx->stream);
  a.flag = flag; a.head = head; a.n_nodes = count;
  const dim3 grid((unsigned)ctx->num_cu * 8), block(256);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(replay_mark_visits_kernel, grid, block, 0, ctx->stream, a);
    if (n_log) hipLaunchKernelGGL(replay_mark_log_kernel, grid, block, 0, ctx->stream, a);
    // the lists: counted, then filled
    hipLaunchKernelGGL(replay_push_visits_kernel, grid, block, 0, ctx->stream, a);
    if (n_log) hipLaunchKernelGGL(replay_push_log_kernel, grid, block, 0, ctx->stream, a);
    e = hipGetLastError();
  }
  unsigned int n_nodes = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&n_nodes, count, sizeof n_nodes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && n_nodes) {
    e = hipMalloc(&nodes, (size_t)n_nodes * sizeof(ReplayNode));
    if (e == hipSuccess) e = hipMemsetAsync(count, 0, sizeof(unsigned int), ctx->stream);
    a.nodes = nodes; a.node_cap = n_nodes;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(replay_push_visits_kernel, grid, block, 0, ctx->stream, a);
      if (n_log) hipLaunchKernelGGL(replay_push_log_kernel, grid, block, 0, ctx->stream, a);
      hipLaunchKernelGGL(replay_resolve_kernel, grid, block, 0, ctx->stream, a);
      // the winners' values once more, from the keys as they stand now
      if (ctx->F.zkey_dbg) hipLaunchKernelGGL(debug_gather_kernel, grid, block, 0, ctx->stream, ctx->F, ctx->V, ctx->P, a.lens_length);
      if (ctx->F.zkey) hipLaunchKernelGGL(closest_gather_kernel, grid, block, 0, ctx->stream, ctx->F, ctx->V);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(flag); (void)hipFree(head); (void)hipFree(count); (void)hipFree(nodes);
  if (e != hipSuccess) rc = fail(ctx, LENTIL_ERR_HIP, std::string("closest-AOV replay: ") + hipGetErrorString(e));
  if (rc == LENTIL_OK) { ctx->resolved_valid = false; ++ctx->n_degenerate_replays; }
  return rc;
}

LENTIL_API int lentil_hip_redistribute(lentil_hip_ctx *ctx) {
  CHECK_CTX_PIPELINED(ctx);
  // A pass nobody has looked at and whose frame has NOT been cleared since: this one adds to the same frame -- it observes.
  // (Abandoned passes wait their turn: begin_slot looks at the oldest when its slot comes round, harvest_ready at those that are through.)
  if (!ctx->inflight.empty() && !ctx->inflight.back().abandoned) { const int rc0 = settle(ctx); if (rc0) return rc0; }
  harvest_ready(ctx);
  const int rc1 = redistribute_top(ctx);
  if (rc1 == LENTIL_OK && (ctx->inflight.empty() || ctx->inflight.back().slot != ctx->slot)) account_pass(ctx);      // (a deferred pass is accounted when its end is known)
  return rc1;
}

static int redistribute_top(lentil_hip_ctx *ctx) {
  ctx->stall_redo = false;
  ctx->degenerate_seen = false;
  const bool clean_entry = ctx->cleared_since_pass;
  // (a context whose frames have needed the replay before keeps a draw log from the start: no pass is run twice again)
  if (ctx->closest_auto_log && ctx->log_cap == 0 && (ctx->F.zkey || ctx->F.zkey_dbg)) {
    const int rc0 = lentil_hip_set_draw_log(ctx, ctx->closest_auto_log_cap);
    if (rc0) return rc0;
  }
  // (per-pass host state the pass advances: should the pass be run a second time below, it starts from the same state)
  uint32_t xor_entry[4];
  memcpy(xor_entry, ctx->xor_state, sizeof xor_entry);
  ht_mark(ctx, "pass{");
  int rc = redistribute_impl(ctx);
  ht_mark(ctx, "}pass");
  if (rc || !ctx->degenerate_seen) return rc;
  // ---- candidates at depth 0 / NaN compete for a closest-filtered AOV: their pixels are replayed in visit order
  if (ctx->comm || ctx->closest_deferred)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "a sample with depth (Z) 0 or NaN competes for a closest-filtered AOV: the reference's result there "
                                             "depends on the order of the samples at the pixel (src/lentil.h:832-837), which the exchange "
                                             "between GPUs does not keep; the pass is refused");
  // The replay walks ONE pass's candidates -- the bound visits and this pass's draw log -- in visit order: a frame that already
  // held an earlier pass's candidates (no clear in between) cannot be replayed from them, with or without a log.
  if (!clean_entry)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "a sample with depth (Z) 0 or NaN competes for a closest-filtered AOV in a frame that was not cleared "
                                             "before this pass: the reference's result depends on the order of ALL candidates at the pixel "
                                             "(src/lentil.h:832-837), and an earlier pass's are gone; clear the frame before the pass");
  bool need_log = false;
  if ((rc = closest_degenerate_replay(ctx, &need_log))) return rc;
  if (!need_log) return LENTIL_OK;
  // no (complete) draw log to replay from.  The frame held nothing before this pass: a log sized from the pass's counters,
  // the frame wiped, the pass once more -- and a log from the start in every later pass of this context.
  unsigned long long accepted = 0;
  for (const DevCounters &k : ctx->h_ctr) accepted += k.accepted;
  ctx->closest_auto_log = true;
  ctx->closest_auto_log_cap = accepted + accepted / 4 + (1ull << 20);
  if ((rc = lentil_hip_set_draw_log(ctx, ctx->closest_auto_log_cap))) return rc;
  if ((rc = lentil_hip_clear_frame(ctx))) return rc;
  ctx->degenerate_seen = false;
  memcpy(ctx->xor_state, xor_entry, sizeof xor_entry);       // (thin lens, abb_chromatic > 0: the same colour channels as the first run)
  if ((rc = redistribute_impl(ctx))) return rc;
  // (counters and timing describe the second run; that there were two is reported like any pass that was redone)
  ++ctx->last_fallback;
  ctx->redo_note = "the pass was run twice: candidates at depth 0 / NaN compete for a closest-filtered AOV and the first run kept no draw log "
                   "to replay their pixels from (lentil_closest_replay.h); later passes of this context keep one from the start";
  if (!ctx->degenerate_seen) return LENTIL_OK;
  if ((rc = closest_degenerate_replay(ctx, &need_log))) return rc;
  if (need_log) return fail(ctx, LENTIL_ERR_NOMEM, "closest-AOV replay: the draw log did not hold the pass's accepted draws");
  return LENTIL_OK;
}


// ---- the asynchronous end of a pass: slots, the look at a pass's end, the totals ------------------------------------------
// one slot's kernel times into the totals, once its last event has passed (blocking: wait for it)
static void flush_timing(lentil_hip_ctx *ctx, int si, bool blocking) {
  lentil_hip_ctx::Slot &sl = ctx->slots[si];
  if (!sl.timing_open) return;
  const bool cur = si == ctx->slot;
  if (cur) { sl.timed_draw = ctx->timed_draw; sl.timed_resolve = ctx->timed_resolve; sl.scan_kernel_timed = ctx->scan_kernel_timed; }
  const bool timed_draw = sl.timed_draw, timed_resolve = sl.timed_resolve, scan_k = sl.scan_kernel_timed;
  if (!timed_draw) { if (!cur) sl.timing_open = false; return; }      // (a pass that failed before its end: nothing to read)
  hipEvent_t last = timed_resolve ? sl.ev[4] : sl.ev[2];
  if (!blocking && hipEventQuery(last) != hipSuccess) { (void)hipGetLastError(); return; }
  if (hipEventSynchronize(last) != hipSuccess) { (void)hipGetLastError(); sl.timing_open = false; return; }
  float a = 0.f, b = 0.f, c = 0.f, k = 0.f;
  if (hipEventElapsedTime(&a, sl.ev[0], sl.ev[1]) != hipSuccess || hipEventElapsedTime(&b, sl.ev[1], sl.ev[2]) != hipSuccess) { (void)hipGetLastError(); a = b = 0.f; }
  if (scan_k && hipEventElapsedTime(&k, sl.ev_scan_k[0], sl.ev_scan_k[1]) == hipSuccess && k > 0.f && k <= a) { b += a - k; a = k; }
  else (void)hipGetLastError();
  if (timed_resolve && hipEventElapsedTime(&c, sl.ev[3], sl.ev[4]) != hipSuccess) { (void)hipGetLastError(); c = 0.f; }
  ctx->totals.scan_ms += a; ctx->totals.draw_ms += b; ctx->totals.resolve_ms += c;
  sl.timing_open = false;
}

// a pass's counters and launch figures into the totals (ctx->h_ctr, last_*: the pass that has just been looked at)
static void account_pass(lentil_hip_ctx *ctx) {
  lentil_pass_totals &T = ctx->totals;
  ++T.passes;
  T.streamed += ctx->last_streamed; T.blind_chunks += ctx->last_blind; T.fallback_chunks += ctx->last_fallback;
  T.visits += ctx->have_visits ? ctx->V.n : 0;
  if (ctx->h_ctr_valid)
    for (const DevCounters &k : ctx->h_ctr) {
      T.redistributed_visits += k.redistributed; T.attempted_draws += k.attempted; T.accepted_draws += k.accepted;
      T.worklist_overflow += k.overflow; T.newton_iterations += k.newton_iters; T.tries += k.tries; T.lane_rounds += k.lane_rounds;
      T.slow_solves += k.slow_solves;
    }
  T.scan_launches += ctx->last_scan_launches ? ctx->last_scan_launches : 1u;
  if ((uint64_t)ctx->last_rounds > T.rounds_max) T.rounds_max = (uint64_t)ctx->last_rounds;
}

// The oldest pass in flight: waits for it, reads its verdict, and -- unless its frame has been cleared since -- does what the
// pass left open, exactly as lentil_hip_redistribute did while it still waited itself.
static int harvest_front(lentil_hip_ctx *ctx) {
  lentil_hip_ctx::Inflight in = ctx->inflight.front();
  ctx->inflight.pop_front();
  lentil_hip_ctx::Slot &sl = ctx->slots[in.slot];
  int rc = LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const hipError_t e = hipEventSynchronize(sl.ev[2]);        // (on the context's stream behind ev_tail: counters copied, every kernel of the pass through)
    if (e != hipSuccess) { release_turn(ctx); return fail(ctx, LENTIL_ERR_HIP, std::string("waiting for a pass's end: ") + hipGetErrorString(e)); }
  }
  {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sl.ev[0], sl.ev[2]) == hipSuccess) {
      const double m = (double)ms + 0.3;      // (+ what the host needs to enqueue a pass: the estimate used to be host time)
      if (!in.t.calibrates_now && m > ctx->longest_pass_ms && m < 200.0) { ctx->longest_pass_ms = m; ctx->longest_pass_visits = in.V.n; ctx->longest_pass_sum = ctx->est_sum_total; }
    } else (void)hipGetLastError();
  }
  const bool can_fix = !in.abandoned;
  // (the per-pass host fields describe the pass being looked at)
  ctx->last_blind = 1; ctx->last_fallback = 0; ctx->last_rounds = 0; ctx->last_scan_launches = 1; ctx->last_streamed = 1;
  ctx->h_ctr_valid = false;
  std::swap(ctx->V, in.V);
  std::swap(ctx->have_visits, in.have_visits);
  bool streamed = true;
  ctx->stall_redo = false;
  rc = streamed_finish(ctx, in.t, sl.h_ctr_pinned, can_fix, &streamed);
  if (rc == LENTIL_OK && can_fix) {
    if (ctx->stall_redo) rc = redo_after_stall(ctx);
    if (rc == LENTIL_OK && in.t.did_more) {
      // the frame moved on behind the resolve the pass had enqueued on its way
      ctx->resolved_valid = false;
      ctx->early_resolve_pending = false;
      (void)hipEventRecord(sl.ev[2], ctx->stream);       // (the pass's end for the totals' times: behind the extra work; the slot's own event, whichever slot is current)
    }
    if (rc == LENTIL_OK) rc = pass_finish(ctx);
    if (rc == LENTIL_OK && in.resolve_requested && !ctx->resolved_valid) rc = resolve_range(ctx, 0, ctx->F.np);
  }
  ctx->stall_redo = false;
  ++ctx->totals.deferred;
  if (in.abandoned) ++ctx->totals.abandoned;
  if (rc == LENTIL_OK) account_pass(ctx);
  std::swap(ctx->V, in.V);
  std::swap(ctx->have_visits, in.have_visits);
  flush_timing(ctx, in.slot, false);
  release_turn(ctx);
  return rc;
}

// every pass in flight, oldest first (the observer's side of the asynchronous end)
static int settle(lentil_hip_ctx *ctx) {
  int rc = LENTIL_OK;
  while (!ctx->inflight.empty()) {
    const int r = harvest_front(ctx);
    if (r != LENTIL_OK && rc == LENTIL_OK) rc = r;
  }
  return rc;
}

// abandoned passes the device is through with: looked at without waiting (their verdicts size the next pass)
static void harvest_ready(lentil_hip_ctx *ctx) {
  while (!ctx->inflight.empty() && ctx->inflight.front().abandoned) {
    if (hipEventQuery(ctx->slots[ctx->inflight.front().slot].ev[2]) != hipSuccess) { (void)hipGetLastError(); return; }
    (void)harvest_front(ctx);
  }
}

// a pass begins: the next slot's events and counter block become the context's (a pass still in flight on it is looked at first)
static int begin_slot(lentil_hip_ctx *ctx) {
  lentil_hip_ctx::Slot &cur = ctx->slots[ctx->slot];
  cur.timed_draw = ctx->timed_draw; cur.timed_resolve = ctx->timed_resolve; cur.scan_kernel_timed = ctx->scan_kernel_timed;
  if (ctx->spin_readback) {        // (LENTIL_SPIN_READBACK: one block, addressed by the device -- no rotation, no asynchronous end)
    flush_timing(ctx, ctx->slot, true);
    ctx->timed_draw = ctx->timed_resolve = false; ctx->scan_kernel_timed = false;
    cur.timing_open = true;
    return LENTIL_OK;
  }
  const int next = (ctx->slot + 1) % 3;
  // (at most two passes in flight unobserved: with two there already this one waits for the older -- the device still has the newer to work on)
  while (ctx->inflight.size() >= 2 || (!ctx->inflight.empty() && ctx->inflight.front().slot == next)) {
    const int rc = harvest_front(ctx);
    if (rc) return rc;
  }
  flush_timing(ctx, ctx->slot, false);       // (saves the context's flags into the slot it leaves, whether or not its events have passed)
  ctx->slot = -1;                             // (no slot is current while the next one's last pass -- three passes ago -- is flushed from its own flags)
  flush_timing(ctx, next, true);              // its events are about to be recorded again
  ctx->slot = next;
  lentil_hip_ctx::Slot &n = ctx->slots[next];
  for (int i = 0; i < 5; ++i) ctx->ev[i] = n.ev[i];
  ctx->ev_scan_k[0] = n.ev_scan_k[0]; ctx->ev_scan_k[1] = n.ev_scan_k[1];
  ctx->h_ctr_pinned = n.h_ctr_pinned;
  ctx->timed_draw = ctx->timed_resolve = false; ctx->scan_kernel_timed = false;
  n.timing_open = true;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_set_async(lentil_hip_ctx *ctx, int on) {
  CHECK_CTX(ctx);
  ctx->async_end = on != 0;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_pass_totals(lentil_hip_ctx *ctx, lentil_pass_totals *out, int reset) {
  CHECK_CTX(ctx);
  if (!out) return fail(ctx, LENTIL_ERR_INVALID, "out is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (int si = 0; si < 3; ++si) flush_timing(ctx, si, true);
  ctx->slots[ctx->slot].timing_open = false;       // (the current pass's times are in: not a second time when its slot is left)
  *out = ctx->totals;
  if (reset) { ctx->totals = lentil_pass_totals{}; ctx->notes.clear(); }
  return LENTIL_OK;
}

// one pass, with the recovery of a streamed pass that stalled after its first accept
static int redistribute_impl(lentil_hip_ctx *ctx) {
  ctx->stall_redo = false;
  int rc = redistribute_pass(ctx);
  if (rc || !ctx->stall_redo) return rc;
  return redo_after_stall(ctx);
}

static int redo_after_stall(lentil_hip_ctx *ctx) {
  int rc;
  // A streamed pass gave up waiting (kStuckTicks) after its first accept had added draws to the frame.  Every kernel of the
  // pass has left by now (they all watch DevCounters::stuck); the frame was clear before the pass, so: clear it again --
  // accumulators through the splat flags, direct sums, cryptomatte -- and run the whole pass once more, chunked.
  ctx->stall_redo = false;
  ++ctx->n_stall_redone;
  g_stat_redone.fetch_add(1, std::memory_order_relaxed);
  if (getenv("LENTIL_STREAM_DEBUG")) fprintf(stderr, "[stream] stalled after the first accept: frame wiped, pass run again chunked\n");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (hipStream_t st : {ctx->stream, ctx->chunks[0].stream, ctx->pub_stream, ctx->slow1_stream, ctx->aux_stream})
    if (st) HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipMemsetAsync((char *)ctx->d_ctr + offsetof(DevCounters, stuck), 0, sizeof(unsigned int), ctx->stream));
  if ((rc = lentil_hip_clear_frame(ctx))) return rc;
  const bool stream_mode = ctx->stream_mode;
  ctx->stream_mode = false;
  rc = redistribute_pass(ctx);
  ctx->stream_mode = stream_mode;
  if (rc) return rc;
  ++ctx->last_fallback;          // (lentil_counters::fallback_chunks: the pass was redone; ::streamed is 0)
  return LENTIL_OK;
}

static int redistribute_pass(lentil_hip_ctx *ctx) {
  if (!ctx->have_params || !ctx->have_frame || !ctx->have_visits)
    return fail(ctx, LENTIL_ERR_INVALID, "redistribute needs set_params, alloc_frame and visits");
  const lentil_params &P = ctx->P;
  if (P.cameraType == LENTIL_POLYNOMIAL_OPTICS && !ctx->have_lens)
    return fail(ctx, LENTIL_ERR_INVALID, "polynomial optics needs set_lens");
  if (P.bokeh_enable_image && !ctx->have_bokeh) return fail(ctx, LENTIL_ERR_INVALID, "bokeh_enable_image needs set_bokeh");
  if (ctx->V.n_extra + 1 != ctx->F.n_aovs)
    return fail(ctx, LENTIL_ERR_INVALID, "visit stream carries a different number of AOVs than the frame");
  for (uint32_t k = 1; k < ctx->F.n_aovs; ++k)
    if (ctx->V.n && !ctx->V.extra[k - 1] && !(ctx->F.debug_mask & (1u << k)))
      return fail(ctx, LENTIL_ERR_INVALID, "an extra AOV column is null");
  if (ctx->probe_fn && P.abb_chromatic != 0.0f)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "occlusion probes with abb_chromatic != 0: the reference draws an attempt's colour channel behind its probe, from "
                                             "one generator in visit order (src/lentil_filter.cpp:356-406); not built");
  if (ctx->F.debug_mask && ctx->closest_deferred && !ctx->comm)
    return fail(ctx, LENTIL_ERR_UNSUPPORTED, "the lentil_debug AOV is exchanged between GPUs by lentil_hip_allreduce / _exchange_bands only");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  { const int rc = crypto_before_pass(ctx); if (rc) return rc; }
  if (ctx->crypto_direct_enqueued) {       // (a pass that failed after it had enqueued them)
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_crypto, 0));
    ctx->crypto_direct_enqueued = false;
  }
  ctx->resolved_valid = false;
  ctx->early_resolve_pending = false;
  ctx->late_resolve_done = false;
  const int C = ctx->n_chunks;
  if (ctx->pass_pending) ctx->dirty_known = false;   // an earlier pass's rows were never asked for: unknown until a full clear
  ctx->pass_pending = true;
  { const int rc = begin_slot(ctx); if (rc) return rc; }      // (this pass's own events and counter block)
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctr, 0, sizeof(DevCounters) * (C + 1), ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  ctx->last_rounds = 0;
  ctx->scan_kernel_timed = false;
  ctx->h_ctr_valid = false;
  ctx->last_blind = ctx->last_fallback = 0;
  ctx->last_streamed = 0;
  for (auto &ch : ctx->chunks) ch.heavy_pending = false;
  ctx->last_scan_launches = 0;
  bool streamed = false, deferred = false;
  const bool tl_chroma = P.cameraType == LENTIL_THINLENS && P.abb_chromatic > 0.0f;
  if (tl_chroma) {
    { const int rc = join_clear(ctx); if (rc) return rc; }
    if (ctx->V.n) { const int rc = redistribute_tl_chroma(ctx); if (rc) return rc; }
    else { HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream)); HIP_TRY(ctx, hipEventRecord(ctx->scans_done, ctx->stream)); }
    streamed = true;          // (nothing of the chunked form below runs)
  } else {
    const int rc = redistribute_streamed(ctx, &streamed, &deferred);
    if (rc) return rc;
    if (ctx->stall_redo) return LENTIL_OK;       // (lentil_hip_redistribute wipes the frame and calls again)
  }
  if (!streamed) { const int rc = join_clear(ctx); if (rc) return rc; }
  if (!streamed && ctx->V.n) {
    // ---- scans: all chunks back to back on the main stream
    ScanPlan plan;
    {
      const int rc = plan_scan(ctx, plan);
      if (rc) return rc;
    }
    const uint32_t M = plan.M;
    for (int ci = 0; ci < C; ++ci) {
      lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
      if (M) {
        // chunk boundaries: the first chunk takes first_chunk_frac of the tiles, the others share the rest
        auto bound = [&](int i) -> uint64_t {
          if (i <= 0) return 0;
          if (i >= C) return plan.n_tiles;
          const double f0 = C > 1 ? ctx->first_chunk_frac : 1.0;
          const double x = f0 + (1.0 - f0) * (double)(i - 1) / (double)(C - 1);
          const uint64_t b = (uint64_t)((double)plan.n_tiles * x);
          return b > plan.n_tiles ? plan.n_tiles : b;
        };
        ch.tile_begin = bound(ci);
        ch.tile_end = bound(ci + 1);
        ch.v_begin = ch.tile_begin * plan.sa.ppt * M;
        ch.v_end = ch.tile_end * plan.sa.ppt * M;
        if (ch.v_begin > ctx->V.n) ch.v_begin = ctx->V.n;
        if (ch.v_end > ctx->V.n) ch.v_end = ctx->V.n;
      } else {
        ch.v_begin = ((ctx->V.n * ci / C) + 63) & ~63ull;
        ch.v_end = ci == C - 1 ? ctx->V.n : (((ctx->V.n * (ci + 1) / C) + 63) & ~63ull);
        if (ch.v_begin > ctx->V.n) ch.v_begin = ctx->V.n;
        if (ch.v_end > ctx->V.n) ch.v_end = ctx->V.n;
      }
      if (ch.v_end > ch.v_begin) {
        const int rc = launch_scan(ctx, plan, ch, ctx->d_ctr + ci, nullptr);
        if (rc) return rc;
        ++ctx->last_scan_launches;
      }
      HIP_TRY(ctx, hipEventRecord(ch.scanned, ctx->stream));
    }
  }
  if (!streamed) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->scans_done, ctx->stream));
  }
  if (!streamed && ctx->V.n) {
    // ---- draws: per chunk on its own stream
    DrawArgs da{};
    init_draw_args(ctx, da);
    // solve/accept rounds enqueued without looking (an unused one costs ~30 us, a missing one a host round trip):
    // per chunk what it needed in the previous pass, at least 2 (3 when nothing is known); LENTIL_BLIND_ROUNDS fixes it
    int forced_rounds = 0;
    if (const char *e = getenv("LENTIL_BLIND_ROUNDS")) { forced_rounds = atoi(e); if (forced_rounds < 1) forced_rounds = 1; if (forced_rounds > 8) forced_rounds = 8; }
    const int blind_rounds = forced_rounds ? forced_rounds : 3;
    std::vector<int> rounds_of(C, blind_rounds);
    std::vector<DrawArgs> das(C, da);
    // chunks whose last pass left an estimate go first, without any host wait; the others wait for their scan
    std::vector<char> enq(C, 0);
    for (int ci = 0; ci < C; ++ci) {
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->chunks[ci].stream, ctx->chunks[ci].scanned, 0));
      bool done = false;
      if (!forced_rounds && ctx->chunks[ci].have_est) {
        const int r = ctx->chunks[ci].est_rounds;
        rounds_of[ci] = r < 2 ? 2 : (r > 6 ? 6 : r);
      }
      const int rc = enqueue_chunk_draws_blind(ctx, ci, das[ci], rounds_of[ci], &done);
      if (rc) return rc;
      enq[ci] = done ? 1 : 0;
      if (done) ++ctx->last_blind;
    }
    for (int ci = 0; ci < C; ++ci) {
      if (enq[ci]) continue;
      const int rc = enqueue_chunk_draws(ctx, ci, das[ci], blind_rounds);
      if (rc) return rc;
    }
    // every chunk's counters follow its last kernel on its own stream (pinned, asynchronous): one host wait below
    bool any_blind = false;
    for (int ci = 0; ci < C; ++ci) any_blind = any_blind || enq[ci];
    if (any_blind)
      for (int ci = 0; ci < C; ++ci)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_ctr_pinned + ci, ctx->d_ctr + ci, sizeof(DevCounters), hipMemcpyDeviceToHost, ctx->chunks[ci].stream));
    // ---- blind chunks: what the scan really found (next pass's estimate); a chunk that did not fit is redone.
    // The counters read back above are there once every chunk stream has drained; they also serve touched_rows.
    if (any_blind) {
      for (int ci = 0; ci < C; ++ci) HIP_TRY(ctx, hipStreamSynchronize(ctx->chunks[ci].stream));
      ctx->h_ctr.assign(ctx->h_ctr_pinned, ctx->h_ctr_pinned + C);
      ctx->h_ctr_valid = true;
      unsigned long long it = 0, tr = 0, sl = 0;
      for (const DevCounters &k : ctx->h_ctr) { it += k.newton_iters; tr += k.tries; sl += k.slow_solves; }
      if (tr) { ctx->mean_iters = (double)it / (double)tr; ctx->parked_frac = (double)sl / (double)tr; }
    }
    for (int ci = 0; ci < C; ++ci) {
      lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
      if (!enq[ci]) continue;
      const DevCounters c = ctx->h_ctr[(size_t)ci];
      if (c.fallback) {
        ctx->h_ctr_valid = false;
        ++ctx->last_fallback;
        // the empty rounds left their queue cursors behind: fresh queues, then the chunk again with exact sizes
        HIP_TRY(ctx, hipMemsetAsync((char *)(ctx->d_ctr + ci) + offsetof(DevCounters, n_tasks), 0,
                                    offsetof(DevCounters, inv_row_min) - offsetof(DevCounters, n_tasks), ch.stream));
        HIP_TRY(ctx, hipMemsetAsync((char *)(ctx->d_ctr + ci) + offsetof(DevCounters, fallback), 0, sizeof(unsigned long long), ch.stream));
        const int rc = enqueue_chunk_draws(ctx, ci, das[ci], blind_rounds);
        if (rc) return rc;
        rounds_of[ci] = blind_rounds;
      } else {
        const uint64_t cap = ch.v_end - ch.v_begin;
        ch.est_items = c.work_count < cap ? c.work_count : cap;
        ch.est_sum = c.sum_samples;
        ch.est_rounds = (int)c.rounds_used;
        // nothing left open after the blind rounds (the usual case): no round-by-round continuation, no second look
        if (ch.est_items == 0 || c.n_active[rounds_of[ci] & 1] == 0) ch.n_items = 0;
      }
    }
    // ---- any chunk with items still missing draws after the blind rounds continues round by round
    int max_rounds = ctx->last_rounds;
    for (int ci = 0; ci < C; ++ci) if (rounds_of[ci] > max_rounds) max_rounds = rounds_of[ci];
    for (int ci = 0; ci < C; ++ci) {
      lentil_hip_ctx::Chunk &ch = ctx->chunks[ci];
      HIP_TRY(ctx, hipStreamSynchronize(ch.stream));
      if (ch.n_items == 0) continue;
      ctx->h_ctr_valid = false;       // more rounds: the counters move on
      int rounds = rounds_of[ci];
      const int rc = finish_rounds(ctx, ci, das[ci], rounds_of[ci], &rounds);
      if (rounds > ch.est_rounds) ch.est_rounds = rounds;
      if (rc) return rc;
      if (rounds > max_rounds) max_rounds = rounds;
    }
    ctx->last_rounds = max_rounds;
    // what the whole stream held: sizes the next pass if it runs streamed
    {
      uint64_t items = 0, sum = 0;
      int rounds = 0;
      bool all = true;
      for (const auto &ch : ctx->chunks) {
        all = all && ch.have_est;
        items += ch.est_items; sum += ch.est_sum;
        if (ch.est_rounds > rounds) rounds = ch.est_rounds;
      }
      if (all) { ctx->have_total_est = true; ctx->est_items_total = items; ctx->est_sum_total = sum; ctx->est_rounds_total = rounds; }
    }
  }
  if (ctx->F.zkey_dbg && ctx->V.n && !ctx->closest_deferred) {
    hipLaunchKernelGGL(debug_gather_kernel, dim3((unsigned)ctx->num_cu * 8), dim3(256), 0, ctx->stream, ctx->F, ctx->V, P,
                       ctx->have_lens ? ctx->hlens.length : 0.0);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->F.zkey && ctx->V.n && !ctx->closest_deferred) {
    // closest-filter AOVs: the winners of this pass (one pass per frame: the keys index the bound stream)
    hipLaunchKernelGGL(closest_gather_kernel, dim3((unsigned)ctx->num_cu * 8), dim3(256), 0, ctx->stream, ctx->F, ctx->V);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->early_resolve_pending) {
    // second half of the resolve that ran beside the pass's second round: the groups of pixels that received draws
    ctx->early_resolve_pending = false;
    // (a pass whose draws were redone the chunked way added first-round draws after the early half: lentil_hip_resolve
    // then does the whole frame)
    if (ctx->F.dir && ctx->F.touched && !ctx->last_fallback) {
      if (!ctx->late_resolve_done) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_res, 0));
        const int rc = launch_resolve_half(ctx, ctx->stream, 2u);
        if (rc) return rc;
      }
      ctx->resolved_valid = true;
    } else {
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_res, 0));     // (the early half must not overwrite a later resolve)
    }
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  ctx->timed_draw = true;
  if (deferred) return LENTIL_OK;      // (the pass's end is looked at by its next observer: harvest_front -> pass_finish)
  return pass_finish(ctx);
}

// what a pass's counters say once they are known: dropped work, candidates the closest-filtered AOVs cannot order
static int pass_finish(lentil_hip_ctx *ctx) {
  const int C = ctx->n_chunks;
  // A work list, task queue or result pool that was too small drops work on the device (DevCounters::overflow):
  // that is an incomplete frame, not a result
  if (ctx->V.n) {
    if (!ctx->h_ctr_valid) {
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_ctr_pinned, ctx->d_ctr, sizeof(DevCounters) * C, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      ctx->h_ctr.assign(ctx->h_ctr_pinned, ctx->h_ctr_pinned + C);
      ctx->h_ctr_valid = true;
    }
    unsigned long long dropped = 0;
    for (const DevCounters &k : ctx->h_ctr) dropped += k.overflow;
    if (dropped)
      return fail(ctx, LENTIL_ERR_NOMEM, "the device dropped " + std::to_string(dropped) +
                                             " work items (work list, task queue or result pool too small): the frame is incomplete");
    // closest-filtered AOVs: a candidate at |Z| == 0 or NaN makes the reference's result depend on the order of the candidates
    // at that pixel (src/lentil.h:832-837, closest_key_of in lentil_kernels.h)
    unsigned int degenerate = 0;
    for (const DevCounters &k : ctx->h_ctr) degenerate |= k.degenerate_depth;
    // (lentil_hip_redistribute replays those pixels' candidates in visit order, lentil_closest_replay.h)
    ctx->degenerate_seen = degenerate && (ctx->F.zkey || ctx->F.zkey_dbg);
  }
  return crypto_after_pass(ctx);
}

LENTIL_API int lentil_hip_set_closest_exchange(lentil_hip_ctx *ctx, int deferred, uint32_t visit_id_base) {
  CHECK_CTX(ctx);
  ctx->closest_deferred = deferred != 0;
  ctx->visit_id_base = visit_id_base;
  ctx->V.id_base = visit_id_base;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_zkey_buffer(lentil_hip_ctx *ctx, void **device_ptr, uint64_t *n_keys) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  if (device_ptr) *device_ptr = ctx->F.zkey;
  if (n_keys) *n_keys = ctx->F.zkey ? ctx->F.np : 0;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_closest_gather(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  ctx->resolved_valid = false;         // (what an early resolve left in d_resolved no longer describes the frame)
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  if (!ctx->V.n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->F.zkey) {
    hipLaunchKernelGGL(closest_gather_kernel, dim3((unsigned)ctx->num_cu * 8), dim3(256), 0, ctx->stream, ctx->F, ctx->V);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->F.zkey_dbg) {      // lentil_debug: its own key plane (min-reduced by lentil_hip_allreduce), value from the winner's columns
    hipLaunchKernelGGL(debug_gather_kernel, dim3((unsigned)ctx->num_cu * 8), dim3(256), 0, ctx->stream, ctx->F, ctx->V, ctx->P,
                       ctx->have_lens ? ctx->hlens.length : 0.0);
    HIP_TRY(ctx, hipGetLastError());
  }
  return LENTIL_OK;
}

static int resolve_range(lentil_hip_ctx *ctx, uint64_t p_begin, uint64_t p_end) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
  const uint64_t total = p_end - p_begin;
  if (total) {
    uint64_t blocks = (total + 255) / 256;
    const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    const size_t lds = (size_t)4 * 64 * ctx->F.stride * sizeof(float);      // 4 waves x 64 records
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, ctx->F, ctx->d_resolved, p_begin, p_end);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  ctx->timed_resolve = true;
  return LENTIL_OK;
}

// one half of a whole-frame resolve on `st`: everything (only_touched = 0) or the groups that received draws
// only_touched: 0 the whole frame; 1 / 2 the 64-pixel groups whose splat flag is at least that (1: any round's draws, 2: a later round's)
static int launch_resolve_half(lentil_hip_ctx *ctx, hipStream_t st, uint32_t only_touched) {
  const size_t lds = (size_t)4 * 64 * ctx->F.stride * sizeof(float);
  if (only_touched) {
    const uint64_t chunks = ((ctx->F.np + 63) / 64 + 63) / 64;         // 64 groups of 64 pixels per block, dealt round its waves
    uint64_t blocks = chunks;
    const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(resolve_touched_kernel, dim3((unsigned)blocks), dim3(256), lds, st, ctx->F, ctx->d_resolved, only_touched);
  } else {
    // (few waves per CU: the second round's solves and parked solves are a chain of latencies on the same SIMDs, and
    // this kernel has that whole round to finish in)
    int per_cu = 4;
    if (const char *e = getenv("LENTIL_EARLY_RESOLVE_BLOCKS")) { per_cu = atoi(e); if (per_cu < 1) per_cu = 1; if (per_cu > 8) per_cu = 8; }
    uint64_t blocks = (ctx->F.np + 255) / 256;
    const uint64_t max_blocks = (uint64_t)ctx->num_cu * (uint64_t)per_cu;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)blocks), dim3(256), lds, st, ctx->F, ctx->d_resolved, (uint64_t)0, ctx->F.np);
  }
  HIP_TRY(ctx, hipGetLastError());
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_resolve(lentil_hip_ctx *ctx) {
  CHECK_CTX_PIPELINED(ctx);
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  // (behind a pass whose end is open: should that pass turn out to need more work, its observer resolves again)
  if (!ctx->inflight.empty() && !ctx->inflight.back().abandoned) ctx->inflight.back().resolve_requested = true;
  { const int rcj = join_clear(ctx); if (rcj) return rcj; }
  ht_mark(ctx, "resolve");
  if (ctx->resolved_valid) {          // the pass resolved the frame on its way (nothing has touched it since)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    ctx->timed_resolve = true;
    return LENTIL_OK;
  }
  return resolve_range(ctx, 0, ctx->F.np);
}

static int check_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows) {
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  if ((uint64_t)row_begin + n_rows > (uint64_t)ctx->P.yres) return fail(ctx, LENTIL_ERR_INVALID, "row range exceeds the frame");
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_resolve_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows) {
  CHECK_CTX(ctx);
  const int rc = check_rows(ctx, row_begin, n_rows);
  if (rc) return rc;
  return resolve_range(ctx, (uint64_t)row_begin * ctx->P.xres, (uint64_t)(row_begin + n_rows) * ctx->P.xres);
}

LENTIL_API int lentil_hip_pack_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, void *dev_dst) {
  CHECK_CTX(ctx);
  int rc = check_rows(ctx, row_begin, n_rows);
  if (rc) return rc;
  if (!n_rows) return LENTIL_OK;
  if (!dev_dst) return fail(ctx, LENTIL_ERR_INVALID, "dev_dst is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t p_begin = (uint64_t)row_begin * ctx->P.xres, n_pix = (uint64_t)n_rows * ctx->P.xres;
  if ((rc = fold_direct(ctx, p_begin, p_begin + n_pix, false))) return rc;
  uint64_t blocks = (n_pix * (4ull * ctx->F.n_aovs + 1ull) + 255) / 256;
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->F, p_begin, n_pix, (float *)dev_dst);
  HIP_TRY(ctx, hipGetLastError());
  return LENTIL_OK;
}

static int merge_rows_impl(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, const void *dev_acc_rows,
                           const void *dev_key_rows, const void *dev_key_rows_dbg, bool packed);

// The public entry points of the tiled exchange carry one key plane; frames with a lentil_debug AOV have two, and only
// the library's own exchange (lentil_comm.h) passes the second
#define LENTIL_NO_DEBUG_AOV(ctx)                                                                                          \
  if ((ctx)->F.debug_mask)                                                                                                \
  return fail(ctx, LENTIL_ERR_UNSUPPORTED, "frames with a lentil_debug AOV are exchanged by lentil_hip_exchange_bands / _allreduce only")

static int compact_rows_impl(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, void *dev_idx, void *dev_vals,
                             void *dev_keys, void *dev_keys_dbg, uint32_t capacity, uint32_t *count, unsigned int *d_count_own);

LENTIL_API int lentil_hip_compact_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, void *dev_idx,
                                       void *dev_vals, void *dev_keys, uint32_t capacity, uint32_t *count) {
  CHECK_CTX(ctx);
  LENTIL_NO_DEBUG_AOV(ctx);
  return compact_rows_impl(ctx, row_begin, n_rows, dev_idx, dev_vals, dev_keys, nullptr, capacity, count, nullptr);
}

// `d_count_own`: the count goes to a word of the caller's (zeroed by the caller) and stays on the device -- nothing waits
static int compact_rows_impl(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, void *dev_idx, void *dev_vals,
                             void *dev_keys, void *dev_keys_dbg, uint32_t capacity, uint32_t *count, unsigned int *d_count_own) {
  CHECK_CTX(ctx);
  int rc = check_rows(ctx, row_begin, n_rows);
  if (rc) return rc;
  if (!count && !d_count_own) return fail(ctx, LENTIL_ERR_INVALID, "count is null");
  if (count) *count = 0;
  if (!n_rows) return LENTIL_OK;
  if (!dev_idx || !dev_vals) return fail(ctx, LENTIL_ERR_INVALID, "dev_idx / dev_vals is null");
  if (ctx->F.zkey && !dev_keys) return fail(ctx, LENTIL_ERR_INVALID, "the frame has closest-filtered AOVs: dev_keys is required");
  if (ctx->F.zkey_dbg && !dev_keys_dbg) return fail(ctx, LENTIL_ERR_INVALID, "the frame has a lentil_debug AOV: its keys are required");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t p_begin = (uint64_t)row_begin * ctx->P.xres, n_pix = (uint64_t)n_rows * ctx->P.xres;
  if ((rc = fold_direct(ctx, p_begin, p_begin + n_pix, false))) return rc;
  // the shared slot of the counter block (draw-log cursor) is free between passes
  unsigned int *d_count = d_count_own ? d_count_own : reinterpret_cast<unsigned int *>(&ctx->d_ctr[ctx->n_chunks].overflow);
  if (!d_count_own) HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned int), ctx->stream));
  uint64_t blocks = (n_pix + 255) / 256;
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->F, p_begin, n_pix,
                     (uint32_t *)dev_idx, (float *)dev_vals, ctx->F.zkey ? (unsigned long long *)dev_keys : nullptr,
                     ctx->F.zkey_dbg ? (unsigned long long *)dev_keys_dbg : nullptr, capacity, d_count);
  HIP_TRY(ctx, hipGetLastError());
  if (d_count_own) return LENTIL_OK;
  unsigned int n = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n, d_count, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *count = n;
  return LENTIL_OK;
}

static int merge_sparse_impl(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, uint32_t n, const void *dev_idx,
                             const void *dev_vals, const void *dev_keys, const void *dev_keys_dbg, const uint32_t *n_dev = nullptr);

LENTIL_API int lentil_hip_merge_sparse(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, uint32_t n,
                                       const void *dev_idx, const void *dev_vals, const void *dev_keys) {
  CHECK_CTX(ctx);
  LENTIL_NO_DEBUG_AOV(ctx);
  return merge_sparse_impl(ctx, row_begin, n_rows, n, dev_idx, dev_vals, dev_keys, nullptr);
}

// `n_dev`: n is the capacity of the arrays and the entry count is read from *n_dev by the kernels (above n: nothing merged)
static int merge_sparse_impl(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, uint32_t n, const void *dev_idx,
                             const void *dev_vals, const void *dev_keys, const void *dev_keys_dbg, const uint32_t *n_dev) {
  ctx->resolved_valid = false;         // (what an early resolve left in d_resolved no longer describes the frame)
  CHECK_CTX(ctx);
  int rc = check_rows(ctx, row_begin, n_rows);
  if (rc) return rc;
  if (!n) return LENTIL_OK;
  if (!dev_idx || !dev_vals) return fail(ctx, LENTIL_ERR_INVALID, "dev_idx / dev_vals is null");
  if (ctx->F.zkey && !dev_keys) return fail(ctx, LENTIL_ERR_INVALID, "the frame has closest-filtered AOVs: dev_keys is required");
  if (ctx->F.zkey_dbg && !dev_keys_dbg) return fail(ctx, LENTIL_ERR_INVALID, "the frame has a lentil_debug AOV: its keys are required");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  untrust_touched(ctx);
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
  uint64_t blocks = ((uint64_t)n * (4ull * ctx->F.n_aovs + 1ull) + 255) / 256;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(merge_sparse_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->F, n, (const uint32_t *)dev_idx,
                     (const float *)dev_vals, ctx->F.zkey ? (const unsigned long long *)dev_keys : nullptr,
                     ctx->F.zkey_dbg ? (const unsigned long long *)dev_keys_dbg : nullptr, n_dev, n);
  HIP_TRY(ctx, hipGetLastError());
  uint64_t kb = ((uint64_t)n + 255) / 256;
  if (kb > max_blocks) kb = max_blocks;
  if (ctx->F.zkey) {
    hipLaunchKernelGGL(merge_sparse_keys_kernel, dim3((unsigned)kb), dim3(256), 0, ctx->stream, ctx->F.zkey, ctx->F.np, n,
                       (const uint32_t *)dev_idx, (const unsigned long long *)dev_keys, n_dev, n);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->F.zkey_dbg) {
    hipLaunchKernelGGL(merge_sparse_keys_kernel, dim3((unsigned)kb), dim3(256), 0, ctx->stream, ctx->F.zkey_dbg, ctx->F.np, n,
                       (const uint32_t *)dev_idx, (const unsigned long long *)dev_keys_dbg, n_dev, n);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->dirty_known) {       // the entries lie in rows [row_begin, row_begin + n_rows), the sender's compact_rows range
    if ((int32_t)row_begin < ctx->dirty_lo || ctx->dirty_hi <= ctx->dirty_lo) ctx->dirty_lo = (int32_t)row_begin;
    if ((int32_t)(row_begin + n_rows) > ctx->dirty_hi) ctx->dirty_hi = (int32_t)(row_begin + n_rows);
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_merge_packed_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows,
                                            const void *dev_packed_rows, const void *dev_key_rows) {
  CHECK_CTX(ctx);
  LENTIL_NO_DEBUG_AOV(ctx);
  return merge_rows_impl(ctx, row_begin, n_rows, dev_packed_rows, dev_key_rows, nullptr, true);
}

LENTIL_API int lentil_hip_merge_rows(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, const void *dev_acc_rows,
                                     const void *dev_key_rows) {
  CHECK_CTX(ctx);
  LENTIL_NO_DEBUG_AOV(ctx);
  return merge_rows_impl(ctx, row_begin, n_rows, dev_acc_rows, dev_key_rows, nullptr, false);
}

static int merge_rows_impl(lentil_hip_ctx *ctx, uint32_t row_begin, uint32_t n_rows, const void *dev_acc_rows,
                           const void *dev_key_rows, const void *dev_key_rows_dbg, bool packed) {
  ctx->resolved_valid = false;         // (what an early resolve left in d_resolved no longer describes the frame)
  CHECK_CTX(ctx);
  int rc = check_rows(ctx, row_begin, n_rows);
  if (rc) return rc;
  if (!n_rows) return LENTIL_OK;
  if (!dev_acc_rows) return fail(ctx, LENTIL_ERR_INVALID, "dev_acc_rows is null");
  if (ctx->F.zkey && !dev_key_rows) return fail(ctx, LENTIL_ERR_INVALID, "the frame has closest-filtered AOVs: key rows are required");
  if (ctx->F.zkey_dbg && !dev_key_rows_dbg) return fail(ctx, LENTIL_ERR_INVALID, "the frame has a lentil_debug AOV: its key rows are required");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  untrust_touched(ctx);
  const uint64_t p_begin = (uint64_t)row_begin * ctx->P.xres, n_pix = (uint64_t)n_rows * ctx->P.xres;
  uint64_t blocks = (n_pix * ctx->F.stride + 255) / 256;
  const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
  if (blocks > max_blocks) blocks = max_blocks;
  if (packed)
    hipLaunchKernelGGL(merge_packed_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->F, p_begin, n_pix,
                       (const float *)dev_acc_rows, ctx->F.zkey ? (const unsigned long long *)dev_key_rows : nullptr,
                       ctx->F.zkey_dbg ? (const unsigned long long *)dev_key_rows_dbg : nullptr);
  else
    hipLaunchKernelGGL(merge_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ctx->F, p_begin, n_pix,
                       (const float *)dev_acc_rows, ctx->F.zkey ? (const unsigned long long *)dev_key_rows : nullptr,
                       ctx->F.zkey_dbg ? (const unsigned long long *)dev_key_rows_dbg : nullptr);
  HIP_TRY(ctx, hipGetLastError());
  uint64_t kb = (n_pix + 255) / 256;
  if (kb > max_blocks) kb = max_blocks;
  if (ctx->F.zkey) {
    hipLaunchKernelGGL(merge_keys_kernel, dim3((unsigned)kb), dim3(256), 0, ctx->stream, ctx->F.zkey, p_begin, n_pix,
                       (const unsigned long long *)dev_key_rows);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->F.zkey_dbg) {
    hipLaunchKernelGGL(merge_keys_kernel, dim3((unsigned)kb), dim3(256), 0, ctx->stream, ctx->F.zkey_dbg, p_begin, n_pix,
                       (const unsigned long long *)dev_key_rows_dbg);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (ctx->dirty_known) {
    if ((int32_t)row_begin < ctx->dirty_lo || ctx->dirty_hi <= ctx->dirty_lo) ctx->dirty_lo = (int32_t)row_begin;
    if ((int32_t)(row_begin + n_rows) > ctx->dirty_hi) ctx->dirty_hi = (int32_t)(row_begin + n_rows);
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_touched_rows(lentil_hip_ctx *ctx, int32_t *row_lo, int32_t *row_hi) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<DevCounters> c((size_t)ctx->n_chunks);
  if (ctx->h_ctr_valid && ctx->h_ctr.size() == c.size()) {
    c = ctx->h_ctr;                 // a blind pass has read them back already, after its last kernel
  } else {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(c.data(), ctx->d_ctr, sizeof(DevCounters) * (size_t)ctx->n_chunks, hipMemcpyDeviceToHost));
  }
  int64_t lo = INT32_MAX, hi = 0;
  for (const DevCounters &k : c) {
    if (!k.row_max_p1) continue;
    const int64_t l = 0x7FFFFFFFll - (int64_t)k.inv_row_min;
    if (l < lo) lo = l;
    if ((int64_t)k.row_max_p1 > hi) hi = (int64_t)k.row_max_p1;
  }
  // a uniform stream adds every non-redistributed visit to its own pixel: the rows of the stream itself
  const VisitsDev &vh = ctx->V;
  if (vh.n && vh.visits_per_pixel) {
    const uint64_t row_visits = (uint64_t)vh.pixels_per_row * vh.visits_per_pixel;
    const uint64_t rows = (vh.n + row_visits - 1) / row_visits;
    const int64_t l = vh.pixel_y0, h = (int64_t)vh.pixel_y0 + (int64_t)(rows - 1) * vh.pixel_row_stride + 1;
    if (l < lo) lo = l;
    if (h > hi) hi = h;
  }
  if (hi <= lo) { lo = 0; hi = 0; }
  if (hi > (int64_t)ctx->P.yres) hi = ctx->P.yres;
  if (lo < 0) lo = 0;
  if (row_lo) *row_lo = (int32_t)lo;
  if (row_hi) *row_hi = (int32_t)hi;
  // everything added since the last clear lies in these rows: the next clear_frame only has to wipe them
  if (ctx->dirty_known && ctx->dirty_hi > ctx->dirty_lo) {
    if (ctx->dirty_lo < lo) lo = ctx->dirty_lo;
    if (ctx->dirty_hi > hi) hi = ctx->dirty_hi;
  }
  ctx->dirty_lo = (int32_t)lo; ctx->dirty_hi = (int32_t)hi;
  ctx->pass_pending = false;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_sync(lentil_hip_ctx *ctx) {
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ht_mark(ctx, "sync{");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ht_mark(ctx, "}sync");
  ht_dump(ctx);
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_aov(lentil_hip_ctx *ctx, uint32_t aov, float *host_rgba) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame || aov >= ctx->F.n_aovs || !host_rgba) return fail(ctx, LENTIL_ERR_INVALID, "bad download_aov arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(host_rgba, ctx->d_resolved + (size_t)aov * ctx->F.np * 4, ctx->F.np * 16,
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_accum(lentil_hip_ctx *ctx, uint32_t aov, float *host_rgba, float *host_weight) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame || aov >= ctx->F.n_aovs) return fail(ctx, LENTIL_ERR_INVALID, "bad download_accum arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // debugging / test path: fetch the interleaved pixel records and pick the requested columns on the host
  {
    const int rc = fold_direct(ctx, 0, ctx->F.np, true);
    if (rc) return rc;
  }
  const uint64_t np = ctx->F.np;
  const uint32_t rec = ctx->F.stride;
  std::vector<float> tmp(np * rec);
  HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), ctx->F.acc, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (uint64_t p = 0; p < np; ++p) {
    if (host_rgba) memcpy(host_rgba + p * 4, tmp.data() + p * rec + 4u * aov, 16);
    if (host_weight) host_weight[p] = tmp[p * rec + 4u * ctx->F.n_aovs];
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_records(lentil_hip_ctx *ctx, float *host_records, uint64_t capacity_floats, uint32_t *stride) {
  CHECK_CTX(ctx);
  if (!ctx->have_frame) return fail(ctx, LENTIL_ERR_INVALID, "no frame allocated");
  if (stride) *stride = ctx->F.stride;
  if (!host_records) return LENTIL_OK;
  const uint64_t n = ctx->F.np * ctx->F.stride;
  if (capacity_floats < n) return fail(ctx, LENTIL_ERR_INVALID, "download_records: the host block is smaller than xres * yres * stride floats");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rc = fold_direct(ctx, 0, ctx->F.np, true);
    if (rc) return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(host_records, ctx->F.acc, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_accum_buffer(lentil_hip_ctx *ctx, void **device_ptr, uint64_t *n_floats) {
  CHECK_CTX(ctx);
  ctx->resolved_valid = false;         // (what an early resolve left in d_resolved no longer describes the frame)
  if (!ctx->have_frame || !device_ptr || !n_floats) return fail(ctx, LENTIL_ERR_INVALID, "bad accum_buffer arguments");
  {
    const int rc = fold_direct(ctx, 0, ctx->F.np, true);      // the caller reduces `acc` across GPUs
    if (rc) return rc;
  }
  *device_ptr = ctx->F.acc;
  *n_floats = ctx->F.np * ctx->F.stride;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_stream(lentil_hip_ctx *ctx, void **hip_stream) {
  CHECK_CTX(ctx);
  if (!hip_stream) return fail(ctx, LENTIL_ERR_INVALID, "hip_stream is null");
  *hip_stream = (void *)ctx->stream;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_get_counters(lentil_hip_ctx *ctx, lentil_counters *out) {
  CHECK_CTX(ctx);
  if (!out) return fail(ctx, LENTIL_ERR_INVALID, "out is null");
  std::vector<DevCounters> c(ctx->n_chunks + 1);
  if (ctx->h_ctr_valid && ctx->h_ctr.size() == (size_t)ctx->n_chunks) {
    // the pass read them back itself, behind its last kernel
    for (int i = 0; i < ctx->n_chunks; ++i) c[i] = ctx->h_ctr[i];
  } else {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(c.data(), ctx->d_ctr, sizeof(DevCounters) * c.size(), hipMemcpyDeviceToHost));
  }
  memset(out, 0, sizeof(*out));
  out->visits = ctx->have_visits ? ctx->V.n : 0;
  for (int i = 0; i < ctx->n_chunks; ++i) {
    out->redistributed_visits += c[i].redistributed;
    out->attempted_draws += c[i].attempted;
    out->accepted_draws += c[i].accepted;
    out->worklist_overflow += c[i].overflow;
    out->newton_iterations += c[i].newton_iters;
    out->tries += c[i].tries;
    out->lane_rounds += c[i].lane_rounds;
    out->slow_solves += c[i].slow_solves;
  }
  out->blind_chunks = ctx->last_blind;
  out->fallback_chunks = ctx->last_fallback;
  out->streamed = ctx->last_streamed;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_last_timing(lentil_hip_ctx *ctx, float ms[3]) {
  CHECK_CTX(ctx);
  if (!ms) return fail(ctx, LENTIL_ERR_INVALID, "ms is null");
  ms[0] = ms[1] = ms[2] = 0.f;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->timed_draw) {
    HIP_TRY(ctx, hipEventElapsedTime(&ms[0], ctx->ev[0], ctx->ev[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&ms[1], ctx->ev[1], ctx->ev[2]));
    if (ctx->scan_kernel_timed) {
      // the scan launch by its own start / stop events; the draws: the rest of scan start -> end of the draws
      float k = 0.f;
      if (hipEventElapsedTime(&k, ctx->ev_scan_k[0], ctx->ev_scan_k[1]) == hipSuccess && k > 0.f && k <= ms[0]) {
        ms[1] += ms[0] - k;
        ms[0] = k;
      }
    }
  }
  if (ctx->timed_resolve) HIP_TRY(ctx, hipEventElapsedTime(&ms[2], ctx->ev[3], ctx->ev[4]));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_last_launches(lentil_hip_ctx *ctx, uint32_t n[2]) {
  CHECK_CTX(ctx);
  if (!n) return fail(ctx, LENTIL_ERR_INVALID, "n is null");
  n[0] = ctx->last_scan_launches;
  n[1] = (uint32_t)ctx->last_rounds;
  return LENTIL_OK;
}

static void crypto_log_set_by_caller(lentil_hip_ctx *ctx);
LENTIL_API int lentil_hip_set_draw_log(lentil_hip_ctx *ctx, uint64_t capacity) {
  CHECK_CTX(ctx);
  crypto_log_set_by_caller(ctx);      // (the cryptomatte replay re-marks a log it allocates itself)
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->d_log);
  ctx->d_log = nullptr;
  ctx->log_cap = 0;
  if (capacity) {
    HIP_TRY(ctx, hipMalloc(&ctx->d_log, capacity * sizeof(lentil_draw_record)));
    ctx->log_cap = capacity;
  }
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_download_draw_log(lentil_hip_ctx *ctx, lentil_draw_record *out, uint64_t capacity,
                                            uint64_t *n_records) {
  CHECK_CTX(ctx);
  if (!n_records) return fail(ctx, LENTIL_ERR_INVALID, "n_records is null");
  DevCounters c;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpyAsync(&c, ctx->d_ctr + ctx->n_chunks, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *n_records = c.log_count;
  uint64_t n = c.log_count < ctx->log_cap ? c.log_count : ctx->log_cap;
  if (n > capacity) n = capacity;
  if (out && n) HIP_TRY(ctx, hipMemcpy(out, ctx->d_log, n * sizeof(lentil_draw_record), hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

// ----- single-function tests -----------------------------------------------------------
template <typename T>
static int dev_copy_in(lentil_hip_ctx *ctx, const T *src, size_t n, T **dst, std::vector<void *> &tmp) {
  *dst = nullptr;
  if (!src || !n) return LENTIL_OK;
  HIP_TRY(ctx, hipMalloc((void **)dst, n * sizeof(T)));
  tmp.push_back(*dst);
  HIP_TRY(ctx, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return LENTIL_OK;
}
template <typename T>
static int dev_alloc(lentil_hip_ctx *ctx, size_t n, T **dst, std::vector<void *> &tmp) {
  HIP_TRY(ctx, hipMalloc((void **)dst, n * sizeof(T)));
  tmp.push_back(*dst);
  return LENTIL_OK;
}

// ---- lentil_hip_box_probe: what the box delivers (see lentil_hip.h) ---------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void probe_fp64_kernel(double *sink, uint32_t rounds, unsigned long long *clk) {
  // eight independent multiply-add chains per lane (the solves interleave about as many): v_mul_f64 / v_add_f64, no FMA
  double a0 = 1.0 + threadIdx.x * 1e-9, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9, a4 = a0 + 4e-9, a5 = a0 + 5e-9, a6 = a0 + 6e-9, a7 = a0 + 7e-9;
  const double m = 1.0000000001, c = 1e-12;
  const unsigned long long t0 = clock64(), r0 = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < rounds; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    }
  }
  const unsigned long long t1 = clock64(), r1 = __builtin_amdgcn_s_memrealtime();
  const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 12345.678) sink[0] = s;          // (keeps the chains)
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ __launch_bounds__(256) void probe_copy_kernel(const float4 *src, float4 *dst, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void probe_read_kernel(const float4 *src, float *sink, uint64_t n) {
  float acc = 0.f;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 v = nt_load(src + i);
    acc += (v.x + v.y) + (v.z + v.w);
  }
  if (acc == 12345.678f) sink[0] = acc;
}

LENTIL_API int lentil_hip_box_probe(lentil_hip_ctx *ctx, double probe[6]) {
  CHECK_CTX(ctx);
  if (!probe) return fail(ctx, LENTIL_ERR_INVALID, "probe is null");
  for (int i = 0; i < 6; ++i) probe[i] = 0.0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t n4 = (512ull << 20) / 16;             // 512 MiB each way: beyond every cache
  float4 *a = nullptr, *b = nullptr;
  unsigned long long *clk = nullptr;
  HIP_TRY(ctx, hipMalloc(&a, n4 * 16));
  hipError_t e = hipMalloc(&b, n4 * 16);
  if (e == hipSuccess) e = hipMalloc(&clk, 2 * sizeof(unsigned long long));
  hipEvent_t ev[2] = {nullptr, nullptr};
  if (e == hipSuccess) e = hipEventCreate(&ev[0]);
  if (e == hipSuccess) e = hipEventCreate(&ev[1]);
  auto timed = [&](auto &&launch, int reps, float &best_ms) {
    best_ms = 1e30f;
    for (int r = 0; r < reps && e == hipSuccess; ++r) {
      e = hipEventRecord(ev[0], ctx->stream);
      launch();
      if (e == hipSuccess) e = hipGetLastError();
      if (e == hipSuccess) e = hipEventRecord(ev[1], ctx->stream);
      if (e == hipSuccess) e = hipEventSynchronize(ev[1]);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev[0], ev[1]);
      if (e == hipSuccess && ms < best_ms) best_ms = ms;
    }
  };
  if (e == hipSuccess) e = hipMemsetAsync(a, 0, n4 * 16, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(b, 0, n4 * 16, ctx->stream);
  const unsigned blocks = (unsigned)ctx->num_cu * 3u;       // three 256-thread blocks per CU: three waves per SIMD
  const uint32_t rounds = 4096;
  float ms = 0.f;
  unsigned long long h_clk[2] = {0, 0};
  // (the first launch brings the clocks up; the best of the following three counts)
  timed([&]() { hipLaunchKernelGGL(probe_fp64_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<double *>(b), rounds, clk); }, 4, ms);
  if (e == hipSuccess) e = hipMemcpy(h_clk, clk, sizeof h_clk, hipMemcpyDeviceToHost);
  if (e == hipSuccess && ms > 0.f) {
    probe[0] = (double)blocks * 256.0 * (double)rounds * 128.0 / ((double)ms * 1e-3) / 1e12;
    if (h_clk[1]) probe[1] = (double)h_clk[0] / (double)h_clk[1] * 100.0;
  }
  timed([&]() { hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)ctx->num_cu * 8u), dim3(256), 0, ctx->stream, a, b, n4); }, 3, ms);
  if (e == hipSuccess && ms > 0.f) probe[2] = 2.0 * (double)n4 * 16.0 / ((double)ms * 1e-3) / 1e9;
  timed([&]() { hipLaunchKernelGGL(probe_read_kernel, dim3((unsigned)ctx->num_cu * 8u), dim3(256), 0, ctx->stream, a, reinterpret_cast<float *>(b), n4); }, 3, ms);
  if (e == hipSuccess && ms > 0.f) probe[3] = (double)n4 * 16.0 / ((double)ms * 1e-3) / 1e9;
  probe[4] = getenv("GPU_MAX_HW_QUEUES") ? atof(getenv("GPU_MAX_HW_QUEUES")) : 0.0;
  probe[5] = (double)ctx->num_cu;
  if (ev[0]) (void)hipEventDestroy(ev[0]);
  if (ev[1]) (void)hipEventDestroy(ev[1]);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(clk);
  if (e != hipSuccess) return fail(ctx, LENTIL_ERR_HIP, std::string("box probe: ") + hipGetErrorString(e));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_process_stats(uint64_t stats[4]) {
  if (!stats) return LENTIL_ERR_INVALID;
  stats[0] = g_stat_streamed.load(); stats[1] = g_stat_stuck.load(); stats[2] = g_stat_stuck_injected.load(); stats[3] = g_stat_redone.load();
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_process_stall_notes(char *buf, uint64_t capacity) {
  if (!buf || !capacity) return LENTIL_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g_stall_notes_mutex);
  std::string all;
  for (const std::string &n : g_stall_notes) all += n + "\n";
  const size_t k = all.size() < capacity - 1 ? all.size() : (size_t)capacity - 1;
  memcpy(buf, all.data(), k);
  buf[k] = 0;
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_batch_model_stats(lentil_hip_ctx *ctx, uint64_t stats[4]) {
  CHECK_CTX(ctx);
  if (!stats) return fail(ctx, LENTIL_ERR_INVALID, "stats is null");
  stats[0] = ctx->n_bm_built; stats[1] = ctx->n_lean; stats[2] = ctx->n_lean_lost; stats[3] = ctx->bm_margin16;
  return LENTIL_OK;
}

// one wave per point: what publish_kernel's waves compute for an item there
__global__ __launch_bounds__(64) void batch_estimate_kernel(BatchModelDev M, uint64_t n, const float *cs, uint32_t samples, uint32_t retries,
                                                            uint32_t extra_num, uint32_t extra_const, float *out) {
  const uint64_t i = blockIdx.x;
  if (i >= n) return;
  const uint32_t plain = first_batch_hi(samples, retries, extra_num, extra_const);
  const BatchEstimate e = batch_estimate(M, cs[3 * i], cs[3 * i + 1], cs[3 * i + 2], threadIdx.x);
  if (threadIdx.x == 0) {
    out[4 * i] = e.q_strict; out[4 * i + 1] = e.q; out[4 * i + 2] = e.fail;
    out[4 * i + 3] = (float)batch_from_estimate(e, samples, retries, plain, M.margin16);
  }
}

LENTIL_API int lentil_hip_debug_batch_estimate(lentil_hip_ctx *ctx, uint64_t n, const float *cs_xyz, uint32_t samples, float *out) {
  CHECK_CTX(ctx);
  if (!cs_xyz || !out) return fail(ctx, LENTIL_ERR_INVALID, "null argument");
  if (!ctx->have_params || !ctx->have_lens || ctx->P.cameraType != LENTIL_POLYNOMIAL_OPTICS)
    return fail(ctx, LENTIL_ERR_INVALID, "the first-batch model needs polynomial-optics parameters and a lens");
  if (!n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure_batch_model(ctx))) return rc;
  if (!ctx->bm_valid) return fail(ctx, LENTIL_ERR_INVALID, "no first-batch model for these parameters");
  std::vector<void *> tmp;
  float *d_cs = nullptr, *d_out = nullptr;
  if ((rc = dev_copy_in(ctx, cs_xyz, (size_t)n * 3, &d_cs, tmp)) == LENTIL_OK && (rc = dev_alloc(ctx, (size_t)n * 4, &d_out, tmp)) == LENTIL_OK) {
    const uint32_t retries = (uint32_t)(ctx->P.vignetting_retries < 0 ? 0 : ctx->P.vignetting_retries);
    hipLaunchKernelGGL(batch_estimate_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, batch_model_dev(ctx), n, d_cs, samples, retries,
                       ctx->extra_num, ctx->extra_const, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(ctx, LENTIL_ERR_HIP, std::string("batch estimate: ") + hipGetErrorString(e));
  }
  for (void *p : tmp) (void)hipFree(p);
  return rc;
}

struct TmpFree {
  std::vector<void *> v;
  ~TmpFree() { for (void *p : v) (void)hipFree(p); }
};

LENTIL_API int lentil_hip_test_lt_sample_aperture(lentil_hip_ctx *ctx, uint64_t n, const double *scene,
                                                  const double *ap, double lambda, double *sensor,
                                                  double *out, double *transmittance) {
  CHECK_CTX(ctx);
  if (!ctx->have_lens) return fail(ctx, LENTIL_ERR_INVALID, "set_lens first");
  if (!n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpFree tf;
  TestArgs t{};
  t.P = ctx->P; t.lens = ctx->d_lens; t.terms = ctx->d_terms; t.bokeh = ctx->bokeh; t.n = n; t.lambda = lambda;
  double *d_scene, *d_ap, *o0, *o1, *o2;
  int rc;
  if ((rc = dev_copy_in(ctx, scene, n * 3, &d_scene, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, ap, n * 2, &d_ap, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 5, &o0, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 5, &o1, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n, &o2, tf.v))) return rc;
  t.in0 = d_scene; t.in1 = d_ap; t.o0 = o0; t.o1 = o1; t.o2 = o2;
  hipLaunchKernelGGL(test_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(sensor, o0, n * 5 * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(out, o1, n * 5 * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(transmittance, o2, n * 8, hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

// Camera::logarithmic_focus_search on the GPU: the candidates on the host exactly as the reference produces them
// (the running fp64 sum of logarithmic_values and libm's pow, src/lens.h:395-407), one lane per candidate, one block
// for the ordered minimum.
static void focus_candidates(std::vector<double> &shift) {
  shift.clear();
  for (double i = -1.0; i <= 1.0; i += 0.0001) shift.push_back((i < 0 ? -1 : 1) * std::pow(i, 2.0) * (45.0 - 0.0) + 0.0);
}

static int run_focus(lentil_hip_ctx *ctx, const double *shift, uint32_t n, double focal_distance, double lambda,
                     double *miss, double *sensor, double *out, double *best) {
  TmpFree tf;
  FocusArgs f{};
  f.lens = ctx->d_lens; f.terms = ctx->d_terms; f.lambda = lambda; f.housing_radius = ctx->lens_housing_radius;
  f.focal_distance = focal_distance; f.n = n;
  double *d_shift, *d_miss, *d_sensor = nullptr, *d_out = nullptr, *d_best;
  int rc;
  if ((rc = dev_copy_in(ctx, shift, n, &d_shift, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n, &d_miss, tf.v))) return rc;
  if (sensor && (rc = dev_alloc(ctx, (uint64_t)n * 5, &d_sensor, tf.v))) return rc;
  if (out && (rc = dev_alloc(ctx, (uint64_t)n * 5, &d_out, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, 2, &d_best, tf.v))) return rc;
  f.shift = d_shift; f.miss = d_miss; f.sensor = d_sensor; f.out = d_out; f.best = d_best;
  hipLaunchKernelGGL(focus_miss_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, f);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(focus_argmin_kernel, dim3(1), dim3(1024), 0, ctx->stream, f);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (miss) HIP_TRY(ctx, hipMemcpy(miss, d_miss, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (sensor) HIP_TRY(ctx, hipMemcpy(sensor, d_sensor, (size_t)n * 40, hipMemcpyDeviceToHost));
  if (out) HIP_TRY(ctx, hipMemcpy(out, d_out, (size_t)n * 40, hipMemcpyDeviceToHost));
  if (best) HIP_TRY(ctx, hipMemcpy(best, d_best, 16, hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_focus_search(lentil_hip_ctx *ctx, double focal_distance, double lambda, double *best_sensor_shift) {
  CHECK_CTX(ctx);
  if (!ctx->have_lens) return fail(ctx, LENTIL_ERR_INVALID, "set_lens first");
  if (!best_sensor_shift) return fail(ctx, LENTIL_ERR_INVALID, "best_sensor_shift is null");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<double> shift;
  focus_candidates(shift);
  double best[2] = {0.0, 0.0};
  const int rc = run_focus(ctx, shift.data(), (uint32_t)shift.size(), focal_distance, lambda, nullptr, nullptr, nullptr, best);
  if (rc) return rc;
  *best_sensor_shift = best[0];
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_test_y0_intersection(lentil_hip_ctx *ctx, uint64_t n, const double *sensor_shift, double lambda,
                                               double *distance, double *sensor, double *out) {
  CHECK_CTX(ctx);
  if (!ctx->have_lens) return fail(ctx, LENTIL_ERR_INVALID, "set_lens first");
  if (!n) return LENTIL_OK;
  if (n > 0x7FFFFFFFull || !sensor_shift || !distance) return fail(ctx, LENTIL_ERR_INVALID, "bad test_y0_intersection arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = run_focus(ctx, sensor_shift, (uint32_t)n, 0.0, lambda, distance, sensor, out, nullptr);
  if (rc) return rc;
  for (uint64_t i = 0; i < n; ++i) distance[i] = 0.0 - distance[i];      // miss = 0 - distance, exactly
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_test_trace_bw_po(lentil_hip_ctx *ctx, uint64_t n, const double *target,
                                           const int32_t *px, const int32_t *py, const int32_t *attempt,
                                           double *sensor_xy, int32_t *ok) {
  CHECK_CTX(ctx);
  if (!ctx->have_lens || !ctx->have_params) return fail(ctx, LENTIL_ERR_INVALID, "set_params and set_lens first");
  if (ctx->P.bokeh_enable_image && !ctx->have_bokeh) return fail(ctx, LENTIL_ERR_INVALID, "set_bokeh first");
  if (!n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpFree tf;
  TestArgs t{};
  t.P = ctx->P; t.lens = ctx->d_lens; t.terms = ctx->d_terms; t.bokeh = ctx->bokeh; t.n = n;
  double *d_t, *o0;
  int32_t *d_px, *d_py, *d_at, *oi;
  int rc;
  if ((rc = dev_copy_in(ctx, target, n * 3, &d_t, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, px, n, &d_px, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, py, n, &d_py, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, attempt, n, &d_at, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 2, &o0, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n, &oi, tf.v))) return rc;
  t.in0 = d_t; t.i0 = d_px; t.i1 = d_py; t.i2 = d_at; t.o0 = o0; t.oi = oi;
  hipLaunchKernelGGL(test_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(sensor_xy, o0, n * 2 * 8, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(ok, oi, n * 4, hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

LENTIL_API int lentil_hip_test_aperture_sample(lentil_hip_ctx *ctx, uint64_t n, const uint32_t *a,
                                               const uint32_t *b, double *xy) {
  CHECK_CTX(ctx);
  if (!ctx->have_params) return fail(ctx, LENTIL_ERR_INVALID, "set_params first");
  if (ctx->P.bokeh_enable_image && !ctx->have_bokeh) return fail(ctx, LENTIL_ERR_INVALID, "set_bokeh first");
  if (!n) return LENTIL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpFree tf;
  TestArgs t{};
  t.P = ctx->P; t.lens = nullptr; t.terms = nullptr; t.bokeh = ctx->bokeh; t.n = n;
  uint32_t *d_a, *d_b;
  double *o0;
  int rc;
  if ((rc = dev_copy_in(ctx, a, n, &d_a, tf.v))) return rc;
  if ((rc = dev_copy_in(ctx, b, n, &d_b, tf.v))) return rc;
  if ((rc = dev_alloc(ctx, n * 2, &o0, tf.v))) return rc;
  t.u0 = d_a; t.u1 = d_b; t.o0 = o0;
  hipLaunchKernelGGL(test_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(xy, o0, n * 2 * 8, hipMemcpyDeviceToHost));
  return LENTIL_OK;
}

#include "lentil_upload.h"
#include "lentil_comm.h"
#include "lentil_crypto.h"

#ifdef LENTIL_TIMELINE
// development aid (tools/timeline.py): copies the event histogram out and clears it
LENTIL_API int lentil_hip_debug_timeline(unsigned int *out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned int) * kTlSub * kTlChannels * kTlBuckets) != hipSuccess) return LENTIL_ERR_HIP;
  if (out && hipMemcpyFromSymbol(out + kTlSub * kTlChannels * kTlBuckets, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 32) != hipSuccess) return LENTIL_ERR_HIP;
  if (out && hipMemcpyFromSymbol(out + kTlSub * kTlChannels * kTlBuckets + 64, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 64) != hipSuccess) return LENTIL_ERR_HIP;
  if (reset) {
    void *sp = nullptr;
    unsigned long long init[32][2];
    for (int i = 0; i < 32; ++i) { init[i][0] = ~0ull; init[i][1] = 0ull; }
    if (hipGetSymbolAddress(&sp, HIP_SYMBOL(g_span)) != hipSuccess || hipMemcpy(sp, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return LENTIL_ERR_HIP;
    void *q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_dbg)) != hipSuccess || hipMemset(q, 0, sizeof(unsigned long long) * 32) != hipSuccess) return LENTIL_ERR_HIP;
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_timeline)) != hipSuccess) return LENTIL_ERR_HIP;
    if (hipMemset(p, 0, sizeof(unsigned int) * kTlSub * kTlChannels * kTlBuckets) != hipSuccess) return LENTIL_ERR_HIP;
  }
  return LENTIL_OK;
}
#endif
