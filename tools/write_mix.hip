// What does a thin stream of record writes cost inside a wide read stream on MI355X?  (the scan
// kernels read ~720-1900 B of visit columns per pixel and update a 32-160 B pixel record.)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/write_mix tools/write_mix.hip && gpurun_out/write_mix
// Every wave step reads 8 KiB (8 v4f per lane from 8 "columns") and owes REC bytes of records; variants:
//   mode 0  no record traffic                      mode 1  store the step's records every step
//   mode 2  read-modify-write them every step      mode 3  collect B steps in LDS, store B*REC contiguous bytes
//   mode 4  as 3 with read-modify-write (records requested at the start of the B steps)
//   mode 5  as 1, the record address wrapped into 1 MiB (stays in L2: is it the memory or the store path?)
//   mode 6  as 3, with a grid-wide barrier before and after the stores: chip-wide read and write phases
//           (spins are bounded: a grid that is not co-resident gives wrong timing, not a hang)
// Waves walk B consecutive tiles, then jump by the grid, so that mode 3/4 bursts are contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kCols = 8;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ void grid_barrier(unsigned int *ctr, unsigned int target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);
    for (int spin = 0; spin < 200000 && __atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target; ++spin) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
}

template <int MODE, int B, int REC4>   // REC4: v4f of records per step (<= 64)
__global__ __launch_bounds__(256) void mix(const v4f *__restrict__ cols, v4f *rec, uint64_t n_tiles, uint64_t col_stride, unsigned int *bar) {
  __shared__ v4f stage[4][B * REC4 > 0 ? B * REC4 : 1];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t wave_global = (uint64_t)blockIdx.x * 4 + wave, wave_stride = (uint64_t)gridDim.x * 4;
  unsigned int phase = 0;
  const uint64_t t_end = MODE == 6 ? ((n_tiles + wave_stride * B - 1) / (wave_stride * B)) * (wave_stride * B) : n_tiles;
  for (uint64_t t0 = wave_global * B; t0 < t_end; t0 += wave_stride * B) {
    v4f old[(B * REC4 + 63) / 64];
    if (MODE == 4) {
#pragma unroll
      for (int u = 0; u < (B * REC4 + 63) / 64; ++u) {
        const uint32_t i = lane + 64u * u;
        old[u] = i < (uint32_t)(B * REC4) ? rec[t0 * REC4 + i] : v4f{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll 1
    for (int b = 0; b < B; ++b) {
      const uint64_t t = t0 + b;
      if (t >= n_tiles) break;
      v4f cur = v4f{0.f, 0.f, 0.f, 0.f};
      if (MODE == 2 && lane < REC4) cur = rec[t * REC4 + lane];
      v4f v[kCols];
#pragma unroll
      for (int c = 0; c < kCols; ++c) v[c] = __builtin_nontemporal_load(cols + (uint64_t)c * col_stride + t * 64 + lane);
      v4f s = cur;
#pragma unroll
      for (int c = 0; c < kCols; ++c) s += v[c];
      if (REC4 < 64) { s.x += __shfl_xor(s.x, 32); s.y += __shfl_xor(s.y, 32); s.z += __shfl_xor(s.z, 32); s.w += __shfl_xor(s.w, 32); }
      if (MODE == 0) { if (s.x == 1.2345e-30f) rec[lane] = s; }
      if ((MODE == 1 || MODE == 2) && lane < REC4) rec[t * REC4 + lane] = s;
      if (MODE == 5 && lane < REC4) rec[(t * REC4 + lane) & 0xFFFFu] = s;
      if ((MODE == 3 || MODE == 4 || MODE == 6) && lane < REC4) stage[wave][b * REC4 + lane] = s;
    }
    if (MODE == 6) grid_barrier(bar, ++phase * gridDim.x);
    if (MODE == 3 || MODE == 4 || MODE == 6) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
      for (int u = 0; u < (B * REC4 + 63) / 64; ++u) {
        const uint32_t i = lane + 64u * u;
        if (i < (uint32_t)(B * REC4) && t0 * REC4 + i < n_tiles * REC4) {
          v4f s = stage[wave][i];
          if (MODE == 4) s += old[u];
          rec[t0 * REC4 + i] = s;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
    }
    if (MODE == 6) grid_barrier(bar, ++phase * gridDim.x);
  }
}

template <int MODE, int B, int REC4>
static int run(const char *name, const v4f *cols, v4f *rec, uint64_t n_tiles, uint64_t col_stride, int blocks, bool clear = true) {
  static unsigned int *bar = nullptr;
  if (!bar) CK(hipMalloc(&bar, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (MODE == 6) {                      // the barrier needs every block resident at once
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mix<MODE, B, REC4>, 256, 0));
    if (occ < 1) { printf("%s: does not fit\n", name); return 0; }
    if (occ < 4) blocks = blocks / 4 * occ;
  }
  float best = 1e30f;
  for (int it = 0; it < 6; ++it) {
    if (clear || it == 0) CK(hipMemsetAsync(rec, 0, n_tiles * REC4 * 16, 0));
    CK(hipMemsetAsync(bar, 0, 4, 0));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mix<MODE, B, REC4>), dim3(blocks), dim3(256), 0, 0, cols, rec, n_tiles, col_stride, bar);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  const double rd = (double)n_tiles * 64 * 16 * kCols, wr = MODE ? (double)n_tiles * REC4 * 16 : 0.0;
  printf("%-44s %7.3f ms  reads %.2f GB at %.2f TB/s  records %.3f GB%s\n", name, best, rd / 1e9, rd / best / 1e9, wr / 1e9,
         (MODE == 2 || MODE == 4) ? " (read + written)" : "");
  return 0;
}

int main() {
  const uint64_t n_tiles = 720000;              // x 8 KiB = 5.9 GB of columns
  const uint64_t col_stride = n_tiles * 64;
  v4f *cols, *rec;
  CK(hipMalloc(&cols, col_stride * kCols * 16));
  CK(hipMalloc(&rec, n_tiles * 64 * 16));
  CK(hipMemset(cols, 0, col_stride * kCols * 16));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 4;
  // records: 32 v4f = 512 B per 8 KiB step (6 %: the single-AOV scan), 64 v4f = 1 KiB (12 %: nine AOVs)
  if (run<0, 1, 32>("no records", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<1, 1, 32>("512 B stored per step", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<2, 1, 32>("512 B read-modify-written per step", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<3, 4, 32>("2 KiB stored every 4 steps", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<3, 16, 32>("8 KiB stored every 16 steps", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<4, 16, 32>("8 KiB read-modify-written every 16 steps", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<1, 1, 64>("1 KiB stored per step", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<2, 1, 64>("1 KiB read-modify-written per step", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<3, 8, 64>("8 KiB stored every 8 steps", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<4, 8, 64>("8 KiB read-modify-written every 8 steps", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<1, 1, 64>("1 KiB stored per step, records not cleared", cols, rec, n_tiles, col_stride, blocks, false)) return 1;
  if (run<2, 1, 64>("1 KiB r-m-w per step, records not cleared", cols, rec, n_tiles, col_stride, blocks, false)) return 1;
  if (run<5, 1, 64>("1 KiB stored per step into 1 MiB", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<1, 1, 16>("256 B stored per step", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<6, 16, 32>("8 KiB every 16 steps, chip-wide phases", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<6, 8, 64>("8 KiB every 8 steps, chip-wide phases", cols, rec, n_tiles, col_stride, blocks)) return 1;
  if (run<6, 16, 64>("16 KiB every 16 steps, chip-wide phases", cols, rec, n_tiles, col_stride, blocks)) return 1;
  return 0;
}
