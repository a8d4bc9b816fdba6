// Microbenchmark: how many scattered "splats" per second does the chip take?  A splat = 5 consecutive fp32 atomic adds into a
// 32-byte pixel record (what accept_kernel does per accepted draw of a beauty-only frame), records chosen at random
// in a 266 MB buffer, lanes transposed as the accept kernel does (lane q adds float q % 5 of splat q / 5).
// usage: atomic_rate [splats] [blocks] [mode]   mode 0: atomicAdd (agent scope), 1: one lane per splat adds float 0 only,
//        2: plain stores instead of atomics (what the memory system does without the read-modify-write)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void splat(float *acc, unsigned char *touched, unsigned long long n_rec, unsigned long long n_splats, int mode) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned long long n_waves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
  const unsigned per = mode == 1 ? 64u : 12u;
  unsigned k = 0;
  for (unsigned long long s0 = wave * per; s0 < n_splats; s0 += n_waves * per) {
    const unsigned d = mode == 1 ? lane : lane / 5u, ch = mode == 1 ? 0u : lane - d * 5u;
    if (d >= per || s0 + d >= n_splats) continue;
    unsigned long long h = (s0 + d) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    unsigned long long rec = h % n_rec;
    if (mode >= 3) {
      // mode 3: the splats of one "item" (1024 consecutive splats) fall inside a disc of `mode3_radius` pixels around the item's
      // own pixel, as the draws of a highlight do; mode 4: additionally flag touched[pix >> 6] with a byte store
      const unsigned long long item = (s0 + d) >> 10;
      unsigned long long hi = item * 0xD6E8FEB86659FD93ull; hi ^= hi >> 32; hi *= 0xD6E8FEB86659FD93ull; hi ^= hi >> 32;
      const int cx = (int)(hi % 3841ull), cy = (int)((hi >> 20) % 2161ull);
      const int R = 100;
      int dx = (int)(h % (2 * R + 1)) - R, dy = (int)((h >> 24) % (2 * R + 1)) - R;
      int x = cx + dx, y = cy + dy;
      x = x < 0 ? 0 : (x > 3840 ? 3840 : x); y = y < 0 ? 0 : (y > 2160 ? 2160 : y);
      rec = (unsigned long long)y * 3841ull + x;
      if (mode == 4 && ch == 0) touched[rec >> 6] = 1;
    }
    float *p = acc + rec * 8ull + ch;
    if (mode == 2) *p = 1.0f; else atomicAdd(p, 1.0f);
    // modes 5..: a wave waits for its atomics after every (mode - 4) x 5 of them, as a kernel does whose next step needs a
    // load (vmcnt counts in order: the load's data only comes once the atomics before it are through)
    if (mode >= 5 && ((++k) % (5u * (unsigned)(mode - 4))) == 0u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
int main(int argc, char **argv) {
  const unsigned long long n_splats = argc > 1 ? strtoull(argv[1], 0, 10) : 1200000ull;
  const int blocks = argc > 2 ? atoi(argv[2]) : 512;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;
  const unsigned long long n_rec = 3841ull * 2161ull;
  float *acc;
  hipMalloc(&acc, n_rec * 32);
  hipMemset(acc, 0, n_rec * 32);
  unsigned char *touched; hipMalloc(&touched, n_rec / 64 + 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(splat, dim3(blocks), dim3(256), 0, 0, acc, touched, n_rec, n_splats, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it) printf("splats %llu blocks %d mode %d: %.3f ms -> %.2f M splats/ms\n", n_splats, blocks, mode, ms, n_splats / ms * 1e-6);
  }
  return 0;
}
