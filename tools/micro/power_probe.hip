// Microbenchmark: what does an fp64-bound kernel lose while an HBM-bound kernel runs beside it -- and is it the clock?
// The streamed pass overlaps its HBM-bound scan with fp64 Newton solves: two solve waves per SIMD run at 15.7 G lane-iterations/s
// with the chip to themselves (LENTIL_SOLVE_BLOCKS=2 on the chunked pass, round 6) and at 10-11 beside and for a while BEHIND the
// scan.  Occupancy it is not.  This asks the chip directly:
//   F = dependent fp64 multiply / add chains (no FMA), two 256-thread blocks of 168 registers per CU, ~6 ms
//   C = a streaming copy (float4, read + write) over 2 x 512 MiB, one 256-thread block per CU, ~6 ms
//   S = one wave that samples the shader clock (clock64() against the 100 MHz counter) every 50 us
// F alone, C alone, F beside C, and F started 1 ms after C has ENDED; the sampler runs through all of it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void f_kernel(double *sink, uint32_t rounds, unsigned long long *clk) {
  asm volatile("v_mov_b32 v167, 0" ::: "v167");
  double a0 = 1.0 + threadIdx.x * 1e-9, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9, a4 = a0 + 4e-9, a5 = a0 + 5e-9, a6 = a0 + 6e-9, a7 = a0 + 7e-9;
  const double m = 1.0000000001, c = 1e-12;
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  for (uint32_t i = 0; i < rounds; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    }
  }
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 12345.678) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; clk[2] = r0; clk[3] = r1; }
}
__global__ __launch_bounds__(256) void c_kernel(const float4 *src, float4 *dst, uint64_t n, uint32_t loops, unsigned long long *clk) {
  const unsigned long long r0 = wall_clock64();
  for (uint32_t l = 0; l < loops; ++l)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
      dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = r0; clk[3] = wall_clock64(); }
}
__global__ void sampler(unsigned long long *out, uint32_t n, uint32_t period_ticks) {
  unsigned long long next = wall_clock64(), c_prev = clock64(), r_prev = next;
  for (uint32_t i = 0; i < n; ++i) {
    next += period_ticks;
    while (wall_clock64() < next) __builtin_amdgcn_s_sleep(16);
    const unsigned long long c = clock64(), r = wall_clock64();
    out[2 * i] = r; out[2 * i + 1] = (c - c_prev) * 100ull / (r - r_prev ? r - r_prev : 1ull);      // MHz over the last period
    c_prev = c; r_prev = r;
  }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const unsigned cus = (unsigned)p.multiProcessorCount;
  hipStream_t sf, sc, ss;
  CK(hipStreamCreateWithFlags(&sf, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
  const uint64_t n4 = (512ull << 20) / 16;
  float4 *a, *b; double *sink; unsigned long long *clkf, *clkc, *samp;
  const uint32_t n_samp = 2400;          // 120 ms at 50 us
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clkf, 64)); CK(hipMalloc(&clkc, 64)); CK(hipMalloc(&samp, 16ull * n_samp));
  CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
  const uint32_t rounds = 12000, loops = 28;
  auto F = [&](int blocks_per_cu) { hipLaunchKernelGGL(f_kernel, dim3(cus * blocks_per_cu), dim3(256), 0, sf, sink, rounds, clkf); };
  auto C = [&]() { hipLaunchKernelGGL(c_kernel, dim3(cus), dim3(256), 0, sc, a, b, n4, loops, clkc); };
  // warm up (clocks, code objects)
  F(2); C(); CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(sampler, dim3(1), dim3(1), 0, ss, samp, n_samp, 5000u);
  usleep(3000);
  struct Mark { const char *what; unsigned long long f[4], c[4]; int bp; };
  std::vector<Mark> marks;
  auto read = [&](const char *what, bool f, bool c, int bp = 2) {
    Mark m{}; m.what = what; m.bp = bp;
    if (f) CK(hipMemcpy(m.f, clkf, 32, hipMemcpyDeviceToHost));
    if (c) CK(hipMemcpy(m.c, clkc, 32, hipMemcpyDeviceToHost));
    marks.push_back(m);
  };
  for (int bp = 2; bp <= 3; ++bp) {
    F(bp); CK(hipStreamSynchronize(sf)); read(bp == 2 ? "F alone, 2 blocks/CU" : "F alone, 3 blocks/CU", true, false, bp); usleep(3000);
  }
  C(); CK(hipStreamSynchronize(sc)); read("C alone", false, true); usleep(3000);
  for (int bp = 2; bp <= 3; ++bp) {
    C(); F(bp); CK(hipStreamSynchronize(sf)); CK(hipStreamSynchronize(sc)); read(bp == 2 ? "F (2 blocks/CU) beside C" : "F (3 blocks/CU) beside C", true, true, bp); usleep(3000);
  }
  C(); CK(hipStreamSynchronize(sc)); F(2); CK(hipStreamSynchronize(sf)); read("F (2 blocks/CU) straight behind C", true, true); usleep(3000);
  F(2); CK(hipStreamSynchronize(sf)); read("F alone again", true, false);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> s(2 * n_samp);
  CK(hipMemcpy(s.data(), samp, 16ull * n_samp, hipMemcpyDeviceToHost));
  const double flop = (double)cus * 256.0 * rounds * 128.0;
  const unsigned long long t_base = s[0];
  for (auto &m : marks) {
    printf("%-40s", m.what);
    if (m.f[1]) printf(" F: %7.3f ms (block 0), shader clock %6.1f MHz over it, %5.1f TFLOP/s", m.f[1] / 1e5, (double)m.f[0] / m.f[1] * 100.0, flop * m.bp / (m.f[1] / 1e8) / 1e12);
    if (m.c[3]) printf(" | C: %7.3f ms (block 0) = %6.0f GB/s", (m.c[3] - m.c[2]) / 1e5, (double)n4 * 32.0 * loops / ((m.c[3] - m.c[2]) / 1e8) / 1e9);
    if (m.f[1]) printf(" | F ran %8.2f .. %8.2f ms", (m.f[2] - (double)t_base) / 1e5, (m.f[3] - (double)t_base) / 1e5);
    if (m.c[3]) printf(" | C ran %8.2f .. %8.2f ms", (m.c[2] - (double)t_base) / 1e5, (m.c[3] - (double)t_base) / 1e5);
    printf("\n");
  }
  printf("# shader clock, MHz, one figure per 0.5 ms (mean of ten 50-us samples), from the sampler's start:\n");
  for (uint32_t i = 0; i + 10 <= n_samp; i += 10) {
    unsigned long long sum = 0; for (int k = 0; k < 10; ++k) sum += s[2 * (i + k) + 1];
    printf("%s%5.1fms:%4llu", (i / 10) % 8 == 0 ? "\n" : "  ", (s[2 * i] - t_base) / 1e5, sum / 10);
  }
  printf("\n");
  return 0;
}
