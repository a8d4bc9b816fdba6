// A one-wave kernel that samples the shader clock while something else runs on the chip (tools/clock_trace.py loads this as a
// shared library into the process that runs the passes): clock64() against the 100 MHz counter every `period_ticks`.
// sampler_stamp puts the 100 MHz counter's value into a slot from a stream of the caller's (a pass's start, in the sampler's time).
#include <hip/hip_runtime.h>
#include <cstdint>
static unsigned long long *g_buf = nullptr, *g_stamps = nullptr;
static hipStream_t g_stream = nullptr;
static uint32_t g_n = 0;
__global__ void sampler_kernel(unsigned long long *out, uint32_t n, uint32_t period_ticks) {
  unsigned long long next = wall_clock64(), c_prev = clock64(), r_prev = next;
  for (uint32_t i = 0; i < n; ++i) {
    next += period_ticks;
    while (wall_clock64() < next) __builtin_amdgcn_s_sleep(8);
    const unsigned long long c = clock64(), r = wall_clock64();
    out[2 * i] = r; out[2 * i + 1] = (c - c_prev) * 100ull / (r - r_prev ? r - r_prev : 1ull);
    c_prev = c; r_prev = r;
  }
}
__global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }
extern "C" int sampler_start(uint32_t n, uint32_t period_ticks) {
  if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
  if (g_buf) { (void)hipFree(g_buf); g_buf = nullptr; }
  if (!g_stamps && hipMalloc(&g_stamps, 8 * 4096) != hipSuccess) return -2;
  if (hipMemset(g_stamps, 0, 8 * 4096) != hipSuccess) return -3;
  if (hipMalloc(&g_buf, 16ull * n) != hipSuccess) return -4;
  g_n = n;
  hipLaunchKernelGGL(sampler_kernel, dim3(1), dim3(1), 0, g_stream, g_buf, n, period_ticks);
  return hipGetLastError() == hipSuccess ? 0 : -5;
}
extern "C" int sampler_stamp(void *stream, uint32_t slot) {
  if (!g_stamps || slot >= 4096) return -1;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g_stamps + slot);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int sampler_read(unsigned long long *host_pairs, unsigned long long *host_stamps) {
  if (!g_buf) return -1;
  if (hipStreamSynchronize(g_stream) != hipSuccess) return -2;
  if (hipMemcpy(host_pairs, g_buf, 16ull * g_n, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  if (host_stamps && hipMemcpy(host_stamps, g_stamps, 8 * 4096, hipMemcpyDeviceToHost) != hipSuccess) return -4;
  return 0;
}
