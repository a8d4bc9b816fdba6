// Microbenchmark: does a SIMD hand the registers a finished wave gives back to a new wave while OLDER waves are still resident?
// The streamed pass keeps two 168-register solve waves per SIMD beside the scan's 80-register wave; when the scan's waves leave, 176
// of 512 registers are free by count, and yet neither a third solve block nor the straggler waves (160) are placed before the first
// solve waves leave ("not in the registers the scan gives back").  A wave's registers are one contiguous range --
// so it depends on WHERE the hole is, and on whether the allocator can reuse a hole behind a live allocation at all.
//
// X = long-lived 168-register blocks (the resident solve blocks), Y = a short-lived block of NY registers with 100 KB of LDS (the
// scan: one per CU), Z = a 168-register block launched once Y has ended.  Z's blocks record when they start; X is released long
// after.  Every spin is bounded by wall-clock ticks (100 MHz), nothing waits for a flag.
// usage: vgpr_alloc          (runs every scenario, prints how many of Z's blocks began before X was released)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void now_kernel(unsigned long long *t) { *t = wall_clock64(); }

// a block that holds NV registers (and LDS) until the absolute tick `until`; rec[2b] = start, rec[2b+1] = end
template <int NV, int THREADS, int LDS>
__global__ __launch_bounds__(THREADS) void hold(unsigned long long until, unsigned long long *rec) {
  __shared__ char s[LDS > 0 ? LDS : 1];
  extern __shared__ char dyn[];
  if (LDS > 0) s[threadIdx.x] = 1;
  if (LDS == 0) dyn[threadIdx.x] = 1;
  const unsigned long long t0 = wall_clock64();
  if (NV == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if (NV == 176) asm volatile("v_mov_b32 v175, 0" ::: "v175");
  if (NV == 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
  if (NV == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (NV == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  unsigned long long t = t0;
  while (t < until && t - t0 < 5000000ull) {      // (50 ms at most, whatever the host asked for)
    __builtin_amdgcn_s_sleep(32);
    t = wall_clock64();
  }
  if (threadIdx.x == 0) { rec[2 * blockIdx.x] = t0; rec[2 * blockIdx.x + 1] = t; }
}

struct Ctx {
  int cus;
  hipStream_t sx, sx2, sy, sz;
  unsigned long long *d_now, *rx, *rx2, *ry, *rz;
};

static unsigned long long device_now(Ctx &c) {
  hipLaunchKernelGGL(now_kernel, dim3(1), dim3(1), 0, c.sz, c.d_now);
  CK(hipStreamSynchronize(c.sz));
  unsigned long long t;
  CK(hipMemcpy(&t, c.d_now, 8, hipMemcpyDeviceToHost));
  return t;
}

enum { ORDER_XXY = 0, ORDER_YXX = 1, ORDER_XYX = 2, ORDER_XX = 3 };

template <int NY, int ZT, int NZ = 168>
static void scenario(Ctx &c, const char *name, int order, int z_per_cu, int x_ms = 8) {
  const unsigned cus = (unsigned)c.cus;
  const unsigned long long T0 = device_now(c);
  const unsigned long long tick = 100;                       // ticks per us
  const unsigned long long y_until = T0 + 3000 * tick;       // Y leaves 3 ms in
  const unsigned long long x_until = T0 + (unsigned long long)x_ms * 1000 * tick;
  const unsigned long long z_until = 0;                      // Z leaves at once (it records its start)
  CK(hipMemset(c.rz, 0, 16 * 4 * cus * 4));
  auto launch_x = [&](hipStream_t st, unsigned long long *rec, unsigned per_cu) {
    hipLaunchKernelGGL((hold<168, 256, 28 * 1024>), dim3(cus * per_cu), dim3(256), 0, st, x_until, rec);
  };
  auto launch_y = [&]() {
    hipLaunchKernelGGL((hold<NY, 256, 0>), dim3(cus), dim3(256), 100 * 1024, c.sy, y_until, c.ry);
  };
  // (X as two launches of one block per CU: a 256-block grid on an idle chip goes one block to a CU, a 512-block grid need not go two)
  if (order == ORDER_XXY) { launch_x(c.sx, c.rx, 1); usleep(400); launch_x(c.sx2, c.rx2, 1); usleep(600); launch_y(); }
  if (order == ORDER_YXX) { launch_y(); usleep(600); launch_x(c.sx, c.rx, 1); usleep(400); launch_x(c.sx2, c.rx2, 1); }
  if (order == ORDER_XYX) { launch_x(c.sx, c.rx, 1); usleep(600); launch_y(); usleep(600); launch_x(c.sx2, c.rx2, 1); }
  if (order == ORDER_XX) { launch_x(c.sx, c.rx, 1); usleep(400); launch_x(c.sx2, c.rx2, 1); }
  CK(hipGetLastError());
  if (order != ORDER_XX) CK(hipStreamSynchronize(c.sy)); else usleep(3000);
  usleep(200);
  const unsigned zb = cus * (unsigned)z_per_cu;
  hipLaunchKernelGGL((hold<NZ, ZT, 7 * 1024>), dim3(zb), dim3(ZT), 0, c.sz, z_until, c.rz);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> z(2 * zb), x(2 * cus * 2), y(2 * cus);
  CK(hipMemcpy(z.data(), c.rz, 16 * zb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(x.data(), c.rx, 16 * cus, hipMemcpyDeviceToHost));
  CK(hipMemcpy(x.data() + 2 * cus, c.rx2, 16 * cus, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y.data(), c.ry, 16 * cus, hipMemcpyDeviceToHost));
  unsigned long long y_end = 0, x_start_max = 0;
  if (order != ORDER_XX) for (unsigned b = 0; b < cus; ++b) y_end = std::max(y_end, y[2 * b + 1]);
  else y_end = T0 + 3000 * tick;
  for (unsigned b = 0; b < cus * 2u; ++b) x_start_max = std::max(x_start_max, x[2 * b]);
  std::vector<double> zs;
  unsigned early = 0;
  for (unsigned b = 0; b < zb; ++b) {
    zs.push_back(((double)z[2 * b] - (double)y_end) / 100.0);
    if (z[2 * b] < x_until) ++early;
  }
  std::sort(zs.begin(), zs.end());
  printf("%-58s Z blocks begun before X left: %4u of %4u | Z start behind Y's end, us: min %8.1f median %8.1f max %8.1f | X left %.1f us behind Y; X all resident %.1f us in\n",
         name, early, zb, zs.front(), zs[zs.size() / 2], zs.back(), ((double)x_until - (double)y_end) / 100.0, ((double)x_start_max - (double)T0) / 100.0);
  fflush(stdout);
}

int main() {
  Ctx c;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  c.cus = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.gcnArchName, c.cus);
  CK(hipStreamCreateWithFlags(&c.sx, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.sx2, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.sy, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.sz, hipStreamNonBlocking));
  CK(hipMalloc(&c.d_now, 8));
  CK(hipMalloc(&c.rx, 16 * c.cus * 2)); CK(hipMalloc(&c.rx2, 16 * c.cus * 2)); CK(hipMalloc(&c.ry, 16 * c.cus)); CK(hipMalloc(&c.rz, 16 * c.cus * 4 * 4));
  CK(hipFuncSetAttribute((const void *)hold<80, 256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute((const void *)hold<96, 256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute((const void *)hold<176, 256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  for (int rep = 0; rep < 2; ++rep) {
    printf("-- repetition %d\n", rep);
    scenario<80, 256>(c, "control: X X resident, no Y; Z 168 x 256 thr (176 free at the top)", ORDER_XX, 1);
    scenario<80, 256>(c, "X X then Y(80); Y leaves; Z 168 x 256 thr", ORDER_XXY, 1);
    scenario<80, 64>(c, "X X then Y(80); Y leaves; Z 168 x 64 thr, 4 per CU", ORDER_XXY, 4);
    scenario<80, 64, 160>(c, "X X then Y(80); Y leaves; Z 160 x 64 thr, 4 per CU", ORDER_XXY, 4);
    scenario<80, 64, 96>(c, "X X then Y(80); Y leaves; Z 96 x 64 thr, 4 per CU", ORDER_XXY, 4);
    scenario<80, 256>(c, "Y(80) then X X; Y leaves; Z 168 x 256 thr (holes 80 + 96)", ORDER_YXX, 1);
    scenario<80, 64, 96>(c, "Y(80) then X X; Y leaves; Z 96 x 64 thr, 4 per CU", ORDER_YXX, 4);
    scenario<80, 64, 80>(c, "Y(80) then X X; Y leaves; Z 80 x 64 thr, 8 per CU", ORDER_YXX, 8);
    scenario<176, 256>(c, "Y(176) then X X; Y leaves; Z 168 x 256 thr", ORDER_YXX, 1);
    scenario<176, 64>(c, "Y(176) then X X; Y leaves; Z 168 x 64 thr, 4 per CU", ORDER_YXX, 4);
    scenario<80, 256>(c, "X, Y(80), X; Y leaves; Z 168 x 256 thr", ORDER_XYX, 1);
    scenario<96, 256>(c, "X, Y(96), X; Y leaves; Z 168 x 256 thr", ORDER_XYX, 1);
  }
  return 0;
}
