// PMC calibration: known byte counts through the scan kernel's access pattern (float4 per lane,
// several column streams), to convert FETCH_SIZE / WRITE_SIZE of rocprofv3 into HBM bytes on gfx950
// (MI355X_MICROARCH.md, section HBM: "calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate tools/pmc_calibrate.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- /tmp/pmc_calibrate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_one_stream(const float4 *a, size_t n, float *out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void read_five_streams(const float4 *a, const float4 *b, const float4 *c, const float4 *d,
                                  const float4 *e, size_t n, float *out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i], w = b[i], x = c[i], y = d[i], z = e[i];
    s += v.x + w.y + x.z + y.w + z.x;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void write_stream(float4 *a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const size_t n = 74649600;          // visits of the 4K bench frame; 16 B each per column
  float4 *col[5];
  float *out;
  for (auto &c : col) { hipMalloc(&c, n * 16); hipMemset(c, 0, n * 16); }
  hipMalloc(&out, 4);
  for (int rep = 0; rep < 3; ++rep) {
    read_one_stream<<<2048, 256>>>(col[0], n, out);
    read_five_streams<<<2048, 256>>>(col[0], col[1], col[2], col[3], col[4], n, out);
    write_stream<<<2048, 256>>>(col[0], n);
  }
  hipDeviceSynchronize();
  printf("bytes: one stream %zu, five streams %zu, write %zu\n", n * 16, n * 80, n * 16);
  return 0;
}
