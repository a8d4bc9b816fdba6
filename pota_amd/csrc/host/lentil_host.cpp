// lentil_host.cpp -- CPU-side setup code of the plugin mirror (liblentil_host.so).
#include "../../../include/lentil_host.h"

#include <algorithm>
#include <vector>

#define HOST_API extern "C" __attribute__((visibility("default")))

namespace {
// descending-by-value index comparator, src/imagebokeh.h:21-27
struct ByValueDesc {
  const float *v;
  bool operator()(int a, int b) const { return v[a] > v[b]; }
};
}  // namespace

HOST_API int lentil_host_bokeh_probability(const float *pix, int32_t x, int32_t y, int32_t nch, float *cdfRow,
                                           int32_t *rowIndices, float *cdfColumn, int32_t *columnIndices) {
  if (!pix || x <= 0 || y <= 0 || nch < 3 || x != y) return -1;
  const int n = x * y;
  // luminance, fp32, sequential total (src/imagebokeh.h:164-168)
  std::vector<float> lum(n), prob(n), rowSum(y), rowProb(n);
  float total = 0.0f;
  for (int i = 0, j = 0; i < n; ++i, j += nch) {
    lum[i] = pix[j] * 0.3f + pix[j + 1] * 0.59f + pix[j + 2] * 0.11f;
    total += lum[i];
  }
  const float invTotal = 1.0f / total;                       // :180
  for (int i = 0; i < n; ++i) prob[i] = lum[i] * invTotal;   // :183-186
  for (int r = 0, k = 0; r < y; ++r) {                       // :204-212
    float s = 0.0f;
    for (int c = 0; c < x; ++c, ++k) s += prob[k];
    rowSum[r] = s;
  }
  for (int r = 0; r < y; ++r) rowIndices[r] = r;
  std::sort(rowIndices, rowIndices + y, ByValueDesc{rowSum.data()});   // :238
  float run = 0.0f;
  for (int r = 0; r < y; ++r) {                              // :254-258
    cdfRow[r] = run + rowSum[rowIndices[r]];
    run = cdfRow[r];
  }
  for (int r = 0, i = 0; r < y; ++r)                         // :273-285
    for (int c = 0; c < x; ++c, ++i)
      rowProb[i] = (prob[i] != 0 && rowSum[r] != 0) ? prob[i] / rowSum[r] : 0.0f;
  for (int i = 0; i < n; ++i) columnIndices[i] = i;
  for (int i = 0; i < n; i += x) std::sort(columnIndices + i, columnIndices + i + x, ByValueDesc{rowProb.data()});  // :301-303
  for (int r = 0, i = 0; r < y; ++r) {                       // :319-328
    run = 0.0f;
    for (int c = 0; c < x; ++c, ++i) {
      cdfColumn[i] = run + rowProb[columnIndices[i]];
      run = cdfColumn[i];
    }
  }
  return 0;
}
