/*
 * lentil_host.h -- host-side (CPU, once-per-render) helpers of the plugin mirror:
 * liblentil_host.so.  These produce the tables / parameters that liblentil_hip.so consumes.
 * C-ABI, plain pointers and sizes.
 */
#ifndef LENTIL_HOST_H
#define LENTIL_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* imageData::bokehProbability (src/imagebokeh.h:143-338): builds the row CDF and the per-row
 * column CDFs, both sorted by descending probability with std::sort (tie order matters and is
 * libstdc++'s), from x*y*nchannels float texels as AiTextureLoad delivers them
 * (src/imagebokeh.h:16-18,107).  Output arrays: cdfRow[y], rowIndices[y], cdfColumn[x*y],
 * columnIndices[x*y].  Returns 0, or -1 for an invalid image (non-square, < 3 channels:
 * src/imagebokeh.h:50-52,97-101). */
int lentil_host_bokeh_probability(const float *pixelData, int32_t x, int32_t y, int32_t nchannels,
                                  float *cdfRow, int32_t *rowIndices, float *cdfColumn,
                                  int32_t *columnIndices);

#ifdef __cplusplus
}
#endif
#endif
