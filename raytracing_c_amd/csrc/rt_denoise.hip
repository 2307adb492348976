// rt_denoise.hip -- the reference's post-process (denoiser.c:51-153) on the gathered u8 frame.
// a 3x3 luminance-sorted median blended in by how much the centre pixel
// deviates from its neighbourhood.  One thread per pixel; the nine RGB8 taps come from L1/L2 (a frame
// is 6 MB), so the kernel moves 3 B in + 3 B out per pixel of HBM traffic and is bandwidth bound.
// Same arithmetic, in the same order, as oracle_denoise_image() (bit-exact tests).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_math.h"

#define DENOISING_THRESHOLD  0.0125f       // denoiser.c:13
#define NEIGHBOURHOOD_WEIGHT 5             // denoiser.c:14

__global__ void rt_denoise_kernel(int width, int height, int src_stride, int src_comp, int dst_stride, int dst_comp,
                                  const uint8_t *src, uint8_t *dst) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= width || y >= height) return;

  const float k = 1.0f / 255.999f;          // u8 * k == u8 / 255.999f for all 256 inputs (tests/test_oracle_kat.py)
  float lr[9], lg[9], lb[9], ll[9];
  float o_r = 0, o_g = 0, o_b = 0, o_l = 0;
  const int nc = src_comp < 3 ? src_comp : 3;
#pragma unroll
  for (int t = 0; t < 9; t++) {
    int xx = x + (t % 3) - 1, yy = y + (t / 3) - 1;           // yo outer, xo inner: denoiser.c:76-77
    xx = xx < 0 ? 0 : (xx >= width ? width - 1 : xx);
    yy = yy < 0 ? 0 : (yy >= height ? height - 1 : yy);
    const uint8_t *p = src + ((size_t)xx + (size_t)yy * src_stride) * src_comp;
    float r = (float)(int)p[0] * k;
    float g = nc > 1 ? (float)(int)p[1] * k : 0.0f;
    float b = nc > 2 ? (float)(int)p[2] * k : 0.0f;
    float l = r * 0.2126f + g * 0.7152f + b * 0.0722f;       // denoiser.c:16-18
    if (t == 4) { o_r = r; o_g = g; o_b = b; o_l = l; }
    lr[t] = r; lg[t] = g; lb[t] = b; ll[t] = l;
    // stable insertion (before the first strictly brighter entry, denoiser.c:85-101) as a bubble from the end
#pragma unroll
    for (int j = t; j > 0; j--) {
      bool sw = ll[j - 1] > ll[j];
      float a0 = ll[j - 1], a1 = ll[j]; ll[j - 1] = sw ? a1 : a0; ll[j] = sw ? a0 : a1;
      a0 = lr[j - 1]; a1 = lr[j]; lr[j - 1] = sw ? a1 : a0; lr[j] = sw ? a0 : a1;
      a0 = lg[j - 1]; a1 = lg[j]; lg[j - 1] = sw ? a1 : a0; lg[j] = sw ? a0 : a1;
      a0 = lb[j - 1]; a1 = lb[j]; lb[j - 1] = sw ? a1 : a0; lb[j] = sw ? a0 : a1;
    }
  }
  float mean = 0.0f;
#pragma unroll
  for (int i = 1; i < 8; i++) mean += ll[i];
  mean /= 7.0f;
  float noisiness = rt_absf(ll[4] - mean);
  float diff = rt_absf(ll[4] - o_l) - noisiness * (float)NEIGHBOURHOOD_WEIGHT;
  diff = rt_clampf(diff, 0.0f, DENOISING_THRESHOLD) / DENOISING_THRESHOLD;
  uint8_t *q = dst + ((size_t)x + (size_t)y * dst_stride) * dst_comp;
  const int dc = dst_comp < 3 ? dst_comp : 3;
  q[0] = (uint8_t)(rt_lerpf(o_r, lr[4], diff) * 255.999f);
  if (dc > 1) q[1] = (uint8_t)(rt_lerpf(o_g, lg[4], diff) * 255.999f);
  if (dc > 2) q[2] = (uint8_t)(rt_lerpf(o_b, lb[4], diff) * 255.999f);
}

extern "C" int rt_launch_denoise(int width, int height, int src_stride, int src_comp, int dst_stride, int dst_comp,
                                 const uint8_t *src, uint8_t *dst, hipStream_t stream) {
  dim3 block(64, 4), grid((width + 63) / 64, (height + 3) / 4);
  hipLaunchKernelGGL(rt_denoise_kernel, grid, block, 0, stream, width, height, src_stride, src_comp, dst_stride, dst_comp,
                     src, dst);
  return (int)hipGetLastError();
}
