// Bicubic x4 (align_corners=False) device code shared by larva_pointwise.hip (bicubic4_kernel) and
// conv3x3_mfma.hip (step_prologue_kernel).  See larva_pointwise.hip for the derivation.
#pragma once
#include "larva_common.h"

namespace larva {

__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
  const float x2 = 1.0f - t;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  const float x3 = x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

// One thread produces the 4 horizontally adjacent outputs of one LR pixel in one HR row; work item i of
// planes * 4H * W, `first` / `stride` = this thread's grid-stride walk.
__device__ __forceinline__ void bicubic4_body(const float* __restrict__ in, float* __restrict__ out, int planes,
                                              int H, int W, long long first, long long stride) {
  const int HH = 4 * H;
  const long long total = (long long)planes * HH * W;
  for (long long i = first; i < total; i += stride) {
    const int x = (int)(i % W);
    const long long t2 = i / W;
    const int Y = (int)(t2 % HH);
    const int p = (int)(t2 / HH);
    const float* src = in + (size_t)p * H * W;
    const float sy = 0.25f * ((float)Y + 0.5f) - 0.5f;
    const float fy = floorf(sy);
    const int iy = (int)fy;
    float wy[4];
    cubic_coeffs(sy - fy, wy);
    // columns x-2 .. x+2 of the 4 source rows
    float v[4][5];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yy = min(max(iy - 1 + r, 0), H - 1);
#pragma unroll
      for (int c = 0; c < 5; ++c) v[r][c] = src[(size_t)yy * W + min(max(x - 2 + c, 0), W - 1)];
    }
    f32x4 o;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int X = 4 * x + jj;
      const float sx = 0.25f * ((float)X + 0.5f) - 0.5f;
      const float fx = floorf(sx);
      const int ix = (int)fx;  // x-1 for jj < 2, x for jj >= 2
      float wx[4];
      cubic_coeffs(sx - fx, wx);
      const int c0 = ix - 1 - (x - 2);  // 0 or 1
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float rowv = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) rowv += wx[c] * (c0 == 0 ? v[r][c] : v[r][c + 1]);
        acc += wy[r] * rowv;
      }
      o[jj] = acc;
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)p * HH + Y) * (4 * W) + 4 * x) = o;
  }
}

}  // namespace larva
