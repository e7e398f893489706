// HBM-bound companions of the conv kernels (gfx950): bicubic x4 base image, L1 loss forward /
// backward, pixel-unshuffle of the exit gradient, fused AdamW.  All are plain streaming
// kernels: 16-byte accesses per lane, grid capped at 2048 blocks, no LDS.
#include "larva_common.h"
#include "larva_bicubic.h"
#include "larva_loss.h"

namespace larva {

// ---------------------------------------------------------------------------------------------
// Bicubic x4, align_corners=False (models/LarvaNet.py:283-285 -> F.interpolate(mode='bicubic')).
// src = (dst + 0.5) / 4 - 0.5; taps at floor(src)-1..+2 with index clamping; Keys kernel
// A = -0.75; row pass first, then the column weights (the order of ATen's separable CPU path).
// One thread produces the 4 horizontally adjacent outputs of one LR pixel in one HR row.
// ---------------------------------------------------------------------------------------------
__global__ void bicubic4_kernel(const float* __restrict__ in, float* __restrict__ out, int planes,
                                int H, int W) {
  bicubic4_body(in, out, planes, H, W, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                (long long)gridDim.x * blockDim.x);
}

// The same arithmetic, one thread per LR pixel = its 4 x 4 block of HR pixels (the stand-alone launch of the inference
// forward: a 339 x 510 image is 33 MB of base image).  Round 6: SEPARABLE inside the thread and free of run-time tap
// selection.  The four sub-pixel phases of a x4 upscale are constants -- src - floor(src) = .625, .875, .125, .375 for output
// j = 0..3 of a block, taps starting at column x - 2 + (j >> 1) -- and their Keys weights are dyadic rationals that every
// evaluation order produces exactly (A = -3/4, t a multiple of 1/8: no rounding anywhere in cubic_coeffs), so the weights
// fold to literals that equal bicubic4_body's run-time ones bit for bit.  The row pass is done ONCE per source row and
// phase (5 x 4 four-tap sums) and shared by the four output rows (16 four-tap column sums): 144 fused multiply-adds per
// block instead of 320 + 768 selects, in the same order per output as bicubic4_body (row sums first, then the column
// weights: ATen's separable path), i.e. the same bits.  Round 5's form took 16.7 us for that image (2.0 TB/s, vector-ALU
// bound), bicubic4_kernel 23.1.
__global__ __launch_bounds__(256) void bicubic4_block_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              unsigned planes, unsigned H, unsigned W) {
  const unsigned total = planes * H * W;   // (< 2^31: checked by the launcher)
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned x = i % W, t2 = i / W, y = t2 % H, p = t2 / H;
  const float* src = in + (size_t)p * H * W;
  int col[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) col[c] = min(max((int)x - 2 + c, 0), (int)W - 1);
  float v[5][5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const float* row = src + (size_t)min(max((int)y - 2 + r, 0), (int)H - 1) * W;
#pragma unroll
    for (int c = 0; c < 5; ++c) v[r][c] = row[col[c]];
  }
  float wq[4][4];   // [phase j][tap]: compile-time constants after unrolling
  cubic_coeffs(0.625f, wq[0]);
  cubic_coeffs(0.875f, wq[1]);
  cubic_coeffs(0.125f, wq[2]);
  cubic_coeffs(0.375f, wq[3]);
  float h[5][4];    // row pass: source row r, output column phase jj
#pragma unroll
  for (int r = 0; r < 5; ++r)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float rowv = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) rowv += wq[jj][c] * v[r][(jj >> 1) + c];
      h[r][jj] = rowv;
    }
  float* o = out + ((size_t)p * (4 * H) + 4 * y) * (4 * W) + 4 * x;
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) {
    f32x4 q;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += wq[ii][r] * h[(ii >> 1) + r][jj];
      q[jj] = acc;
    }
    *reinterpret_cast<f32x4*>(o + (size_t)ii * (4 * W)) = q;
  }
}

// The other F.interpolate mode the reference's --interpolate flag can reach (models/LarvaNet.py:57,283-285 always
// passes align_corners=False, which F.interpolate refuses for nearest / area with a ValueError; 'linear' and
// 'trilinear' do not take 4-D input): BILINEAR x4.  src = max(0.25 (dst + 0.5) - 0.5, 0), i0 = floor(src),
// i1 = min(i0 + 1, size - 1), w1 = src - i0; out = w0y (w0x v00 + w1x v01) + w1y (w0x v10 + w1x v11) (ATen
// upsample_bilinear2d: rows of x-interpolations).  One thread produces the 4 horizontally adjacent outputs of one
// LR pixel in one HR row, like bicubic4_body.
__global__ void bilinear4_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int H, int W) {
  const int HH = 4 * H;
  const long long total = (long long)planes * HH * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long long t2 = i / W;
    const int Y = (int)(t2 % HH);
    const int p = (int)(t2 / HH);
    const float* src = in + (size_t)p * H * W;
    const float sy = fmaxf(0.25f * ((float)Y + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, y1 = min(y0 + 1, H - 1);
    const float wy1 = sy - (float)y0, wy0 = 1.f - wy1;
    // columns x-1, x, x+1 of the two source rows
    float v[2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int xx = min(max(x - 1 + c, 0), W - 1);
      v[0][c] = src[(size_t)y0 * W + xx];
      v[1][c] = src[(size_t)y1 * W + xx];
    }
    f32x4 o;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float sx = fmaxf(0.25f * ((float)(4 * x + jj) + 0.5f) - 0.5f, 0.f);
      const int x0 = (int)sx;                       // x - 1 for jj < 2 (x itself at the left border), else x
      const float wx1 = sx - (float)x0, wx0 = 1.f - wx1;
      const int c0 = x0 - (x - 1), c1 = min(x0 + 1, W - 1) - (x - 1);
      const float a0 = c0 == 0 ? v[0][0] : v[0][1], a1 = c1 == 0 ? v[0][0] : c1 == 1 ? v[0][1] : v[0][2];
      const float b0 = c0 == 0 ? v[1][0] : v[1][1], b1 = c1 == 0 ? v[1][0] : c1 == 1 ? v[1][1] : v[1][2];
      o[jj] = wy0 * (wx0 * a0 + wx1 * a1) + wy1 * (wx0 * b0 + wx1 * b1);
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)p * HH + Y) * (4 * W) + 4 * x) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// LarvaHead (models/LarvaNet.py:223-233): conv3x3 3 -> COUT, bias, no activation, as a DIRECT
// convolution.  K = 27: on the MFMA kernel the image is zero-padded to 16 channels (two 8-channel
// K chunks, 13/16 of the multiplies are zeros) and the launch costs what a quarter of a 48 -> 48
// layer costs; here the layer is what it is -- 0.44 MB in, 7.08 MB out, bandwidth-bound.  One thread
// = one output pixel x 16 output channels: its 27 inputs in registers (neighbouring lanes share them
// through L1), the 16 x 27 weights wave-uniform (scalar loads), 16 coalesced dword stores per thread.
// `x` is the unpadded [N][3][H][W] image; `out` rows are `pitch` >= W floats apart, columns [W, pitch)
// written as zeros (the layout the next layer's 16-byte staging path expects).
// ---------------------------------------------------------------------------------------------
constexpr int kHeadCoutsPerThread = 16;

__global__ __launch_bounds__(256) void head_conv3_direct_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ out, int N, int cout, int H,
                                                                int W, int pitch) {
  // this block's 16 x 27 weights (+ bias) -> LDS, rows padded to 28 floats: every lane then reads the
  // same address (a broadcast, no bank conflict).  Scalar loads were tried first: 16 dependent
  // s_load round trips per thread, 12.8 us for the launch.
  __shared__ __attribute__((aligned(16))) float wl[kHeadCoutsPerThread][28];
  const int co0 = blockIdx.y * kHeadCoutsPerThread;
  for (int i = threadIdx.x; i < kHeadCoutsPerThread * 28; i += 256) {
    const int j = i / 28, k = i - j * 28;
    wl[j][k] = k < 27 ? w[(co0 + j) * 27 + k] : (bias ? bias[co0 + j] : 0.f);   // slot 27 = the bias
  }
  __syncthreads();
  const int total = N * H * pitch;                      // (< 2^31: checked by the launcher)
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int xx = pix % pitch;
  const int t = pix / pitch;
  const int y = t % H;
  const int n = t / H;
  // the 3 x 3 window: clamped row / column offsets + validity, shared by the three channels
  int roff[3], coff[3];
  bool rok[3], cok[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int yy = y + k - 1, xc = xx + k - 1;
    rok[k] = yy >= 0 && yy < H;
    cok[k] = xc >= 0 && xc < W;
    roff[k] = min(max(yy, 0), H - 1) * W;
    coff[k] = min(max(xc, 0), W - 1);
  }
  const float* xb = x + (size_t)n * 3 * H * W;
  const int cplane = H * W;
  float v[27];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float val = xb[c * cplane + roff[ky] + coff[kx]];
        v[(c * 3 + ky) * 3 + kx] = (rok[ky] && cok[kx]) ? val : 0.f;
      }
  const size_t plane = (size_t)H * pitch;
  float* o = out + ((size_t)n * cout + co0) * plane + (size_t)y * pitch + xx;
  const bool real = xx < W;
#pragma unroll
  for (int j = 0; j < kHeadCoutsPerThread; ++j) {
    float wj[28];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(&wl[j][4 * q]);
      wj[4 * q] = w4[0]; wj[4 * q + 1] = w4[1]; wj[4 * q + 2] = w4[2]; wj[4 * q + 3] = w4[3];
    }
    float acc = wj[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc = fmaf(v[k], wj[k], acc);
    o[j * plane] = real ? acc : 0.f;
  }
}

// The same layer for LARGE images (the head of a 339 x 510 validation image writes 33 MB; on the padded MFMA launch it took
// 31.6 us = 1.05 TB/s): one thread = 4 consecutive pixels of a row x 8 output channels -- 54 inputs in registers, every
// output channel's 4 pixels leave as ONE 16-byte store (a wave writes 1 KiB runs instead of 256-byte ones).  Same fma chain
// per output as head_conv3_direct_kernel.  pitch % 4 == 0.
constexpr int kHead4Couts = 8;

// Round 6 (profiles/r06_head_bicubic_ab.txt: 19.8 -> 15.2 us for that image):
//  * the block's weights are NOT staged in LDS: `w + (co0 + j) * 27` is a block-uniform address, the compiler fetches the
//    27 taps + bias of an output channel with scalar loads into SGPRs and the FMAs take them as scalar operands -- no
//    ds_read_b128 broadcast per lane (56 per thread before), no 28 VGPRs of weights, no barrier at the head of the block;
//  * threads whose 3 x 6 input window lies inside the image (all but the border ring) load it without clamps or selects,
//    a window row as one 16-byte + one 8-byte load (global loads need dword alignment only).
__global__ __launch_bounds__(256) void head_conv3_direct4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int N, int cout, int H, int W, int pitch) {
  const int co0 = blockIdx.y * kHead4Couts;
  const unsigned q4 = (unsigned)pitch >> 2;
  const unsigned total = (unsigned)N * H * q4;          // groups of 4 pixels (< 2^31: checked by the launcher)
  const unsigned gidx = blockIdx.x * 256u + threadIdx.x;
  if (gidx >= total) return;
  const int xq = (int)(gidx % q4), t = (int)(gidx / q4), y = t % H, n = t / H;
  const int x0 = 4 * xq;
  const float* xb = x + (size_t)n * 3 * H * W;
  const int cplane = H * W;
  float v[3][3][6];   // [channel][row y-1..y+1][column x0-1..x0+4], zero outside the image
  if (y >= 1 && y + 1 < H && x0 >= 1 && x0 + 4 < W) {
    // (global loads need dword alignment only: the six floats of a window row travel as one 16-byte + one 8-byte load)
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
    const float* p0 = xb + (size_t)(y - 1) * W + (x0 - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4_u a = *reinterpret_cast<const f32x4_u*>(p0 + c * cplane + ky * W);
        const f32x2_u b = *reinterpret_cast<const f32x2_u*>(p0 + c * cplane + ky * W + 4);
        v[c][ky][0] = a[0]; v[c][ky][1] = a[1]; v[c][ky][2] = a[2]; v[c][ky][3] = a[3]; v[c][ky][4] = b[0]; v[c][ky][5] = b[1];
      }
  } else {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      const bool rok = yy >= 0 && yy < H;
      const int roff = min(max(yy, 0), H - 1) * W;
#pragma unroll
      for (int kx = 0; kx < 6; ++kx) {
        const int xc = x0 + kx - 1;
        const bool ok = rok && xc >= 0 && xc < W;
        const int off = roff + min(max(xc, 0), W - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float val = xb[c * cplane + off];
          v[c][ky][kx] = ok ? val : 0.f;
        }
      }
    }
  }
  const size_t plane = (size_t)H * pitch;
  float* o = out + ((size_t)n * cout + co0) * plane + (size_t)y * pitch + x0;
#pragma unroll
  for (int j = 0; j < kHead4Couts; ++j) {
    const float* wj = w + (co0 + j) * 27;            // block-uniform: scalar loads
    const float bj = bias ? bias[co0 + j] : 0.f;
    f32x4 r;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float acc = bj;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc = fmaf(v[c][ky][px + kx], wj[(c * 3 + ky) * 3 + kx], acc);
      r[px] = (x0 + px < W) ? acc : 0.f;
    }
    *reinterpret_cast<f32x4*>(o + (size_t)j * plane) = r;
  }
}

// ---------------------------------------------------------------------------------------------
// L1 loss (nn.L1Loss(), models/LarvaNet.py:85,108): sum |a - b| -> per-block partials -> a
// one-block launch adds them in index order (reproducible) and writes sum / numel.
// ---------------------------------------------------------------------------------------------
constexpr int kL1Blocks = 1024;

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ b,
                                                         long long numel, float* __restrict__ partial) {
  float s = 0.f;
  const long long n4 = numel >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 va = reinterpret_cast<const f32x4*>(a)[i];
    const f32x4 vb = reinterpret_cast<const f32x4*>(b)[i];
    s += fabsf(va[0] - vb[0]) + fabsf(va[1] - vb[1]) + fabsf(va[2] - vb[2]) + fabsf(va[3] - vb[3]);
  }
  if (blockIdx.x == 0) {
    for (long long i = (n4 << 2) + threadIdx.x; i < numel; i += 256) s += fabsf(a[i] - b[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// Second launch on purpose: folding it into the first with a last-arriver ticket was measured
// 2-3x slower (1024 serialized arrivals on one counter, ~12 ns each, vs a 3 us kernel).
__global__ __launch_bounds__(256) void l1_finish_kernel(const float* __restrict__ partial, int n,
                                                        float inv_numel, float* __restrict__ loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = ((ws[0] + ws[1]) + (ws[2] + ws[3])) * inv_numel;
}

// d/da mean|a-b| * gout = sign(a-b) * gout / numel, sign(0) = 0 (ATen's l1 backward).
__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ gout, float inv_numel, long long numel,
                              float* __restrict__ ga) {
  const float g = gout[0] * inv_numel;
  const long long n4 = numel >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const f32x4 va = reinterpret_cast<const f32x4*>(a)[i];
    const f32x4 vb = reinterpret_cast<const f32x4*>(b)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = va[e] - vb[e];
      o[e] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
    reinterpret_cast<f32x4*>(ga)[i] = o;
  }
  if (blockIdx.x == 0) {
    for (long long i = (n4 << 2) + threadIdx.x; i < numel; i += blockDim.x) {
      const float d = a[i] - b[i];
      ga[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
  }
}

// L1 backward fused with the PixelShuffle(4) backward of the exit it feeds: reads the exit
// output and the truth in HR layout, writes sign(out - truth) * gout / numel straight into the
// [N][16C][H][W] layout the leg's dgrad / wgrad consume (no HR-layout gradient tensor).
__global__ void l1_bwd_unshuffle4_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                         const float* __restrict__ gout, float gscale, float inv_numel,
                                         float* __restrict__ out, int planes, int H, int W) {
  const float g = (gout[0] * gscale) * inv_numel;
  const int HH = 4 * H;
  const long long total = (long long)planes * HH * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    const long long t2 = idx / W;
    const int Y = (int)(t2 % HH);
    const int p = (int)(t2 / HH);
    const int y = Y >> 2, i = Y & 3;
    const size_t src = ((size_t)p * HH + Y) * (4 * W) + 4 * x;
    const f32x4 va = *reinterpret_cast<const f32x4*>(a + src);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(b + src);
    float* o = out + (((size_t)p * 16 + 4 * i) * H + y) * W + x;
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = va[e] - vb[e];
      o[e * plane] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
  }
}

// L1 forward and backward of one exit in ONE pass over (a, b): block partial sums of |a - b| (as
// l1_partial_kernel, same block/thread decomposition of the flat index: identical sums) AND the
// gradient sign(a - b) * g in pixel-unshuffled layout, for a gradient value g known on the host
// (training: d loss / d loss = 1 times the 1/M of the mean over exits, times 1 / numel).
__global__ __launch_bounds__(256) void l1_partial_grad_kernel(const float* __restrict__ a,
                                                              const float* __restrict__ b, long long numel,
                                                              float g, float* __restrict__ partial,
                                                              float* __restrict__ grad, int H, int W) {
  float s = 0.f;
  const long long n4 = numel >> 2;  // one f32x4 = the 4 sub-pixel columns (j) of one LR pixel
  const int HH = 4 * H;
  const size_t plane = (size_t)H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 va = reinterpret_cast<const f32x4*>(a)[i];
    const f32x4 vb = reinterpret_cast<const f32x4*>(b)[i];
    s += fabsf(va[0] - vb[0]) + fabsf(va[1] - vb[1]) + fabsf(va[2] - vb[2]) + fabsf(va[3] - vb[3]);
    const int x = (int)(i % W);
    const long long t2 = i / W;
    const int Y = (int)(t2 % HH);
    const long long p = t2 / HH;  // image * HR channels + HR channel
    float* o = grad + ((size_t)(p * 16 + 4 * (Y & 3)) * H + (Y >> 2)) * W + x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = va[e] - vb[e];
      o[e * plane] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// The same for up to 8 exits scored against ONE truth image in one launch (blockIdx.y = exit).
struct L1Jobs {
  const float* a[8];
  float* partial[8];
  float* grad[8];
};
__global__ __launch_bounds__(256) void l1_partial_grad_batch_kernel(L1Jobs jobs, const float* __restrict__ b,
                                                                    long long numel, float g, int H, int W) {
  const float* __restrict__ a = jobs.a[blockIdx.y];
  float* __restrict__ partial = jobs.partial[blockIdx.y];
  float* __restrict__ grad = jobs.grad[blockIdx.y];
  float s = 0.f;
  const long long n4 = numel >> 2;
  const int HH = 4 * H;
  const size_t plane = (size_t)H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 va = reinterpret_cast<const f32x4*>(a)[i];
    const f32x4 vb = reinterpret_cast<const f32x4*>(b)[i];
    s += fabsf(va[0] - vb[0]) + fabsf(va[1] - vb[1]) + fabsf(va[2] - vb[2]) + fabsf(va[3] - vb[3]);
    const int x = (int)(i % W);
    const long long t2 = i / W;
    const int Y = (int)(t2 % HH);
    const long long p = t2 / HH;
    float* o = grad + ((size_t)(p * 16 + 4 * (Y & 3)) * H + (Y >> 2)) * W + x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = va[e] - vb[e];
      o[e * plane] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// out[0] = ( sum_i  scale_i * (sum of the count_i floats at p_i) ) / divisor: the mean over the
// exits of their L1 terms straight from the block partial sums of l1_partial_kernel (count_i =
// its block count, scale_i = 1 / numel), or of ready scalars (count 1, scale 1).  Each term is
// reduced exactly like l1_finish_kernel does, the terms are added in index order: bit-identical
// to l1_finish + sum_scalars, in one launch instead of n + 1.
__global__ __launch_bounds__(256) void loss_from_partials_kernel(TermList l, float divisor, float* __restrict__ out,
                                                                 float* __restrict__ host_cell, unsigned* __restrict__ dev_seq) {
  loss_terms_block(l, divisor, out, host_cell, dev_seq);
}

// out[0] = (t0 + t1 + ... ) / divisor over up to 8 device scalars, added in index order.
struct ScalarList {
  const float* p[8];
  int n;
};
__global__ void sum_scalars_kernel(ScalarList l, float divisor, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < l.n; ++i) s += l.p[i][0];
    out[0] = s / divisor;
  }
}

// ---------------------------------------------------------------------------------------------
// Inverse of PixelShuffle(4) (the backward of models/LarvaNet.py:261):
//   out[n][16c + 4i + j][y][x] = in[n][c][4y+i][4x+j]
// ---------------------------------------------------------------------------------------------
__global__ void pixel_unshuffle4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                        int planes /* N*C_hr */, int H, int W) {
  const int HH = 4 * H;
  const long long total = (long long)planes * HH * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    const long long t2 = idx / W;
    const int Y = (int)(t2 % HH);
    const int p = (int)(t2 / HH);
    const int y = Y >> 2, i = Y & 3;
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((size_t)p * HH + Y) * (4 * W) + 4 * x);
    float* o = out + (((size_t)p * 16 + 4 * i) * H + y) * W + x;
    const size_t plane = (size_t)H * W;
    o[0] = v[0];
    o[plane] = v[1];
    o[2 * plane] = v[2];
    o[3 * plane] = v[3];
  }
}

// ---------------------------------------------------------------------------------------------
// AdamW over one flat parameter buffer (torch.optim.AdamW semantics, models/LarvaNet.py:86-88:
// decoupled weight decay, bias-corrected moments, eps added after the sqrt).  `step_lr` holds
// {step count as float, lr} on the device so that a captured graph can be replayed.
// ---------------------------------------------------------------------------------------------
struct AdamwCoef {   // per-step scalars of torch.optim.AdamW: step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t), 1 - lr * wd
  float step_size, bc2_sqrt, decay;
};

__device__ __forceinline__ AdamwCoef adamw_coef_device(float step, float lr, double beta1, double beta2, float wd) {
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  return AdamwCoef{(float)((double)lr / bc1), (float)sqrt(bc2), (float)(1.0 - (double)lr * (double)wd)};
}

// A value the compiler must materialise: keeps a product from being contracted into the sum that consumes it
// (hipcc's default is -ffp-contract=fast, and `#pragma clang fp contract(off)` did not survive the inlining here).
__device__ __forceinline__ float rounded(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// torch.optim.AdamW's single-tensor step, operation by operation, every product and sum rounded where ATen's CPU
// kernels (AVX2 build) round them, so the 16-byte walk and the element-wise walk also give the same bits.  Against
// torch 2.10 on the CPU the moments come out bit for bit and the parameters on all but ~0.1 % of the elements
// (1 ulp; emulated in numpy in round 3 and tested on the device):
//   param.mul_(1 - lr * wd); exp_avg.lerp_(grad, 1 - beta1)            [lerp: fma(weight, end - start, start)]
//   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)        [fma(value * t1, t2, self)]
//   denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)       [a division, then the add]
//   param.addcdiv_(exp_avg, denom, value=-step_size)                    [self + (value * t1) / t2]
struct AdamwBetas {   // beta2, and 1 - beta1 / 1 - beta2 computed in DOUBLE like torch's Python code, then rounded: 0.1f and
  float beta2, omb1, omb2;   // 0.001f -- in float arithmetic 1.f - 0.9f is 0.100000024 and 1.f - 0.999f is 0.00100004673
};

__device__ __forceinline__ void adamw_update(float& param, float& mi, float& vi, float grad, const AdamwCoef& c,
                                             const AdamwBetas& b, float eps) {
  mi = __fmaf_rn(b.omb1, rounded(grad - mi), mi);
  vi = __fmaf_rn(rounded(b.omb2 * grad), grad, rounded(vi * b.beta2));   // (ATen's AVX2 addcmul: the last product is fused)
  const float denom = rounded(__fdiv_rn(__fsqrt_rn(vi), c.bc2_sqrt)) + eps;
  param = rounded(param * c.decay) + rounded(__fdiv_rn(rounded(-c.step_size * mi), denom));
}

// step_lr != null: {step count as float, lr} live on the device (a captured graph can be replayed) and the
// bias corrections are computed per thread; else `coef` comes ready from the host (computed in double like
// torch does) and the flat buffers are walked 16 bytes per lane.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ step_lr, AdamwCoef coef, double beta1,
                                                    double beta2, float eps, float wd, float gscale, long long n, int vec,
                                                    const float* __restrict__ copy_src, float* __restrict__ copy_dst) {
  if (copy_dst && blockIdx.x == 0 && threadIdx.x == 0) copy_dst[0] = copy_src[0];   // (the step's loss: see the launcher)
  if (step_lr) coef = adamw_coef_device(step_lr[0], step_lr[1], beta1, beta2, wd);
  const AdamwBetas bt{(float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2)};
  const long long n4 = vec ? n >> 2 : 0;   // (vec: all four buffers are 16-byte aligned)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      adamw_update(pe, me, ve, __fmul_rn(gg[e], gscale), coef, bt, eps);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float pp = p[i], mm = m[i], vv = v[i];
    adamw_update(pp, mm, vv, __fmul_rn(g[i], gscale), coef, bt, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

static int adamw_vec_ok(const void* p, const void* g, const void* m, const void* v) {
  return ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
           reinterpret_cast<uintptr_t>(v)) & 15) == 0;
}

// ---------------------------------------------------------------------------------------------
// Device-resident patch sampler: crop + rot90 + horizontal flip + uint8 -> float of one training
// batch, straight out of a dataset kept in HBM (replaces the host loop of
// dataloaders/div2k_train_loader.py:72-98 and the H2D copy of train_larva.py:123-124).
// draws[b] = {image, x, y, k, flip} (k = number of counter-clockwise quarter turns, np.rot90);
// the crop starts at (y*mult, x*mult) and is P x P pixels; images are uint8 CHW at
// data + offsets[image] with (H, W) = hw[2*image .. +1].
//   rot90^k then flip:  out[c][i][j] = crop[c][si][sj],  j' = flip ? P-1-j : j,
//     k%4==0: (si,sj) = (i, j')   1: (j', P-1-i)   2: (P-1-i, P-1-j')   3: (P-1-j', i)
// ---------------------------------------------------------------------------------------------
__global__ void gather_patches_kernel(const unsigned char* __restrict__ data,
                                      const long long* __restrict__ offsets, const int* __restrict__ hw,
                                      const int* __restrict__ draws, float* __restrict__ out, int B, int P,
                                      int mult) {
  const long long total = (long long)B * 3 * P * P;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % P);
    long long t = idx / P;
    const int i = (int)(t % P); t /= P;
    const int c = (int)(t % 3);
    const int b = (int)(t / 3);
    const int* d = draws + 5 * b;
    const int img = d[0], x0 = d[1] * mult, y0 = d[2] * mult, k = d[3] & 3, flip = d[4];
    const int H = hw[2 * img], W = hw[2 * img + 1];
    const int jj = flip ? P - 1 - j : j;
    int si, sj;
    if (k == 0) { si = i; sj = jj; }
    else if (k == 1) { si = jj; sj = P - 1 - i; }
    else if (k == 2) { si = P - 1 - i; sj = P - 1 - jj; }
    else { si = P - 1 - jj; sj = i; }
    out[idx] = (float)data[offsets[img] + ((long long)c * H + (y0 + si)) * W + (x0 + sj)];
  }
}

// ---------------------------------------------------------------------------------------------
// Validation metric on the device (validate.py:17-27): the output image is converted with
// clip(round-half-to-even(x), 0, 255) and compared with the uint8 truth cropped top-left to the
// output size; sum of squared differences as an exact 64-bit integer (order-independent).
// out: [C][H][W] float, truth: [C][TH][TW] uint8 with TH >= H, TW >= W.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqerr_u8_kernel(const float* __restrict__ out,
                                                       const unsigned char* __restrict__ truth, int C, int H,
                                                       int W, int TH, int TW,
                                                       unsigned long long* __restrict__ acc) {
  const long long total = (long long)C * H * W;
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long long t = i / W;
    const int y = (int)(t % H);
    const int c = (int)(t / H);
    const float q = fminf(fmaxf(rintf(out[i]), 0.f), 255.f);
    const int d = (int)truth[((size_t)c * TH + y) * TW + x] - (int)q;
    s += (unsigned long long)(d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(acc, s);
}

static inline int grid_for(long long work, int block) {
  long long g = (work + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}


}  // namespace larva

using namespace larva;

extern "C" {

int larva_bicubic4_fwd(const float* in, float* out, int N, int C, int H, int W, void* stream) {
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
  const long long px = (long long)N * C * H * W;
  if (px < (1ll << 31) - 256 && !(reinterpret_cast<uintptr_t>(out) & 15)) {   // one thread per LR pixel = a 4 x 4 block of HR pixels
    hipLaunchKernelGGL(bicubic4_block_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                       (unsigned)(N * C), (unsigned)H, (unsigned)W);
    return (int)hipGetLastError();
  }
  const long long work = (long long)N * C * 4 * H * W;
  hipLaunchKernelGGL(bicubic4_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     N * C, H, W);
  return (int)hipGetLastError();
}

// F.interpolate(x, scale_factor=4, mode, align_corners=False) for mode 0 = bicubic (larva_bicubic4_fwd) and
// 1 = bilinear -- the two modes for which the reference's call (models/LarvaNet.py:283-285) does not raise.
// out must be 16-byte aligned.
int larva_upsample4_fwd(const float* in, float* out, int N, int C, int H, int W, int mode, void* stream) {
  if (mode == 0) return larva_bicubic4_fwd(in, out, N, C, H, W, stream);
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || mode != 1 || (reinterpret_cast<uintptr_t>(out) & 15))
    return (int)hipErrorInvalidValue;
  const long long work = (long long)N * C * 4 * H * W;
  hipLaunchKernelGGL(bilinear4_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, in, out, N * C, H, W);
  return (int)hipGetLastError();
}

// Direct 3 -> cout head convolution (cout % 16 == 0): x [N][3][H][W], w [cout][3][3][3] (PyTorch layout,
// no packing), bias [cout] or NULL, out [N][cout][H][pitch] with columns [W, pitch) zeroed.
int larva_head_conv3_direct(const float* x, const float* w, const float* bias, float* out, int N, int cout,
                            int H, int W, int pitch, void* stream) {
  if (pitch == 0) pitch = W;
  if (!x || !w || !out || N <= 0 || cout <= 0 || cout % kHeadCoutsPerThread || H <= 0 || W <= 0 || pitch < W)
    return (int)hipErrorInvalidValue;
  const long long total = (long long)N * H * pitch;
  if (total >= (1ll << 31) - 256 || (long long)N * 3 * H * W >= (1ll << 31)) return (int)hipErrorInvalidValue;
  if (pitch % 4 == 0 && cout % kHead4Couts == 0 && !(reinterpret_cast<uintptr_t>(out) & 15)) {
    // 4 pixels x 8 output channels per thread, 16-byte stores
    hipLaunchKernelGGL(head_conv3_direct4_kernel, dim3((unsigned)((total / 4 + 255) / 256), cout / kHead4Couts), dim3(256), 0,
                       (hipStream_t)stream, x, w, bias, out, N, cout, H, W, pitch);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(head_conv3_direct_kernel, dim3((unsigned)((total + 255) / 256), cout / kHeadCoutsPerThread), dim3(256),
                     0, (hipStream_t)stream, x, w, bias, out, N, cout, H, W, pitch);
  return (int)hipGetLastError();
}

int larva_l1_workspace_floats(void) { return kL1Blocks; }

// loss[0] = mean |a - b|; `workspace` holds larva_l1_workspace_floats() floats (block partials,
// added in index order by a second tiny launch: reproducible).
int larva_l1_fwd(const float* a, const float* b, long long numel, float* workspace, float* loss,
                 void* stream) {
  if (!a || !b || !workspace || !loss || numel <= 0) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return (int)hipErrorInvalidValue;
  int blocks = grid_for(numel / 4, 256);
  if (blocks > kL1Blocks) blocks = kL1Blocks;
  hipLaunchKernelGGL(l1_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, numel, workspace);
  hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, blocks,
                     1.0f / (float)numel, loss);
  return (int)hipGetLastError();
}

// Block partial sums of sum|a - b| only (no finishing launch): `partial` receives *blocks <=
// larva_l1_workspace_floats() floats, to be consumed by larva_loss_from_partials.
int larva_l1_partial(const float* a, const float* b, long long numel, float* partial, int* blocks_out,
                     void* stream) {
  if (!a || !b || !partial || !blocks_out || numel <= 0) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return (int)hipErrorInvalidValue;
  int blocks = grid_for(numel / 4, 256);
  if (blocks > kL1Blocks) blocks = kL1Blocks;
  *blocks_out = blocks;
  hipLaunchKernelGGL(l1_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, numel, partial);
  return (int)hipGetLastError();
}

// larva_l1_partial plus, in the same pass, the gradient the exit's backward will need:
// grad [N][16C][H][W] = sign(a - b) * (gvalue * gscale / numel) for a, b [N][C][4H][4W] -- valid when
// the gradient arriving at this L1 term is the host-known constant gvalue (the seed of
// loss.backward(), 1).  Same partial sums as larva_l1_partial, bit for bit.
int larva_l1_partial_grad(const float* a, const float* b, float gvalue, float gscale, float* partial,
                          int* blocks_out, float* grad, int N, int C, int H, int W, void* stream) {
  if (!a || !b || !partial || !blocks_out || !grad || N <= 0 || C <= 0 || H <= 0 || W <= 0)
    return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return (int)hipErrorInvalidValue;
  const long long numel = (long long)N * C * 16 * H * W;
  int blocks = grid_for(numel / 4, 256);
  if (blocks > kL1Blocks) blocks = kL1Blocks;
  *blocks_out = blocks;
  const float g = (gvalue * gscale) * (1.0f / (float)numel);  // the arithmetic of l1_bwd_unshuffle4_kernel
  hipLaunchKernelGGL(l1_partial_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, numel, g, partial,
                     grad, H, W);
  return (int)hipGetLastError();
}

// larva_l1_partial_grad for n <= 8 images a[i] against the same b, one launch.
int larva_l1_partial_grad_batch(const float* const* a, const float* b, int n, float gvalue, float gscale,
                                float* const* partial, int* blocks_out, float* const* grad, int N, int C, int H,
                                int W, void* stream) {
  if (!a || !b || !partial || !blocks_out || !grad || n < 1 || n > 8 || N <= 0 || C <= 0 || H <= 0 || W <= 0)
    return (int)hipErrorInvalidValue;
  L1Jobs jobs{};
  uintptr_t bits = reinterpret_cast<uintptr_t>(b);
  for (int i = 0; i < n; ++i) {
    if (!a[i] || !partial[i] || !grad[i]) return (int)hipErrorInvalidValue;
    jobs.a[i] = a[i]; jobs.partial[i] = partial[i]; jobs.grad[i] = grad[i];
    bits |= reinterpret_cast<uintptr_t>(a[i]);
  }
  if (bits & 15) return (int)hipErrorInvalidValue;
  const long long numel = (long long)N * C * 16 * H * W;
  int blocks = grid_for(numel / 4, 256);
  if (blocks > kL1Blocks) blocks = kL1Blocks;
  *blocks_out = blocks;
  const float g = (gvalue * gscale) * (1.0f / (float)numel);
  hipLaunchKernelGGL(l1_partial_grad_batch_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, jobs, b, numel,
                     g, H, W);
  return (int)hipGetLastError();
}

// out[0] = ( sum_i scale[i] * (sum of count[i] floats at terms[i]) ) / divisor, n <= 8 terms; host_cell (may be
// NULL, else from larva_host_cell_alloc) receives the same float and the next sequence number with one system-scope
// 8-byte release store.
int larva_loss_from_partials_to_host_seq(const float* const* terms, const int* count, const float* scale, int n,
                                         float divisor, float* out, float* host_cell, unsigned* dev_seq, void* stream) {
  if (!terms || !count || !scale || n < 1 || n > 8 || !out || (dev_seq && !host_cell)) return (int)hipErrorInvalidValue;
  TermList l{};
  for (int i = 0; i < n; ++i) {
    if (!terms[i] || count[i] < 1) return (int)hipErrorInvalidValue;
    l.p[i] = terms[i];
    l.count[i] = count[i];
    l.scale[i] = scale[i];
  }
  l.n = n;
  hipLaunchKernelGGL(loss_from_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, l, divisor, out, host_cell, dev_seq);
  return (int)hipGetLastError();
}

int larva_loss_from_partials_to_host(const float* const* terms, const int* count, const float* scale, int n,
                                     float divisor, float* out, float* host_cell, void* stream) {
  return larva_loss_from_partials_to_host_seq(terms, count, scale, n, divisor, out, host_cell, nullptr, stream);
}

int larva_loss_from_partials(const float* const* terms, const int* count, const float* scale, int n,
                             float divisor, float* out, void* stream) {
  return larva_loss_from_partials_to_host(terms, count, scale, n, divisor, out, nullptr, stream);
}

// One {float value, uint32 sequence} pair (in a 64-byte line of its own) of coherent, device-mapped pinned host
// memory: a kernel stores into it with system scope (one 8-byte store: the value and the sequence number of the
// store, which starts at 0 and grows by one per launch), the host reads it without synchronising with any stream
// (`return loss.item()`, models/LarvaNet.py:139, while the rest of the step still runs).  On return the value is NaN
// and the sequence number 0.
int larva_host_cell_alloc(float** cell) {
  if (!cell) return (int)hipErrorInvalidValue;
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, 64, hipHostMallocCoherent | hipHostMallocMapped);
  if (e != hipSuccess) return (int)e;
  *cell = static_cast<float*>(p);
  (*cell)[0] = __builtin_nanf("");
  reinterpret_cast<unsigned*>(p)[1] = 0u;
  return 0;
}

// Waits for the device first: a launch (or a captured graph's replay) that stores into the cell may be in flight.
int larva_host_cell_free(float* cell) {
  if (!cell) return 0;
  (void)hipDeviceSynchronize();
  return (int)hipHostFree(cell);
}

// L1 backward written in the pixel-unshuffled layout: a, b [N][C][4H][4W] -> ga [N][16C][H][W],
// ga = sign(a - b) * (gout[0] * gscale) / numel (gscale: e.g. the 1/M of the mean over exits).
int larva_l1_bwd_unshuffle4(const float* a, const float* b, const float* gout, float gscale, float* ga, int N,
                            int C, int H, int W, void* stream) {
  if (!a || !b || !gout || !ga || N <= 0 || C <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return (int)hipErrorInvalidValue;
  const long long work = (long long)N * C * 4 * H * W;
  const float inv = 1.0f / (float)(work * 4);
  hipLaunchKernelGGL(l1_bwd_unshuffle4_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                     gout, gscale, inv, ga, N * C, H, W);
  return (int)hipGetLastError();
}


// out[0] = (sum of n <= 8 device scalars, added in index order) / divisor.
int larva_sum_scalars(const float* const* terms, int n, float divisor, float* out, void* stream) {
  if (!terms || n < 1 || n > 8 || !out) return (int)hipErrorInvalidValue;
  ScalarList l{};
  for (int i = 0; i < n; ++i) {
    if (!terms[i]) return (int)hipErrorInvalidValue;
    l.p[i] = terms[i];
  }
  l.n = n;
  hipLaunchKernelGGL(sum_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, l, divisor, out);
  return (int)hipGetLastError();
}

// ga = sign(a - b) * gout[0] / numel   (gout is a device scalar: the upstream gradient).
int larva_l1_bwd(const float* a, const float* b, const float* gout, long long numel, float* ga,
                 void* stream) {
  if (!a || !b || !gout || !ga || numel <= 0) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(ga)) & 15)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(numel / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                     gout, 1.0f / (float)numel, numel, ga);
  return (int)hipGetLastError();
}

// in [N][C][4H][4W] -> out [N][16C][H][W]
int larva_pixel_unshuffle4(const float* in, float* out, int N, int C, int H, int W, void* stream) {
  if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
  const long long work = (long long)N * C * 4 * H * W;
  hipLaunchKernelGGL(pixel_unshuffle4_kernel, dim3(grid_for(work, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, N * C, H, W);
  return (int)hipGetLastError();
}

int larva_adamw_step(float* p, const float* g, float* m, float* v, const float* step_lr, double beta1,
                     double beta2, double eps, double weight_decay, float grad_scale, long long n,
                     void* stream) {
  if (!p || !g || !m || !v || !step_lr || n <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     step_lr, AdamwCoef{}, beta1, beta2, (float)eps, (float)weight_decay, grad_scale, n, adamw_vec_ok(p, g, m, v),
                     (const float*)nullptr,
                     (float*)nullptr);
  return (int)hipGetLastError();
}

int larva_adamw_step_host_copy(float* p, const float* g, float* m, float* v, int step, double lr, double beta1,
                               double beta2, double eps, double weight_decay, float grad_scale, long long n,
                               const float* copy_src, float* copy_dst, void* stream);

// Same update with the step count (1-based) and learning rate passed from the host.
int larva_adamw_step_host(float* p, const float* g, float* m, float* v, int step, double lr, double beta1,
                          double beta2, double eps, double weight_decay, float grad_scale, long long n,
                          void* stream) {
  return larva_adamw_step_host_copy(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, grad_scale, n, nullptr, nullptr,
                                    stream);
}

// larva_adamw_step_host that also copies ONE float (copy_dst[0] = copy_src[0]) -- the step's loss out of the
// captured graph's static buffer into a tensor of the caller's (models/LarvaNet.py:139 returns the loss):
// saves the 4-byte device-to-device copy launch.
int larva_adamw_step_host_copy(float* p, const float* g, float* m, float* v, int step, double lr, double beta1,
                               double beta2, double eps, double weight_decay, float grad_scale, long long n,
                               const float* copy_src, float* copy_dst, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1 || (copy_dst && !copy_src)) return (int)hipErrorInvalidValue;
  // the per-step scalars in double, like torch.optim.AdamW computes them on the host (its hyper-parameters are Python
  // floats = doubles: they arrive here as doubles)
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const AdamwCoef coef{(float)(lr / bc1), (float)sqrt(bc2), (float)(1.0 - lr * weight_decay)};
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (const float*)nullptr, coef, beta1, beta2, (float)eps, (float)weight_decay, grad_scale, n, adamw_vec_ok(p, g, m, v),
                     copy_src, copy_dst);
  return (int)hipGetLastError();
}

// out[b][c][i][j] (float) = augmented P x P crop of image draws[b][0]; see gather_patches_kernel.
int larva_gather_patches(const unsigned char* data, const long long* offsets, const int* hw,
                         const int* draws, float* out, int B, int P, int mult, void* stream) {
  if (!data || !offsets || !hw || !draws || !out || B <= 0 || P <= 0 || mult <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_patches_kernel, dim3(grid_for((long long)B * 3 * P * P, 256)), dim3(256), 0,
                     (hipStream_t)stream, data, offsets, hw, draws, out, B, P, mult);
  return (int)hipGetLastError();
}

// acc[0] += sum over the C x H x W output of (truth_u8 - clip(rint(out), 0, 255))^2; the caller
// zeroes acc.  PSNR = 10 log10(255^2 * C*H*W / acc).
int larva_sqerr_u8(const float* out, const unsigned char* truth, int C, int H, int W, int TH, int TW,
                   unsigned long long* acc, void* stream) {
  if (!out || !truth || !acc || C <= 0 || H <= 0 || W <= 0 || TH < H || TW < W) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sqerr_u8_kernel, dim3(grid_for((long long)C * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     out, truth, C, H, W, TH, TW, acc);
  return (int)hipGetLastError();
}

const char* larva_error_string(int code) { return hipGetErrorString((hipError_t)code); }

int larva_abi_version(void) { return 5; }   // 5: product entry points only (measurement entry points live in the -DLARVA_DIAG_API build, tools/larva_diag.h); larva_conv3x3_fwd_tiled; larva_conv3x3_fwd_strips takes the host table; the *_mb sign-bit entry points are gone.  4: ReLU sign-bit operands.  3: unpadded, swizzled packed weight rows at 32 / 64 channels; AdamW hyper-parameters as doubles

}  // extern "C"
