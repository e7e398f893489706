/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * Plain-C restatement of the arithmetic on the LarvaNet hot path.  The reference
 * (Geunwoo-Jeon/LarvaNet) performs these steps through PyTorch operators (third-party
 * dependency, not vendored in the reference tree; README.md:12-20 names "pytorch 1.5", the
 * build container has torch 2.10.0); this file restates their published definitions and is
 * pinned against outputs of the imported reference (tests/golden/, made by
 * tests/golden/make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * Convolutions accumulate in double and round once, so the result is the correctly rounded
 * value up to one ulp -- a tighter truth than either ATen's or the MFMA kernel's fp32 order.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

/* nn.Conv2d(k=3, s=1, p=1): models/LarvaNet.py:210,212,227,256,258.  b may be NULL. */
void ref_conv3x3(const float* x, const float* w, const float* b, float* out, int N, int Cin,
                 int Cout, int H, int W) {
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx) {
          double acc = b ? (double)b[co] : 0.0;
          for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 3; ++ky) {
              const int iy = y + ky - 1;
              if (iy < 0 || iy >= H) continue;
              for (int kx = 0; kx < 3; ++kx) {
                const int ix = xx + kx - 1;
                if (ix < 0 || ix >= W) continue;
                acc += (double)w[((size_t)(co * Cin + ci) * 3 + ky) * 3 + kx] *
                       (double)x[((size_t)(n * Cin + ci) * H + iy) * W + ix];
              }
            }
          out[((size_t)(n * Cout + co) * H + y) * W + xx] = (float)acc;
        }
}

/* Gradient of ref_conv3x3 w.r.t. its input (autograd of nn.Conv2d; loss.backward(),
 * models/LarvaNet.py:113): dx[n][ci][y][x] = sum_{co,ky,kx} dy[n][co][y-ky+1][x-kx+1] w[co][ci][ky][kx] */
void ref_conv3x3_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout,
                       int H, int W) {
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx) {
          double acc = 0.0;
          for (int co = 0; co < Cout; ++co)
            for (int ky = 0; ky < 3; ++ky) {
              const int oy = y - ky + 1;
              if (oy < 0 || oy >= H) continue;
              for (int kx = 0; kx < 3; ++kx) {
                const int ox = xx - kx + 1;
                if (ox < 0 || ox >= W) continue;
                acc += (double)w[((size_t)(co * Cin + ci) * 3 + ky) * 3 + kx] *
                       (double)dy[((size_t)(n * Cout + co) * H + oy) * W + ox];
              }
            }
          dx[((size_t)(n * Cin + ci) * H + y) * W + xx] = (float)acc;
        }
}

/* Gradient w.r.t. weight and bias: dw[co][ci][ky][kx] = sum dy[n][co][y][x] x[n][ci][y+ky-1][x+kx-1]. */
void ref_conv3x3_wgrad(const float* dy, const float* x, float* dw, float* db, int N, int Cin,
                       int Cout, int H, int W) {
  for (int co = 0; co < Cout; ++co) {
    for (int ci = 0; ci < Cin; ++ci)
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          double acc = 0.0;
          for (int n = 0; n < N; ++n)
            for (int y = 0; y < H; ++y) {
              const int iy = y + ky - 1;
              if (iy < 0 || iy >= H) continue;
              for (int xx = 0; xx < W; ++xx) {
                const int ix = xx + kx - 1;
                if (ix < 0 || ix >= W) continue;
                acc += (double)dy[((size_t)(n * Cout + co) * H + y) * W + xx] *
                       (double)x[((size_t)(n * Cin + ci) * H + iy) * W + ix];
              }
            }
          dw[((size_t)(co * Cin + ci) * 3 + ky) * 3 + kx] = (float)acc;
        }
    if (db) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n)
        for (size_t i = 0; i < (size_t)H * W; ++i) acc += (double)dy[(size_t)(n * Cout + co) * H * W + i];
      db[co] = (float)acc;
    }
  }
}

/* nn.PixelShuffle(r), models/LarvaNet.py:261: out[n][c][y*r+i][x*r+j] = in[n][c*r*r + i*r + j][y][x].
 * Works on raw 32-bit words so integer fixtures stay bit-exact. */
void ref_pixel_shuffle(const unsigned* in, unsigned* out, int N, int Cout, int H, int W, int r) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < Cout; ++c)
      for (int y = 0; y < H; ++y)
        for (int i = 0; i < r; ++i)
          for (int xx = 0; xx < W; ++xx)
            for (int j = 0; j < r; ++j)
              out[((size_t)(n * Cout + c) * (H * r) + y * r + i) * (W * r) + xx * r + j] =
                  in[((size_t)(n * Cout * r * r + c * r * r + i * r + j) * H + y) * W + xx];
}

void ref_pixel_unshuffle(const unsigned* in, unsigned* out, int N, int Cout, int H, int W, int r) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < Cout; ++c)
      for (int y = 0; y < H; ++y)
        for (int i = 0; i < r; ++i)
          for (int xx = 0; xx < W; ++xx)
            for (int j = 0; j < r; ++j)
              out[((size_t)(n * Cout * r * r + c * r * r + i * r + j) * H + y) * W + xx] =
                  in[((size_t)(n * Cout + c) * (H * r) + y * r + i) * (W * r) + xx * r + j];
}

/* F.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False), models/LarvaNet.py:283-285.
 * Keys cubic convolution, A = -0.75; source coordinate (dst + 0.5) / scale - 0.5; the four taps
 * floor(src)-1 .. floor(src)+2 are index-clamped to the image.  Weights in double. */
static void cubic_w(double t, double w[4]) {
  const double A = -0.75;
  const double x0 = t + 1.0, x2 = 1.0 - t, x3 = 2.0 - t;
  w[0] = ((A * x0 - 5.0 * A) * x0 + 8.0 * A) * x0 - 4.0 * A;
  w[1] = ((A + 2.0) * t - (A + 3.0)) * t * t + 1.0;
  w[2] = ((A + 2.0) * x2 - (A + 3.0)) * x2 * x2 + 1.0;
  w[3] = ((A * x3 - 5.0 * A) * x3 + 8.0 * A) * x3 - 4.0 * A;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void ref_bicubic_up(const float* in, float* out, int planes, int H, int W, int scale) {
  const int HH = H * scale, WW = W * scale;
  for (int p = 0; p < planes; ++p)
    for (int Y = 0; Y < HH; ++Y) {
      const double sy = ((double)Y + 0.5) / scale - 0.5;
      const int iy = (int)floor(sy);
      double wy[4];
      cubic_w(sy - iy, wy);
      for (int X = 0; X < WW; ++X) {
        const double sx = ((double)X + 0.5) / scale - 0.5;
        const int ix = (int)floor(sx);
        double wx[4];
        cubic_w(sx - ix, wx);
        double acc = 0.0;
        for (int r = 0; r < 4; ++r) {
          const int yy = clampi(iy - 1 + r, 0, H - 1);
          double row = 0.0;
          for (int c = 0; c < 4; ++c) row += wx[c] * (double)in[((size_t)p * H + yy) * W + clampi(ix - 1 + c, 0, W - 1)];
          acc += wy[r] * row;
        }
        out[((size_t)p * HH + Y) * WW + X] = (float)acc;
      }
    }
}

/* nn.L1Loss() (mean reduction), models/LarvaNet.py:85,108. */
double ref_l1_mean(const float* a, const float* b, long long n) {
  double s = 0.0;
  for (long long i = 0; i < n; ++i) s += fabs((double)a[i] - (double)b[i]);
  return s / (double)n;
}

/* Its gradient w.r.t. a, times the upstream scalar g: sign(a-b) * g / n with sign(0) = 0. */
void ref_l1_grad(const float* a, const float* b, float g, long long n, float* ga) {
  const float s = g / (float)n;
  for (long long i = 0; i < n; ++i) {
    const float d = a[i] - b[i];
    ga[i] = d > 0.f ? s : (d < 0.f ? -s : 0.f);
  }
}

/* One optim.AdamW step (models/LarvaNet.py:86-88,114; defaults betas=(.9,.999), eps=1e-8,
 * weight_decay=0.01): decoupled decay, then the bias-corrected Adam update. step is 1-based. */
void ref_adamw(float* p, const float* g, float* m, float* v, long long n, int step, double lr,
               double beta1, double beta2, double eps, double wd) {
  /* (hyper-parameters are doubles, as torch.optim.AdamW's Python floats are) */
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  for (long long i = 0; i < n; ++i) {
    double pi = (double)p[i] * (1.0 - lr * wd);
    const double mi = beta1 * m[i] + (1.0 - beta1) * g[i];
    const double vi = beta2 * v[i] + (1.0 - beta2) * (double)g[i] * g[i];
    pi -= (lr / bc1) * mi / (sqrt(vi) / sqrt(bc2) + eps);
    p[i] = (float)pi;
    m[i] = (float)mi;
    v[i] = (float)vi;
  }
}

/* validate.py:17-27 helpers.  np.round is round-half-to-even = rint() in the default mode. */
void ref_image_to_uint8(const float* in, unsigned char* out, long long n) {
  for (long long i = 0; i < n; ++i) {
    double v = rint((double)in[i]);
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[i] = (unsigned char)v;
  }
}

/* PSNR over all RGB pixels of two uint8 CHW images; truth is cropped top-left to the output
 * size (validate.py:20-21), differences and their squares are float32, the mean is taken over
 * float32 squares (np.mean accumulates pairwise; double here is within 1e-6 dB of it). */
double ref_image_psnr(const unsigned char* out, int C, int H, int W, const unsigned char* truth,
                      int TH, int TW) {
  double s = 0.0;
  for (int c = 0; c < C; ++c)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const float d = (float)truth[((size_t)c * TH + y) * TW + x] - (float)out[((size_t)c * H + y) * W + x];
        s += (double)(d * d);
      }
  const double mse = s / ((double)C * H * W);
  return 10.0 * log10(255.0 * 255.0 / mse);
}
