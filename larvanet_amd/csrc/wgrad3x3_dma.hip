// EXPERIMENTAL VARIANT of wgrad3x3_mfma.hip (same math, same partial-image format), selected
// with LARVA_WGRAD=dma: 1-row tiles streamed by LDS-DMA into a 2-stage ring, <= 256 registers
// and 86 KiB LDS so that it can share a CU with a conv workgroup.  Alone it is slower than the
// register-staged kernel (3x halo re-read of x makes it bandwidth-bound: 127 vs 81 us per
// 8-layer batch); kept to measure whether sharing CUs with the dgrad chain pays.
//
// Weight / bias gradient of the 3x3 convolutions (the autograd backward of nn.Conv2d at
// models/LarvaNet.py:210-212, 227, 256-258 and models/LarvaNetV2.py:318-323) for gfx950:
//
//   dW[co][ci][ky][kx] = sum_{n,y,x} dy[n][co][y][x] * x[n][ci][y+ky-1][x+kx-1]
//   db[co]             = sum_{n,y,x} dy[n][co][y][x]
//
// as a split-K GEMM on v_mfma_f32_16x16x4_f32: M = co, N = (ci, tap), K = pixels.  A workgroup
// walks a contiguous run of 1x48-pixel tiles; per tile dy (1 row) and x (3 halo rows) are
// streamed into a 2-stage LDS ring by LDS-DMA (no staging registers), the next tile landing
// under the current tile's MFMAs; every wave keeps the accumulators of ALL co groups x its share
// of the (ci group, tap) operands in registers across the whole run, so a partial image is
// written exactly once per workgroup.  A second, batched kernel sums the per-workgroup partial images in a fixed
// order (bitwise reproducible, no float atomics) and writes the PyTorch-layout gradient.
//
// Several layers are processed by ONE launch (blockIdx.y = job): in the backward pass no
// weight gradient is on the critical path, so the caller queues them and fills the chip with
// few, long-running workgroups instead of 256 short ones per layer.
//
// Roofline: fp32 MFMA, 2*9*Cin*Cout FLOP per pixel (same as the forward conv).
#include "larva_common.h"

namespace larva {
namespace dma {

constexpr int kMaxJobs = 16;

struct WgradJob {
  const float* dy;   // [N][COUT][H][W]
  const float* x;    // [N][CIN][H][W]
  float* partial;    // [splits][ NB*CT*256 + COUT ]
};

struct WgradBatch {
  WgradJob job[kMaxJobs];
  int N, H, W;
  int tiles_x, tiles_y;
  int vec_ok;
};

// Source of every LDS-DMA lane that is zero padding (outside the image, or layout padding).
__device__ __attribute__((aligned(16))) float g_wg_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

// One tile = ONE image row x 48 columns (12 k-steps of 4 pixels).  Per tile the stage holds
//   dy [COUT][PSD = 52]            (48 pixels + 4 pad)
//   x  [CIN][3 halo rows][kRS] at channel stride PSX = 172 (row idx 3 = x0-1, 4..51 = x0..x0+47, 52 = x0+48)
// in 1 KiB (VEC: 16 B per lane) or 256 B (!VEC: 4 B per lane) LDS-DMA pieces.  Channel strides
// are multiples of 4 floats (DMA lanes are 16-byte slots) with stride/4 odd: the 16 channels x 2
// pixels of a 32-lane ds_read_b32 group then fall on 16 banks, 2 lanes each -- a 2-way conflict
// that costs nothing here (10 LDS reads per 21 MFMAs).
template <int COUT, int CIN, bool VEC>
struct WgCfg {
  static constexpr int CT = COUT / 16;
  static constexpr int NB = (CIN / 16) * 9;  // B operands: (ci group, tap)
  static constexpr int PSD = 52;
  static constexpr int PSX = 172;
  static constexpr int EL = VEC ? 4 : 1;               // floats per DMA lane
  static constexpr int PIECE = 64 * EL;                // floats per DMA instruction
  static constexpr int DY_PIECES = (COUT * PSD + PIECE - 1) / PIECE;
  static constexpr int X_PIECES = (CIN * PSX + PIECE - 1) / PIECE;
  static constexpr int DY_FLOATS = DY_PIECES * PIECE;
  static constexpr int X_FLOATS = X_PIECES * PIECE;
  static constexpr int PIECES = DY_PIECES + X_PIECES;
  static constexpr int NPW = (PIECES + 3) / 4;         // pieces per wave per tile
  static constexpr int STAGE_FLOATS = DY_FLOATS + X_FLOATS;
  static constexpr size_t LDS_BYTES = 2 * STAGE_FLOATS * sizeof(float);
  static constexpr int KSTEPS = kTileCols / 4;         // 12
  // The next tile's pieces are all issued in the FIRST half of this tile's k-steps: issued -> landed
  // takes ~1 us, and the tile after the next barrier needs them (issued evenly over the 12 steps
  // the last pieces were ~0.5 us old at the barrier and every tile stalled on them).
  static constexpr int ISSUE_STEPS = KSTEPS / 2;
  static constexpr int PER_STEP = (NPW + ISSUE_STEPS - 1) / ISSUE_STEPS;
  static constexpr int PARTIAL_FLOATS = NB * CT * 256 + COUT;
};

template <int COUT, int CIN, bool VEC>
struct WgPlan {
  // piece i of this wave: lane's source offset (floats) relative to dy/x at (image n, row y,
  // column 0), LDS offset, and validity classes (bit i): always-padding lanes have no bit set.
  int off[WgCfg<COUT, CIN, VEC>::NPW];
  unsigned long long ok_mid;    // valid whatever the row
  unsigned long long need_top;  // valid only if row y-1 exists
  unsigned long long need_bot;  // valid only if row y+1 exists
};

template <int COUT, int CIN, bool VEC>
__device__ __forceinline__ void wg_make_plan(const WgradBatch& b, int wave, int lane, int x0,
                                             WgPlan<COUT, CIN, VEC>& pl) {
  using C = WgCfg<COUT, CIN, VEC>;
  static_assert(C::NPW <= 64, "validity masks are 64 bits");
  pl.ok_mid = pl.need_top = pl.need_bot = 0;
#pragma unroll
  for (int i = 0; i < C::NPW; ++i) {
    const int p = min(wave + 4 * i, C::PIECES - 1);
    const bool isx = p >= C::DY_PIECES;
    // dy element
    const int de = (p * 64 + lane) * C::EL;
    const int dco = de / C::PSD, dcol = de - dco * C::PSD;
    const bool d_ok = dco < COUT && dcol < kTileCols && x0 + dcol < b.W;
    const int d_off = dco * b.H * b.W + x0 + dcol;
    // x element
    const int xe = ((p - C::DY_PIECES) * 64 + lane) * C::EL;
    const int xci = xe / C::PSX, xrem = xe - xci * C::PSX;
    const int xr = xrem / kRS, xidx = xrem - xr * kRS;
    const int gx = x0 - 4 + xidx;
    const bool x_ok = xci < CIN && xr < 3 && gx >= 0 && gx < b.W;
    const int x_off = (xci * b.H + (xr - 1)) * b.W + gx;
    const bool ok = isx ? x_ok : d_ok;
    pl.off[i] = ok ? (isx ? x_off : d_off) : 0;
    const unsigned long long bit = (ok ? 1ull : 0ull) << i;
    if (isx && xr == 0) pl.need_top |= bit;
    else if (isx && xr == 2) pl.need_bot |= bit;
    else pl.ok_mid |= bit;
  }
}

struct WgTile {
  const float* dy;  // dy at (n, channel 0, row y, column 0)
  const float* x;   // x  at (n, channel 0, row y, column 0)
  unsigned long long ok;  // per-lane validity bits for this tile's rows
};

template <int COUT, int CIN, bool VEC>
__device__ __forceinline__ void wg_dma_piece(const WgPlan<COUT, CIN, VEC>& pl, int i, int wave, const WgTile& t,
                                             float* stage) {
  using C = WgCfg<COUT, CIN, VEC>;
  const int p = min(wave + 4 * i, C::PIECES - 1);  // scalar; the x pieces follow the dy pieces in LDS
  const bool isx = p >= C::DY_PIECES;
  float* dst = stage + p * C::PIECE;
  const uint64_t base = reinterpret_cast<uint64_t>(isx ? t.x : t.dy);
  const uint64_t zero = reinterpret_cast<uint64_t>(&g_wg_zero_page[0]);
  // 64-bit signed offset: the x base may sit one row before the tensor start for row y-1 of y = 0
  // (those lanes are invalid and read the zero page instead).
  const uint64_t addr = ((t.ok >> i) & 1ull) ? base + 4ll * (long long)pl.off[i] : zero;
  lds_dma<VEC ? 16 : 4>(reinterpret_cast<const void*>(addr), dst);
}

// k-step `ks` covers pixels (4 ks .. 4 ks + 3) of the tile's row.
template <int COUT, int CIN, bool VEC, int B0, int NBW>
__device__ __forceinline__ void wg_read(const float* a_base, const float* b_base, int ks,
                                        float (&av)[COUT / 16], float (&bv)[NBW]) {
  using C = WgCfg<COUT, CIN, VEC>;
  const int col = 4 * ks;
#pragma unroll
  for (int c = 0; c < C::CT; ++c) av[c] = a_base[c * 16 * C::PSD + col];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int bi = B0 + j, cit = bi / 9, tap = bi % 9, ky = tap / 3, kx = tap % 3;
    bv[j] = b_base[cit * 16 * C::PSX + ky * kRS + col + kx];
  }
}

template <int KEEP>
__device__ __forceinline__ void wg_wait_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(KEEP) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int COUT, int CIN, bool VEC, int B0, int NBW>
__device__ __forceinline__ void wg_role(const WgradBatch& b, const WgradJob& j, float* smem,
                                        int split, int splits, int wave, int tid, bool bias_wave) {
  using C = WgCfg<COUT, CIN, VEC>;
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;

  f32x4 acc[C::CT][NBW];
#pragma unroll
  for (int c = 0; c < C::CT; ++c)
#pragma unroll
    for (int k = 0; k < NBW; ++k) acc[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[C::CT];
#pragma unroll
  for (int c = 0; c < C::CT; ++c) bsum[c] = 0.f;

  // tiles are (n, y, tx) with tx fastest; this workgroup owns a contiguous run
  const int total = b.N * b.H * b.tiles_x;
  const int t_begin = (int)(((long long)total * split) / splits);
  const int t_end = (int)(((long long)total * (split + 1)) / splits);
  const size_t plane = (size_t)b.H * b.W;

  WgPlan<COUT, CIN, VEC> pl;
  int plan_tx = -1;
  auto tile_of = [&](int t, WgTile& wt) {
    const int tx = t % b.tiles_x;
    const int t2 = t / b.tiles_x;
    const int y = t2 % b.H;
    const int n = t2 / b.H;
    if (tx != plan_tx) {  // wave-uniform; only ever re-planned for images wider than one tile
      wg_make_plan<COUT, CIN, VEC>(b, wave, lane, tx * kTileCols, pl);
      plan_tx = tx;
    }
    wt.dy = j.dy + (size_t)n * COUT * plane + (size_t)y * b.W;
    wt.x = j.x + (size_t)n * CIN * plane + (size_t)y * b.W;
    wt.ok = pl.ok_mid | (y > 0 ? pl.need_top : 0ull) | (y + 1 < b.H ? pl.need_bot : 0ull);
  };

  WgTile cur{}, nxt{};
  if (t_begin < t_end) {
    tile_of(t_begin, cur);
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) wg_dma_piece<COUT, CIN, VEC>(pl, i, wave, cur, smem);
  }
  for (int t = t_begin; t < t_end; ++t) {
    // tile t has landed (every wave's pieces, after the barrier) and every wave is done with the
    // other stage, which tile t+1 now streams into underneath this tile's MFMAs.  Past the end
    // the last tile is streamed once more (nobody reads it): no branch inside the MFMA loop.
    wg_wait_and_barrier<0>();
    float* stage = smem + ((t - t_begin) & 1) * C::STAGE_FLOATS;
    float* ostage = smem + (((t - t_begin) & 1) ^ 1) * C::STAGE_FLOATS;
    tile_of(min(t + 1, t_end - 1), nxt);
    const float* a_base = stage + lr * C::PSD + lq;
    const float* b_base = stage + C::DY_FLOATS + lr * C::PSX + lq + 3;

    float av[2][C::CT], bv[2][NBW];
    wg_read<COUT, CIN, VEC, B0, NBW>(a_base, b_base, 0, av[0], bv[0]);
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
      if (ks + 1 < C::KSTEPS)
        wg_read<COUT, CIN, VEC, B0, NBW>(a_base, b_base, ks + 1, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
#pragma unroll
      for (int u = 0; u < C::PER_STEP; ++u) {
        const int i = ks * C::PER_STEP + u;
        if (i < C::NPW) wg_dma_piece<COUT, CIN, VEC>(pl, i, wave, nxt, ostage);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < C::CT; ++c)
#pragma unroll
        for (int k = 0; k < NBW; ++k)
          acc[c][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1][c], bv[ks & 1][k], acc[c][k], 0, 0, 0);
      if (bias_wave) {
#pragma unroll
        for (int c = 0; c < C::CT; ++c) bsum[c] += av[ks & 1][c];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // no LDS-DMA may be in flight when the workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // Partial image: [b][ct][lane][4] = the accumulator registers as they stand (1 KiB per tile
  // per store instruction, fully coalesced).  acc[c][k][r] = dW[co = 16c + 4lq + r][ci = 16*cit + lr][tap].
  float* part = j.partial + (size_t)split * C::PARTIAL_FLOATS;
#pragma unroll
  for (int c = 0; c < C::CT; ++c)
#pragma unroll
    for (int k = 0; k < NBW; ++k)
      *reinterpret_cast<f32x4*>(part + (((B0 + k) * C::CT + c) * 64 + lane) * 4) = acc[c][k];
  if (bias_wave) {
#pragma unroll
    for (int c = 0; c < C::CT; ++c) {
      float v = bsum[c];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 16) part[C::NB * C::CT * 256 + c * 16 + lane] = v;
    }
  }
}

// 2 waves/SIMD launch bound: <= 256 registers, so a conv workgroup (72 KiB LDS, < 128 VGPRs)
// fits on the same CU beside this one (86 KiB LDS at 48 channels).
template <int COUT, int CIN, bool VEC>
__global__ __launch_bounds__(256, (VEC && COUT * CIN <= 48 * 48) ? 2 : 1) void wgrad3x3_kernel(WgradBatch b) {
  using C = WgCfg<COUT, CIN, VEC>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WgradJob& j = b.job[blockIdx.y];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, splits = gridDim.x;
  // (ci group, tap) operands dealt to the 4 waves: NB = 9 -> 3,2,2,2; 18 -> 5,5,4,4;
  // 27 -> 7,7,7,6; 36 -> 9,9,9,9.
  constexpr int NB = C::NB;
  constexpr int W0 = (NB + 3) / 4, W1 = (NB + 2) / 4, W2 = (NB + 1) / 4, W3 = NB / 4;
  if (wave == 0) wg_role<COUT, CIN, VEC, 0, W0>(b, j, smem, split, splits, wave, tid, true);
  else if (wave == 1) wg_role<COUT, CIN, VEC, W0, W1>(b, j, smem, split, splits, wave, tid, false);
  else if (wave == 2) wg_role<COUT, CIN, VEC, W0 + W1, W2>(b, j, smem, split, splits, wave, tid, false);
  else wg_role<COUT, CIN, VEC, W0 + W1 + W2, W3>(b, j, smem, split, splits, wave, tid, false);
}

// ---------------------------------------------------------------------------------------------
// Fixed-order reduction of the partial images into PyTorch-layout gradients.
// ---------------------------------------------------------------------------------------------
struct ReduceJob {
  const float* partial;  // [splits][PARTIAL_FLOATS]
  float* dw;             // [cout][w_cin_total][3][3]
  float* db;             // [cout] or null
  int cin_off;           // first input channel of this job inside dw
  int cin_valid;         // channels of this job that exist in dw (3 for the head, else CIN)
  int w_cin_total;
};

struct ReduceBatch {
  ReduceJob job[kMaxJobs];
  int splits;
  int cout, cin;  // kernel shape of every job in the batch
};

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceBatch rb) {
  const ReduceJob& j = rb.job[blockIdx.y];
  const int ct_n = rb.cout / 16, nb = (rb.cin / 16) * 9;
  const int n_w = nb * ct_n * 256;
  const int pf = n_w + rb.cout;  // multiple of 4
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= pf) return;
  // one 16-byte column of the [splits][pf] partial matrix per thread; 8 independent loads in flight
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const float* p = j.partial + i;
  int k = 0;
  for (; k + 8 <= rb.splits; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(k + u) * pf);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < rb.splits; ++k) s += *reinterpret_cast<const f32x4*>(p + (size_t)k * pf);
  if (i < n_w) {
    const int lane = (i >> 2) & 63;
    const int t = i >> 8;
    const int c = t % ct_n, bi = t / ct_n;
    const int cit = bi / 9, tap = bi % 9;
    const int co0 = c * 16 + (lane >> 4) * 4;
    const int ci = cit * 16 + (lane & 15);
    if (ci < j.cin_valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) j.dw[((size_t)(co0 + r) * j.w_cin_total + j.cin_off + ci) * 9 + tap] = s[r];
    }
  } else if (j.db) {
#pragma unroll
    for (int r = 0; r < 4; ++r) j.db[i - n_w + r] = s[r];
  }
}

template <int COUT, int CIN>
static hipError_t launch_wgrad(const WgradBatch& b, int njobs, int splits, hipStream_t stream) {
  constexpr size_t lds_v = WgCfg<COUT, CIN, true>::LDS_BYTES, lds_s = WgCfg<COUT, CIN, false>::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_kernel<COUT, CIN, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_kernel<COUT, CIN, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (b.vec_ok)
    hipLaunchKernelGGL((wgrad3x3_kernel<COUT, CIN, true>), dim3(splits, njobs), dim3(256), lds_v, stream, b);
  else
    hipLaunchKernelGGL((wgrad3x3_kernel<COUT, CIN, false>), dim3(splits, njobs), dim3(256), lds_s, stream, b);
  return hipGetLastError();
}

}  // namespace dma
}  // namespace larva

using namespace larva;
using namespace larva::dma;

extern "C" {

// Weight + bias gradients of `njobs` (<= 16) same-shape 3x3 convolutions in two launches.
// Job i: dy[i] [N][cout][H][W], x[i] [N][cin][H][W] -> partial[i] (workspace of
// larva_wgrad_partial_floats floats) -> dw[i] [cout][w_cin_total[i]][3][3] at input-channel
// offset cin_off[i] (only the first cin_valid[i] channels are written; the rest of `cin` is
// zero padding of x), db[i] [cout] (may be null).  Gradients are OVERWRITTEN, not accumulated.
int larva_conv3x3_wgrad_dma(const float* const* dy, const float* const* x, float* const* partial,
                        float* const* dw, float* const* db, const int* cin_off,
                        const int* cin_valid, const int* w_cin_total, int njobs, int splits,
                        int N, int cout, int cin, int H, int W, void* stream) {
  if (njobs < 1 || njobs > kMaxJobs || splits < 1 || N <= 0 || H <= 0 || W <= 0)
    return (int)hipErrorInvalidValue;
  WgradBatch b{};
  ReduceBatch rb{};
  bool aligned = (W % 4 == 0);
  for (int i = 0; i < njobs; ++i) {
    if (!dy[i] || !x[i] || !partial[i] || !dw[i]) return (int)hipErrorInvalidValue;
    b.job[i] = WgradJob{dy[i], x[i], partial[i]};
    aligned = aligned && ((reinterpret_cast<uintptr_t>(dy[i]) & 15) == 0) &&
              ((reinterpret_cast<uintptr_t>(x[i]) & 15) == 0);
    rb.job[i] = ReduceJob{partial[i], dw[i], db ? db[i] : nullptr, cin_off[i], cin_valid[i], w_cin_total[i]};
  }
  b.N = N; b.H = H; b.W = W;
  b.tiles_x = (W + kTileCols - 1) / kTileCols;
  b.tiles_y = H;  // one image row per tile
  b.vec_ok = aligned ? 1 : 0;
  if ((long long)(cin > cout ? cin : cout) * H * W >= (1ll << 31)) return (int)hipErrorInvalidValue;
  const int total = N * b.tiles_x * b.tiles_y;
  if (splits > total) splits = total;
  rb.splits = splits; rb.cout = cout; rb.cin = cin;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (cout == 48 && cin == 48) e = launch_wgrad<48, 48>(b, njobs, splits, s);
  else if (cout == 48 && cin == 16) e = launch_wgrad<48, 16>(b, njobs, splits, s);
  else if (cout == 32 && cin == 32) e = launch_wgrad<32, 32>(b, njobs, splits, s);
  else if (cout == 64 && cin == 64) e = launch_wgrad<64, 64>(b, njobs, splits, s);
  else return (int)hipErrorInvalidValue;
  if (e != hipSuccess) return (int)e;
  const int pf = (cin / 16) * 9 * (cout / 16) * 256 + cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((pf / 4 + 255) / 256, njobs), dim3(256), 0, s, rb);
  return (int)hipGetLastError();
}

}  // extern "C"
