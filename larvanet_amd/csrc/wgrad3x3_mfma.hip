// Weight / bias gradient of the 3x3 convolutions (the autograd backward of nn.Conv2d at
// models/LarvaNet.py:210-212, 227, 256-258 and models/LarvaNetV2.py:318-323) for gfx950:
//
//   dW[co][ci][ky][kx] = sum_{n,y,x} dy[n][co][y][x] * x[n][ci][y+ky-1][x+kx-1]
//   db[co]             = sum_{n,y,x} dy[n][co][y][x]
//
// as a split-K GEMM on v_mfma_f32_16x16x4_f32: M = co, N = (ci, tap), K = pixels.  A workgroup
// walks a contiguous run of 3x48-pixel tiles; per tile it stages dy (3 rows) and x (5 halo rows)
// in LDS, and every wave keeps the accumulators of ALL co groups x its share of the (ci group,
// tap) operands in registers across the whole run, so a partial image is written exactly once
// per workgroup.  A second, batched kernel sums the per-workgroup partial images in a fixed
// order (bitwise reproducible, no float atomics) and writes the PyTorch-layout gradient.
//
// Several layers are processed by ONE launch (blockIdx.y = job): in the backward pass no
// weight gradient is on the critical path, so the caller queues them and fills the chip with
// few, long-running workgroups instead of 256 short ones per layer.
//
// Two kernels: wgrad3x3_pipe_kernel (48x48 channels on 16-byte aligned tensors: LDS double buffer,
// the next tile's staging laid out in the MFMA shadows, 12.5 us per tile against 11.1 us for its
// MFMAs alone = 120 TFLOP/s in the training step) and the generic register-staged wgrad3x3_kernel
// (every other shape; 16 us per layer at 48x48).  Staging is through registers in both: an LDS-DMA
// variant with 1-row tiles (tried, removed) re-read the x halo 3x and became bandwidth-bound, and
// 3-row tiles with the 16-byte channel strides DMA needs do not fit the LDS twice.
//
// Roofline: fp32 MFMA, 2*9*Cin*Cout FLOP per pixel (same as the forward conv).
#include "larva_common.h"
#include "larva_loss.h"

#include <stdlib.h>

#include <type_traits>

// Timing-only ablations of the pipelined kernel (tools/diag_wgrad.py): 1 no MFMA, 2 no staging
// (loads + LDS writes of the next tile), 4 no operand reads, 8 no bias sums, 16 no LDS writes,
// 32 no address arithmetic, 64 no global loads, 128 (with 16) wait for the loads without writing.  Results are wrong
// by construction when nonzero.
#ifndef WG_DIAG
#define WG_DIAG 0
#endif
// (Settled by earlier rounds' A/Bs and no longer switchable: the pipelined kernel's MFMAs are inline asm with the
// accumulator tied in an AGPR quad (r1: -7 %); operand RUNS for the shapes whose lightest wave has fewer than 17 MFMAs per
// k-step -- (32,32), (64,32), (48,32), the heads -- and not for (48,48) (r4: its spelled-out layout is 0.7 % faster); the
// load -> LDS-write distance of 6 k-steps (r4: 10 / 14 no better).)
constexpr int kWgMinGaps = 6;   // MFMAs per k-step of its lightest wave below which a shape stays off the pipelined kernel ((48,16): 6)


namespace larva {

constexpr int kMaxJobs = 64;

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
  // flat launch (wgrad3x3_pipe_flat_kernel): the tiles of all njobs layers form one sequence that is
  // cut evenly over the workgroups; first_wg[j] = the first workgroup whose share touches layer j
  // (its partial image for that layer is image 0, the next workgroup's image 1, ...)
  int njobs;
  short first_wg[kMaxJobs];
  // optional tail of the flat sequence: ONE (COUT, 16) layer -- the 3 -> 48 head on its zero-padded input --
  // on the register-staged role.  It enters the sequence as head_units pseudo-tiles (its tiles priced in units of
  // a 48 -> 48 tile, see flat_head_units); pseudo-tile p covers its tiles [p T / head_units, (p + 1) T / head_units).
  WgradJob head;
  int head_units;       // 0: no head job
  int head_first_wg;    // the first workgroup whose share reaches into the head
  // pipelined kernels only (round 4): x has halves * CIN channels and a workgroup walks its tiles once per half -- a
  // (64, 64) layer, whose two tile buffers (217 KB) do not fit the LDS, runs as two (64, 32) passes into ONE partial
  // image of the (64, 64) layout ((ci group, tap) operands 18 h .. 18 h + 17 come from pass h).  0 / 1: one pass.
  int halves;
#if defined(LARVA_DIAG) && (LARVA_DIAG & 512)
  unsigned long long* diag_area;   // stamp area of this launch (tools/diag_step.py), null: none
#endif
};

#if defined(LARVA_DIAG) && (LARVA_DIAG & 512)
// In-kernel timeline of the flat weight-gradient launches (with the conv kernels' own stamps, conv3x3_mfma.hip): wave 0
// of every workgroup writes the 100 MHz wall clock at entry and before it ends, and its HW_ID | XCC_ID << 32, into
// area[workgroup * 4 + {0, 1, 2}] (the launch carries its area's address in its arguments).
__device__ __forceinline__ void wstamp(unsigned long long* area, int k) {
  if (!area || threadIdx.x != 0 || blockIdx.x >= 256) return;
  unsigned long long* p = area + (size_t)blockIdx.x * 4;
  p[k] = __builtin_amdgcn_s_memrealtime();
  if (k == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    p[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  }
}
#define WSTAMP(b, k) wstamp((b).diag_area, (k))
#else
#define WSTAMP(b, k) ((void)0)
#endif

// A (48, 16) tile on the register-staged role against a (48, 48) tile on the pipelined one: a third of the MFMAs,
// but two workgroup barriers and an exposed LDS write per tile -- measured 0.58 of the time (same-box builds with
// the price at 0.4 / 0.5 / 0.6 / 0.7: step 1.835 / 1.709 / 1.636 / 1.634 ms: under-priced, the workgroups whose whole
// share is head tiles finish last and set the launch's duration).  Priced at 0.7: finishing early costs a fraction
// of one CU.  Round 4: the tail runs the PIPELINED role at 16 input channels (6-9 MFMAs per k-step, operand runs):
// 0.47 of a (48, 48) tile launched on its own (tools/bench_wgrad_head.py, 0.55 register-staged on the same box);
// priced at 0.6.
constexpr int kHeadCost10 = 6;
__host__ __device__ constexpr int flat_head_units(int tiles) { return (tiles * kHeadCost10 + 9) / 10; }

template <int COUT, int CIN>
struct WgCfg {
  static constexpr int CT = COUT / 16;
  static constexpr int NB = (CIN / 16) * 9;  // B operands: (ci group, tap)
  static constexpr int NBW = (NB + 3) / 4;   // per wave (last waves may own one fewer)
  // Channel strides == 2*odd (mod 32): 16 channels x 2 adjacent pixels hit 32 distinct banks.
  static constexpr int PSD = 146;            // dy: 3 rows x 48
  static constexpr int PSX = 278;            // x : 5 rows x kRS (tail of the last row unused)
  static constexpr int DY_FLOATS = COUT * PSD;
  static constexpr int X_FLOATS = CIN * PSX + 8;
  static constexpr size_t LDS_BYTES = (DY_FLOATS + X_FLOATS) * sizeof(float);
  static constexpr int DY_SLOTS = COUT * kTileRows * (kTileCols / 4);
  static constexpr int X_SLOTS = CIN * kHaloRows * (kRS / 4);
  static constexpr int DY_ITERS = (DY_SLOTS + 255) / 256;
  static constexpr int X_ITERS = (X_SLOTS + 255) / 256;
  static constexpr int PARTIAL_FLOATS = NB * CT * 256 + COUT;
};

template <int COUT, int CIN>
struct WgStaging {
  f32x4 dy[WgCfg<COUT, CIN>::DY_ITERS];
  f32x4 x[WgCfg<COUT, CIN>::X_ITERS];
  uint32_t ok_dy, ok_x;
};

template <int COUT, int CIN, bool VEC>
__device__ __forceinline__ void wg_load(const WgradBatch& b, const WgradJob& j, int tile, int tid,
                                        WgStaging<COUT, CIN>& st) {
  using C = WgCfg<COUT, CIN>;
  const int tx = tile % b.tiles_x;
  const int t2 = tile / b.tiles_x;
  const int ty = t2 % b.tiles_y;
  const int n = t2 / b.tiles_y;
  const int x0 = tx * kTileCols, y0 = ty * kTileRows;
  const size_t plane = (size_t)b.H * b.W;
  uint32_t ok_dy = 0, ok_x = 0;
  const float* dyimg = j.dy + (size_t)n * COUT * plane;
#pragma unroll
  for (int i = 0; i < C::DY_ITERS; ++i) {
    int s = tid + i * 256;
    const bool live = s < C::DY_SLOTS;
    s = live ? s : 0;
    const int co = s / (kTileRows * 12);
    const int rem = s - co * (kTileRows * 12);
    const int r = rem / 12;
    const int q = rem - r * 12;
    const int gy = y0 + r, gx = x0 + 4 * q;
    const bool row_ok = live && gy < b.H;
    const float* row = dyimg + (size_t)co * plane + (size_t)min(gy, b.H - 1) * b.W;
    f32x4 v;
    if constexpr (VEC) {
      v = *reinterpret_cast<const f32x4*>(row + min(gx, b.W - 4));
      ok_dy |= ((row_ok && gx < b.W) ? 1u : 0u) << i;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (row_ok && gx + e < b.W) ? row[min(gx + e, b.W - 1)] : 0.f;
    }
    st.dy[i] = v;
  }
  const float* ximg = j.x + (size_t)n * CIN * plane;
#pragma unroll
  for (int i = 0; i < C::X_ITERS; ++i) {
    int s = tid + i * 256;
    const bool live = s < C::X_SLOTS;
    s = live ? s : 0;
    const int ci = s / (kHaloRows * 14);
    const int rem = s - ci * (kHaloRows * 14);
    const int r = rem / 14;
    const int q = rem - r * 14;
    const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * q;
    const bool row_ok = live && gy >= 0 && gy < b.H;
    const float* row = ximg + (size_t)ci * plane + (size_t)min(max(gy, 0), b.H - 1) * b.W;
    f32x4 v;
    if constexpr (VEC) {
      v = *reinterpret_cast<const f32x4*>(row + min(max(gx, 0), b.W - 4));
      ok_x |= ((row_ok && gx >= 0 && gx < b.W) ? 1u : 0u) << i;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = gx + e;
        v[e] = (row_ok && x >= 0 && x < b.W) ? row[min(max(x, 0), b.W - 1)] : 0.f;
      }
    }
    st.x[i] = v;
  }
  st.ok_dy = ok_dy;
  st.ok_x = ok_x;
}

__device__ __forceinline__ void lds_store4(float* p, f32x4 v) {
  // Channel bases are only 8-byte aligned (odd strides/2), so two 8-byte stores.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
  *reinterpret_cast<f32x2*>(p + 2) = f32x2{v[2], v[3]};
}

template <int COUT, int CIN, bool VEC>
__device__ __forceinline__ void wg_store(float* s_dy, float* s_x, int tid,
                                         const WgStaging<COUT, CIN>& st) {
  using C = WgCfg<COUT, CIN>;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < C::DY_ITERS; ++i) {
    const int s = tid + i * 256;
    if (i * 256 + 255 < C::DY_SLOTS || s < C::DY_SLOTS) {
      const int co = s / (kTileRows * 12);
      const int rem = s - co * (kTileRows * 12);
      const int r = rem / 12;
      const int q = rem - r * 12;
      f32x4 v = st.dy[i];
      if constexpr (VEC) v = ((st.ok_dy >> i) & 1u) ? v : zero;
      lds_store4(s_dy + co * C::PSD + r * kTileCols + 4 * q, v);
    }
  }
#pragma unroll
  for (int i = 0; i < C::X_ITERS; ++i) {
    const int s = tid + i * 256;
    if (i * 256 + 255 < C::X_SLOTS || s < C::X_SLOTS) {
      const int ci = s / (kHaloRows * 14);
      const int rem = s - ci * (kHaloRows * 14);
      const int r = rem / 14;
      const int q = rem - r * 14;
      f32x4 v = st.x[i];
      if constexpr (VEC) v = ((st.ok_x >> i) & 1u) ? v : zero;
      lds_store4(s_x + ci * C::PSX + r * kRS + 4 * q, v);
    }
  }
}

// k-step `ks` of a tile covers pixels (row = ks / 12, cols 4*(ks%12) .. +3).
template <int COUT, int CIN, int B0, int NBW>
__device__ __forceinline__ void wg_read(const float* a_base, const float* b_base, int ks,
                                        float (&av)[COUT / 16], float (&bv)[NBW]) {
  using C = WgCfg<COUT, CIN>;
  const int row = ks / 12, col = 4 * (ks % 12);
#pragma unroll
  for (int c = 0; c < C::CT; ++c) av[c] = a_base[c * 16 * C::PSD + row * kTileCols + col];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int bi = B0 + j, cit = bi / 9, tap = bi % 9, ky = tap / 3, kx = tap % 3;
    bv[j] = b_base[cit * 16 * C::PSX + (row + ky) * kRS + col + kx];
  }
}

template <int COUT, int CIN, bool VEC, int B0, int NBW>
__device__ __forceinline__ void wg_role_range(const WgradBatch& b, const WgradJob& j, float* smem,
                                              int t_begin, int t_end, float* part, int tid, bool bias_wave) {
  using C = WgCfg<COUT, CIN>;
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  float* s_dy = smem;
  float* s_x = smem + C::DY_FLOATS;
  const float* a_base = s_dy + lr * C::PSD + lq;
  const float* b_base = s_x + lr * C::PSX + lq + 3;

  f32x4 acc[C::CT][NBW];
#pragma unroll
  for (int c = 0; c < C::CT; ++c)
#pragma unroll
    for (int k = 0; k < NBW; ++k) acc[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[C::CT];
#pragma unroll
  for (int c = 0; c < C::CT; ++c) bsum[c] = 0.f;

  // The next tile's global loads ride in registers under the MFMA block of the current one;
  // at 64x64 channels accumulators + staging exceed the register file, so that shape loads
  // its tile synchronously instead.
  constexpr bool kPrefetch = (COUT * CIN <= 48 * 48);
  WgStaging<COUT, CIN> st;
  if (kPrefetch && t_begin < t_end) wg_load<COUT, CIN, VEC>(b, j, t_begin, tid, st);
  for (int t = t_begin; t < t_end; ++t) {
    if (!kPrefetch) wg_load<COUT, CIN, VEC>(b, j, t, tid, st);
    __syncthreads();  // every wave is done reading the previous tile
    wg_store<COUT, CIN, VEC>(s_dy, s_x, tid, st);
    __syncthreads();
    if (kPrefetch && t + 1 < t_end) wg_load<COUT, CIN, VEC>(b, j, t + 1, tid, st);

    float av[2][C::CT], bv[2][NBW];
    wg_read<COUT, CIN, B0, NBW>(a_base, b_base, 0, av[0], bv[0]);
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) {
      if (ks + 1 < 36) wg_read<COUT, CIN, B0, NBW>(a_base, b_base, ks + 1, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
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

  // Partial image: [b][ct][lane][4] = the accumulator registers as they stand (1 KiB per tile
  // per store instruction, fully coalesced).  acc[c][k][r] = dW[co = 16c + 4lq + r][ci = 16*cit + lr][tap].
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

template <int COUT, int CIN, bool VEC, int B0, int NBW>
__device__ __forceinline__ void wg_role(const WgradBatch& b, const WgradJob& j, float* smem,
                                        int split, int splits, int tid, bool bias_wave) {
  const int total = b.N * b.tiles_x * b.tiles_y;
  const int t_begin = (int)(((long long)total * split) / splits);
  const int t_end = (int)(((long long)total * (split + 1)) / splits);
  wg_role_range<COUT, CIN, VEC, B0, NBW>(b, j, smem, t_begin, t_end,
                                         j.partial + (size_t)split * WgCfg<COUT, CIN>::PARTIAL_FLOATS, tid, bias_wave);
}

// ---------------------------------------------------------------------------------------------
// Pipelined variant (16-byte staging path, 48x48 channels): the tile run is double-buffered in
// LDS and ALL staging work for tile t+1 -- address arithmetic, global loads, LDS writes -- is
// spread over the 36 k-steps of tile t.  With one wave per SIMD the matrix pipe idles whenever
// the wave issues anything else for longer than an MFMA's shadow: a v_mfma_f32_16x16x4_f32
// occupies the pipe for 32 cycles and the vector issue port for 8 of them, so ~24 cycles of other
// instructions per MFMA are free and every cycle beyond that is lost (measured: the same work
// lumped between MFMA blocks cost 15.2 us per tile against 10.9 us for the MFMAs alone).  So
// each k-step is laid out as "gaps" of one MFMA + at most a handful of filler instructions:
//     gaps 0..7     one LDS read of the NEXT k-step's operands each
//     gaps 8..11    address + bounds arithmetic of staging slot KS of the next tile (4 VALU each)
//     gap  12 + w   its global load (out-of-image slots read a zero page instead), w = wave
//     gap  16 + w   LDS write of slot KS-LAG, loaded LAG k-steps earlier
// (sched_group_barrier pipeline; no VALU accumulation anywhere: db is an MFMA against ones).
// One workgroup barrier per tile.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_wg_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

template <int COUT, int CIN>
struct WgPipe {
  using C = WgCfg<COUT, CIN>;
  static constexpr int NSLOT = C::DY_ITERS + C::X_ITERS;
  // k-steps between a slot's global load and its LDS write (~0.35 us each).  Longer only costs
  // registers: at 15 the allocator starts shuttling values through AGPRs (v_accvgpr_* in the loop).
  static constexpr int LAG = 6;
  static constexpr int MIN_GAPS = C::CT * (C::NB / 4);          // MFMAs per k-step of the lightest wave
  // The k-step's fillers -- CT + NBW operand reads, two address steps, the global load, the LDS write -- are laid
  // out one per MFMA gap where the lightest wave has >= 17 MFMAs per k-step ((48,48): 21); with fewer (round 3:
  // (32,32), 8-10 MFMAs) they are dealt evenly over the gaps there are, up to two per gap (gap_of in pipe_gap_asm) --
  // an MFMA's shadow holds ~24 issue cycles, a filler costs 4-16.
  static constexpr bool SPREAD = MIN_GAPS >= 17;
  // Operand runs (round 4).  The K index of an MFMA (lane quarter lq) is free to mean any pixel as long as A and B agree.
  // Until round 3 k-step ks of a tile row covered pixels 4 ks' .. 4 ks' + 3 (ks' = ks % 12), lane quarter lq pixel
  // 4 ks' + lq: a lane's operands of successive k-steps lie 16 bytes apart and every one is a 4-byte LDS read -- CT + NBW
  // reads per k-step beside CT * NBW MFMAs, which at 8-10 MFMAs per k-step ((32,32)) no longer fit the MFMAs' shadows
  // (0.60 of peak against (48,48)'s 0.81).  With RUNS lane quarter lq owns the twelfth-of-a-tile-row run of pixels
  // 12 lq .. 12 lq + 11 and k-step ks' takes pixel 12 lq + ks' of it: the dy operands of FOUR successive k-steps are one
  // 16-byte read, and the x operands of a (ci group, ky) row for all three kx and twelve k-steps are the 14 floats
  // 12 lq - 1 .. 12 lq + 12 = three 16-byte reads + two edge floats, kept in registers and addressed by compile-time
  // element indices: 1.5-2 reads per k-step instead of 7-10.  Same products, another summation order.
  // 16-byte aligned channel strides (4 x odd: the 16 lanes of a quarter hit 16 x 4 distinct banks) where two tile
  // buffers still fit; else the old strides with 8-byte aligned 16-byte reads (ds_read2_b64).
  static constexpr bool RUNS = !WG_DIAG && !SPREAD;
  static constexpr bool ALIGNED = RUNS && 2 * (size_t)(COUT * 148 + CIN * 284 + 8) * sizeof(float) <= 160 * 1024;
  static constexpr int PSD = ALIGNED ? 148 : C::PSD;
  static constexpr int PSX = ALIGNED ? 284 : C::PSX;
  static constexpr int DY_FLOATS = COUT * PSD;
  static constexpr int X_FLOATS = CIN * PSX + 8;
  static constexpr int BUF_FLOATS = DY_FLOATS + X_FLOATS;
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_FLOATS * sizeof(float);
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024 && NSLOT + LAG <= 36 && MIN_GAPS >= kWgMinGaps;
};

// Per-thread, tile-invariant description of staging slot I: element offset of its 16 bytes
// relative to the image base at tile origin (0, 0); its float index inside an LDS tile buffer
// packed with its row / column inside the tile.
template <int COUT, int CIN>
struct PipeGeom {
  int goff[WgPipe<COUT, CIN>::NSLOT];
  int pos[WgPipe<COUT, CIN>::NSLOT];   // LDS float index | r << 16 | 4q << 20 | unused << 30
};

template <int COUT, int CIN, int I>
__device__ __forceinline__ void pipe_geom_slot(PipeGeom<COUT, CIN>& g, int tid, int plane, int W) {
  using C = WgCfg<COUT, CIN>;
  if constexpr (I < C::DY_ITERS) {
    int s = tid + I * 256;
    const bool live = s < C::DY_SLOTS;
    s = live ? s : 0;
    const int co = s / (kTileRows * 12);
    const int rem = s - co * (kTileRows * 12);
    const int r = rem / 12;
    const int q = rem - r * 12;
    g.goff[I] = co * plane + r * W + 4 * q;
    g.pos[I] = (co * WgPipe<COUT, CIN>::PSD + r * kTileCols + 4 * q) | (r << 16) | ((4 * q) << 20) | (live ? 0 : (1 << 30));
  } else {
    int s = tid + (I - C::DY_ITERS) * 256;
    const bool live = s < C::X_SLOTS;
    s = live ? s : 0;
    const int ci = s / (kHaloRows * 14);
    const int rem = s - ci * (kHaloRows * 14);
    const int r = rem / 14;
    const int q = rem - r * 14;
    g.goff[I] = ci * plane + (r - 1) * W + 4 * q - 4;
    g.pos[I] = (WgPipe<COUT, CIN>::DY_FLOATS + ci * WgPipe<COUT, CIN>::PSX + r * kRS + 4 * q) | (r << 16) | ((4 * q) << 20) |
               (live ? 0 : (1 << 30));
  }
}

template <int COUT, int CIN, int I>
__device__ __forceinline__ void pipe_geom_all(PipeGeom<COUT, CIN>& g, int tid, int plane, int W) {
  if constexpr (I < WgPipe<COUT, CIN>::NSLOT) {
    pipe_geom_slot<COUT, CIN, I>(g, tid, plane, W);
    pipe_geom_all<COUT, CIN, I + 1>(g, tid, plane, W);
  }
}

struct PipeTile {   // wave-uniform: the tile being staged
  const float* dyimg;   // image n of dy
  const float* ximg;    // image n of x
  int y0, x0, org;      // org = y0 * W + x0
};

// Staging slot I, first half: row / column of the slot in the image.
template <int COUT, int CIN, int I>
__device__ __forceinline__ void pipe_addr_a(const PipeGeom<COUT, CIN>& g, const PipeTile& t, int& gy, int& gx) {
  using C = WgCfg<COUT, CIN>;
  constexpr int dy0 = (I < C::DY_ITERS) ? 0 : -1, dx0 = (I < C::DY_ITERS) ? 0 : -4;
  gy = t.y0 + dy0 + ((g.pos[I] >> 16) & 0xf);
  gx = t.x0 + dx0 + (g.pos[I] >> 20);   // an unused slot gets a huge column: out of range below
}

// Second half: the address (zero page when the slot lies outside the image) -- as an integer,
// so that the select is two v_cndmask and not a branch.
template <int COUT, int CIN, int I>
__device__ __forceinline__ uint64_t pipe_addr_b(const PipeGeom<COUT, CIN>& g, const PipeTile& t,
                                                const WgradBatch& b, int gy, int gx) {
  using C = WgCfg<COUT, CIN>;
  const bool ok = (unsigned)gy < (unsigned)b.H && (unsigned)gx < (unsigned)b.W;
  const float* img = (I < C::DY_ITERS) ? t.dyimg : t.ximg;
  const uint64_t real = reinterpret_cast<uint64_t>(img + (g.goff[I] + t.org));
  const uint64_t zero = reinterpret_cast<uint64_t>(&g_wg_zero_page[0]);
  return ok ? real : zero;
}

template <int COUT, int CIN, int I>
__device__ __forceinline__ void pipe_lds_write(const PipeGeom<COUT, CIN>& g, float* buf, f32x4 v) {
  using C = WgCfg<COUT, CIN>;
  constexpr int slots = (I < C::DY_ITERS) ? C::DY_SLOTS : C::X_SLOTS;
  constexpr int i = (I < C::DY_ITERS) ? I : I - C::DY_ITERS;
  auto put = [&]() {
    if constexpr (WgPipe<COUT, CIN>::ALIGNED) *reinterpret_cast<f32x4*>(buf + (g.pos[I] & 0xffff)) = v;   // one ds_write_b128
    else lds_store4(buf + (g.pos[I] & 0xffff), v);
  };
  if constexpr (i * 256 + 255 < slots) {
    put();
  } else {  // only some threads own a slot in the last round
    if (!(g.pos[I] >> 30)) put();
  }
}

// Operand runs of one wave (WgPipe::RUNS): the (ci group, ky) rows its (ci group, tap) operands B0 .. B0 + NBW - 1 touch
// are groups GF .. GF + NG - 1 (group = ci group * 3 + ky).  Per tile row (double-buffered by row parity) and group: the 14
// floats at run positions -1 .. 12 -- eL, three 16-byte windows, eR; per quad of k-steps (double-buffered by quad
// parity) and output-channel group: the four dy operands.
template <int B0, int NBW>
struct RunGroups {
  static constexpr int GF = B0 / 3;
  static constexpr int NG = (B0 + NBW - 1) / 3 - GF + 1;
};

template <int CT, int NG>
struct RunOps {
  f32x4 aq[2][CT];
  float eL[2][NG];
  f32x4 w[2][NG][3];
  float eR[2][NG];
};

// Reads of quad Q (k-steps 4 Q .. 4 Q + 3; tile row Q / 3, third t = Q % 3 of the runs) in issue order: the CT dy
// quads of quad Q + 1, then per group -- t = 0: window 2 of this row; t = 1: eR of this row; t = 2: eL and windows
// 0, 1 of the NEXT row (what its first quad needs).  They are dealt over the quad's first three k-steps, so that the
// youngest has a k-step's MFMAs to land under.
template <int CT, int NG>
struct RunPlan {
  static constexpr int n(int Q) { return (Q < 8 ? CT : 0) + (Q % 3 < 2 ? NG : (Q / 3 < 2 ? 3 * NG : 0)); }
  static constexpr int first(int Q, int e) {   // the first read of quad Q issued in its k-step e or later
    int i = 0;
    while (i < n(Q) && i * 3 / n(Q) < e) ++i;
    return i;
  }
  static constexpr int count(int Q, int e) { return e >= 3 ? 0 : first(Q, e + 1) - first(Q, e); }
};

template <int COUT, int CIN, int NBW, int NG>
struct PipeCtx {
  const WgradBatch& b;
  const PipeGeom<COUT, CIN>& g;
  PipeTile next;
  const float* a_base;
  const float* b_base;
  float* nxt;
  f32x4 (&stage)[WgPipe<COUT, CIN>::NSLOT];
  float (&av)[2][COUT / 16];
  float (&bv)[2][NBW];
  f32x4 (&acc)[COUT / 16][NBW];
  f32x4 (&bacc)[COUT / 16];   // BIAS wave: rows = co, every column = sum over pixels of dy
  float one;                   // 1.0f in a VGPR (B operand of the bias MFMAs in the asm form)
  RunOps<COUT / 16, NG>& ro;   // WgPipe::RUNS: the operand runs and this lane's run bases in the current buffer
  const float* ra;
  const float* rb;
};

// 16 bytes of LDS at a float index that is a multiple of 4 (ALIGNED strides: one ds_read_b128) or of 2 (ds_read2_b64)
template <bool ALIGNED>
__device__ __forceinline__ f32x4 lds_load4(const float* p) {
  if constexpr (ALIGNED) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
    return *reinterpret_cast<const f32x4a8*>(p);
  }
}

// Read I of quad Q (RunPlan's order).  ra = this lane's dy run base (buffer + lr * PSD + 12 lq), rb = its x run base
// (x part + lr * PSX + 12 lq + 4: run position 0 = the pixel under the tap centre).
template <int COUT, int CIN, int B0, int NBW, int Q, int I>
__device__ __forceinline__ void run_read(RunOps<COUT / 16, RunGroups<B0, NBW>::NG>& o, const float* ra, const float* rb) {
  using P = WgPipe<COUT, CIN>;
  using G = RunGroups<B0, NBW>;
  constexpr int CT = COUT / 16, t = Q % 3, row = Q / 3, NA = Q < 8 ? CT : 0;
  if constexpr (I < NA) {
    constexpr int q1 = Q + 1;
    o.aq[q1 & 1][I] = lds_load4<P::ALIGNED>(ra + I * 16 * P::PSD + (q1 / 3) * kTileCols + 4 * (q1 % 3));
  } else {
    constexpr int J = I - NA;
    constexpr int gi = t < 2 ? J : J / 3, gg = G::GF + gi, cit = gg / 3, ky = gg % 3;
    if constexpr (t == 0) {
      o.w[row & 1][gi][2] = lds_load4<P::ALIGNED>(rb + cit * 16 * P::PSX + (row + ky) * kRS + 8);
    } else if constexpr (t == 1) {
      o.eR[row & 1][gi] = rb[cit * 16 * P::PSX + (row + ky) * kRS + 12];
    } else {
      constexpr int r2 = row + 1, what = J % 3;
      const float* p = rb + cit * 16 * P::PSX + (r2 + ky) * kRS;
      if constexpr (what == 0) o.eL[r2 & 1][gi] = p[-1];
      else o.w[r2 & 1][gi][what - 1] = lds_load4<P::ALIGNED>(p + 4 * (what - 1));
    }
  }
}

// What the first quad of a tile needs, read behind the tile barrier: the dy quads of quad 0, eL and windows 0, 1 of row 0.
template <int COUT, int CIN, int B0, int NBW>
__device__ __forceinline__ void run_prime(RunOps<COUT / 16, RunGroups<B0, NBW>::NG>& o, const float* ra, const float* rb) {
  using P = WgPipe<COUT, CIN>;
  using G = RunGroups<B0, NBW>;
#pragma unroll
  for (int c = 0; c < COUT / 16; ++c) o.aq[0][c] = lds_load4<P::ALIGNED>(ra + c * 16 * P::PSD);
#pragma unroll
  for (int gi = 0; gi < G::NG; ++gi) {
    const int gg = G::GF + gi, cit = gg / 3, ky = gg % 3;
    const float* p = rb + cit * 16 * P::PSX + ky * kRS;
    o.eL[0][gi] = p[-1];
    o.w[0][gi][0] = lds_load4<P::ALIGNED>(p);
    o.w[0][gi][1] = lds_load4<P::ALIGNED>(p + 4);
  }
}

// The operands of k-step KS: dy of output-channel group c / x of (ci group, tap) operand B0 + k.
template <int CT, int NG, int KS, int C_>
__device__ __forceinline__ float run_a(const RunOps<CT, NG>& o) { return o.aq[(KS / 4) & 1][C_][KS % 4]; }

template <int B0, int NBW, int CT, int KS, int K>
__device__ __forceinline__ float run_b(const RunOps<CT, RunGroups<B0, NBW>::NG>& o) {
  constexpr int bi = B0 + K, gi = bi / 3 - RunGroups<B0, NBW>::GF, kx = bi % 3, row = KS / 12, rel = KS % 12 + kx - 1;
  if constexpr (rel < 0) return o.eL[row & 1][gi];
  else if constexpr (rel >= 12) return o.eR[row & 1][gi];
  else return o.w[row & 1][gi][rel / 4][rel % 4];
}

// k-step KS of the pipelined tile loop: one scheduling region holding the step's MFMAs, the
// operand reads of step KS+1, the address arithmetic + global load of staging slot KS and the LDS
// write of slot KS-LAG; the sched_group_barrier sequence at the end tells the scheduler how to
// lay them out (one MFMA, then at most ~24 issue cycles of the rest, repeat).
// On the BIAS wave (the one with the fewest (ci group, tap) operands) db rides on the matrix pipe
// too: CT extra MFMAs per k-step against an all-ones B operand, so that no wave issues VALU adds
// and all four carry the same number of MFMAs.
template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV, int KS>
__device__ __forceinline__ void pipe_kstep_builtin(PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG>& x) {
  using C = WgCfg<COUT, CIN>;
  using P = WgPipe<COUT, CIN>;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KS + 1 < 36 && !(WG_DIAG & 4))
    wg_read<COUT, CIN, B0, NBW>(x.a_base, x.b_base, KS + 1, x.av[(KS + 1) & 1], x.bv[(KS + 1) & 1]);
  if constexpr (KS < P::NSLOT && !(WG_DIAG & 2)) {
    int gy, gx;
    uint64_t addr;
    if constexpr (WG_DIAG & 32) {
      addr = reinterpret_cast<uint64_t>(&g_wg_zero_page[0]);
    } else {
      pipe_addr_a<COUT, CIN, KS>(x.g, x.next, gy, gx);
      addr = pipe_addr_b<COUT, CIN, KS>(x.g, x.next, x.b, gy, gx);
    }
    if constexpr (WG_DIAG & 64) x.stage[KS] = f32x4{(float)(addr & 7), 0.f, 0.f, 0.f};
    else x.stage[KS] = *reinterpret_cast<const f32x4*>(addr);
  }
  constexpr bool kWrites = KS >= P::LAG && KS - P::LAG < P::NSLOT;
  if constexpr (kWrites && !(WG_DIAG & 2) && !(WG_DIAG & 16))
    pipe_lds_write<COUT, CIN, KS - P::LAG>(x.g, x.nxt, x.stage[KS - P::LAG]);
  if constexpr (kWrites && (WG_DIAG & 128)) asm volatile("" ::"v"(x.stage[KS - P::LAG]));  // wait, no write
#pragma unroll
  for (int c = 0; c < C::CT; ++c)
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
      if constexpr (!(WG_DIAG & 1))
        x.acc[c][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.av[KS & 1][c], x.bv[KS & 1][k], x.acc[c][k], 0, 0, 0);
      else
        x.acc[c][k][0] += x.av[KS & 1][c] * x.bv[KS & 1][k];
    }
  if constexpr (BIAS && !(WG_DIAG & 8)) {
#pragma unroll
    for (int c = 0; c < C::CT; ++c)
      x.bacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.av[KS & 1][c], 1.0f, x.bacc[c], 0, 0, 0);
  }
  // masks: 0x8 MFMA, 0x2 VALU, 0x20 VMEM read, 0x100 DS read, 0x200 DS write
  constexpr int NMF = C::CT * NBW + (BIAS ? C::CT : 0);
#pragma unroll
  for (int m = 0; m < NMF; ++m) {
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
    // same-box A/B of four layouts (reads doubled up, write first, no stagger) and of LAG 3..15:
    // all within 1.5 % of each other, 12.5 us per tile against 11.1 us for the MFMAs alone
    if (m < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    else if (m < 12) __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
    else if (m == 12 + WV) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
    else if (m == 16 + WV) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// The same k-step with the MFMAs as inline asm, accumulator tied to one AGPR quad ("+a").  With the
// builtin the register allocator carries the loop's accumulators partly in VGPRs across the tile
// loop's edge and shuttles ~190 of them per tile through v_accvgpr_read / v_accvgpr_write (the
// MFMAs then read one AGPR quad and write another); tied operands leave it no such choice.  Asm
// statements keep their order, so the layout is spelled out gap by gap instead of through
// sched_group_barrier: gap m = MFMA m + its filler, pinned by a sched_barrier.
__device__ __forceinline__ void mfma_tied(f32x4& acc, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// Filler event E of a k-step -> the MFMA gap it is issued in.  Events in program order: E < G0 = CT + NBW: the
// operand reads of the next k-step; G0: address step a; G0 + 1: address step b; G0 + 2: the global load of staging
// slot KS; G0 + 3: the LDS write of slot KS - LAG.  SPREAD (>= 17 gaps on every wave): one event per gap with the
// load / write staggered over the waves, as measured best for (48,48); else dealt evenly over the NG gaps.
template <int G0, int NG, bool SPREAD, int WV>
__host__ __device__ constexpr int gap_of(int e) {
  if (SPREAD) return e < G0 + 2 ? e : (e == G0 + 2 ? G0 + 2 + (WV & 1) : G0 + 4 + WV);
  return e * NG / (G0 + 4);
}

template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV, int KS, int M, int E>
__device__ __forceinline__ void pipe_gap_events(PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG>& x, int& gy, int& gx, uint64_t& addr) {
  using C = WgCfg<COUT, CIN>;
  using P = WgPipe<COUT, CIN>;
  constexpr int G0 = C::CT + NBW;
  constexpr int NG = C::CT * NBW + (BIAS ? C::CT : 0);
  if constexpr (E < G0 + 4) {
    if constexpr (gap_of<G0, NG, P::SPREAD, WV>(E) == M) {
      if constexpr (E < G0) {
        if constexpr (KS + 1 < 36) {
          constexpr int ks = KS + 1, row = ks / 12, col = 4 * (ks % 12);
          if constexpr (E < C::CT) {
            x.av[ks & 1][E] = x.a_base[E * 16 * C::PSD + row * kTileCols + col];
          } else {
            constexpr int bi = B0 + (E - C::CT), cit = bi / 9, tap = bi % 9, ky = tap / 3, kx = tap % 3;
            x.bv[ks & 1][E - C::CT] = x.b_base[cit * 16 * C::PSX + (row + ky) * kRS + col + kx];
          }
        }
      } else if constexpr (E == G0 + 3) {
        constexpr bool kWrites = KS >= P::LAG && KS - P::LAG < P::NSLOT;
        if constexpr (kWrites) pipe_lds_write<COUT, CIN, KS - P::LAG>(x.g, x.nxt, x.stage[KS - P::LAG]);
      } else if constexpr (KS < P::NSLOT) {
        if constexpr (E == G0) pipe_addr_a<COUT, CIN, KS>(x.g, x.next, gy, gx);
        if constexpr (E == G0 + 1) addr = pipe_addr_b<COUT, CIN, KS>(x.g, x.next, x.b, gy, gx);
        if constexpr (E == G0 + 2) x.stage[KS] = *reinterpret_cast<const f32x4*>(addr);
      }
    }
    pipe_gap_events<COUT, CIN, B0, NBW, BIAS, WV, KS, M, E + 1>(x, gy, gx, addr);
  }
}

// The same walk with operand runs: the k-step's events are the reads RunPlan deals to it, then the four staging events.
template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV, int KS, int M, int E>
__device__ __forceinline__ void pipe_gap_events_runs(PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG>& x, int& gy, int& gx, uint64_t& addr) {
  using C = WgCfg<COUT, CIN>;
  using P = WgPipe<COUT, CIN>;
  using PL = RunPlan<C::CT, RunGroups<B0, NBW>::NG>;
  constexpr int Q = KS / 4, e = KS % 4;
  constexpr int G0 = PL::count(Q, e);
  constexpr int NG = C::CT * NBW + (BIAS ? C::CT : 0);
  if constexpr (E < G0 + 4) {
    if constexpr (E * NG / (G0 + 4) == M) {
      if constexpr (E < G0) {
        run_read<COUT, CIN, B0, NBW, Q, PL::first(Q, e) + E>(x.ro, x.ra, x.rb);
      } else if constexpr (E == G0 + 3) {
        constexpr bool kWrites = KS >= P::LAG && KS - P::LAG < P::NSLOT;
        if constexpr (kWrites) pipe_lds_write<COUT, CIN, KS - P::LAG>(x.g, x.nxt, x.stage[KS - P::LAG]);
      } else if constexpr (KS < P::NSLOT) {
        if constexpr (E == G0) pipe_addr_a<COUT, CIN, KS>(x.g, x.next, gy, gx);
        if constexpr (E == G0 + 1) addr = pipe_addr_b<COUT, CIN, KS>(x.g, x.next, x.b, gy, gx);
        if constexpr (E == G0 + 2) x.stage[KS] = *reinterpret_cast<const f32x4*>(addr);
      }
    }
    pipe_gap_events_runs<COUT, CIN, B0, NBW, BIAS, WV, KS, M, E + 1>(x, gy, gx, addr);
  }
}

template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV, int KS, int M>
__device__ __forceinline__ void pipe_gap_asm(PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG>& x, int& gy, int& gx, uint64_t& addr) {
  using C = WgCfg<COUT, CIN>;
  using P = WgPipe<COUT, CIN>;
  constexpr int NMAIN = C::CT * NBW;
  if constexpr (P::RUNS) {
    constexpr int NGR = RunGroups<B0, NBW>::NG;
    if constexpr (M < NMAIN) {
      mfma_tied(x.acc[M / NBW][M % NBW], run_a<C::CT, NGR, KS, M / NBW>(x.ro), run_b<B0, NBW, C::CT, KS, M % NBW>(x.ro));
    } else {
      mfma_tied(x.bacc[M - NMAIN], run_a<C::CT, NGR, KS, M - NMAIN>(x.ro), x.one);
    }
    pipe_gap_events_runs<COUT, CIN, B0, NBW, BIAS, WV, KS, M, 0>(x, gy, gx, addr);
    __builtin_amdgcn_sched_barrier(0);
    return;
  }
  if constexpr (M < NMAIN) {
    mfma_tied(x.acc[M / NBW][M % NBW], x.av[KS & 1][M / NBW], x.bv[KS & 1][M % NBW]);
  } else {
    mfma_tied(x.bacc[M - NMAIN], x.av[KS & 1][M - NMAIN], x.one);
  }
  // this gap's fillers: operand reads of the next k-step, then the staging slot's address, load, LDS write
  if constexpr (P::SPREAD) {
    // one filler per gap, spelled out (the (48,48) layout as measured in round 1; same-box it is 0.7 % faster than the
    // same assignment produced by the generic walk below)
    constexpr int G0 = C::CT + NBW;
    if constexpr (KS + 1 < 36 && M < G0) {
      constexpr int ks = KS + 1, row = ks / 12, col = 4 * (ks % 12);
      if constexpr (M < C::CT) {
        x.av[ks & 1][M] = x.a_base[M * 16 * C::PSD + row * kTileCols + col];
      } else {
        constexpr int bi = B0 + (M - C::CT), cit = bi / 9, tap = bi % 9, ky = tap / 3, kx = tap % 3;
        x.bv[ks & 1][M - C::CT] = x.b_base[cit * 16 * C::PSX + (row + ky) * kRS + col + kx];
      }
    }
    if constexpr (KS < P::NSLOT) {
      if constexpr (M == G0) pipe_addr_a<COUT, CIN, KS>(x.g, x.next, gy, gx);
      if constexpr (M == G0 + 1) addr = pipe_addr_b<COUT, CIN, KS>(x.g, x.next, x.b, gy, gx);
      if constexpr (M == G0 + 2 + (WV & 1)) x.stage[KS] = *reinterpret_cast<const f32x4*>(addr);
    }
    constexpr bool kWrites = KS >= P::LAG && KS - P::LAG < P::NSLOT;
    if constexpr (kWrites && M == G0 + 4 + WV) pipe_lds_write<COUT, CIN, KS - P::LAG>(x.g, x.nxt, x.stage[KS - P::LAG]);
  } else {
    pipe_gap_events<COUT, CIN, B0, NBW, BIAS, WV, KS, M, 0>(x, gy, gx, addr);
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV, int KS, int M>
__device__ __forceinline__ void pipe_gaps_asm(PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG>& x, int& gy, int& gx, uint64_t& addr) {
  if constexpr (M < (COUT / 16) * NBW + (BIAS ? COUT / 16 : 0)) {
    pipe_gap_asm<COUT, CIN, B0, NBW, BIAS, WV, KS, M>(x, gy, gx, addr);
    pipe_gaps_asm<COUT, CIN, B0, NBW, BIAS, WV, KS, M + 1>(x, gy, gx, addr);
  }
}

template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV, int KS>
__device__ __forceinline__ void pipe_kstep(PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG>& x) {
  if constexpr (!WG_DIAG) {
    int gy = 0, gx = 0;
    uint64_t addr = 0;
    __builtin_amdgcn_sched_barrier(0);
    pipe_gaps_asm<COUT, CIN, B0, NBW, BIAS, WV, KS, 0>(x, gy, gx, addr);
  } else {
    pipe_kstep_builtin<COUT, CIN, B0, NBW, BIAS, WV, KS>(x);
  }
}

template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV, int KS>
__device__ __forceinline__ void pipe_ksteps(PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG>& x) {
  if constexpr (KS < 36) {
    pipe_kstep<COUT, CIN, B0, NBW, BIAS, WV, KS>(x);
    pipe_ksteps<COUT, CIN, B0, NBW, BIAS, WV, KS + 1>(x);
  }
}

template <int COUT, int CIN, int I>
__device__ __forceinline__ void pipe_first_tile(const WgradBatch& b, const PipeGeom<COUT, CIN>& g, const PipeTile& t,
                                                float* buf) {
  if constexpr (I < WgPipe<COUT, CIN>::NSLOT) {
    int gy, gx;
    pipe_addr_a<COUT, CIN, I>(g, t, gy, gx);
    const uint64_t addr = pipe_addr_b<COUT, CIN, I>(g, t, b, gy, gx);
    pipe_lds_write<COUT, CIN, I>(g, buf, *reinterpret_cast<const f32x4*>(addr));
    pipe_first_tile<COUT, CIN, I + 1>(b, g, t, buf);
  }
}

// One pass of the pipelined kernel over a layer: CIN of the x tensor's channels (x already points at the first of them,
// x_nstride = floats per image of x) against all of dy.
struct PipeSeg {
  const float* dy;
  const float* x;
  size_t x_nstride;
  float* part_w;   // this pass's (ci group, tap) operands of the partial image
  float* part_b;   // the image's bias sums; null: another pass writes them
};

__device__ __forceinline__ PipeSeg pipe_seg(const WgradBatch& b, const WgradJob& j, float* image, int cin, int nb_ct_floats,
                                            int h, int halves) {
  const size_t plane = (size_t)b.H * b.W;
  return PipeSeg{j.dy, j.x + (size_t)h * cin * plane, (size_t)cin * halves * plane, image + (size_t)h * nb_ct_floats,
                 h == 0 ? image + (size_t)halves * nb_ct_floats : nullptr};
}

// Tiles [t_begin, t_end) of one pass -> its part of the partial image.
template <int COUT, int CIN, int B0, int NBW, bool BIAS, int WV>
__device__ __forceinline__ void wg_role_pipe(const WgradBatch& b, const PipeSeg& j, float* smem, int t_begin,
                                             int t_end, int tid) {
  using C = WgCfg<COUT, CIN>;
  using P = WgPipe<COUT, CIN>;
  const int lane = tid & 63, lr = lane & 15, lq = lane >> 4;

  f32x4 acc[C::CT][NBW];
#pragma unroll
  for (int c = 0; c < C::CT; ++c)
#pragma unroll
    for (int k = 0; k < NBW; ++k) acc[c][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc[C::CT];
#pragma unroll
  for (int c = 0; c < C::CT; ++c) bacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int plane = b.H * b.W;

  PipeGeom<COUT, CIN> geom;
  pipe_geom_all<COUT, CIN, 0>(geom, tid, plane, b.W);

  int tx = __builtin_amdgcn_readfirstlane(t_begin % b.tiles_x);
  int ty = __builtin_amdgcn_readfirstlane((t_begin / b.tiles_x) % b.tiles_y);
  int n = __builtin_amdgcn_readfirstlane(t_begin / (b.tiles_x * b.tiles_y));
  auto tile_at = [&]() {
    PipeTile t;
    t.dyimg = j.dy + (size_t)n * COUT * plane;
    t.ximg = j.x + (size_t)n * j.x_nstride;
    t.y0 = ty * kTileRows;
    t.x0 = tx * kTileCols;
    t.org = t.y0 * b.W + t.x0;
    return t;
  };
  if (t_begin < t_end) pipe_first_tile<COUT, CIN, 0>(b, geom, tile_at(), smem);
  __syncthreads();

  f32x4 stage[P::NSLOT];
  float av[2][C::CT], bv[2][NBW];
  RunOps<C::CT, RunGroups<B0, NBW>::NG> ro;
  float one = 1.0f;
  asm volatile("" : "+v"(one));  // keep it in a register: the asm MFMA takes no literal
  int par = 0;
  for (int t = t_begin; t < t_end; ++t) {
    // the next tile (the last tile stages itself once more into the idle buffer: keeps the loop
    // free of branches, nobody reads that copy)
    if (t + 1 < t_end) {
      if (++tx == b.tiles_x) {
        tx = 0;
        if (++ty == b.tiles_y) {
          ty = 0;
          ++n;
        }
      }
    }
    float* cur = smem + par * P::BUF_FLOATS;
    PipeCtx<COUT, CIN, NBW, RunGroups<B0, NBW>::NG> x{b, geom, tile_at(), cur + lr * P::PSD + lq,
                                                       cur + P::DY_FLOATS + lr * P::PSX + lq + 3,
                                                       smem + (par ^ 1) * P::BUF_FLOATS, stage, av, bv, acc, bacc, one, ro,
                                                       cur + lr * P::PSD + 12 * lq, cur + P::DY_FLOATS + lr * P::PSX + 12 * lq + 4};
    if constexpr (P::RUNS) run_prime<COUT, CIN, B0, NBW>(ro, x.ra, x.rb);
    else wg_read<COUT, CIN, B0, NBW>(x.a_base, x.b_base, 0, av[0], bv[0]);
    __builtin_amdgcn_sched_barrier(0);
    pipe_ksteps<COUT, CIN, B0, NBW, BIAS, WV, 0>(x);
    // pin the accumulators to AGPRs across the loop edge: left alone, the allocator carries them
    // in VGPRs there and re-copies all of them (v_accvgpr_write x 96) at the top of every tile
#pragma unroll
    for (int c = 0; c < C::CT; ++c) {
#pragma unroll
      for (int k = 0; k < NBW; ++k) asm volatile("" : "+a"(acc[c][k]));
      if constexpr (BIAS) asm volatile("" : "+a"(bacc[c]));
    }
    __syncthreads();  // tile t fully read, tile t+1 fully written
    par ^= 1;
  }

  // (the compiler does not see the asm MFMAs: cover their write-back before the accumulators are read)
  if constexpr (!WG_DIAG) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int c = 0; c < C::CT; ++c)
#pragma unroll
    for (int k = 0; k < NBW; ++k)
      *reinterpret_cast<f32x4*>(j.part_w + (((B0 + k) * C::CT + c) * 64 + lane) * 4) = acc[c][k];
  if constexpr (BIAS) {  // column 0 of the bias accumulators: lane 16*lq holds co = 16c + 4lq + r
    if (lr == 0 && j.part_b) {
#pragma unroll
      for (int c = 0; c < C::CT; ++c) *reinterpret_cast<f32x4*>(j.part_b + c * 16 + 4 * lq) = bacc[c];
    }
  }
}

template <int COUT, int CIN>
__device__ __forceinline__ void wg_pipe_pass(const WgradBatch& b, const PipeSeg& j, float* smem, int t_begin, int t_end,
                                             int tid, int wave) {
  using C = WgCfg<COUT, CIN>;
  constexpr int NB = C::NB;
  constexpr int W0 = (NB + 3) / 4, W1 = (NB + 2) / 4, W2 = (NB + 1) / 4, W3 = NB / 4;
  // the bias sums ride on the last wave: it owns the fewest (ci group, tap) operands
  if (wave == 0) wg_role_pipe<COUT, CIN, 0, W0, false, 0>(b, j, smem, t_begin, t_end, tid);
  else if (wave == 1) wg_role_pipe<COUT, CIN, W0, W1, false, 1>(b, j, smem, t_begin, t_end, tid);
  else if (wave == 2) wg_role_pipe<COUT, CIN, W0 + W1, W2, false, 2>(b, j, smem, t_begin, t_end, tid);
  else wg_role_pipe<COUT, CIN, W0 + W1 + W2, W3, true, 3>(b, j, smem, t_begin, t_end, tid);
}

// Tiles [t_begin, t_end) of layer `j` -> the partial image at `image`: one pass per half of x's channels.
template <int COUT, int CIN>
__device__ __forceinline__ void wg_pipe_waves(const WgradBatch& b, const WgradJob& j, float* smem, int t_begin, int t_end,
                                              float* image, int tid, int wave) {
  using C = WgCfg<COUT, CIN>;
  const int halves = b.halves > 1 ? b.halves : 1;   // (uniform)
  for (int h = 0; h < halves; ++h) {
    if (h) __syncthreads();   // every wave is done with the tile buffers before the next pass restages them
    wg_pipe_pass<COUT, CIN>(b, pipe_seg(b, j, image, CIN, C::NB * C::CT * 256, h, halves), smem, t_begin, t_end, tid, wave);
  }
}

// floats of one partial image of a launch whose x tensors have halves * CIN channels
template <int COUT, int CIN>
__device__ __forceinline__ int pipe_image_floats(const WgradBatch& b) {
  return WgCfg<COUT, CIN>::NB * WgCfg<COUT, CIN>::CT * 256 * (b.halves > 1 ? b.halves : 1) + COUT;
}

template <int COUT, int CIN>
__global__ __launch_bounds__(256, 1) void wgrad3x3_pipe_kernel(WgradBatch b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WgradJob& j = b.job[blockIdx.y];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, splits = gridDim.x;
  const int total = b.N * b.tiles_x * b.tiles_y;
  // (64-bit divisions run on the vector ALU: bring the wave-uniform results back to SGPRs)
  const int t_begin = __builtin_amdgcn_readfirstlane((int)(((long long)total * split) / splits));
  const int t_end = __builtin_amdgcn_readfirstlane((int)(((long long)total * (split + 1)) / splits));
  wg_pipe_waves<COUT, CIN>(b, j, smem, t_begin, t_end, j.partial + (size_t)split * pipe_image_floats<COUT, CIN>(b), tid, wave);
}

// One grid over ALL layers of a launch: the njobs x (tiles per layer) tiles form one sequence, workgroup w
// owns [G w / nwg, G (w+1) / nwg) of it.  Where that share crosses a layer boundary the workgroup finishes
// the layer (writes its partial image), restarts its staging pipeline on the next layer and goes on: no
// CU idles at any layer count (40 layers on 256 CUs: 40 tiles each; as 32 x 8 + 8 x 32 workgroups the
// second launch ran at 12.8 us per layer against 11.6), and a step writes 256 + (layers - 1) partial
// images instead of 2 x 256.
template <int COUT, int CIN>
__global__ __launch_bounds__(256, 1) void wgrad3x3_pipe_flat_kernel(WgradBatch b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  WSTAMP(b, 0);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = blockIdx.x, nwg = gridDim.x;
  const int total = b.N * b.tiles_x * b.tiles_y;
  const long long T = (long long)total * b.njobs;       // tiles of the njobs layers; the head's pseudo-tiles follow
  const long long G = T + b.head_units;
  const int g0 = __builtin_amdgcn_readfirstlane((int)((G * w) / nwg));
  const int g1 = __builtin_amdgcn_readfirstlane((int)((G * (w + 1)) / nwg));
  int g = g0;
  const int g_end = min(g1, (int)T);
  int jb = __builtin_amdgcn_readfirstlane(g / total);
  while (g < g_end) {   // one pass per layer this workgroup touches (bounded: at most njobs)
    const int base = jb * total;
    const int seg_end = min(g_end, base + total);
    const WgradJob& j = b.job[jb];
    float* part = j.partial + (size_t)(w - b.first_wg[jb]) * pipe_image_floats<COUT, CIN>(b);
    wg_pipe_waves<COUT, CIN>(b, j, smem, g - base, seg_end - base, part, tid, wave);
    __syncthreads();    // every wave is done with the tile buffers before the next layer restages them
    g = seg_end;
    ++jb;
  }
  if (b.head_units > 0 && g1 > (int)T) {   // (wave-uniform: whole workgroup)
    using H = WgCfg<COUT, 16>;
    const long long p0 = max(g0, (int)T) - T, p1 = g1 - T;
    const int h_begin = __builtin_amdgcn_readfirstlane((int)(p0 * total / b.head_units));
    const int h_end = __builtin_amdgcn_readfirstlane((int)(p1 * total / b.head_units));
    float* part = b.head.partial + (size_t)(w - b.head_first_wg) * H::PARTIAL_FLOATS;
    if constexpr (WgPipe<COUT, 16>::FITS) {
      // (round 4) the pipelined role at 16 input channels: its two tile buffers lie inside this launch's LDS
      static_assert(WgPipe<COUT, 16>::LDS_BYTES <= WgPipe<COUT, CIN>::LDS_BYTES, "the head's tile buffers fit the launch's LDS");
      const size_t plane = (size_t)b.H * b.W;
      const PipeSeg seg{b.head.dy, b.head.x, 16 * plane, part, part + H::NB * H::CT * 256};
      wg_pipe_pass<COUT, 16>(b, seg, smem, h_begin, h_end, tid, wave);
    } else {
      // (ci group, tap) operands of the 16-channel job dealt to the 4 waves: NB = 9 -> 3, 2, 2, 2
      constexpr int NB = H::NB;
      constexpr int W0 = (NB + 3) / 4, W1 = (NB + 2) / 4, W2 = (NB + 1) / 4, W3 = NB / 4;
      if (wave == 0) wg_role_range<COUT, 16, true, 0, W0>(b, b.head, smem, h_begin, h_end, part, tid, true);
      else if (wave == 1) wg_role_range<COUT, 16, true, W0, W1>(b, b.head, smem, h_begin, h_end, part, tid, false);
      else if (wave == 2) wg_role_range<COUT, 16, true, W0 + W1, W2>(b, b.head, smem, h_begin, h_end, part, tid, false);
      else wg_role_range<COUT, 16, true, W0 + W1 + W2, W3>(b, b.head, smem, h_begin, h_end, part, tid, false);
    }
  }
#if defined(LARVA_DIAG) && (LARVA_DIAG & 512)
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP(b, 1);
#endif
}

// Workgroups per CU of the register-staged kernel.  Its tile loop is "stage (global loads -> LDS, two barriers), then
// the MFMA block": alone on a CU the matrix pipe idles during every staging phase (58-60 % of peak).  The small shapes
// -- (32,32), the (C,16) heads -- need 54 / 46 KB of LDS and ~130 registers, so TWO workgroups fit a CU and run out of
// phase (the caller then launches twice as many splits, larva_wgrad_cu_share); the larger shapes do not fit twice.
template <int COUT, int CIN>
constexpr int kWgradPerCu = (COUT * CIN <= 32 * 32 && 2 * WgCfg<COUT, CIN>::LDS_BYTES <= 160 * 1024) ? 2 : 1;

template <int COUT, int CIN, bool VEC>
__global__ __launch_bounds__(256, (kWgradPerCu<COUT, CIN>)) void wgrad3x3_kernel(WgradBatch b) {
  using C = WgCfg<COUT, CIN>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const WgradJob& j = b.job[blockIdx.y];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, splits = gridDim.x;
  // (ci group, tap) operands dealt to the 4 waves: NB = 9 -> 3,2,2,2; 18 -> 5,5,4,4;
  // 27 -> 7,7,7,6; 36 -> 9,9,9,9.
  constexpr int NB = C::NB;
  constexpr int W0 = (NB + 3) / 4, W1 = (NB + 2) / 4, W2 = (NB + 1) / 4, W3 = NB / 4;
  if (wave == 0) wg_role<COUT, CIN, VEC, 0, W0>(b, j, smem, split, splits, tid, true);
  else if (wave == 1) wg_role<COUT, CIN, VEC, W0, W1>(b, j, smem, split, splits, tid, false);
  else if (wave == 2) wg_role<COUT, CIN, VEC, W0 + W1, W2>(b, j, smem, split, splits, tid, false);
  else wg_role<COUT, CIN, VEC, W0 + W1 + W2, W3>(b, j, smem, split, splits, tid, false);
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
  int splits;            // partial images of this job
  int cout, cin;         // kernel shape of this job (cin = the padded channel count of x)
};

constexpr int kMaxReduceJobs = 64;

struct ReduceBatch {
  ReduceJob job[kMaxReduceJobs];
  // optional extra block (blockIdx.y == njobs): the step's loss from its terms (larva_loss.h) -- the value is not
  // needed before the step ends, so it rides on this launch instead of having one of its own
  int njobs;
  TermList loss_terms;
  float loss_divisor;
  float* loss_out;
};

// One block per (layer, 16 output channels, 16 input channels): threads 0..191 each sum three 16-byte columns of
// the [splits][pf] partial matrix (tap = 3 u + tid / 64; up to 24 independent loads in flight), the sums cross an LDS
// transpose from accumulator layout to [co][ci][tap], and all 256 threads write the block's 16 runs of 144
// contiguous floats of dw.  (Until round 2 every thread stored its 4 sums straight from accumulator layout:
// 4-byte stores 36 bytes apart, 16.7 MB of HBM writes for 2.65 MB of gradients.)
// A layer with MANY partial images (the 3 -> 48 head: one image per tile, 256 of them, for 1296 gradients) keeps the
// older form instead -- 16-byte columns stored straight from accumulator layout, 64 columns per block with the
// images dealt to the block's four waves: its (cout / 16) x (cin / 16) blocks would each pull megabytes through
// one CU (and did so in the column-per-thread form too: 7 blocks, 1 MB each, were the launch's long pole).
constexpr int kReduceRow = 16 * 9 + 1;   // LDS row of one output channel: [ci 16][tap 9] + 1 float of bank skew
constexpr int kReduceColumnwiseAbove = 16;   // partial images per layer above which the column-per-thread form runs

__device__ __forceinline__ void reduce_columnwise(const ReduceJob& j) {
  // 64 columns per block; wave w sums the images k = w, w + 4, ... (8 loads in flight), the four sums meet in LDS
  // in a fixed order
  __shared__ f32x4 meet[3][64];
  const int ct_n = j.cout / 16, nb = (j.cin / 16) * 9;
  const int n_w = nb * ct_n * 256;
  const int pf = n_w + j.cout;  // multiple of 4
  const int col = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = (blockIdx.x * 64 + col) * 4;
  if (blockIdx.x * 256 >= pf) return;   // (whole block)
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < pf) {
    const float* p = j.partial + i;
    int k = wave;
    for (; k + 28 < j.splits; k += 32) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(k + 4 * u) * pf);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < j.splits; k += 4) s += *reinterpret_cast<const f32x4*>(p + (size_t)k * pf);
  }
  if (wave) meet[wave - 1][col] = s;
  __syncthreads();
  if (wave || i >= pf) return;
  s = (s + meet[0][col]) + (meet[1][col] + meet[2][col]);
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

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceBatch rb) {
  if ((int)blockIdx.y == rb.njobs) {   // (only launched when there is a loss to finish)
    if (blockIdx.x == 0) loss_terms_block(rb.loss_terms, rb.loss_divisor, rb.loss_out);
    return;
  }
  __shared__ float tile[16 * kReduceRow];
  const ReduceJob& j = rb.job[blockIdx.y];
  if (j.splits > kReduceColumnwiseAbove) return reduce_columnwise(j);
  const int ct_n = j.cout / 16;
  const int c = blockIdx.x % ct_n, cit = blockIdx.x / ct_n;
  if (cit >= j.cin / 16) return;
  const int n_w = (j.cin / 16) * 9 * ct_n * 256;
  const int pf = n_w + j.cout;  // multiple of 4
  const int tid = threadIdx.x;
  if (tid < 192) {
    const int lane = tid & 63, tap0 = tid >> 6;
    f32x4 s[3];
    const float* p[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      p[u] = j.partial + ((size_t)((cit * 9 + tap0 + 3 * u) * ct_n + c) * 256 + lane * 4);
    }
    int k = 0;
    for (; k + 8 <= j.splits; k += 8) {   // the images are added in index order, whatever is in flight
      f32x4 v[3][8];
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[u][q] = *reinterpret_cast<const f32x4*>(p[u] + (size_t)(k + q) * pf);
#pragma unroll
      for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) s[u] += v[u][q];
    }
    for (; k < j.splits; ++k)
#pragma unroll
      for (int u = 0; u < 3; ++u) s[u] += *reinterpret_cast<const f32x4*>(p[u] + (size_t)k * pf);
    const int co = (lane >> 4) * 4, ci = lane & 15;   // accumulator layout: lane -> (4 output channels, input channel)
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(co + r) * kReduceRow + ci * 9 + tap0 + 3 * u] = s[u][r];
  } else if (cit == 0 && j.db && tid < 192 + 16) {   // the bias gradient: the last `cout` floats of every image
    const float* p = j.partial + n_w + c * 16 + (tid - 192);
    float sb = 0.f;
    for (int k = 0; k < j.splits; ++k) sb += p[(size_t)k * pf];
    j.db[c * 16 + (tid - 192)] = sb;
  }
  __syncthreads();
  const int ci_n = min(16, j.cin_valid - cit * 16);   // (the head: 3 of its 16 padded input channels exist)
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const int idx = e * 256 + tid, co = idx / 144, rem = idx - co * 144;
    if (rem < ci_n * 9)
      j.dw[((size_t)(c * 16 + co) * j.w_cin_total + j.cin_off + cit * 16) * 9 + rem] = tile[co * kReduceRow + rem];
  }
}

static bool wgrad_use_pipe() {
  static const bool on = [] {
    const char* e = getenv("LARVA_WGRAD_PIPE");
    return !(e && e[0] == '0');
  }();
  return on;
}

// How a (COUT, CIN) layer gets onto the pipelined kernel: 1 = as it is, 2 = as two passes over 32 input channels each
// ((64, 64) and the 64-filter legs' (48, 64): two tile buffers of the whole layer do not fit the LDS), 0 = not at all.
template <int COUT, int CIN>
constexpr int kPipeHalves = WgPipe<COUT, CIN>::FITS ? 1 : ((CIN == 64 && WgPipe<COUT, 32>::FITS) ? 2 : 0);

template <int COUT, int CIN>
static hipError_t launch_wgrad(const WgradBatch& b, int njobs, int splits, hipStream_t stream) {
  using C = WgCfg<COUT, CIN>;
  using P = WgPipe<COUT, CIN>;
  if constexpr (kPipeHalves<COUT, CIN> == 2) {
    if (b.vec_ok && wgrad_use_pipe()) {
      using P2 = WgPipe<COUT, CIN / 2>;
      static PerDeviceOnce lds_half;
      if (const hipError_t e = ensure_dynamic_lds(lds_half, reinterpret_cast<const void*>(wgrad3x3_pipe_kernel<COUT, CIN / 2>), P2::LDS_BYTES);
          e != hipSuccess)
        return e;
      WgradBatch b2 = b;
      b2.halves = 2;
      hipLaunchKernelGGL((wgrad3x3_pipe_kernel<COUT, CIN / 2>), dim3(splits, njobs), dim3(256), P2::LDS_BYTES, stream, b2);
      return hipGetLastError();
    }
  }
  static PerDeviceOnce lds_a, lds_b, lds_c;
  if (const hipError_t e = ensure_dynamic_lds(lds_a, reinterpret_cast<const void*>(wgrad3x3_kernel<COUT, CIN, true>), C::LDS_BYTES); e != hipSuccess)
    return e;
  if (const hipError_t e = ensure_dynamic_lds(lds_b, reinterpret_cast<const void*>(wgrad3x3_kernel<COUT, CIN, false>), C::LDS_BYTES); e != hipSuccess)
    return e;
  if constexpr (P::FITS) {
    if (const hipError_t e = ensure_dynamic_lds(lds_c, reinterpret_cast<const void*>(wgrad3x3_pipe_kernel<COUT, CIN>), P::LDS_BYTES); e != hipSuccess)
      return e;
  }
  if constexpr (P::FITS) {
    if (b.vec_ok && wgrad_use_pipe()) {
      hipLaunchKernelGGL((wgrad3x3_pipe_kernel<COUT, CIN>), dim3(splits, njobs), dim3(256), P::LDS_BYTES, stream, b);
      return hipGetLastError();
    }
  }
  if (b.vec_ok)
    hipLaunchKernelGGL((wgrad3x3_kernel<COUT, CIN, true>), dim3(splits, njobs), dim3(256), C::LDS_BYTES, stream, b);
  else
    hipLaunchKernelGGL((wgrad3x3_kernel<COUT, CIN, false>), dim3(splits, njobs), dim3(256), C::LDS_BYTES, stream, b);
  return hipGetLastError();
}

template <int COUT, int CIN>
static hipError_t launch_wgrad_flat(const WgradBatch& b, int nwg, hipStream_t stream) {
  using P = WgPipe<COUT, CIN>;
  static_assert(P::FITS, "the flat launch exists for the pipelined kernel only");
  static PerDeviceOnce lds_set;
  if (const hipError_t e = ensure_dynamic_lds(lds_set, reinterpret_cast<const void*>(wgrad3x3_pipe_flat_kernel<COUT, CIN>), P::LDS_BYTES);
      e != hipSuccess)
    return e;
  hipLaunchKernelGGL((wgrad3x3_pipe_flat_kernel<COUT, CIN>), dim3(nwg), dim3(256), P::LDS_BYTES, stream, b);
  return hipGetLastError();
}

}  // namespace larva

using namespace larva;

extern "C" {

// Phase 1 as ONE grid of `nwg` workgroups over all njobs (<= 64) layers (48 -> 48, 32 -> 32 or 64 -> 64 channels, W % 4 == 0,
// 16-byte aligned tensors; hipErrorNotSupported otherwise: use larva_conv3x3_wgrad_partial).  The layers'
// tiles form one sequence cut evenly over the workgroups; a workgroup whose share crosses a layer boundary
// contributes a partial image to both layers.  splits_out[i] = the number of partial images of layer i
// (what larva_wgrad_reduce needs); partial[i] must hold larva_wgrad_flat_max_splits(njobs, nwg, tiles)
// images.  Deterministic like the per-layer launch.
int larva_wgrad_flat_max_splits(int njobs, int nwg, int tiles_per_layer) {
  if (njobs < 1 || nwg < 1 || tiles_per_layer < 1) return 0;
  const long long G = (long long)njobs * tiles_per_layer;
  if (nwg > G) nwg = (int)G;
  const long long smallest = G / nwg;                                   // the smallest share (>= 1)
  return (int)((tiles_per_layer + smallest - 1) / smallest + 1);        // shares that can touch one layer
}

int larva_wgrad_flat_head_splits(int njobs, int nwg, int tiles_per_layer);

#if defined(LARVA_DIAG) && (LARVA_DIAG & 512)
static int g_wdiag_next = -1, g_wdiag_cap = 0;
static unsigned long long* g_wdiag_base = nullptr;
#endif
static int wgrad_flat_impl(const float* const* dy, const float* const* x, float* const* partial, int njobs,
                           const float* head_dy, const float* head_x16, float* head_partial, int nwg, int N, int cout,
                           int cin, int H, int W, int* splits_out, int* head_splits_out, void* stream) {
  if (njobs < 1 || njobs > kMaxJobs || nwg < 1 || nwg > 32767 || N <= 0 || H <= 0 || W <= 0 || !splits_out)
    return (int)hipErrorInvalidValue;
  if (!((cout == 48 && cin == 48) || (cout == 32 && cin == 32) || (cout == 64 && cin == 64)) || W % 4 || !wgrad_use_pipe())
    return (int)hipErrorNotSupported;
  const bool head = head_dy || head_x16 || head_partial;
  if (head && cout != 48) return (int)hipErrorNotSupported;   // (the head's tail role is priced for the 48-channel grid)
  if (head && (!head_dy || !head_x16 || !head_partial || !head_splits_out)) return (int)hipErrorInvalidValue;
  WgradBatch b{};
  for (int i = 0; i < njobs; ++i) {
    if (!dy[i] || !x[i] || !partial[i]) return (int)hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(dy[i]) | reinterpret_cast<uintptr_t>(x[i])) & 15) return (int)hipErrorNotSupported;
    b.job[i] = WgradJob{dy[i], x[i], partial[i]};
  }
  if (head && ((reinterpret_cast<uintptr_t>(head_dy) | reinterpret_cast<uintptr_t>(head_x16)) & 15))
    return (int)hipErrorNotSupported;
  b.N = N; b.H = H; b.W = W;
  b.tiles_x = (W + kTileCols - 1) / kTileCols;
  b.tiles_y = (H + kTileRows - 1) / kTileRows;
  b.vec_ok = 1;
  b.njobs = njobs;
  const long long total = (long long)N * b.tiles_x * b.tiles_y;
  const long long T = total * njobs;
  if (T >= (1ll << 30)) return (int)hipErrorInvalidValue;
  if (head) {
    b.head = WgradJob{head_dy, head_x16, head_partial};
    b.head_units = flat_head_units((int)total);
  }
  const long long G = T + b.head_units;
  if (nwg > G) nwg = (int)G;
  // the same integer arithmetic as the kernel: share of workgroup w = [G w / nwg, G (w+1) / nwg)
  int w = 0;
  for (int i = 0; i < njobs; ++i) {
    const long long lo = total * i, hi = total * (i + 1);
    while ((G * (w + 1)) / nwg <= lo) ++w;                    // first workgroup whose share ends beyond lo
    b.first_wg[i] = (short)w;
    int last = w;
    while (last + 1 < nwg && (G * (last + 1)) / nwg < hi) ++last;
    splits_out[i] = last - w + 1;
    if (splits_out[i] > larva_wgrad_flat_max_splits(njobs, nwg, (int)total)) return (int)hipErrorInvalidValue;
  }
  if (head) {
    while ((G * (w + 1)) / nwg <= T) ++w;                     // first workgroup whose share reaches beyond T
    b.head_first_wg = w;
    *head_splits_out = nwg - w;
    if (*head_splits_out > larva_wgrad_flat_head_splits(njobs, nwg, (int)total)) return (int)hipErrorInvalidValue;
  }
#if defined(LARVA_DIAG) && (LARVA_DIAG & 512)
  b.diag_area = (g_wdiag_base && g_wdiag_next >= 0 && g_wdiag_next < g_wdiag_cap) ? g_wdiag_base + (size_t)(g_wdiag_next++) * 256 * 4 : nullptr;
#endif
  if (cout == 32) return (int)launch_wgrad_flat<32, 32>(b, nwg, (hipStream_t)stream);
  if (cout == 64) {   // two (64, 32) passes per workgroup and layer
    b.halves = 2;
    return (int)launch_wgrad_flat<64, 32>(b, nwg, (hipStream_t)stream);
  }
  return (int)launch_wgrad_flat<48, 48>(b, nwg, (hipStream_t)stream);
}

#if defined(LARVA_DIAG) && (LARVA_DIAG & 512)
int larva_diag_set_wgrad_stamps(unsigned long long* buf) {
  g_wdiag_base = buf;
  return 0;
}
// flat weight-gradient launches built from now on take slots first, first + 1, ... < cap (first < 0: off); returns
// the slot the next launch would have taken
int larva_diag_arm_wgrad_slots(int first, int cap) {
  const int was = g_wdiag_next;
  g_wdiag_next = first;
  g_wdiag_cap = cap;
  return was;
}
#endif

int larva_conv3x3_wgrad_partial_flat(const float* const* dy, const float* const* x, float* const* partial,
                                     int njobs, int nwg, int N, int cout, int cin, int H, int W,
                                     int* splits_out, void* stream) {
  return wgrad_flat_impl(dy, x, partial, njobs, nullptr, nullptr, nullptr, nwg, N, cout, cin, H, W, splits_out, nullptr,
                         stream);
}

// The same grid with ONE (48, 16) layer behind the njobs (48, 48) layers -- LarvaHead's weight gradient
// (models/LarvaNet.py:227-233; head_x16 = its input zero-padded to 16 channels, [N][16][H][W]) -- on the
// register-staged role: the workgroups at the end of the sequence take its tiles instead of a launch of its own
// (256 workgroups x 1 tile, 256 partial images) running after this one.  head_partial must hold
// larva_wgrad_flat_head_splits(njobs, nwg, tiles) images of larva_wgrad_partial_floats(48, 16, 1) floats;
// *head_splits_out = the images written.
int larva_wgrad_flat_head_splits(int njobs, int nwg, int tiles_per_layer) {
  if (njobs < 1 || nwg < 1 || tiles_per_layer < 1) return 0;
  const long long T = (long long)njobs * tiles_per_layer, G = T + flat_head_units(tiles_per_layer);
  if (nwg > G) nwg = (int)G;
  int w = 0;
  while ((G * (w + 1)) / nwg <= T) ++w;
  return nwg - w;
}

int larva_conv3x3_wgrad_partial_flat_head(const float* const* dy, const float* const* x, float* const* partial,
                                          int njobs, const float* head_dy, const float* head_x16, float* head_partial,
                                          int nwg, int N, int H, int W, int* splits_out, int* head_splits_out,
                                          void* stream) {
  if (!head_dy || !head_x16 || !head_partial || !head_splits_out) return (int)hipErrorInvalidValue;
  return wgrad_flat_impl(dy, x, partial, njobs, head_dy, head_x16, head_partial, nwg, N, 48, 48, H, W, splits_out,
                         head_splits_out, stream);
}

// How many workgroups of the (cout, cin) weight-gradient kernel share a CU (1 or 2): a caller that wants the chip
// full launches 256 * this many workgroups in total.
int larva_wgrad_cu_share(int cout, int cin) {
  if (cout == 48 && cin == 48) return 1;   // (the pipelined kernel: 159 KB of LDS)
  if (cout == 32 && cin == 32) return (WgPipe<32, 32>::FITS && wgrad_use_pipe()) ? 1 : kWgradPerCu<32, 32>;
  // (the pipelined kernel, where a shape is on it, owns its CU: two tile buffers)
  if (cout == 48 && cin == 16) return (WgPipe<48, 16>::FITS && wgrad_use_pipe()) ? 1 : kWgradPerCu<48, 16>;
  if (cout == 32 && cin == 16) return (WgPipe<32, 16>::FITS && wgrad_use_pipe()) ? 1 : kWgradPerCu<32, 16>;
  if (cout == 64 && cin == 16) return (WgPipe<64, 16>::FITS && wgrad_use_pipe()) ? 1 : kWgradPerCu<64, 16>;
  return 1;
}

// Floats of partial-image workspace one job needs for `splits` workgroups.
long long larva_wgrad_partial_floats(int cout, int cin, int splits) {
  return (long long)splits * ((long long)(cin / 16) * 9 * (cout / 16) * 256 + cout);
}

// Phase 1: partial images.  njobs (<= 64) same-shape layers in one launch; job i reads dy[i]
// [N][cout][H][W] and x[i] [N][cin][H][W] and writes `splits` partial images to partial[i]
// (larva_wgrad_partial_floats(cout, cin, splits) floats).  Returns the number of partial images
// actually written per job in *splits_used (splits clamped to the number of tiles).
int larva_conv3x3_wgrad_partial(const float* const* dy, const float* const* x, float* const* partial,
                                int njobs, int splits, int N, int cout, int cin, int H, int W,
                                int* splits_used, void* stream) {
  if (njobs < 1 || njobs > kMaxJobs || splits < 1 || N <= 0 || H <= 0 || W <= 0)
    return (int)hipErrorInvalidValue;
  WgradBatch b{};
  bool aligned = (W % 4 == 0);
  for (int i = 0; i < njobs; ++i) {
    if (!dy[i] || !x[i] || !partial[i]) return (int)hipErrorInvalidValue;
    b.job[i] = WgradJob{dy[i], x[i], partial[i]};
    aligned = aligned && ((reinterpret_cast<uintptr_t>(dy[i]) & 15) == 0) &&
              ((reinterpret_cast<uintptr_t>(x[i]) & 15) == 0);
  }
  b.N = N; b.H = H; b.W = W;
  b.tiles_x = (W + kTileCols - 1) / kTileCols;
  b.tiles_y = (H + kTileRows - 1) / kTileRows;
  b.vec_ok = aligned ? 1 : 0;
  const int total = N * b.tiles_x * b.tiles_y;
  if (splits > total) splits = total;
  if (splits_used) *splits_used = splits;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (cout == 48 && cin == 48) e = launch_wgrad<48, 48>(b, njobs, splits, s);
  else if (cout == 48 && cin == 16) e = launch_wgrad<48, 16>(b, njobs, splits, s);
  else if (cout == 32 && cin == 32) e = launch_wgrad<32, 32>(b, njobs, splits, s);
  else if (cout == 64 && cin == 64) e = launch_wgrad<64, 64>(b, njobs, splits, s);
  // --num_filters = 32 / 64 networks: the head (on its 16-channel padded input) and the legs' last conv (48 outputs)
  else if (cout == 32 && cin == 16) e = launch_wgrad<32, 16>(b, njobs, splits, s);
  else if (cout == 64 && cin == 16) e = launch_wgrad<64, 16>(b, njobs, splits, s);
  else if (cout == 48 && cin == 32) e = launch_wgrad<48, 32>(b, njobs, splits, s);
  else if (cout == 48 && cin == 64) e = launch_wgrad<48, 64>(b, njobs, splits, s);
  else return (int)hipErrorInvalidValue;
  return (int)e;
}

// Phase 2: fixed-order sum of the partial images of njobs (<= 64) layers -- possibly produced by
// several phase-1 launches of different kernel shapes, each layer with its own split count and
// (cout, cin) -- into PyTorch-layout
// gradients: dw[i] [cout][w_cin_total[i]][3][3] at input-channel offset cin_off[i] (the first
// cin_valid[i] channels are written; the rest of `cin` is zero padding of x), db[i] [cout] (may
// be null).  Gradients are OVERWRITTEN, not accumulated.  One launch: a training step reduces
// all of its layers at the end of backward instead of once per module.
static int reduce_launch(const float* const* partial, float* const* dw, float* const* db, const int* cin_off,
                         const int* cin_valid, const int* w_cin_total, const int* splits, const int* cout, const int* cin,
                         int njobs, const TermList* loss, float divisor, float* loss_out, void* stream) {
  if (njobs < 1 || njobs > kMaxReduceJobs) return (int)hipErrorInvalidValue;
  ReduceBatch rb{};
  int pf_max = 0;
  for (int i = 0; i < njobs; ++i) {
    if (!partial[i] || !dw[i] || splits[i] < 1 || cout[i] <= 0 || cin[i] <= 0 || cout[i] % 16 || cin[i] % 16)
      return (int)hipErrorInvalidValue;
    rb.job[i] = ReduceJob{partial[i], dw[i], db ? db[i] : nullptr, cin_off[i], cin_valid[i], w_cin_total[i],
                          splits[i], cout[i], cin[i]};
    const int pf = splits[i] > kReduceColumnwiseAbove
                       ? (((cin[i] / 16) * 9 * (cout[i] / 16) * 256 + cout[i]) / 4 + 63) / 64
                       : (cout[i] / 16) * (cin[i] / 16);   // blocks of this layer
    pf_max = pf > pf_max ? pf : pf_max;
  }
  rb.njobs = njobs;
  if (loss) {
    rb.loss_terms = *loss;
    rb.loss_divisor = divisor;
    rb.loss_out = loss_out;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(pf_max, njobs + (loss ? 1 : 0)), dim3(256), 0,
                     (hipStream_t)stream, rb);
  return (int)hipGetLastError();
}

int larva_wgrad_reduce(const float* const* partial, float* const* dw, float* const* db,
                       const int* cin_off, const int* cin_valid, const int* w_cin_total,
                       const int* splits, const int* cout, const int* cin, int njobs, void* stream) {
  return reduce_launch(partial, dw, db, cin_off, cin_valid, w_cin_total, splits, cout, cin, njobs, nullptr, 1.f, nullptr,
                       stream);
}

// larva_wgrad_reduce + larva_loss_from_partials (same arguments, same arithmetic) in one launch: the loss of a
// training step (models/LarvaNet.py:104-109) is not needed before the step ends, so finishing it rides on
// the last launch of backward.
int larva_wgrad_reduce_with_loss(const float* const* partial, float* const* dw, float* const* db,
                                 const int* cin_off, const int* cin_valid, const int* w_cin_total,
                                 const int* splits, const int* cout, const int* cin, int njobs,
                                 const float* const* terms, const int* count, const float* scale, int nterms,
                                 float divisor, float* loss_out, void* stream) {
  if (!terms || !count || !scale || nterms < 1 || nterms > 8 || !loss_out) return (int)hipErrorInvalidValue;
  TermList l{};
  for (int i = 0; i < nterms; ++i) {
    if (!terms[i] || count[i] < 1) return (int)hipErrorInvalidValue;
    l.p[i] = terms[i];
    l.count[i] = count[i];
    l.scale[i] = scale[i];
  }
  l.n = nterms;
  return reduce_launch(partial, dw, db, cin_off, cin_valid, w_cin_total, splits, cout, cin, njobs, &l, divisor, loss_out,
                       stream);
}

// Both phases for njobs (<= 64) layers.
int larva_conv3x3_wgrad(const float* const* dy, const float* const* x, float* const* partial,
                        float* const* dw, float* const* db, const int* cin_off,
                        const int* cin_valid, const int* w_cin_total, int njobs, int splits,
                        int N, int cout, int cin, int H, int W, void* stream) {
  int used = 0;
  int rc = larva_conv3x3_wgrad_partial(dy, x, partial, njobs, splits, N, cout, cin, H, W, &used, stream);
  if (rc) return rc;
  int sp[kMaxJobs], co[kMaxJobs], ci[kMaxJobs];
  for (int i = 0; i < njobs; ++i) { sp[i] = used; co[i] = cout; ci[i] = cin; }
  return larva_wgrad_reduce(partial, dw, db, cin_off, cin_valid, w_cin_total, sp, co, ci, njobs, stream);
}

}  // extern "C"
