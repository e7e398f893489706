// Fused 3x3 convolution (stride 1, zero pad 1, NCHW fp32) for gfx950 as an implicit GEMM on
// v_mfma_f32_16x16x4_f32, with the reference's elementwise neighbours fused into the epilogue.
//
// Replaces, per launch, these reference call sites (file:line into the reference tree):
//   conv+ReLU                      models/LarvaNet.py:210-211, 256-257
//   conv + torch.add(x, res)       models/LarvaNet.py:212, 217-220
//   last block conv + outer skip   models/LarvaNet.py:246-248   (two residual operands)
//   conv -> PixelShuffle(4) -> += base   models/LarvaNet.py:258, 261, 263-267   (mode 1)
//   torch.cat(features) + merge conv     models/LarvaNetV2.py:328-330  (several source tensors)
// and, with tap-mirrored / channel-transposed packed weights, the input-gradient (dgrad) of each
// of them, with the ReLU-backward mask and the skip-gradient adds fused the same way.
//
// GEMM view per workgroup (one 3 x 48 pixel tile):  D[cout][pixel] = sum_k W[cout][k] * im2col[k][pixel],
//   k = (chunk of 8 input channels) x (tap 0..8) x (2 k-steps of 4 channels).
//   MFMA A operand = weights (M = cout), B operand = activations (N = pixel), so each lane ends
//   up with 4 consecutive output channels of ONE pixel -- exactly the (i = lane>>4, j = reg)
//   sub-pixels of PixelShuffle(4), which makes the shuffle a single aligned 16-byte store.
//
// Staging: each K chunk (8-channel halo tile, 10 KiB, + its 9x8xCOUT weights) is one LDS stage.
//   VEC  (W % 4 == 0, 16-byte aligned tensors): LDS-DMA (global_load_lds_dwordx4) into a ring of
//        3 stages, two chunks in flight behind a counted vmcnt, one barrier per chunk, no staging
//        registers; 72 KiB of LDS and < 128 VGPRs per workgroup, so two workgroups (of this or
//        another kernel) share a CU and fill each other's prologue / epilogue bubbles.
//   !VEC (any W): the same stage image filled through registers (scalar loads), 2 stages.
//
// Roofline: fp32 MFMA (2*9*Cin*Cout FLOP per pixel; 41 472 at 48 channels) -- see DESIGN.md.
#include "larva_common.h"
#include "larva_bicubic.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>

// LARVA_DIAG (never defined in the product build; tools/diag_conv.py, diag_overlap.py, diag_step.py, diag_wide.py build
// their own libraries with it): timing-only ablations and in-kernel stamps.  bit 0 = skip the MFMA blocks, bit 1 = skip
// the global->LDS staging, bit 2 = skip the epilogue's global traffic, bit 3 = return at kernel entry, bit 4 = plain (not
// non-temporal) output stores, bits 5 / 6 / 9 / 11 = stamps (below), bit 7 = operands from registers instead of LDS,
// bit 8 = no chunk barriers (with bit 1).
// What earlier rounds measured through compile-time switches and settled (the losing sides are gone from this file; the
// numbers are in DESIGN_HISTORY.md and profiles/): MFMA waves issuing the LDS-DMA themselves (r1), epilogue operands
// fetched after the K loop / in front of chunk 0 / through LDS (r1, r3, r4: profiles/r03_ab_epilogue_operands_lds_v2.txt,
// r04_ab_aux_late.txt), operand reads in a block in front of each k-step (r1), channel-major accumulators in every
// epilogue (r2), prologue / epilogue at wave priority 3 (r3), the tile table through a scalar load (r3) or always
// from memory (r4), a loader wave that streams nothing past the last chunk (r4: r04_ab_loader_tail.txt), plain stores
// from the 48-column tiles (r4: r04_ab_wide_plain.txt), the exits' operands requested late (r4: r04_ab_aux_late_exits.txt),
// un-pinned epilogue scalars (r4), a ring of two stages with three workgroups per CU (r3), operands one or three
// k-steps ahead (r2, r3), ReLU-backward masks as sign bits (r4: r04_ab_maskbits_*.txt), eight MFMA waves per strip
// workgroup (r5: r05_ab_strip8.txt).
#ifndef LARVA_DIAG
#define LARVA_DIAG 0
#endif
constexpr int kWgPerCu = 2;          // workgroups per CU the 16-byte-path kernels are sized for (72 KiB of LDS, <= 128 VGPRs)
constexpr int kOperandDepth = 2;     // k-steps between an operand's LDS read and the MFMAs that use it


// LARVA_DIAG bit 5 (32): in-kernel timeline.  Wave 0 of every workgroup writes 100 MHz wall-clock
// stamps (s_memrealtime) at kernel entry, after the DMA issue, after the first chunk landed, after
// the K loop, after the stores were issued and after they drained, to the buffer given to
// larva_diag_set_stamps (16 x uint64 per workgroup; slots 8.. = exit of chunk c's barrier).  tools/diag_conv.py --timeline.
#ifndef LARVA_DIAG_ONLY48
#define LARVA_DIAG_ONLY48 0
#endif

namespace larva {

#if defined(LARVA_DIAG) && (LARVA_DIAG & 32)
__device__ unsigned long long* g_stamps = nullptr;
#if LARVA_DIAG & 512
// bit 9 (512, with bit 5): per-LAUNCH stamp areas for a whole graph of launches (tools/diag_overlap.py: which
// workgroups of the two half-batch chains are resident together?).  Every launch built while a slot counter is
// armed (larva_diag_arm_slots) takes the next area of kDiagWgPerSlot x 16 stamps: [0] kernel entry, [1] chunk 0
// landed, [2] K loop done, [3] stores drained (100 MHz wall clock), [4] HW_ID | XCC_ID << 32 of wave 0, and the
// prologue's steps: [5] arguments + tile decode done, [6] chunk 0's weight pieces issued, [7] bias / epilogue operands
// requested, [8] chunk 0's input pieces issued, [9] chunk 1 issued.
constexpr int kDiagWgPerSlot = 256;
__device__ int g_slot_of_launch = 0;   // (unused on the device: the slot travels in ConvArgs)
// (round 4: the launch carries the ADDRESS of its stamp area in its arguments -- null when no slot is armed.  Until
// then every stamp() first loaded g_stamps from memory, armed or not: ten dependent scalar round trips per workgroup,
// which is what made this build ~1 us per layer slower than the product even with the stamps disarmed.)
__device__ __forceinline__ void stamp_slot(unsigned long long* area, int k) {
  if (!area || threadIdx.x != 0 || blockIdx.x >= kDiagWgPerSlot) return;
  // (bit 11, 2048: also the prologue's steps -- five more stamps per workgroup cost ~1 us of its life)
  const int idx = k == 0 ? 0 : k == 2 ? 1 : k == 3 ? 2 : k == 5 ? 3
                  : !(LARVA_DIAG & 2048) ? -1 : k == 6 ? 5 : k == 7 ? 6 : k == 14 ? 7 : k == 15 ? 8 : k == 1 ? 9 : -1;
  if (idx < 0) return;
  unsigned long long* p = area + (size_t)blockIdx.x * 16;
  p[idx] = __builtin_amdgcn_s_memrealtime();
  if (k == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    p[4] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  }
}
#define stamp(k) stamp_slot(a.diag_area, (k))
#else
__device__ __forceinline__ void stamp(int k) {
  // bit 6 (64): stamps in SHADER-CLOCK cycles (s_memtime) instead of the 100 MHz wall clock: the two runs
  // together give the clock the kernel actually ran at (tools/diag_conv.py --timeline --clock)
  if (g_stamps && threadIdx.x == 0)
    g_stamps[blockIdx.x * 16 + k] = (LARVA_DIAG & 64) ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime();
}
#endif
#else
__device__ __forceinline__ void stamp(int) {}
#endif

constexpr int kCh = 8;  // input channels per K chunk

// Tile geometry of one workgroup: ROWS x COLS output pixels = ROWS * COLS/16 pixel groups of 16
// consecutive pixels of a row; the input halo tile is (ROWS+2) rows of COLS+2 columns, staged with
// the first wanted column (x0-1) at index 3 of a row of RS floats so that x0 sits on a 16-byte
// boundary, channel planes PS floats apart (the smallest stride >= HALO*RS that is == 16 mod 32:
// the two 16-lane halves of a ds_read_b32 group then land on disjoint banks).
//   GeoWide  3 x 48: 9 pixel groups -- one workgroup per CU at the training shape (16 x 48 x 48:
//            256 tiles), waves 7/7/7/6 of the 27 (cout group x pixel group) units;
//   GeoS5 / GeoS4  5 x 16 and 4 x 16 "strip" tiles: 5 and 4 pixel groups, waves 4/4/4/3 and
//            3/3/3/3 of 15 and 12 units -- the same 7 MFMAs per k-step per CU when one of each
//            shares a CU.  Half a batch (8 x 48 x 48) is then ALSO 256 workgroups, so the two halves
//            of a batch run as two independent layer chains on two streams, two workgroups per CU
//            out of phase: one chain's prologue / store burst / launch boundary hides under the other
//            chain's K loop (what a single dependent chain of 256-workgroup launches cannot do).
template <int ROWS_, int COLS_>
struct TileGeo {
  static constexpr int ROWS = ROWS_, COLS = COLS_;
  static constexpr int HALO = ROWS + 2;
  static constexpr int PC = COLS / 16;
  static constexpr int RS = COLS + 8;
  static constexpr int PS = ((HALO * RS + 15) / 32) * 32 + 16;
  static_assert(COLS % 16 == 0 && PS >= HALO * RS && PS % 32 == 16, "tile geometry");
};
using GeoWide = TileGeo<kTileRows, kTileCols>;
// 4 x 48 (round 4): 12 pixel groups, 36 (cout group x pixel group) units at 48 channels = 9 / 9 / 9 / 9 over the four
// waves, 24 = 6 / 6 / 6 / 6 at 32 channels.  The host picks the height per launch (conv_tile_rows: what was measured);
// results are bit-identical (an output pixel's K loop does not depend on its tile).
using GeoWide4 = TileGeo<4, kTileCols>;
using GeoS5 = TileGeo<5, 16>;
using GeoS4 = TileGeo<4, 16>;
static_assert(GeoWide::RS == kRS && GeoWide::PS == 304, "the 3 x 48 tile of larva_common.h");

struct ConvArgs {
  const float* src[kMaxSrc];  // channel-concatenated inputs, each [N][cin_per_src][H][W]
  const float* wpk;           // packed weights [n_chunks][9][8][CS]
  const float* bias;          // [COUT] or null
  const float* res0;          // [N][COUT][H][W]            (added first)
  const float* res1;          // [N][COUT][H][W]            (added second)
  const float* mask;          // [N][COUT][H][W]            (v = mask > 0 ? v : 0, before the adds)
  const float* base;          // pixel-shuffle store: [N][COUT/16][4H][4W]
  float* out;                 // [N][COUT][H][W], or [N][COUT/16][4H][4W] for the pixel-shuffle store
  int cin_per_src;            // multiple of 8
  int n_chunks;               // total input channels / 8
  int N, H, W;
  int pitch;                  // row stride (floats) of every [..][H][W] tensor above; >= W.  Columns
                              // [W, pitch) of the inputs hold zeros and are written as zeros.
  int tiles_x, tiles_y;
  int nwg;                    // N * tiles_x * tiles_y = workgroups of this conv (gridDim.x)
  // strip tiles (conv3x3_mfma_strip_kernel): tile_tab[t] = y0 | x0 << 12 | (5-row tile ? 1u << 31 : 0)
  // for the tiles_x entries of ONE image (tiles_x = tiles per image, tiles_y = 1)
  const unsigned* tile_tab;
  // kEpiShuffleL1 (an exit of the training step scored by nn.L1Loss in the same launch):
  const float* truth;         // [N][COUT/16][4H][4W], the image the exit is compared with
  float* grad;                // [N][COUT][H][pitch]: sign(out - truth) * gval, pixel-unshuffled
  float* partial;             // [4 * nwg]: sum |out - truth| of every MFMA wave's share of its tile
  int plain_stores;           // strip kernel, mode-0 output: plain instead of non-temporal stores (see the epilogue)
  // Strip kernel, round 4: ONE image's tile table inline, 16 bits per tile (y0 | (x0 / 16) << 8 | (5-row tile ? 1 << 15 : 0)),
  // when it fits (tab_n = its entry count <= 64, H <= 256, pitch <= 2048; else tab_n = 0 and the workgroup looks its
  // tile up in tile_tab).  It then arrives with the kernel's other arguments in the first batch of scalar loads and
  // the entry is picked out of SGPRs: the table look-up used to be a vector load whose address needs the arguments,
  // i.e. a second dependent memory round trip (~0.5 us) in front of every workgroup's first LDS-DMA request.
  int tab_n;
  alignas(64) int tab16[32];   // (two 64-byte lines of the kernarg segment: see fetch_args)
  float gval;                 // d loss / d out element = seed * (1/M) / numel
                              // (`out` may be null with this epilogue: the exit's image is not wanted)
  // ceil(2^40 / d) for d = tiles_x, tiles_y, cin_per_src / 8: the kernel's wave-uniform divisions
  // as one 64-bit multiply + shift (a runtime integer division is ~40 vector instructions, three of
  // them stood at the very start of every workgroup); exact while dividend * divisor < 2^40
  unsigned long long magic_tx, magic_ty, magic_cps;
#if LARVA_DIAG & 512
  unsigned long long* diag_area;   // stamp area of this launch (null: none)
#endif
};

__host__ __device__ __forceinline__ unsigned long long div_magic(int d) { return ((1ull << 40) + d - 1) / (unsigned)d; }
__device__ __forceinline__ int div_by_magic(int n, unsigned long long magic) {
  return (int)(((unsigned long long)(unsigned)n * magic) >> 40);
}

// Epilogue variants (compile-time so that every residual/mask load of a wave is in flight at
// once instead of one dependent L2 round trip per runtime branch).
enum Epi : int {
  kEpiPlain = 0,        // out = acc + bias
  kEpiRelu = 1,         // out = max(acc + bias, 0)
  kEpiMask = 2,         // out = mask > 0 ? acc + bias : 0            (ReLU backward)
  kEpiRes1 = 3,         // out = (acc + bias) + res0
  kEpiRes2 = 4,         // out = ((acc + bias) + res0) + res1
  kEpiShuffle = 5,      // pixel-shuffle(4) store
  kEpiShuffleBase = 6,  // pixel-shuffle(4) store, + base
  kEpiShuffleL1 = 7,    // + base, then L1 against `truth`: partial sums of |out - truth| and the sign
                        // gradient in pixel-unshuffled layout from the registers (image store optional)
  kEpiCount = 8
};

template <int COUT, typename G = GeoWide>
struct ConvCfg {
  static constexpr int CT = COUT / 16;
  static constexpr int CS = cout_stride(COUT);
  static constexpr int PS = G::PS;  // plane stride of a staged channel
  // One stage = [8 channel planes, padded to whole KiB][9*8 weight rows, padded to whole KiB]; every
  // 1 KiB "piece" is what one wave-level LDS-DMA instruction writes.  (3 x 48 tile: 10 + 14 pieces.)
  static constexpr int IN_PIECES = (kCh * PS + 255) / 256;
  static constexpr int IN_FLOATS = IN_PIECES * 256;
  static constexpr int IN_SLOTS_PER_PLANE = PS / 4;                         // 76 float4 slots
  static constexpr int W_USED = 9 * kCh * CS;
  static constexpr int W_PIECES = (W_USED + 255) / 256;
  static constexpr int W_FLOATS = W_PIECES * 256;
  static constexpr int PIECES = IN_PIECES + W_PIECES;
  static constexpr int NPW = (PIECES + 3) / 4;                              // pieces per wave per chunk
  static constexpr int STAGE_FLOATS = IN_FLOATS + W_FLOATS;
  static constexpr int STEPS = 9 * (kCh / 4);                               // 18 k-steps per chunk
  static constexpr int NST = 3;                                              // ring stages (loader-wave path)
  static constexpr int AHEAD = NST - 1;                                      // chunks in flight beside the one multiplying
  static constexpr size_t LDS_BYTES_DMA = NST * STAGE_FLOATS * sizeof(float);
  // LDS-DMA path: a fifth wave issues the pieces of chunks >= 2 (see run_loader); it keeps two
  // chunks in flight, and vmcnt counts 63 operations at most.
  static constexpr bool LOADER = 2 * PIECES <= 63;
  static constexpr int THREADS_DMA = LOADER ? 320 : 256;
  static constexpr size_t LDS_BYTES_REG = 2 * STAGE_FLOATS * sizeof(float);
  // register-staged path
  static constexpr int RIN_SLOTS = kCh * G::HALO * (G::RS / 4);             // 560 float4 slots (3 x 48)
  static constexpr int RIN_ITERS = (RIN_SLOTS + 255) / 256;
  static constexpr int RW_SLOTS = W_USED / 4;
  static constexpr int RW_ITERS = (RW_SLOTS + 255) / 256;
  static_assert(kCh * PS <= IN_FLOATS, "input planes must fit their pieces");
};

// ---------------------------------------------------------------------------------------------
// LDS-DMA staging (VEC path)
// ---------------------------------------------------------------------------------------------
template <int COUT, typename G>
struct DmaPlan {
  // Per piece i of this wave (piece index p = wave + 4 i, clamped): the lane's source offset in
  // BYTES relative to the chunk's image base (input pieces) or weight base (weight pieces), or
  // kDmaZero when the lane's 16 bytes are zero padding (the buffer range check then writes zeros).
  // Fixed for the whole kernel: chunks only move the two base addresses.
  unsigned voff[ConvCfg<COUT, G>::NPW];
};

// WEIGHTS = the offsets of this wave's weight pieces (trivial: a linear image), else those of its
// input pieces (halo tile decomposition + bounds).  Two passes so that the prologue can issue the
// weight pieces before it has worked out the input pieces.
template <int COUT, typename G, bool WEIGHTS>
__device__ __forceinline__ void make_plan(const ConvArgs& a, int wave, int lane, int y0, int x0,
                                          DmaPlan<COUT, G>& pl) {
  using C = ConvCfg<COUT, G>;
#pragma unroll
  for (int i = 0; i < C::NPW; ++i) {
    const int p = min(wave + 4 * i, C::PIECES - 1);  // surplus pieces repeat the last one (same bytes)
    const bool isw = p >= C::IN_PIECES;              // wave-uniform
    if constexpr (WEIGHTS) {
      if (4 * i + 3 < C::IN_PIECES) continue;        // an input piece for every wave
      const int ws = (p - C::IN_PIECES) * 64 + lane;
      if (isw) pl.voff[i] = (ws * 4 < C::W_USED) ? 16u * (unsigned)ws : kDmaZero;
    } else {
      if (4 * i >= C::IN_PIECES) continue;           // a weight piece for every wave
      const int slot = p * 64 + lane;
      const int ci = slot / C::IN_SLOTS_PER_PLANE;
      const int rem = slot - ci * C::IN_SLOTS_PER_PLANE;
      const int r = rem / (G::RS / 4);
      const int q = rem - r * (G::RS / 4);
      const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * q;
      const bool ok = ci < kCh && r < G::HALO && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      if (!isw) pl.voff[i] = ok ? 4u * (unsigned)((ci * a.H + gy) * a.pitch + gx) : kDmaZero;
    }
  }
}

struct ChunkSrc {
  i32x4 img;  // buffer descriptor at the first of the chunk's 8 channel planes of image n
  i32x4 wgt;  // buffer descriptor at the chunk's packed weights
  const float* img_ptr;
  const float* wgt_ptr;
};

template <int COUT, typename G>
__device__ __forceinline__ ChunkSrc chunk_src(const ConvArgs& a, int chunk, int n, const float* src_override = nullptr) {
  using C = ConvCfg<COUT, G>;
  const int c0 = chunk * kCh;
  const int s_idx = div_by_magic(chunk, a.magic_cps);
  const int c_in_src = c0 - s_idx * a.cin_per_src;
  ChunkSrc cs;
  // (one source tensor -- every launch of the layer chains: src[0] came with the first batch of arguments; only a
  // concatenated input pays the dependent kernarg load of its pointer, a scalar round trip per chunk)
  const float* src = src_override ? src_override : a.src[0];
  if (s_idx != 0) {
    asm volatile("" ::: "memory");   // (keeps the branch: hipcc otherwise folds it back into one unconditional indexed load)
    src = a.src[s_idx];
  }
  cs.img_ptr = src + ((size_t)n * a.cin_per_src + c_in_src) * ((size_t)a.H * a.pitch);
  cs.wgt_ptr = a.wpk + (size_t)chunk * C::W_USED;
  cs.img = dma_rsrc(cs.img_ptr);
  cs.wgt = dma_rsrc(cs.wgt_ptr);
  return cs;
}

// One 1 KiB piece: lane l's 16 bytes go to LDS byte address stage_addr + 1024 p + 16 l.  Scalar
// work only: pick the descriptor, form the LDS address, issue.
template <int COUT, typename G>
__device__ __forceinline__ void dma_piece(const DmaPlan<COUT, G>& pl, int i, int wave, const ChunkSrc& cs,
                                          unsigned stage_addr) {
  using C = ConvCfg<COUT, G>;
  const int p = min(wave + 4 * i, C::PIECES - 1);  // scalar; weight pieces follow the input pieces in LDS
  const bool isw = p >= C::IN_PIECES;
  i32x4 rsrc;
#pragma unroll
  for (int k = 0; k < 4; ++k) rsrc[k] = isw ? cs.wgt[k] : cs.img[k];
  lds_dma16_buf(rsrc, pl.voff[i], 0, stage_addr + 1024u * (unsigned)p);
}

// ---------------------------------------------------------------------------------------------
// Loader wave.  Issuing one 1 KiB LDS-DMA piece costs the issuing wave ~100 cycles, and a wave that
// is not issuing MFMAs idles its SIMD's matrix pipe: with the 28 pieces of a chunk dealt to the four
// MFMA waves the K loop ran at 13.4 us against 10.1 us of MFMAs (in-kernel timeline,
// tools/diag_conv.py --timeline).  So the workgroup has a FIFTH wave that does nothing but stream:
// after the barrier that opens chunk c it issues all pieces of chunk c+2 into the stage chunk c-1
// has just vacated, and before the next barrier it waits until chunk c+1 has landed.  Chunks 0 and
// 1 are still issued by the MFMA waves (four waves start the ring faster than one).
// ---------------------------------------------------------------------------------------------
template <int COUT, typename G>
struct LoaderPlan {
  unsigned voff[ConvCfg<COUT, G>::PIECES];   // this lane's source offset of EVERY piece, or kDmaZero
};

template <int COUT, typename G>
__device__ __forceinline__ void make_loader_plan(const ConvArgs& a, int lane, int y0, int x0, LoaderPlan<COUT, G>& pl) {
  using C = ConvCfg<COUT, G>;
#pragma unroll
  for (int p = 0; p < C::PIECES; ++p) {
    if (p < C::IN_PIECES) {
      const int slot = p * 64 + lane;
      const int ci = slot / C::IN_SLOTS_PER_PLANE;
      const int rem = slot - ci * C::IN_SLOTS_PER_PLANE;
      const int r = rem / (G::RS / 4);
      const int q = rem - r * (G::RS / 4);
      const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * q;
      const bool ok = ci < kCh && r < G::HALO && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      pl.voff[p] = ok ? 4u * (unsigned)((ci * a.H + gy) * a.pitch + gx) : kDmaZero;
    } else {
      const int ws = (p - C::IN_PIECES) * 64 + lane;
      pl.voff[p] = (ws * 4 < C::W_USED) ? 16u * (unsigned)ws : kDmaZero;
    }
  }
}

template <int COUT, typename G>
__device__ __forceinline__ void run_loader(const ConvArgs& a, float* smem, int lane, int n, int y0, int x0) {
  using C = ConvCfg<COUT, G>;
  if constexpr ((LARVA_DIAG & 256) != 0) return;   // (no barriers to meet the MFMA waves at)
  LoaderPlan<COUT, G> pl;
  make_loader_plan<COUT, G>(a, lane, y0, x0, pl);
  const int last = a.n_chunks - 1;
  int stage = 0;
  for (int chunk = 0; chunk <= last; ++chunk) {
    // everything but the youngest chunk's pieces has landed: for chunk >= 2 that is chunk `chunk` itself (chunks 0 and
    // 1 come from the MFMA waves, which wait for them on their side)
    asm volatile("; LARVA_RING loader chunk_pieces=%1 ahead=%2\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((C::AHEAD - 1) * C::PIECES), "n"(C::PIECES), "n"(C::AHEAD) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int nstage = (stage + C::AHEAD) % C::NST;   // the stage chunk - 1 has just vacated
    const unsigned dst = lds_addr_of(smem + nstage * C::STAGE_FLOATS);
    // Past the end the last chunk is streamed again into a stage nobody reads: every wait stays the same counted
    // vmcnt(PIECES).  (Those 2 x PIECES issues lie behind the K loop's last barrier, so the loader is the last wave of a
    // workgroup to finish -- not on the critical path: without them the step was 4-5 us SLOWER, profiles/r04_ab_loader_tail.txt.)
    const ChunkSrc cs = chunk_src<COUT, G>(a, min(chunk + C::AHEAD, last), n);
#pragma unroll
    for (int p = 0; p < C::PIECES; ++p)
      lds_dma16_buf(p < C::IN_PIECES ? cs.img : cs.wgt, pl.voff[p], 0, dst + 1024u * (unsigned)p);
    stage = stage == C::NST - 1 ? 0 : stage + 1;
  }
  // no LDS-DMA may be in flight when the workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// Register staging (!VEC path): same stage image, scalar global loads with per-element masks.
// ---------------------------------------------------------------------------------------------
template <int COUT, typename G>
struct RegStaging {
  f32x4 in[ConvCfg<COUT, G>::RIN_ITERS];
  f32x4 w[ConvCfg<COUT, G>::RW_ITERS];
};

template <int COUT, typename G>
__device__ __forceinline__ void reg_load(const ConvArgs& a, const ChunkSrc& cs, int y0, int x0, int tid,
                                         RegStaging<COUT, G>& st) {
  using C = ConvCfg<COUT, G>;
  const size_t plane = (size_t)a.H * a.pitch;
#pragma unroll
  for (int i = 0; i < C::RIN_ITERS; ++i) {
    const int s = min(tid + i * 256, C::RIN_SLOTS - 1);
    const int ci = s / (G::HALO * (G::RS / 4));
    const int rem = s - ci * (G::HALO * (G::RS / 4));
    const int r = rem / (G::RS / 4);
    const int q = rem - r * (G::RS / 4);
    const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * q;
    const bool row_ok = gy >= 0 && gy < a.H;
    const float* row = cs.img_ptr + (size_t)ci * plane + (size_t)min(max(gy, 0), a.H - 1) * a.pitch;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = gx + e;
      const float t = row[min(max(x, 0), a.W - 1)];
      v[e] = (row_ok && x >= 0 && x < a.W) ? t : 0.f;
    }
    st.in[i] = v;
  }
#pragma unroll
  for (int i = 0; i < C::RW_ITERS; ++i) {
    const int s = min(tid + i * 256, C::RW_SLOTS - 1);
    const float* p = cs.wgt_ptr + 4 * s;  // packed images are only guaranteed 4-byte aligned here
    st.w[i] = f32x4{p[0], p[1], p[2], p[3]};
  }
}

template <int COUT, typename G>
__device__ __forceinline__ void reg_store(float* stage, int tid, const RegStaging<COUT, G>& st) {
  using C = ConvCfg<COUT, G>;
#pragma unroll
  for (int i = 0; i < C::RIN_ITERS; ++i) {
    const int s = tid + i * 256;
    if (s < C::RIN_SLOTS) {
      const int ci = s / (G::HALO * (G::RS / 4));
      const int rem = s - ci * (G::HALO * (G::RS / 4));
      const int r = rem / (G::RS / 4);
      const int q = rem - r * (G::RS / 4);
      *reinterpret_cast<f32x4*>(stage + ci * C::PS + r * G::RS + 4 * q) = st.in[i];
    }
  }
#pragma unroll
  for (int i = 0; i < C::RW_ITERS; ++i) {
    const int s = tid + i * 256;
    if (s < C::RW_SLOTS) *reinterpret_cast<f32x4*>(stage + C::IN_FLOATS + 4 * s) = st.w[i];
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA block of one chunk
// ---------------------------------------------------------------------------------------------
// Operands of k-step `step` (= tap*2 + kk) of one chunk.
template <int COUT, typename G, int NCT, int PG0, int NPG>
__device__ __forceinline__ void read_operands(const float* const (&a_base)[NCT], const float* b_base, int step,
                                              float (&av)[NCT], float (&bv)[NPG]) {
  using C = ConvCfg<COUT, G>;
  if constexpr ((LARVA_DIAG & 128) != 0) {   // timing ablation: operands from registers, no LDS reads
#pragma unroll
    for (int c = 0; c < NCT; ++c) av[c] = (float)step;
#pragma unroll
    for (int p = 0; p < NPG; ++p) bv[p] = (float)(step + p);
    return;
  }
  const int tap = step / (kCh / 4), kk = step % (kCh / 4);
  const int ky = tap / 3, kx = tap % 3;
#pragma unroll
  for (int c = 0; c < NCT; ++c) av[c] = a_base[c][(tap * kCh + kk * 4) * C::CS];
#pragma unroll
  for (int p = 0; p < NPG; ++p) {
    const int pg = PG0 + p, prow = pg / G::PC, pcol = pg % G::PC;
    bv[p] = b_base[kk * 4 * C::PS + (prow + ky) * G::RS + pcol * 16 + kx];
  }
}

// Issue order of one k-step: MFMA, its share of the NRD operand reads (0x100 = DS read), MFMA, ...
template <int NMF, int NRD, int M>
__device__ __forceinline__ void shadow_groups() {
  if constexpr (M < NMF) {
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
    constexpr int share = (NRD + NMF - 1 - M) / NMF;
    if constexpr (share > 0) __builtin_amdgcn_sched_group_barrier(0x100, share, 0);
    shadow_groups<NMF, NRD, M + 1>();
  }
}

// One chunk of K (8 channels x 9 taps = 18 k-steps) for a wave that owns cout groups
// [ct0, ct0+NCT) and pixel groups [PG0, PG0+NPG) of the tile.  One wave per SIMD means nobody
// else hides the LDS latency, so the operands of step s+1 are read while step s multiplies
// (hipcc otherwise sinks every ds_read to just above its first use, lgkmcnt(0) per pair of
// MFMAs; the sched_barriers pin "reads of s+1, [one LDS-DMA piece], MFMAs of s").
// SWAP: the MFMA's A operand = activations (M = 16 pixels), B = weights (N = 16 output channels), so
// that a lane ends up with 4 CONSECUTIVE PIXELS of one output channel (see run_role's epilogue) instead
// of 4 consecutive output channels of one pixel.  Same products, same k order: identical results.
// wstage: the chunk's weight rows when they do not sit behind the input planes of `stage` (a weight image kept apart
// from the input ring: the pair-chain measurement kernel), else null.
template <int COUT, typename G, int NCT, int PG0, int NPG, bool SWAP>
__device__ __forceinline__ void mfma_chunk(const float* stage, int ct0, int lane, f32x4 (&acc)[NCT][NPG],
                                           const float* wstage = nullptr) {
  using C = ConvCfg<COUT, G>;
  const int lr = lane & 15, lq = lane >> 4;
  const float* const wrows = wstage ? wstage : stage + C::IN_FLOATS;
  // Weight rows: lane (lr, lq) reads row (tap, k = 4 kk + lq), channel 16 (ct0 + c) + lr.  At 32 / 64 output channels
  // the rows are unpadded and odd rows (lq odd) are stored with their 16-channel groups swapped in pairs (column
  // c ^ 16, cout_swizzled): the group index becomes (ct0 + c) ^ (lq & 1) -- a per-lane base per output-channel
  // group, the k-step offsets stay immediates.  (48 channels: plain rows, the bases differ by constants.)
  const float* a_base[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
    a_base[c] = wrows + lq * C::CS + lr + (cout_swizzled(COUT) ? (((ct0 + c) ^ (lq & 1)) * 16) : (ct0 + c) * 16);
  const float* b_base = stage + lq * C::PS + lr + 3;
  // (Tried and dropped: scheduling regions of TWO k-steps.  tools/probe_mfma_rate.hip shows that a wave
  // streaming MFMAs beside LDS reads loses a fixed ~28 cycles per region -- 36.98 ticks per MFMA with 7
  // MFMAs per region against 33.44 with 14 -- but in this kernel both two-k-step layouts (a tap's 16
  // operand reads within its first 7 MFMAs, or dealt over all 14 with the ring three regions deep) were
  // slower: the two-chain layer 15.6 / 13.9 us against 13.8, the step 1.726 / 1.712 ms against 1.669; so
  // was the probe's exact layout with inline-asm MFMAs, accumulators tied in place and one read behind
  // every MFMA (single chain 16.2 us against 15.5).)
  // Operands are read kOperandDepth k-steps ahead of the MFMAs that use them (ring of D + 1 register sets).
  // With depth 1 the wait in front of a k-step's first MFMA was lgkmcnt(0) on reads issued only 3-4 MFMAs
  // earlier; with depth 2 the compiler's wait is a COUNTED one that leaves the newest k-step's reads in flight
  // (depth 3: step 1.675-1.677 against 1.659-1.661 ms).
  constexpr int D = kOperandDepth;
  float av[D + 1][NCT], bv[D + 1][NPG];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < C::STEPS) read_operands<COUT, G, NCT, PG0, NPG>(a_base, b_base, d, av[d], bv[d]);
#pragma unroll
  for (int step = 0; step < C::STEPS; ++step) {
    // One scheduling region per k-step.  A v_mfma_f32_16x16x4_f32 keeps the matrix pipe busy for
    // 32 cycles but the wave's issue port for only 8 of them: ~24 cycles of other instructions
    // per MFMA are free, everything issued OUTSIDE such a shadow idles the pipe (one wave per
    // SIMD).  So the operand reads of a later step are dealt out one or two per MFMA of step s
    // instead of in a block in front of them (hipcc left alone sinks every ds_read under an lgkmcnt(0)).
    __builtin_amdgcn_sched_barrier(0);
    if (step + D < C::STEPS)
      read_operands<COUT, G, NCT, PG0, NPG>(a_base, b_base, step + D, av[(step + D) % (D + 1)], bv[(step + D) % (D + 1)]);
    constexpr int NMF = NCT * NPG;
    constexpr int NRD = NCT + NPG;
#pragma unroll
    for (int m = 0; m < NMF; ++m) {
      const float wv = av[step % (D + 1)][m / NPG], xv = bv[step % (D + 1)][m % NPG];
      acc[m / NPG][m % NPG] = __builtin_amdgcn_mfma_f32_16x16x4f32(SWAP ? xv : wv, SWAP ? wv : xv, acc[m / NPG][m % NPG], 0, 0, 0);
    }
    shadow_groups<NMF, NRD, 0>();
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Wait until all but the `KEEP` youngest vector-memory operations of this wave are done, then
// meet the other waves.  Raw barrier on purpose: __syncthreads() would drain every LDS-DMA.
// PIECES / AUX: what KEEP is made of -- the LDS-DMA pieces of the NEXT chunk this wave has in flight and the epilogue
// operand loads behind them (kAuxLoads).  They travel into the device assembly as a comment in front of the wait
// ("; LARVA_RING pieces=P aux=A"): tools/check_aux_loads.py compares the loads hipcc really emitted against these VALUES,
// not against plausible ranges.
template <int KEEP, int PIECES, int AUX>
__device__ __forceinline__ void wait_and_barrier() {
  static_assert(KEEP == PIECES + AUX, "a ring wait keeps exactly the next chunk's pieces and the operand loads in flight");
  asm volatile("; LARVA_RING pieces=%1 aux=%2\n\ts_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(KEEP), "n"(PIECES), "n"(AUX) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// virtual block index -> (image, tile origin) of a whole-tensor launch (a.nwg tiles of G::ROWS x G::COLS pixels)
template <typename G>
__device__ __forceinline__ void decode_tile(const ConvArgs& a, int vb, int& n, int& y0, int& x0) {
  const int tile = xcd_remap(vb, a.nwg);
  const int t2 = div_by_magic(tile, a.magic_tx);
  const int tx = tile - t2 * a.tiles_x;
  n = div_by_magic(t2, a.magic_ty);
  const int ty = t2 - n * a.tiles_y;
  x0 = tx * G::COLS;
  y0 = ty * G::ROWS;
}

// One MFMA wave's share of a tile: cout groups [ct0, ct0 + NCT) x pixel groups [PG0, PG0 + NPG).
// PERSIST (conv3x3_mfma_persist_kernel): the workgroup walks tiles blockIdx.x, + gridDim.x, ... of the launch; the loader
// wave streams EVERY chunk (run_loader_persist), the ring runs on across the tiles, and this wave only meets the loader
// at one barrier per chunk, multiplies, and stores its tile while the next tile's first chunks are already in LDS.
template <int COUT, typename G, bool VEC, int EPI, int NCT, int PG0, int NPG, bool PERSIST = false>
__device__ __forceinline__ void run_role(const ConvArgs& a, float* smem, int ct0, int wave, int n, int y0, int x0, int tid) {
  using C = ConvCfg<COUT, G>;
  static_assert(!VEC || C::LOADER, "the 16-byte path runs with a loader wave (2 * PIECES <= 63 for every instantiated shape)");
  static_assert(!PERSIST || VEC, "persistent tiles: the loader-wave path");
  const int lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  float* const out_ptr = a.out;
  // The epilogue's scalars, PINNED in SGPRs from here on (round 4).  Left alone hipcc drops them after the prologue and
  // re-loads each from the kernarg segment where the epilogue uses it -- H / pitch / plain_stores after the K loop and
  // W once per (channel group, pixel group) unit, every one an s_load + s_waitcnt lgkmcnt(0) in front of that unit's
  // stores: five dependent scalar round trips at the tail of every workgroup's life (read off the ISA of round 3's
  // binary).  An opaque asm operand cannot be rematerialised from memory.
  struct { int H, W, pitch, plain; } k = {a.H, a.W, a.pitch, a.plain_stores};
  asm volatile("" : "+s"(k.H), "+s"(k.W), "+s"(k.pitch), "+s"(k.plain));

  // Orientation of the accumulators.  Channel-major (the MFMA's A operand = weights): lane (lr, lq)
  // holds output channels 4 lq .. 4 lq + 3 of pixel lr of its group -- what the pixel-shuffle store wants
  // (4 sub-pixels of one HR row = 16 contiguous bytes).  Pixel-major (A = activations; every other epilogue of the
  // 16-byte path): lane (lr, lq) holds pixels 4 lq .. 4 lq + 3 of output channel lr -- 16 contiguous bytes of an NCHW
  // row, so the ordinary epilogues store (and fetch their mask / residual operands) with ONE 16-byte
  // access per (channel group, pixel group) instead of four 4-byte ones: in the batched launch the
  // store phase was 1.3 us of a conv's 12.7 us (un-instrumented ablation, tools/bench_conv_batch.py).
  constexpr bool kShuffleEpi = (EPI == kEpiShuffle || EPI == kEpiShuffleBase || EPI == kEpiShuffleL1);
  constexpr bool kPixMajor = VEC && !kShuffleEpi;
  f32x4 bias[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) bias[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_bias = [&]() {
    if (a.bias) {
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[c][r] = a.bias[(ct0 + c) * 16 + (kPixMajor ? lr : lq * 4 + r)];
    }
  };

  // The epilogue's other operands (ReLU mask, residuals, bicubic base, truth) are fetched BEFORE the K loop so that
  // they land under it instead of being waited for after it (the mask / residual variants ran 1-1.3 us longer).
  constexpr int NAUX = (EPI == kEpiMask || EPI == kEpiRes1 || EPI == kEpiShuffleBase)
                           ? 1 : ((EPI == kEpiRes2 || EPI == kEpiShuffleL1) ? 2 : 0);
  f32x4 aux[NAUX > 0 ? NAUX : 1][NAUX > 0 ? NCT : 1][NAUX > 0 ? NPG : 1];
  // PERSIST: a tile's operands are requested under its K loop, wave w's in front of chunk w (part = kEarlyAll; see the
  // loop) -- but only the first kEarlyPairs (operand, unit) pairs q = op * NU + u (28 registers); the rest (part =
  // kLatePart) at the head of the epilogue, into the registers the MFMA operands have just left.  Round 5: a 5-wave workgroup puts waves 0 and 4 on the same SIMD, two
  // workgroups per CU are FOUR waves on that SIMD = at most 128 VGPRs, and the two-residual kernel held 147 (64
  // channels: 141 with one residual, 190 with two): the second workgroup of every CU entered only when the first had
  // left (in-kernel stamps, profiles/r05_infer_wide_layer_stamps.txt: 256 of 512 workgroups enter 28-46 us late).  Capped
  // by the compiler instead (launch bounds) it spills 32 registers and the layer takes 78 against 70 us.
  constexpr int NU = NCT * NPG;
  // (48 channels only: at 32 every variant is below 128 anyway, at 64 one workgroup per CU already keeps the matrix pipe
  // as busy as two -- residual layers 110-113 us against 107 for conv + ReLU -- and late operands would stand exposed)
  constexpr int kEarlyPairs = PERSIST && COUT == 48 ? (NAUX * NU < 7 ? NAUX * NU : 7) : NAUX * NU;
  constexpr int kLatePart = 4, kEarlyAll = 5;
  // (tried: `part` as a compile-time tag behind a switch, so that the units of the unrolled nest lose their scalar
  // branches -- the four copies of the loads cost registers instead: 71 / 80 us per residual layer against 63 / 66)
  auto load_aux = [&](int part = -1) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int p = 0; p < NPG; ++p) {
        const int pg = PG0 + p, prow = pg / G::PC, pcol = pg % G::PC;
        const int y = min(y0 + prow, a.H - 1), x = min(x0 + pcol * 16 + lr, a.W - 1);
        // does this call request operand `op` of unit (c, p)?   (wave-uniform, constant after unrolling)
        auto want = [&](int op) {
          const int q = op * NU + c * NPG + p;
          return part < 0 ? true : part == kLatePart ? q >= kEarlyPairs : q < kEarlyPairs;
        };
        if constexpr (kShuffleEpi) {
          const int HH = 4 * a.H, WW = 4 * a.W;
          const size_t idx = (((size_t)n * C::CT + (ct0 + c)) * HH + (4 * y + lq)) * WW + 4 * x;
          if (want(0)) aux[0][c][p] = *reinterpret_cast<const f32x4*>(a.base + idx);
          if constexpr (EPI == kEpiShuffleL1) {
            if (want(1)) aux[1][c][p] = *reinterpret_cast<const f32x4*>(a.truth + idx);
          }
        } else if constexpr (kPixMajor && (PERSIST || G::COLS == 16)) {
          // Buffer loads (persistent tiles and strip tiles; the 3 x 48 / 4 x 48 one-workgroup-per-tile launches keep their
          // global loads below: with these a full-image residual layer took 75.9 instead of 66.4 us): the lane's own offset is ONE VGPR for the whole launch, the unit's offset inside image n a
          // wave-uniform scalar operand -- no vector address arithmetic per unit.  Persistent tiles: 63.3 -> 62.8 / 66.2 ->
          // 65.4 us per residual layer of a full image, 10 / 3 VGPRs fewer (such a layer takes longer than conv + ReLU even
          // with its operand HOT in L2, tools/probe_res_operand.py: issue slots of the MFMA waves, not memory); the
          // training step's strip launches: 1.619-1.620 against 1.625-1.632 ms per step, same box, three alternating
          // rounds.  What a lane of an edge tile requests past the image comes back as zeros (the descriptor covers
          // exactly the image's COUT planes) and is never used.
          const unsigned plane = (unsigned)a.H * (unsigned)a.pitch;
          const unsigned lane_off = 4u * ((unsigned)lr * plane + (unsigned)lq * 4u);
          const int soff = __builtin_amdgcn_readfirstlane(
              (int)(4u * ((unsigned)((ct0 + c) * 16) * plane + (unsigned)(y0 + prow) * (unsigned)a.pitch + (unsigned)(x0 + pcol * 16))));
          auto image = [&](const float* t) {
            const unsigned long long b = reinterpret_cast<unsigned long long>(t + (size_t)n * COUT * plane);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                     (int)(COUT * plane * 4u), 0x00020000);
          };
          if constexpr (EPI == kEpiMask) {
            if (want(0)) aux[0][c][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(image(a.mask), lane_off, soff, 0));
          }
          if constexpr (EPI == kEpiRes1 || EPI == kEpiRes2) {
            if (want(0)) aux[0][c][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(image(a.res0), lane_off, soff, 0));
          }
          if constexpr (EPI == kEpiRes2) {
            if (want(1)) aux[1][c][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(image(a.res1), lane_off, soff, 0));
          }
        } else if constexpr (kPixMajor) {
          const size_t plane = (size_t)a.H * a.pitch;
          const int xb = min(x0 + pcol * 16 + lq * 4, a.pitch - 4);   // (pitch % 4 == 0 on this path)
          const size_t idx = ((size_t)n * COUT + (ct0 + c) * 16 + lr) * plane + (size_t)y * a.pitch + xb;
          if constexpr (EPI == kEpiMask) {
            if (want(0)) aux[0][c][p] = *reinterpret_cast<const f32x4*>(a.mask + idx);
          }
          if constexpr (EPI == kEpiRes1 || EPI == kEpiRes2) {
            if (want(0)) aux[0][c][p] = *reinterpret_cast<const f32x4*>(a.res0 + idx);
          }
          if constexpr (EPI == kEpiRes2) {
            if (want(1)) aux[1][c][p] = *reinterpret_cast<const f32x4*>(a.res1 + idx);
          }
        } else {
          const size_t plane = (size_t)a.H * a.pitch;
          const size_t idx0 = ((size_t)n * COUT + (ct0 + c) * 16 + lq * 4) * plane + (size_t)y * a.pitch + x;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (EPI == kEpiMask) aux[0][c][p][r] = a.mask[idx0 + r * plane];
            if constexpr (EPI == kEpiRes1 || EPI == kEpiRes2) aux[0][c][p][r] = a.res0[idx0 + r * plane];
            if constexpr (EPI == kEpiRes2) aux[1][c][p][r] = a.res1[idx0 + r * plane];
          }
        }
      }
  };
  // Round 4: on the loader-wave path the mask / residual operands are requested BEHIND chunk 1's pieces, not in front of
  // chunk 0's.  vmcnt retires in order: requested first (rounds 1-3), the first counted wait of the ring -- "chunk 0 has
  // landed" -- also waited for them, and in a training step they are COLD (the ReLU mask is an activation the forward
  // pass wrote a millisecond ago, the residuals come from two layers back): the K loop of every mask / residual link
  // started ~1.3 us later than a conv + ReLU link's (in-kernel stamps of the captured step, profiles/r04_step_timeline_
  // stamped.txt: workgroup life 10.4-11.5 against 9.0-10.0 us at the same 5.8-5.9 us of K loop).  Requested last, every
  // wait of the ring leaves exactly these kAuxLoads vector loads in flight (one 16-byte load per operand and (channel
  // group, pixel group) unit with pixel-major accumulators) and the vmcnt(0) behind the K loop collects them.
  // kAuxLoads must not EXCEED the loads the compiler really emits between chunk 1's pieces and the first wait (a
  // smaller real count would let a wait pass with a piece still in flight): tools/check_aux_loads.py counts them in
  // the ISA of every instantiation, and larvanet_amd/build.py runs it on every build.
  // (not for the pixel-shuffle exits: their base / truth operands are 2 x 7 sixteen-byte loads per wave from the HR
  // images, and requested late the batched L1 launch got SLOWER -- 66 against 58 us under rocprofv3, the exits' forward
  // 120 against 112 us in the stamped step: they then queue in front of the loader wave's chunk-2 pieces)
  constexpr bool kAuxLate = VEC && !kShuffleEpi && !(LARVA_DIAG & 6) && NAUX > 0 && !PERSIST;
  constexpr int kAuxLoads = kAuxLate ? NAUX * NCT * NPG : 0;
  auto load_early = [&]() {
    load_bias();
    if constexpr (NAUX > 0 && !kAuxLate && !(LARVA_DIAG & 4)) load_aux();
  };
  int pstage = 0;   // PERSIST: the ring's stage, running on across the workgroup's tiles
  for (int vb = blockIdx.x;; vb += gridDim.x) {   // (one trip unless PERSIST)
  if constexpr (PERSIST) {
    if (vb >= a.nwg) break;
    decode_tile<G>(a, vb, n, y0, x0);
  }
  if constexpr (!VEC || (LARVA_DIAG & 2)) load_early();

  f32x4 acc[NCT][NPG];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int p = 0; p < NPG; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (PERSIST) {
    // bias and the epilogue's operands of THIS tile have the K loop to arrive (the vmcnt(0) behind it also covers the
    // previous tile's stores, long drained)
    load_bias();
    const int my_chunk = a.n_chunks >= 4 ? (wave & 3) : 0;
    __builtin_amdgcn_sched_barrier(0);
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
      asm volatile("s_barrier" ::: "memory");   // the loader has seen the chunk land; everybody is done with the stage before it
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NAUX > 0) {
        // wave w requests its early operands in front of chunk w: the workgroup's operand traffic is spread over the first
        // four chunks (as one burst at the tile's start, on every CU at once, it stood in the memory pipeline in front of
        // the loader wave's next chunks: +14 us per layer with two residual operands), and a wave pays one uniform branch
        // per chunk.  (Until late in round 5: a quarter of EVERY wave's units per chunk -- a scalar branch per unit in
        // front of four chunks' first MFMAs: 62.8 against 61.2 us for a one-residual layer, same box.)
        if (chunk == my_chunk) load_aux(kEarlyAll);
        __builtin_amdgcn_sched_barrier(0);
      }
      mfma_chunk<COUT, G, NCT, PG0, NPG, kPixMajor>(smem + pstage * C::STAGE_FLOATS, ct0, lane, acc);
      pstage = pstage == C::NST - 1 ? 0 : pstage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kEarlyPairs < NAUX * NU) {
      load_aux(kLatePart);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (VEC) {
    // ---- LDS-DMA ring: chunk c lives in stage c % 3; chunks c+1 and c+2 are in flight --------
    // Prologue: the weight pieces of chunk 0 go out first (their offsets need no arithmetic), then bias (and the
    // pixel-shuffle exits' operands), then the input pieces of chunk 0, then all of chunk 1, then the mask / residual
    // operands -- so that the first counted wait leaves chunk 1's NPW pieces and those loads in flight.  From chunk 2 on
    // the loader wave issues (run_loader), and this wave's later waits find only its operand loads outstanding.
    DmaPlan<COUT, G> pl;
    const int last = a.n_chunks - 1;
    if constexpr (!(LARVA_DIAG & 2)) {
      const ChunkSrc cs0 = chunk_src<COUT, G>(a, 0, n);
      const unsigned st0 = lds_addr_of(smem), st1 = lds_addr_of(smem + C::STAGE_FLOATS);
      stamp(6);
      make_plan<COUT, G, true>(a, wave, lane, y0, x0, pl);
#pragma unroll
      for (int i = 0; i < C::NPW; ++i)
        if (wave + 4 * i >= C::IN_PIECES) dma_piece<COUT, G>(pl, i, wave, cs0, st0);
      __builtin_amdgcn_sched_barrier(0);
      stamp(7);
      load_early();
      stamp(14);
      make_plan<COUT, G, false>(a, wave, lane, y0, x0, pl);
#pragma unroll
      for (int i = 0; i < C::NPW; ++i)
        if (wave + 4 * i < C::IN_PIECES) dma_piece<COUT, G>(pl, i, wave, cs0, st0);
      stamp(15);
      const ChunkSrc cs1 = chunk_src<COUT, G>(a, min(1, last), n);
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) dma_piece<COUT, G>(pl, i, wave, cs1, st1);
      if constexpr (kAuxLate) {   // the youngest vector loads of the wave (see above)
        __builtin_amdgcn_sched_barrier(0);
        load_aux();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stamp(1);
    int stage = 0;
    for (int chunk = 0; chunk <= last; ++chunk) {
      // The chunk's own pieces have landed and, after the barrier, everybody's have, and everybody is done with the
      // stage that chunk + 2 is about to overwrite.
      if constexpr ((LARVA_DIAG & 256) != 0) {
        // timing ablation (with bit 1: nothing is staged): no barrier between the chunks
      } else if (chunk == 0) wait_and_barrier<((LARVA_DIAG & 2) ? 0 : C::NPW) + kAuxLoads, ((LARVA_DIAG & 2) ? 0 : C::NPW), kAuxLoads>();
      else wait_and_barrier<kAuxLoads, 0, kAuxLoads>();
      if (chunk == 0) stamp(2);
      stamp(8 + (chunk < 7 ? chunk : 7));
      if constexpr (!(LARVA_DIAG & 1)) mfma_chunk<COUT, G, NCT, PG0, NPG, kPixMajor>(smem + stage * C::STAGE_FLOATS, ct0, lane, acc);
      stage = stage == C::NST - 1 ? 0 : stage + 1;
    }
    // no LDS-DMA may be in flight when the workgroup's LDS is released (and the operand loads have arrived)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);
  } else {
    // ---- register staging, 2 stages ------------------------------------------------------------
    RegStaging<COUT, G> st;
    if constexpr (!(LARVA_DIAG & 2)) {
      reg_load<COUT, G>(a, chunk_src<COUT, G>(a, 0, n), y0, x0, tid, st);
      reg_store<COUT, G>(smem, tid, st);
    }
    __syncthreads();
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
      float* cur = smem + (chunk & 1) * C::STAGE_FLOATS;
      float* nxt = smem + ((chunk & 1) ^ 1) * C::STAGE_FLOATS;
      const bool more = chunk + 1 < a.n_chunks;
      if constexpr (!(LARVA_DIAG & 2)) {
        if (more) reg_load<COUT, G>(a, chunk_src<COUT, G>(a, chunk + 1, n), y0, x0, tid, st);
      }
      if constexpr (!(LARVA_DIAG & 1)) mfma_chunk<COUT, G, NCT, PG0, NPG, kPixMajor>(cur, ct0, lane, acc);
      if constexpr (!(LARVA_DIAG & 2)) {
        if (more) reg_store<COUT, G>(nxt, tid, st);
      }
      __syncthreads();
    }
  }

  if constexpr ((LARVA_DIAG & 4) != 0) {  // keep the accumulators alive, touch no global memory
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int p = 0; p < NPG; ++p) asm volatile("" ::"v"(acc[c][p]));
    return;
  }

  // Epilogue.  Channel-major: lane (lr, lq) holds, in acc[c][p][r], output channel (ct0+c)*16 + lq*4 + r of
  // pixel (y0 + pg / PC, x0 + (pg % PC) * 16 + lr).
  if constexpr (EPI == kEpiShuffleL1) {
    // The exit's image never has to reach memory to be scored: out = shuffle(conv) + base sits in
    // the accumulators in exactly the (16C + 4i + j) channel order its L1 gradient is consumed in
    // by the leg's dgrad / wgrad (models/LarvaNet.py:107-109 + autograd of :261).  sign(0) = 0 as
    // in ATen's l1 backward.  The |out - truth| sum of this wave's share goes to partial[4 tile + wave]
    // (fixed order everywhere: the loss is reproducible run to run).
    const int HH = 4 * k.H, WW = 4 * k.W;
    const size_t plane = (size_t)k.H * k.pitch;
    const float g = a.gval;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int p = 0; p < NPG; ++p) {
        const int pg = PG0 + p, prow = pg / G::PC, pcol = pg % G::PC;
        const int y = y0 + prow, x = x0 + pcol * 16 + lr;
        const f32x4 v = (acc[c][p] + bias[c]) + aux[0][c][p];
        if (y < k.H && x < k.W) {
          if (out_ptr) {
            const size_t idx = (((size_t)n * C::CT + (ct0 + c)) * HH + (4 * y + lq)) * WW + 4 * x;
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_ptr + idx));
          }
          const f32x4 d = v - aux[1][c][p];
          s += fabsf(d[0]) + fabsf(d[1]) + fabsf(d[2]) + fabsf(d[3]);
          const size_t idx0 = ((size_t)n * COUT + (ct0 + c) * 16 + lq * 4) * plane + (size_t)y * k.pitch + x;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_nontemporal_store(d[r] > 0.f ? g : (d[r] < 0.f ? -g : 0.f), &a.grad[idx0 + r * plane]);
        }
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) a.partial[4 * blockIdx.x + wave] = s;
  } else if constexpr (EPI == kEpiShuffle || EPI == kEpiShuffleBase) {
    // PixelShuffle(4): out[n, C, 4y+i, 4x+j] = conv[n, 16C + 4i + j, y, x]; here C = ct0+c,
    // i = lq, j = r -> one aligned 16-byte store per lane (models/LarvaNet.py:261,265-266).
    const int HH = 4 * k.H, WW = 4 * k.W;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int p = 0; p < NPG; ++p) {
        const int pg = PG0 + p, prow = pg / G::PC, pcol = pg % G::PC;
        const int y = y0 + prow, x = x0 + pcol * 16 + lr;
        const size_t idx = (((size_t)n * C::CT + (ct0 + c)) * HH + (4 * y + lq)) * WW + 4 * x;
        f32x4 v = acc[c][p] + bias[c];
        if constexpr (EPI == kEpiShuffleBase) v += aux[0][c][p];
        if (y < k.H && x < k.W) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_ptr + idx));
      }
  } else if constexpr (kPixMajor) {
    // Pixel-major accumulators: acc[c][p][r] = output channel (ct0+c)*16 + lr of pixel
    // (y0 + pg / PC, x0 + (pg % PC) * 16 + 4 lq + r): one 16-byte store per (c, p).
    // Store policy.  Non-temporal for the 48-column tiles: the 7 MB store burst of the 256 workgroups drains faster
    // (-0.7 us per launch in a chain of whole-batch launches that writes a tensor per layer: 15.3 vs 16.1 us;
    // re-measured in round 4, profiles/r04_ab_wide_plain.txt).  The strip kernel takes the policy per launch (k.plain):
    // with two half-batch chains sharing the CUs plain stores are faster in both chains (step 1.634-1.641 ms all-plain,
    // 1.648-1.650 forward only, 1.657-1.660 all-non-temporal: profiles/r04_ab_strip_plain.txt).
    const size_t plane = (size_t)k.H * k.pitch;
    // INTERIOR (a tile wholly inside the image -- every tile of a 48 x 48 training patch, all but the last column of tiles
    // of a 339 x 510 image): no per-lane bounds, no per-element zero fill.  The epilogue's vector instructions compete
    // with the co-resident workgroup's MFMAs for the SIMD's issue port, so what is not issued here is its K loop's gain.
    auto store_all = [&](auto plain_tag, auto interior_tag) {
      constexpr bool kPlain = decltype(plain_tag)::value;
      constexpr bool kInterior = decltype(interior_tag)::value;
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int p = 0; p < NPG; ++p) {
          const int pg = PG0 + p, prow = pg / G::PC, pcol = pg % G::PC;
          const int y = y0 + prow, xb = x0 + pcol * 16 + lq * 4;
          const size_t idx = ((size_t)n * COUT + (ct0 + c) * 16 + lr) * plane + (size_t)y * k.pitch + xb;
          f32x4 v = acc[c][p] + bias[c];
          if (kInterior || (y < k.H && xb < k.pitch)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float o = v[r];
              if constexpr (EPI == kEpiRelu) o = fmaxf(o, 0.f);
              if constexpr (EPI == kEpiMask) o = (aux[0][c][p][r] > 0.f) ? o : 0.f;
              if constexpr (EPI == kEpiRes1 || EPI == kEpiRes2) o += aux[0][c][p][r];
              if constexpr (EPI == kEpiRes2) o += aux[1][c][p][r];
              v[r] = (kInterior || xb + r < k.W) ? o : 0.f;   // columns [W, pitch) are kept at zero for the next layer
            }
            if constexpr (kPlain) *reinterpret_cast<f32x4*>(out_ptr + idx) = v;
            else __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out_ptr + idx));
          }
        }
    };
    // (not for the two-residual and mask epilogues: the second copy of the epilogue costs the two-residual persistent
    // kernel its 128 registers -- 71.0 instead of 65.6 us per layer -- and the training step gains nothing either way)
    constexpr bool kFastPath = EPI == kEpiPlain || EPI == kEpiRelu || EPI == kEpiRes1;
    const bool interior = kFastPath && y0 + G::ROWS <= k.H && x0 + G::COLS <= k.W;   // (wave-uniform)
    auto store_policy = [&](auto plain_tag) {
      if constexpr (kFastPath) {
        if (interior) {
          store_all(plain_tag, std::true_type{});
          return;
        }
      }
      store_all(plain_tag, std::false_type{});
    };
    if constexpr ((LARVA_DIAG & 16) != 0) {
      store_policy(std::true_type{});
    } else if constexpr (G::COLS == 16) {
      if (k.plain) store_policy(std::true_type{});
      else store_policy(std::false_type{});
    } else {
      store_policy(std::false_type{});
    }
  } else {
    // Register-staged path (any width): channel-major accumulators, four 4-byte stores per (c, p)
    const size_t plane = (size_t)k.H * k.pitch;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int p = 0; p < NPG; ++p) {
        const int pg = PG0 + p, prow = pg / G::PC, pcol = pg % G::PC;
        const int y = y0 + prow, x = x0 + pcol * 16 + lr;
        const size_t idx0 = ((size_t)n * COUT + (ct0 + c) * 16 + lq * 4) * plane + (size_t)y * k.pitch + x;
        const f32x4 v = acc[c][p] + bias[c];
        if (y < k.H && x < k.pitch) {
          const bool real = x < k.W;  // columns [W, pitch) are kept at zero for the next layer
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float o = v[r];
            if constexpr (EPI == kEpiRelu) o = fmaxf(o, 0.f);
            if constexpr (EPI == kEpiMask) o = (aux[0][c][p][r] > 0.f) ? o : 0.f;
            if constexpr (EPI == kEpiRes1 || EPI == kEpiRes2) o += aux[0][c][p][r];
            if constexpr (EPI == kEpiRes2) o += aux[1][c][p][r];
            o = real ? o : 0.f;
            if constexpr ((LARVA_DIAG & 16) != 0) out_ptr[idx0 + r * plane] = o;
            else __builtin_nontemporal_store(o, &out_ptr[idx0 + r * plane]);
          }
        }
      }
  }
#if LARVA_DIAG & 32
  stamp(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(5);
#endif
  if constexpr (!PERSIST) break;
  }
}

// Fetch every kernel argument NOW, in one batch of scalar loads.  Left alone the compiler loads
// each field in the block that first uses it: three dependent s_load round trips (~0.15 us each)
// stood between kernel entry and the first LDS-DMA request.
__device__ __forceinline__ void fetch_args(const ConvArgs& a) {
  asm volatile("" ::"s"(a.src[0]), "s"(a.wpk), "s"(a.bias), "s"(a.out), "s"(a.cin_per_src), "s"(a.n_chunks), "s"(a.N),
               "s"(a.H), "s"(a.W), "s"(a.pitch), "s"(a.tiles_x), "s"(a.tiles_y), "s"(a.magic_tx), "s"(a.magic_ty),
               "s"(a.magic_cps), "s"(a.nwg), "s"(a.tile_tab), "s"(a.plain_stores), "s"(a.tab_n),
               "s"(a.tab16[0]), "s"(a.tab16[16]));
}

// VEC: 2 workgroups per CU (launch bound 2 waves/SIMD caps the registers at 256).
template <int COUT, bool VEC, int EPI, typename G = GeoWide>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, float* smem) {
  if constexpr ((LARVA_DIAG & 8) != 0) return;
  stamp(0);
  fetch_args(a);
  const int tile = xcd_remap(blockIdx.x, a.nwg);
  const int t2 = div_by_magic(tile, a.magic_tx);
  const int tx = tile - t2 * a.tiles_x;
  const int n = div_by_magic(t2, a.magic_ty);
  const int ty = t2 - n * a.tiles_y;
  const int x0 = tx * G::COLS, y0 = ty * G::ROWS;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Convs form the dependent chain of a step; weight-gradient workgroups that share the CU run
  // at priority 0 and take the matrix pipe only when these waves cannot use it.
  __builtin_amdgcn_s_setprio(1);

  if constexpr (VEC) {
    if (wave == 4) {
      if constexpr (!(LARVA_DIAG & 2)) run_loader<COUT, G>(a, smem, tid & 63, n, y0, x0);
      return;
    }
  }
  // ROWS * 3 pixel groups x CT cout groups, dealt to the 4 waves (one per SIMD) as evenly as a
  // rectangular (cout groups) x (pixel groups) ownership allows.
  if constexpr (G::ROWS == 4) {
    static_assert(VEC && (COUT == 48 || COUT == 32), "4 x 48 tiles: the 16-byte path at 48 / 32 channels");
    if constexpr (COUT == 48) {  // 36 -> 9,9,9,9
      if (wave < 3) run_role<COUT, G, VEC, EPI, 1, 0, 9>(a, smem, wave, wave, n, y0, x0, tid);
      else run_role<COUT, G, VEC, EPI, 3, 9, 3>(a, smem, 0, wave, n, y0, x0, tid);
    } else {                     // 24 -> 6,6,6,6
      if (wave < 2) run_role<COUT, G, VEC, EPI, 1, 0, 6>(a, smem, wave, wave, n, y0, x0, tid);
      else run_role<COUT, G, VEC, EPI, 1, 6, 6>(a, smem, wave - 2, wave, n, y0, x0, tid);
    }
  } else if constexpr (COUT == 48) {  // 27 -> 7,7,7,6
    if (wave < 3) run_role<COUT, G, VEC, EPI, 1, 0, 7>(a, smem, wave, wave, n, y0, x0, tid);
    else run_role<COUT, G, VEC, EPI, 3, 7, 2>(a, smem, 0, wave, n, y0, x0, tid);
  } else if constexpr (COUT == 32) {  // 18 -> 5,5,4,4
    if (wave < 2) run_role<COUT, G, VEC, EPI, 1, 0, 5>(a, smem, wave, wave, n, y0, x0, tid);
    else run_role<COUT, G, VEC, EPI, 1, 5, 4>(a, smem, wave - 2, wave, n, y0, x0, tid);
  } else {  // COUT == 64: 36 -> 9,9,9,9
    static_assert(COUT == 64, "unsupported channel count");
    run_role<COUT, G, VEC, EPI, 1, 0, 9>(a, smem, wave, wave, n, y0, x0, tid);
  }
}

template <int COUT, bool VEC, int EPI>
__global__ __launch_bounds__(VEC ? ConvCfg<COUT>::THREADS_DMA : 256, VEC ? kWgPerCu : 1) void conv3x3_mfma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_tile<COUT, VEC, EPI>(a, smem);
}

// the same launch on 4 x 48 tiles (16-byte path, 48 / 32 channels)
template <int COUT, int EPI>
__global__ __launch_bounds__((ConvCfg<COUT, GeoWide4>::THREADS_DMA), kWgPerCu) void conv3x3_mfma_rows4_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_tile<COUT, true, EPI, GeoWide4>(a, smem);
}
static_assert(2 * ConvCfg<48, GeoWide4>::LDS_BYTES_DMA <= 160 * 1024, "two 4 x 48 workgroups per CU");

// ---------------------------------------------------------------------------------------------
// Persistent whole-tensor launch (round 5): more tiles than the chip has workgroup slots -- a 339 x 510 image is 1243 tiles
// for 512 slots.  As one workgroup per tile the slots turn over in lockstep: in-kernel stamps of such a launch
// (tools/diag_wide.py, profiles/r05_infer_wide_layer_stamps.txt) show three synchronized rounds -- 512 workgroups enter within
// a microsecond, share the matrix pipes two by two through a 21 us K loop, store, and the next 512 all sit in their
// prologues (kernarg fetch, tile decode, first LDS-DMA round trip: 3.4 us; 6.5 us with a residual operand to fetch) while
// NO K loop runs -- then 219 workgroups on half-empty CUs.  Here every slot is one workgroup that walks its tiles: the
// loader wave streams the flattened (tile, chunk) sequence through the same three-stage ring, so the next tile's first
// two chunks are in LDS before this tile's K loop ends, the MFMA waves' stores drain under the next K loop, and kernarg
// fetch / role set-up happen once per slot.  Results are bit-identical (a pixel's K loop does not depend on who runs it).
// ---------------------------------------------------------------------------------------------
template <int COUT, typename G>
__device__ __forceinline__ void run_loader_persist(const ConvArgs& a, float* smem, int lane) {
  using C = ConvCfg<COUT, G>;
  static_assert(C::NST == 3 && 2 * C::PIECES <= 63, "three stages, two chunks in flight");
  LoaderPlan<COUT, G> pl;
  // issue cursor: tile vb_i (image n_i), chunk c_i, stage s_i; it runs two chunks ahead of the chunk being multiplied
  int vb_i = blockIdx.x, n_i, y0, x0, c_i = 0, s_i = 0;
  decode_tile<G>(a, vb_i, n_i, y0, x0);
  make_loader_plan<COUT, G>(a, lane, y0, x0, pl);
  bool more = true;
  auto issue = [&]() {
    const ChunkSrc cs = chunk_src<COUT, G>(a, c_i, n_i);
    const unsigned dst = lds_addr_of(smem + s_i * C::STAGE_FLOATS);
#pragma unroll
    for (int p = 0; p < C::PIECES; ++p)
      lds_dma16_buf(p < C::IN_PIECES ? cs.img : cs.wgt, pl.voff[p], 0, dst + 1024u * (unsigned)p);
    s_i = s_i == C::NST - 1 ? 0 : s_i + 1;
    if (++c_i == a.n_chunks) {
      c_i = 0;
      vb_i += gridDim.x;
      more = vb_i < a.nwg;
      if (more) {
        decode_tile<G>(a, vb_i, n_i, y0, x0);
        make_loader_plan<COUT, G>(a, lane, y0, x0, pl);
      }
    }
  };
  int ahead = 0;   // chunks issued and not yet handed over
  issue();
  ++ahead;
  if (more) {
    issue();
    ++ahead;
  }
  while (ahead > 0) {
    // the oldest chunk in flight has landed once all but the younger one's pieces are done
    if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");   // hand-over; and everybody has left the stage of the chunk before it
    __builtin_amdgcn_sched_barrier(0);
    --ahead;
    if (more) {
      issue();
      ++ahead;
    }
  }
}

template <int COUT, int EPI>
__global__ __launch_bounds__((ConvCfg<COUT>::THREADS_DMA), kWgPerCu) void conv3x3_mfma_persist_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using G = GeoWide;
  stamp(0);
  fetch_args(a);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_amdgcn_s_setprio(1);
  if (wave == 4) {
    run_loader_persist<COUT, G>(a, smem, tid & 63);
    return;
  }
  if constexpr (COUT == 48) {   // 27 units -> 7,7,7,6 (as conv_tile)
    if (wave < 3) run_role<COUT, G, true, EPI, 1, 0, 7, true>(a, smem, wave, wave, 0, 0, 0, tid);
    else run_role<COUT, G, true, EPI, 3, 7, 2, true>(a, smem, 0, wave, 0, 0, 0, tid);
  } else if constexpr (COUT == 32) {   // 18 -> 5,5,4,4
    if (wave < 2) run_role<COUT, G, true, EPI, 1, 0, 5, true>(a, smem, wave, wave, 0, 0, 0, tid);
    else run_role<COUT, G, true, EPI, 1, 5, 4, true>(a, smem, wave - 2, wave, 0, 0, 0, tid);
  } else {   // 64: 36 -> 9,9,9,9
    static_assert(COUT == 64, "unsupported channel count");
    run_role<COUT, G, true, EPI, 1, 0, 9, true>(a, smem, wave, wave, 0, 0, 0, tid);
  }
}

// Strip tiles (48 output channels, LDS-DMA path): every workgroup looks its tile up in a table of
// ONE image's tiles -- 5 x 16 or 4 x 16 pixels -- and runs the matching instantiation.  ROWS pixel
// groups x 3 cout groups: waves 0..2 own one cout group x the first ROWS-1 pixel groups, wave 3 all
// three cout groups of the last pixel group (15 units -> 4,4,4,3; 12 -> 3,3,3,3).
template <int COUT, int EPI, typename G>
__device__ __forceinline__ void strip_roles(const ConvArgs& a, float* smem, int wave, int n, int y0, int x0, int tid) {
  static_assert(ConvCfg<COUT, G>::LOADER, "the strip kernel is launched with a loader wave");
  static_assert(COUT == 48 || COUT == 32 || COUT == 64, "strip tiles: 32, 48 or 64 output channels");
  if (wave == 4) {
    run_loader<COUT, G>(a, smem, tid & 63, n, y0, x0);
    return;
  }
  if constexpr (COUT == 64) {
    // ROWS pixel groups x 4 cout groups = 20 / 16 units: every wave one cout group x all pixel groups (5,5,5,5 / 4,4,4,4)
    run_role<COUT, G, true, EPI, 1, 0, G::ROWS>(a, smem, wave, wave, n, y0, x0, tid);
  } else if constexpr (COUT == 48) {
    // ROWS pixel groups x 3 cout groups: waves 0..2 own one cout group x the first ROWS-1 pixel groups, wave 3
    // all three cout groups of the last pixel group (15 units -> 4,4,4,3; 12 -> 3,3,3,3)
    if (wave < 3) run_role<COUT, G, true, EPI, 1, 0, G::ROWS - 1>(a, smem, wave, wave, n, y0, x0, tid);
    else run_role<COUT, G, true, EPI, 3, G::ROWS - 1, 1>(a, smem, 0, wave, n, y0, x0, tid);
  } else if constexpr (G::ROWS == 5) {
    // 32 channels, 5 pixel groups x 2 cout groups = 10 units -> 3,3,2,2
    if (wave < 2) run_role<COUT, G, true, EPI, 1, 0, 3>(a, smem, wave, wave, n, y0, x0, tid);
    else if (wave == 2) run_role<COUT, G, true, EPI, 2, 3, 1>(a, smem, 0, wave, n, y0, x0, tid);
    else run_role<COUT, G, true, EPI, 2, 4, 1>(a, smem, 0, wave, n, y0, x0, tid);
  } else {
    // 32 channels, 4 pixel groups x 2 cout groups = 8 units -> 2,2,2,2
    if (wave < 2) run_role<COUT, G, true, EPI, 1, 0, 2>(a, smem, wave, wave, n, y0, x0, tid);
    else run_role<COUT, G, true, EPI, 1, 2, 2>(a, smem, wave - 2, wave, n, y0, x0, tid);
  }
}

template <int COUT, int EPI>
__global__ __launch_bounds__(320, kWgPerCu) void conv3x3_mfma_strip_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  stamp(0);
  fetch_args(a);
  const int tile = xcd_remap(blockIdx.x, a.nwg);
  const int n = div_by_magic(tile, a.magic_tx);   // tiles_x = tiles per image
  const int slot = tile - n * a.tiles_x;
  unsigned e;
  if (a.tab_n) {   // (uniform)
    // a scalar load at a computed offset of the kernarg segment, whose two table lines fetch_args has just pulled into
    // the scalar cache with the other arguments: a cache hit instead of a vector load's trip to L2
    const unsigned pair = (unsigned)a.tab16[slot >> 1];
    const unsigned h = (slot & 1) ? pair >> 16 : pair & 0xffffu;
    e = (h & 0xffu) | (((h >> 8) & 0x7fu) << 16) | ((h >> 15) << 31);   // -> y0 | x0 << 12 | five << 31
  } else {
    e = a.tile_tab[slot];
  }
  const int y0 = (int)(e & 0xfffu), x0 = (int)((e >> 12) & 0xfffu);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_amdgcn_s_setprio(1);
  if (e >> 31) strip_roles<COUT, EPI, GeoS5>(a, smem, wave, n, y0, x0, tid);
  else strip_roles<COUT, EPI, GeoS4>(a, smem, wave, n, y0, x0, tid);
}
template <int COUT>
constexpr size_t kStripLdsBytes = ConvCfg<COUT, GeoS5>::LDS_BYTES_DMA > ConvCfg<COUT, GeoS4>::LDS_BYTES_DMA ? ConvCfg<COUT, GeoS5>::LDS_BYTES_DMA
                                                                                                            : ConvCfg<COUT, GeoS4>::LDS_BYTES_DMA;
static_assert(kWgPerCu * kStripLdsBytes<48> <= 160 * 1024 && kWgPerCu * kStripLdsBytes<64> <= 160 * 1024, "strip workgroups per CU");

// Several INDEPENDENT convolutions of one shape and one epilogue in one launch (blockIdx.y = job).
// A single conv launch at the training shape is one workgroup per CU and spends half of its time
// outside the MFMA loop (launch floor, first-chunk latency, store burst); the kernel is sized for
// two workgroups per CU, so the workgroups of two jobs share each CU and fill each other's
// bubbles (two launches 2 x 17.6 us, one batched launch of two jobs ~27 us).
constexpr int kMaxConvJobs = 4;
struct ConvBatch {
  ConvArgs job[kMaxConvJobs];
};

template <int COUT, int EPI>
__global__ __launch_bounds__(ConvCfg<COUT>::THREADS_DMA, kWgPerCu) void conv3x3_mfma_batch_kernel(ConvBatch b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv_tile<COUT, true, EPI>(b.job[blockIdx.y], smem);
}

// ---------------------------------------------------------------------------------------------
// Weight packing: [Cout][Cin][3][3] (PyTorch layout, models/LarvaNet.py:210) ->
//   fwd  [Cin/8][9][8][CS(Cout)]   wpk[chunk][tap][k][co]  = W[co][chunk*8+k][tap]
//   bwd  [Cout/8][9][8][CS(Cin)]   wpk[chunk][tap][k][ci]  = W[chunk*8+k][ci][8-tap]
// (dgrad is the same convolution with the channel roles swapped and the taps point-mirrored).
// `w_cin_total`/`w_cin_off` select a channel slice of a wider weight (merge conv of LarvaNetV2);
// channels at or beyond w_cin_total pack as zero (the 3-channel head conv padded to 8).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxPackJobs = 64;

struct PackJob {
  const float* w;
  float* fwd;
  float* bwd;
  int cout, cin, w_cin_total, w_cin_off;
};

struct PackBatch {
  PackJob job[kMaxPackJobs];
};

__host__ __device__ constexpr int packed_floats(int cout, int cin) {
  return (cin / kCh) * 9 * kCh * cout_stride(cout);
}

// All layers of a network in one launch: 41 packs per training step would otherwise be 41 launch boundaries for
// 3 us of work each.  One block per (layer, direction, chunk of 8 channels): it reads the chunk's source rows
// (fwd: 72 contiguous floats per output channel; bwd: cin x 9 contiguous floats per output channel of the chunk)
// with consecutive lanes on consecutive floats, transposes them in LDS and writes the chunk's 72 x CS packed rows
// (64 columns per pass: one contiguous run when CS <= 64).  (Until round 2 every thread gathered ONE packed float: three integer divisions and a
// 36- or 1728-byte-strided load each, 7872 blocks per step.)
constexpr int kPackTileCols = 64;                       // columns of a packed row one pass of the LDS tile holds
constexpr int kPackLd = kPackTileCols + 1;
constexpr int kPackTileFloats = 9 * kCh * kPackLd;

__host__ __device__ constexpr int pack_slots(int cout, int cin) { return cin / kCh + cout / kCh; }

__device__ __forceinline__ void pack_block(const PackJob& j, int slot, float* __restrict__ tile) {
  const int nfc = j.cin / kCh;
  const bool fwd = slot < nfc;
  const int chunk = fwd ? slot : slot - nfc;
  float* __restrict__ out = fwd ? j.fwd : j.bwd;
  if ((!fwd && chunk >= j.cout / kCh) || !out) return;   // (whole block)
  const int cols = fwd ? j.cout : j.cin;                 // fastest index of a packed row
  const int cs = cout_stride(cols);
  const int tid = threadIdx.x;
  out += (size_t)chunk * 72 * cs;
  for (int c0 = 0; c0 < cs; c0 += kPackTileCols) {       // 64 columns of the chunk's 72 rows per pass
    const int ncol = min(kPackTileCols, cols - c0);      // source columns of this pass (<= 0: padding only)
    if (fwd) {
      for (int idx = tid; idx < ncol * 72; idx += 256) {
        const int c = idx / 72, r = idx - c * 72, k = r / 9, tap = r - k * 9;
        const int cin_abs = j.w_cin_off + chunk * kCh + k;
        tile[(tap * kCh + k) * kPackLd + c] =
            cin_abs < j.w_cin_total ? j.w[((size_t)(c0 + c) * j.w_cin_total + cin_abs) * 9 + tap] : 0.f;
      }
    } else {
      const int row = ncol * 9;
      for (int idx = tid; idx < kCh * row; idx += 256) {
        const int k = idx / row, rem = idx - k * row, c = rem / 9, tap = rem - c * 9;
        tile[((8 - tap) * kCh + k) * kPackLd + c] =
            j.w_cin_off + c0 + c < j.w_cin_total
                ? j.w[((size_t)(chunk * kCh + k) * j.w_cin_total + j.w_cin_off + c0) * 9 + rem] : 0.f;
      }
    }
    __syncthreads();
    const int wcols = min(kPackTileCols, cs - c0);
    const bool swz = cout_swizzled(cols);   // unpadded rows, odd rows (k odd) with column ^ 16 (larva_common.h)
    for (int idx = tid; idx < 72 * wcols; idx += 256) {
      const int r = idx / wcols, c = idx - r * wcols;
      const int col = swz ? ((c0 + c) ^ ((r & 1) << 4)) : c0 + c;
      out[r * cs + col] = c < ncol ? tile[r * kPackLd + c] : 0.f;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void pack_weights_batch_kernel(PackBatch b) {
  __shared__ float tile[kPackTileFloats];
  pack_block(b.job[blockIdx.y], blockIdx.x, tile);
}

// Everything a training step does before its first convolution, in ONE launch (blockIdx.y = role):
// the kernel-layout weight images of all layers (the optimizer has just changed the weights), the
// head's input zero-padded to 16 channels (HeadFn) and the bicubic x4 base image
// (models/LarvaNet.py:283-285).  As three launches these were 8 + 4 + 7 us of launch-latency-bound
// work in front of the layer chain.
constexpr int kPrologueBlocks = 192;   // blocks per pad / bicubic slice
constexpr int kProloguePadSlices = 2, kPrologueBicSlices = 8;
struct PrologueArgs {
  PackBatch packs;
  int npack, slots;   // pack blocks: npack x slots (slots = the largest pack_slots() of the jobs), the first in the grid
  const float* x;     // [N][C][H][W], C <= 16
  float* x16;         // [N][16][H][W]: channels [0, C) copied (the rest stay zero: written once by the host)
  float* base;        // [N][C][4H][4W]
  int N, C, H, W;
};

__global__ __launch_bounds__(256) void step_prologue_kernel(PrologueArgs a) {
  __shared__ float tile[kPackTileFloats];
  int blk = blockIdx.x;
  if (blk < a.npack * a.slots) {
    const int job = blk / a.slots;
    pack_block(a.packs.job[job], blk - job * a.slots, tile);
    return;
  }
  blk -= a.npack * a.slots;
  constexpr int nthr = kPrologueBlocks * 256;
  const int slice = blk / kPrologueBlocks, tid = (blk - slice * kPrologueBlocks) * 256 + threadIdx.x;
  const int nbic = a.base ? kPrologueBicSlices : 0;
  if (slice < nbic) {
    bicubic4_body(a.x, a.base, a.N * a.C, a.H, a.W, (long long)slice * nthr + tid, (long long)nthr * kPrologueBicSlices);
  } else {
    const int plane = a.H * a.W, per_img = a.C * plane, total = a.N * per_img;
    for (int i = (slice - nbic) * nthr + tid; i < total; i += nthr * kProloguePadSlices) {
      const int n = i / per_img, r = i - n * per_img;
      a.x16[(size_t)n * 16 * plane + r] = a.x[i];
    }
  }
}

// Optional kernel-attached events (hipExtLaunchKernelGGL): start/stop carry the kernel's own
// begin/end timestamps, i.e. the duration a profiler reports, without launch gaps.
struct LaunchTiming {
  hipEvent_t start, stop;
};

template <int COUT, bool VEC, int EPI>
static hipError_t launch_conv_e(const ConvArgs& a, hipStream_t stream, const LaunchTiming* tm) {
  using C = ConvCfg<COUT>;
  constexpr size_t lds = VEC ? C::LDS_BYTES_DMA : C::LDS_BYTES_REG;
  static PerDeviceOnce lds_set;
  if (const hipError_t e = ensure_dynamic_lds(lds_set, reinterpret_cast<const void*>(conv3x3_mfma_kernel<COUT, VEC, EPI>), lds); e != hipSuccess) return e;
  const int grid = a.N * a.tiles_x * a.tiles_y;
  constexpr int threads = VEC ? C::THREADS_DMA : 256;
  if (tm)
    hipExtLaunchKernelGGL((conv3x3_mfma_kernel<COUT, VEC, EPI>), dim3(grid), dim3(threads), lds, stream,
                          tm->start, tm->stop, 0, a);
  else
    hipLaunchKernelGGL((conv3x3_mfma_kernel<COUT, VEC, EPI>), dim3(grid), dim3(threads), lds, stream, a);
  return hipGetLastError();
}

template <int COUT, int EPI>
static hipError_t launch_rows4_e(const ConvArgs& a, hipStream_t stream, const LaunchTiming* tm) {
  using C = ConvCfg<COUT, GeoWide4>;
  constexpr size_t lds = C::LDS_BYTES_DMA;
  static PerDeviceOnce lds_set;
  if (const hipError_t e = ensure_dynamic_lds(lds_set, reinterpret_cast<const void*>(conv3x3_mfma_rows4_kernel<COUT, EPI>), lds); e != hipSuccess) return e;
  if (tm)
    hipExtLaunchKernelGGL((conv3x3_mfma_rows4_kernel<COUT, EPI>), dim3(a.nwg), dim3(C::THREADS_DMA), lds, stream,
                          tm->start, tm->stop, 0, a);
  else
    hipLaunchKernelGGL((conv3x3_mfma_rows4_kernel<COUT, EPI>), dim3(a.nwg), dim3(C::THREADS_DMA), lds, stream, a);
  return hipGetLastError();
}

// workgroup slots of the device for the 16-byte-path kernels (kWgPerCu per CU)
static int conv_slots() { return kWgPerCu * device_cu_count(); }

template <int COUT, int EPI>
static hipError_t launch_persist_e(const ConvArgs& a, hipStream_t stream, const LaunchTiming* tm) {
  using C = ConvCfg<COUT>;
  constexpr size_t lds = C::LDS_BYTES_DMA;
  static PerDeviceOnce lds_set;
  if (const hipError_t e = ensure_dynamic_lds(lds_set, reinterpret_cast<const void*>(conv3x3_mfma_persist_kernel<COUT, EPI>), lds); e != hipSuccess) return e;
  const int grid = a.nwg < conv_slots() ? a.nwg : conv_slots();
  if (tm)
    hipExtLaunchKernelGGL((conv3x3_mfma_persist_kernel<COUT, EPI>), dim3(grid), dim3(C::THREADS_DMA), lds, stream, tm->start, tm->stop, 0, a);
  else
    hipLaunchKernelGGL((conv3x3_mfma_persist_kernel<COUT, EPI>), dim3(grid), dim3(C::THREADS_DMA), lds, stream, a);
  return hipGetLastError();
}

// persistent tiles exist for the epilogues of an inference forward; anything else: hipErrorNotSupported
template <int COUT>
static hipError_t launch_persist(const ConvArgs& a, int epi, hipStream_t stream, const LaunchTiming* tm) {
  // (the residual / mask operands of these kernels come by buffer loads with 32-bit byte offsets inside one image)
  if ((long long)COUT * a.H * a.pitch * 4 >= (1ll << 31)) return hipErrorNotSupported;
  switch (epi) {
    case kEpiPlain: return launch_persist_e<COUT, kEpiPlain>(a, stream, tm);
    case kEpiRelu: return launch_persist_e<COUT, kEpiRelu>(a, stream, tm);
    case kEpiRes1: return launch_persist_e<COUT, kEpiRes1>(a, stream, tm);
    case kEpiRes2: return launch_persist_e<COUT, kEpiRes2>(a, stream, tm);
    case kEpiShuffle: return launch_persist_e<COUT, kEpiShuffle>(a, stream, tm);
    case kEpiShuffleBase: return launch_persist_e<COUT, kEpiShuffleBase>(a, stream, tm);
    default: return hipErrorNotSupported;
  }
}

// 4 x 48 tiles exist for the epilogues of an inference forward (head, conv + ReLU, the two residual forms, the
// pixel-shuffle exits); anything else: hipErrorNotSupported, the caller launches the 3 x 48 tiles
template <int COUT>
static hipError_t launch_rows4(const ConvArgs& a, int epi, hipStream_t stream, const LaunchTiming* tm) {
  switch (epi) {
    case kEpiPlain: return launch_rows4_e<COUT, kEpiPlain>(a, stream, tm);
    case kEpiRelu: return launch_rows4_e<COUT, kEpiRelu>(a, stream, tm);
    case kEpiRes1: return launch_rows4_e<COUT, kEpiRes1>(a, stream, tm);
    case kEpiRes2: return launch_rows4_e<COUT, kEpiRes2>(a, stream, tm);
    case kEpiShuffle: return launch_rows4_e<COUT, kEpiShuffle>(a, stream, tm);
    case kEpiShuffleBase: return launch_rows4_e<COUT, kEpiShuffleBase>(a, stream, tm);
    default: return hipErrorNotSupported;
  }
}

template <int COUT, bool VEC>
static hipError_t launch_conv_v(const ConvArgs& a, int epi, hipStream_t stream, const LaunchTiming* tm) {
  switch (epi) {
    case kEpiPlain: return launch_conv_e<COUT, VEC, kEpiPlain>(a, stream, tm);
    case kEpiRelu: return launch_conv_e<COUT, VEC, kEpiRelu>(a, stream, tm);
    case kEpiMask: return launch_conv_e<COUT, VEC, kEpiMask>(a, stream, tm);
    case kEpiRes1: return launch_conv_e<COUT, VEC, kEpiRes1>(a, stream, tm);
    case kEpiRes2: return launch_conv_e<COUT, VEC, kEpiRes2>(a, stream, tm);
    case kEpiShuffle: return launch_conv_e<COUT, VEC, kEpiShuffle>(a, stream, tm);
    case kEpiShuffleBase: return launch_conv_e<COUT, VEC, kEpiShuffleBase>(a, stream, tm);
    default: return hipErrorInvalidValue;
  }
}

template <int COUT>
static hipError_t launch_conv(const ConvArgs& a, bool vec, int epi, hipStream_t stream, const LaunchTiming* tm) {
  return vec ? launch_conv_v<COUT, true>(a, epi, stream, tm) : launch_conv_v<COUT, false>(a, epi, stream, tm);
}

template <int COUT, int EPI>
static hipError_t launch_batch_e(const ConvBatch& b, int njobs, hipStream_t stream) {
  using C = ConvCfg<COUT>;
  static PerDeviceOnce lds_set;
  if (const hipError_t e = ensure_dynamic_lds(lds_set, reinterpret_cast<const void*>(conv3x3_mfma_batch_kernel<COUT, EPI>), C::LDS_BYTES_DMA); e != hipSuccess) return e;
  const ConvArgs& a = b.job[0];
  hipLaunchKernelGGL((conv3x3_mfma_batch_kernel<COUT, EPI>), dim3(a.N * a.tiles_x * a.tiles_y, njobs),
                     dim3(C::THREADS_DMA), C::LDS_BYTES_DMA, stream, b);
  return hipGetLastError();
}

template <int COUT>
static hipError_t launch_batch(const ConvBatch& b, int njobs, int epi, hipStream_t stream) {
  switch (epi) {
    case kEpiPlain: return launch_batch_e<COUT, kEpiPlain>(b, njobs, stream);
    case kEpiRelu: return launch_batch_e<COUT, kEpiRelu>(b, njobs, stream);
    case kEpiMask: return launch_batch_e<COUT, kEpiMask>(b, njobs, stream);
    case kEpiRes1: return launch_batch_e<COUT, kEpiRes1>(b, njobs, stream);
    case kEpiRes2: return launch_batch_e<COUT, kEpiRes2>(b, njobs, stream);
    case kEpiShuffle: return launch_batch_e<COUT, kEpiShuffle>(b, njobs, stream);
    case kEpiShuffleBase: return launch_batch_e<COUT, kEpiShuffleBase>(b, njobs, stream);
    case kEpiShuffleL1:
      if constexpr (COUT == 48) return launch_batch_e<COUT, kEpiShuffleL1>(b, njobs, stream);
      return hipErrorNotSupported;
    default: return hipErrorInvalidValue;
  }
}

template <int COUT, int EPI>
static hipError_t launch_strip_e(const ConvArgs& a, hipStream_t stream, const LaunchTiming* tm) {
  constexpr size_t lds = kStripLdsBytes<COUT>;
  static PerDeviceOnce lds_set;
  if (const hipError_t e = ensure_dynamic_lds(lds_set, reinterpret_cast<const void*>(conv3x3_mfma_strip_kernel<COUT, EPI>), lds); e != hipSuccess) return e;
  if (tm)
    hipExtLaunchKernelGGL((conv3x3_mfma_strip_kernel<COUT, EPI>), dim3(a.nwg), dim3(320), lds, stream,
                          tm->start, tm->stop, 0, a);
  else
    hipLaunchKernelGGL((conv3x3_mfma_strip_kernel<COUT, EPI>), dim3(a.nwg), dim3(320), lds, stream, a);
  return hipGetLastError();
}

template <int COUT>
static hipError_t launch_strip(const ConvArgs& a, int epi, hipStream_t stream, const LaunchTiming* tm = nullptr) {
  switch (epi) {
    case kEpiPlain: return launch_strip_e<COUT, kEpiPlain>(a, stream, tm);
    case kEpiRelu: return launch_strip_e<COUT, kEpiRelu>(a, stream, tm);
    case kEpiMask: return launch_strip_e<COUT, kEpiMask>(a, stream, tm);
    case kEpiRes1: return launch_strip_e<COUT, kEpiRes1>(a, stream, tm);
    case kEpiRes2: return launch_strip_e<COUT, kEpiRes2>(a, stream, tm);
    case kEpiShuffle: return launch_strip_e<COUT, kEpiShuffle>(a, stream, tm);
    case kEpiShuffleBase: return launch_strip_e<COUT, kEpiShuffleBase>(a, stream, tm);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace larva

using namespace larva;

extern "C" {

// Number of floats of one packed weight image for a conv with `cin` input channels (multiple of
// 8) and `cout` output channels (32, 48 or 64).
long long larva_packed_weight_floats(int cout, int cin) { return (long long)packed_floats(cout, cin); }

// njobs (<= 64) packs in one launch; arrays are host arrays of length njobs.
static int fill_pack_jobs(PackBatch& b, const float* const* w, float* const* wpk_fwd, float* const* wpk_bwd,
                          const int* cout, const int* cin, const int* w_cin_total, const int* w_cin_off, int njobs) {
  int slots = 0;   // -> blocks per job; < 0: invalid
  for (int i = 0; i < njobs; ++i) {
    if (!w[i] || cout[i] % kCh || cin[i] % kCh || cout[i] <= 0 || cin[i] <= 0) return -1;
    b.job[i] = PackJob{w[i], wpk_fwd[i], wpk_bwd[i], cout[i], cin[i], w_cin_total[i], w_cin_off[i]};
    const int n = pack_slots(cout[i], cin[i]);
    slots = n > slots ? n : slots;
  }
  return slots;
}

int larva_pack_weights_batch(const float* const* w, float* const* wpk_fwd, float* const* wpk_bwd,
                             const int* cout, const int* cin, const int* w_cin_total,
                             const int* w_cin_off, int njobs, void* stream) {
  if (njobs < 1 || njobs > kMaxPackJobs) return (int)hipErrorInvalidValue;
  PackBatch b{};
  const int slots = fill_pack_jobs(b, w, wpk_fwd, wpk_bwd, cout, cin, w_cin_total, w_cin_off, njobs);
  if (slots < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(slots, njobs), dim3(256), 0, (hipStream_t)stream, b);
  return (int)hipGetLastError();
}

// Training-step prologue in one launch: larva_pack_weights_batch of njobs (<= 64) weights + the head's
// 16-channel padded copy of x ([N][C][H][W], C <= 16; x16 [N][16][H][W], channels >= C untouched) + the
// bicubic x4 base image of x (base [N][C][4H][4W]).  x16 / base may be NULL to skip that part.
int larva_step_prologue(const float* const* w, float* const* wpk_fwd, float* const* wpk_bwd, const int* cout,
                        const int* cin, const int* w_cin_total, const int* w_cin_off, int njobs, const float* x,
                        float* x16, float* base, int N, int C, int H, int W, void* stream) {
  if (njobs < 0 || njobs > kMaxPackJobs || !x || N <= 0 || C <= 0 || C > 16 || H <= 0 || W <= 0)
    return (int)hipErrorInvalidValue;
  if ((long long)N * 16 * H * W >= (1ll << 31)) return (int)hipErrorInvalidValue;
  if (base && (reinterpret_cast<uintptr_t>(base) & 15)) return (int)hipErrorInvalidValue;
  PrologueArgs a{};
  a.slots = njobs ? fill_pack_jobs(a.packs, w, wpk_fwd, wpk_bwd, cout, cin, w_cin_total, w_cin_off, njobs) : 0;
  if (a.slots < 0) return (int)hipErrorInvalidValue;
  a.npack = njobs;
  a.x = x; a.x16 = x16; a.base = base;
  a.N = N; a.C = C; a.H = H; a.W = W;
  if (!x16 && base) return (int)hipErrorInvalidValue;   // (slices are positional: the bicubic ones precede the pad ones)
  const int blocks = njobs * a.slots + kPrologueBlocks * ((base ? kPrologueBicSlices : 0) + (x16 ? kProloguePadSlices : 0));
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(step_prologue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int larva_pack_weights(const float* w, float* wpk_fwd, float* wpk_bwd, int cout, int cin,
                       int w_cin_total, int w_cin_off, void* stream) {
  return larva_pack_weights_batch(&w, &wpk_fwd, &wpk_bwd, &cout, &cin, &w_cin_total, &w_cin_off, 1, stream);
}

#if LARVA_DIAG & 512
static int g_diag_next_slot = -1, g_diag_slot_cap = 0;
static unsigned long long* g_diag_host_base = nullptr;   // larva_diag_set_stamps' buffer, as the host knows it
static int diag_take_slot() {
  if (g_diag_next_slot < 0 || g_diag_next_slot >= g_diag_slot_cap) return -1;
  return g_diag_next_slot++;
}
// Arm the per-launch stamp areas: launches built from now on take slots first, first + 1, ... < cap (first < 0: off).
// Returns the next slot that would be handed out.
int larva_diag_arm_slots(int first, int cap) {
  const int was = g_diag_next_slot;
  g_diag_next_slot = first;
  g_diag_slot_cap = cap;
  return was;
}
int larva_diag_next_slot(void) { return g_diag_next_slot; }
#endif

// Fused 3x3 convolution.  `src` is an array of `n_src` device pointers (channel concatenation,
// each tensor [N][cin_per_src][H][W]); `wpk` a packed weight image from larva_pack_weights for
// (cout, n_src*cin_per_src).  Epilogue, in this order: relu -> mask -> +res0 -> +res1 -> store
// (mode 0, [N][cout][H][W]) or pixel-shuffle(4) store with optional +base (mode 1,
// [N][cout/16][4H][4W]).  Stream-ordered, never allocates or synchronises.
static int conv_build(const float* const* src, int n_src, int cin_per_src, const float* wpk, const float* bias,
                      const float* res0, const float* res1, const float* mask, const float* base, float* out,
                      int N, int H, int W, int pitch, int relu, int mode, ConvArgs& a, bool& aligned, int& epi) {
  if (pitch == 0) pitch = W;
  if (pitch < W) return (int)hipErrorInvalidValue;
  if (n_src < 1 || n_src > kMaxSrc || cin_per_src % kCh || cin_per_src <= 0 || N <= 0 || H <= 0 || W <= 0)
    return (int)hipErrorInvalidValue;
  if (mode != 0 && mode != 1) return (int)hipErrorInvalidValue;
  if ((long long)cin_per_src * H * pitch >= (1ll << 31)) return (int)hipErrorInvalidValue;  // 32-bit lane offsets
  if (!wpk || !out) return (int)hipErrorInvalidValue;
  a = ConvArgs{};
#if LARVA_DIAG & 512
  {
    const int slot = diag_take_slot();
    a.diag_area = (slot >= 0 && g_diag_host_base) ? g_diag_host_base + (size_t)slot * kDiagWgPerSlot * 16 : nullptr;
  }
#endif
  aligned = (pitch % 4 == 0) && ((reinterpret_cast<uintptr_t>(wpk) & 15) == 0);
  for (int i = 0; i < n_src; ++i) {
    if (!src[i]) return (int)hipErrorInvalidValue;
    a.src[i] = src[i];
    aligned = aligned && ((reinterpret_cast<uintptr_t>(src[i]) & 15) == 0);
  }
  // (16-byte accesses to the mode-0 output and to the mask / residual operands)
  aligned = aligned && (((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res0) |
                          reinterpret_cast<uintptr_t>(res1) | reinterpret_cast<uintptr_t>(mask)) & 15) == 0);
  a.wpk = wpk; a.bias = bias; a.res0 = res0; a.res1 = res1; a.mask = mask; a.base = base;
  a.out = out;
  a.cin_per_src = cin_per_src;
  a.n_chunks = n_src * cin_per_src / kCh;
  a.N = N; a.H = H; a.W = W; a.pitch = pitch;
  a.tiles_x = (pitch + kTileCols - 1) / kTileCols;
  a.tiles_y = (H + kTileRows - 1) / kTileRows;
  if ((long long)N * a.tiles_x * a.tiles_y >= (1ll << 20)) return (int)hipErrorInvalidValue;  // div_by_magic range
  a.magic_tx = div_magic(a.tiles_x);
  a.magic_ty = div_magic(a.tiles_y);
  a.magic_cps = div_magic(cin_per_src / kCh);
  a.nwg = N * a.tiles_x * a.tiles_y;
  // Map the requested fusion onto a compiled epilogue (relu -> mask -> +res0 -> +res1).
  if (mode == 1) {
    if (relu || mask || res0 || res1) return (int)hipErrorInvalidValue;
    epi = base ? kEpiShuffleBase : kEpiShuffle;
  } else {
    if (base) return (int)hipErrorInvalidValue;
    const int code = (relu ? 1 : 0) | (mask ? 2 : 0) | (res0 ? 4 : 0) | (res1 ? 8 : 0);
    switch (code) {
      case 0: epi = kEpiPlain; break;
      case 1: epi = kEpiRelu; break;
      case 2: epi = kEpiMask; break;
      case 4: epi = kEpiRes1; break;
      case 12: epi = kEpiRes2; break;
      default: return (int)hipErrorInvalidValue;
    }
  }
  return 0;
}

// Tile height of a whole-tensor launch: 3 (the tile every shape has) or 4 (16-byte path, 48 / 32 channels, inference
// epilogues).  Measured on a 339 x 510 image (tools/ab_tile_rows.py, profiles/r04_ab_tile_rows.txt): at 32 channels
// 4 x 48 tiles take 32.9 instead of 39.1 us per layer (0.62 against 0.52 of the matrix peak: the short K loop of a
// 32-channel layer amortises its prologue / epilogue over 6 instead of 4-5 MFMAs per k-step and wave); at 48 channels
// 65.1 against 64.0 us -- the "fewer rounds" argument (935 instead of 1243 tiles) does not hold, workgroups are placed
// as slots free up, not in rounds.  So: 4 rows for 32-channel launches of more than one round's worth of tiles.
static int conv_tile_rows(int N, int H, int pitch, int cout, int epi, bool aligned, int forced) {
  const bool can4 = aligned && (cout == 48 || cout == 32) &&
                    (epi == kEpiPlain || epi == kEpiRelu || epi == kEpiRes1 || epi == kEpiRes2 || epi == kEpiShuffle || epi == kEpiShuffleBase);
  if (forced == 3 || !can4) return 3;
  if (forced == 4) return 4;
  const long long tiles3 = (long long)N * ((pitch + kTileCols - 1) / kTileCols) * ((H + 2) / 3);
  return (cout == 32 && tiles3 > conv_slots()) ? 4 : 3;
}

static int conv_dispatch(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                         const float* bias, const float* res0, const float* res1, const float* mask,
                         const float* base, float* out, int N, int cout, int H, int W, int pitch, int relu,
                         int mode, void* stream, const LaunchTiming* tm, int tile_rows = 0) {
  ConvArgs a;
  bool aligned;
  int epi;
  const int rc = conv_build(src, n_src, cin_per_src, wpk, bias, res0, res1, mask, base, out, N, H, W, pitch, relu,
                            mode, a, aligned, epi);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (tile_rows != 0 && tile_rows != 3 && tile_rows != 4) return (int)hipErrorInvalidValue;
  if (conv_tile_rows(N, H, a.pitch, cout, epi, aligned, tile_rows) == 4) {
    a.tiles_y = (H + 3) / 4;
    a.magic_ty = div_magic(a.tiles_y);
    a.nwg = N * a.tiles_x * a.tiles_y;
#if !LARVA_DIAG_ONLY48
    if (cout == 32) return (int)launch_rows4<32>(a, epi, s, tm);
#endif
    return (int)launch_rows4<48>(a, epi, s, tm);
  }
  if (tile_rows == 4) return (int)hipErrorNotSupported;
  // more tiles than workgroup slots (a full image): one persistent workgroup per slot (conv3x3_mfma_persist_kernel)
  // (LARVA_PERSIST=0: one workgroup per tile as before -- read per call, the tests compare the two)
  const char* pe = getenv("LARVA_PERSIST");
  const bool persist_on = !(pe && pe[0] == '0');
  if (persist_on && aligned && a.nwg > conv_slots()) {
    hipError_t e = hipErrorNotSupported;
    switch (cout) {
#if !LARVA_DIAG_ONLY48
      case 32: e = launch_persist<32>(a, epi, s, tm); break;
      case 64: e = launch_persist<64>(a, epi, s, tm); break;
#endif
      case 48: e = launch_persist<48>(a, epi, s, tm); break;
      default: break;
    }
    if (e != hipErrorNotSupported) return (int)e;
  }
  switch (cout) {
#if !LARVA_DIAG_ONLY48
    case 32: return (int)launch_conv<32>(a, aligned, epi, s, tm);
    case 64: return (int)launch_conv<64>(a, aligned, epi, s, tm);
#endif
    case 48: return (int)launch_conv<48>(a, aligned, epi, s, tm);
    default: return (int)hipErrorInvalidValue;
  }
}

int larva_conv3x3_fwd(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                      const float* bias, const float* res0, const float* res1, const float* mask,
                      const float* base, float* out, int N, int cout, int H, int W, int relu,
                      int mode, void* stream) {
  return conv_dispatch(src, n_src, cin_per_src, wpk, bias, res0, res1, mask, base, out, N, cout, H, W, 0, relu,
                       mode, stream, nullptr);
}

// Same, for tensors whose rows are `pitch` >= W floats apart (every [..][H][W] operand: src,
// res0, res1, mask and the mode-0 output).  Columns [W, pitch) of the inputs must hold zeros and
// are written as zeros, so a width that is not a multiple of 4 can still use the 16-byte
// LDS-DMA staging path (pitch = W rounded up to 4).  The pixel-shuffle output stays [..][4H][4W].
int larva_conv3x3_fwd_pitched(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                              const float* bias, const float* res0, const float* res1, const float* mask,
                              const float* base, float* out, int N, int cout, int H, int W, int pitch,
                              int relu, int mode, void* stream) {
  return conv_dispatch(src, n_src, cin_per_src, wpk, bias, res0, res1, mask, base, out, N, cout, H, W, pitch,
                       relu, mode, stream, nullptr);
}

// ... and with the tile height of a whole-tensor launch chosen by the caller: tile_rows 0 = the library's choice (3 x 48
// tiles; 4 x 48 for 32-channel launches of more tiles than workgroup slots), 3 / 4 = that height (tests, A/B timing;
// 4 rows exist on the 16-byte path at 48 / 32 channels for the epilogues of an inference forward: hipErrorNotSupported
// otherwise).  Results do not depend on the tiling.
int larva_conv3x3_fwd_tiled(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                            const float* bias, const float* res0, const float* res1, const float* mask,
                            const float* base, float* out, int N, int cout, int H, int W, int pitch,
                            int relu, int mode, int tile_rows, void* stream) {
  return conv_dispatch(src, n_src, cin_per_src, wpk, bias, res0, res1, mask, base, out, N, cout, H, W, pitch,
                       relu, mode, stream, nullptr, tile_rows);
}

// njobs (2..4) INDEPENDENT convolutions of one shape and one fusion in ONE launch (their workgroups
// share the CUs two by two).  src: njobs * n_src pointers (job-major); every other operand an array
// of njobs pointers, or NULL when no job uses it (a fusion is either used by all jobs or by none).
// Only the 16-byte staging path (pitch % 4 == 0, 16-byte aligned tensors): otherwise
// hipErrorNotSupported, and the caller issues the jobs one by one.
int larva_conv3x3_fwd_batch(int njobs, const float* const* src, int n_src, int cin_per_src,
                            const float* const* wpk, const float* const* bias, const float* const* res0,
                            const float* const* res1, const float* const* mask, const float* const* base,
                            float* const* out, int N, int cout, int H, int W, int pitch, int relu, int mode,
                            void* stream) {
  if (njobs < 2 || njobs > kMaxConvJobs || !src || !wpk || !out) return (int)hipErrorInvalidValue;
  ConvBatch b{};
  int epi0 = -1;
  for (int j = 0; j < njobs; ++j) {
    bool aligned;
    int epi;
    const int rc = conv_build(src + (size_t)j * n_src, n_src, cin_per_src, wpk[j], bias ? bias[j] : nullptr,
                              res0 ? res0[j] : nullptr, res1 ? res1[j] : nullptr, mask ? mask[j] : nullptr,
                              base ? base[j] : nullptr, out[j], N, H, W, pitch, relu, mode, b.job[j], aligned, epi);
    if (rc) return rc;
    if (!aligned) return (int)hipErrorNotSupported;
    if (j > 0 && epi != epi0) return (int)hipErrorInvalidValue;
    epi0 = epi;
  }
  hipStream_t s = (hipStream_t)stream;
  switch (cout) {
#if !LARVA_DIAG_ONLY48
    case 32: return (int)launch_batch<32>(b, njobs, epi0, s);
    case 64: return (int)launch_batch<64>(b, njobs, epi0, s);
#endif
    case 48: return (int)launch_batch<48>(b, njobs, epi0, s);
    default: return (int)hipErrorInvalidValue;
  }
}

// njobs (2..4) EXITS of the training step in one launch (models/LarvaNet.py:104-109 for several i):
// out_j = PixelShuffle(4)(conv(src_j)) + base_j scored against truth_j by nn.L1Loss, without a pass
// over the images: every MFMA wave adds up |out - truth| of its share of the tile
// (partial[j]: 4 * N * tiles floats, see larva_exit_l1_partials) and writes
// grad[j] = sign(out - truth) * gval in the pixel-unshuffled [N][cout][H][pitch] layout the leg's
// dgrad / wgrad read (= larva_l1_bwd_unshuffle4 bit for bit).  out[j] may be NULL (image not wanted).
// cout = 48, 16-byte staging path only (hipErrorNotSupported otherwise).
int larva_exit_l1_partials(int N, int H, int pitch) {
  return 4 * N * ((pitch + kTileCols - 1) / kTileCols) * ((H + kTileRows - 1) / kTileRows);
}

int larva_conv3x3_exit_l1_batch(int njobs, const float* const* src, int n_src, int cin_per_src,
                                const float* const* wpk, const float* const* bias, const float* const* base,
                                const float* const* truth, float* const* out, float* const* grad,
                                float* const* partial, float gval, int N, int cout, int H, int W, int pitch,
                                void* stream) {
  if (njobs < 2 || njobs > kMaxConvJobs || !src || !wpk || !base || !truth || !out || !grad || !partial)
    return (int)hipErrorInvalidValue;
  if (cout != 48) return (int)hipErrorNotSupported;
  ConvBatch b{};
  for (int j = 0; j < njobs; ++j) {
    bool aligned;
    int epi;
    if (!base[j] || !truth[j] || !grad[j] || !partial[j]) return (int)hipErrorInvalidValue;
    const int rc = conv_build(src + (size_t)j * n_src, n_src, cin_per_src, wpk[j], bias ? bias[j] : nullptr, nullptr,
                              nullptr, nullptr, base[j], grad[j], N, H, W, pitch, 0, 1, b.job[j], aligned, epi);
    if (rc) return rc;
    if (!aligned || ((reinterpret_cast<uintptr_t>(truth[j]) | reinterpret_cast<uintptr_t>(base[j])) & 15))
      return (int)hipErrorNotSupported;
    b.job[j].out = out[j];
    b.job[j].truth = truth[j];
    b.job[j].grad = grad[j];
    b.job[j].partial = partial[j];
    b.job[j].gval = gval;
  }
  return (int)launch_batch<48>(b, njobs, kEpiShuffleL1, (hipStream_t)stream);
}

// Strip tiles.  larva_strip_tile_table fills `tab` (host memory, `cap` entries) with the tiles of ONE
// H x W image -- column strips 16 pixels wide, each cut into 5-row and 4-row tiles (their counts kept
// as equal as the heights allow: a 5-row and a 4-row tile sharing a CU issue 4 + 3 MFMAs per k-step on
// every SIMD, exactly what the 3 x 48 tile does) -- and returns their number (if it exceeds `cap`
// nothing beyond `cap` is written), or a negative value for an H that cannot be cut (1, 2, 3, 6, 7,
// 11).  Entry = y0 | x0 << 12 | (5 rows ? 1u << 31 : 0).  phase 0 / 1: the table starts with a
// 5-row / 4-row tile.
int larva_strip_tile_table(int H, int W, int phase, unsigned* tab, int cap) {
  if (H <= 0 || W <= 0 || H >= 4096 || W >= 4096 || !tab || cap < 0) return -1;
  // pass 1: how each 16-column strip is cut (5 a + 4 b = H, the running counts of the two heights kept close)
  int n5 = 0, n4 = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int fives = 0, fours = 0, i5 = 0, i4 = 0;
    for (int x0 = 0; x0 < W; x0 += 16) {
      int best_a = -1, best_gap = 1 << 30;
      for (int a5 = 0; 5 * a5 <= H; ++a5) {
        if ((H - 5 * a5) % 4) continue;
        const int b4 = (H - 5 * a5) / 4;
        const int gap = (fives + a5) - (fours + b4);
        const int g = gap < 0 ? -gap : gap;
        if (g < best_gap) { best_gap = g; best_a = a5; }
      }
      if (best_a < 0) return -2;
      const int a5 = best_a, b4 = (H - 5 * a5) / 4;
      fives += a5; fours += b4;
      if (pass == 0) continue;
      // pass 2: table slots.  The two heights ALTERNATE along the table as far as their counts allow
      // (slot 2k = the k-th tile of the kind `phase` starts with, slot 2k+1 = the k-th of the other
      // kind, the surplus kind behind): two launches whose tables start with different kinds put
      // tiles of different heights at equal block indices.
      int y = 0, ra = a5, rb = b4;
      while (ra > 0 || rb > 0) {
        const bool five = ra > 0 && (rb == 0 || (long long)ra * b4 >= (long long)rb * a5);
        const int k = five ? i5++ : i4++;
        const int mine = five ? n5 : n4, other = five ? n4 : n5;
        const bool first = five == (phase == 0);   // does this kind take the even slots?
        int slot;
        if (k < other) slot = 2 * k + (first ? 0 : 1);
        else slot = 2 * other + (k - other);       // surplus of the larger kind
        (void)mine;
        if (slot < cap) tab[slot] = (unsigned)y | ((unsigned)x0 << 12) | (five ? 0x80000000u : 0u);
        y += five ? 5 : 4;
        if (five) --ra; else --rb;
      }
    }
    n5 = fives; n4 = fours;
  }
  return n5 + n4;
}

// larva_conv3x3_fwd_pitched on strip tiles: `tile_tab` = DEVICE copy of larva_strip_tile_table(H,
// pitch) with `tiles_per_image` entries, `tile_tab_host` = the HOST array larva_strip_tile_table filled (the same
// entries) or NULL: given, and small enough (<= 64 tiles, H <= 256, pitch <= 2048), the table travels inside the kernel
// arguments and a workgroup finds its tile without a dependent memory round trip (ConvArgs::tab16).  plain_stores:
// mode-0 output written with plain instead of non-temporal stores (faster when the next launch reads it at once, see
// the kernel's epilogue).  cout = 32, 48 or 64 and the 16-byte staging path only (pitch % 4 == 0, 16-byte aligned
// tensors), otherwise hipErrorNotSupported.  Results are bit-identical to larva_conv3x3_fwd_pitched: every output's K
// loop runs in the same order, only the assignment of pixels to workgroups differs.
static int strips_dispatch(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                           const float* bias, const float* res0, const float* res1, const float* mask,
                           const float* base, float* out, int N, int cout, int H, int W, int pitch,
                           int relu, int mode, const unsigned* tile_tab, const unsigned* tile_tab_host,
                           int tiles_per_image, int plain_stores, void* stream, const LaunchTiming* tm) {
  if (cout != 48 && cout != 32 && cout != 64) return (int)hipErrorNotSupported;
  if (!tile_tab || tiles_per_image < 1) return (int)hipErrorInvalidValue;
  ConvArgs a;
  bool aligned;
  int epi;
  const int rc = conv_build(src, n_src, cin_per_src, wpk, bias, res0, res1, mask, base, out, N, H, W, pitch, relu,
                            mode, a, aligned, epi);
  if (rc) return rc;
  if (!aligned) return (int)hipErrorNotSupported;
  if ((long long)N * tiles_per_image >= (1ll << 20)) return (int)hipErrorInvalidValue;  // div_by_magic range
  // (the strip kernel's residual / mask operands come by buffer loads with 32-bit byte offsets inside one image)
  if ((long long)cout * H * a.pitch * 4 >= (1ll << 31)) return (int)hipErrorNotSupported;
  a.tile_tab = tile_tab;
  if (tile_tab_host && tiles_per_image <= 64 && H <= 256 && pitch <= 2048) {
    // the same table inline in the kernel arguments, 16 bits per tile (ConvArgs::tab16)
    a.tab_n = tiles_per_image;
    for (int t = 0; t < tiles_per_image; ++t) {
      const unsigned e = tile_tab_host[t];
      const unsigned y0 = e & 0xfffu, x0 = (e >> 12) & 0xfffu;
      if (y0 > 255u || (x0 & 15u) || (x0 >> 4) > 127u) { a.tab_n = 0; break; }
      const unsigned h = y0 | ((x0 >> 4) << 8) | ((e >> 31) << 15);
      a.tab16[t >> 1] |= (int)(h << ((t & 1) * 16));
    }
  }
  a.plain_stores = plain_stores ? 1 : 0;
  a.tiles_x = tiles_per_image;
  a.tiles_y = 1;
  a.magic_tx = div_magic(tiles_per_image);
  a.magic_ty = div_magic(1);
  a.nwg = N * tiles_per_image;
#if !LARVA_DIAG_ONLY48
  if (cout == 32) return (int)launch_strip<32>(a, epi, (hipStream_t)stream, tm);
  if (cout == 64) return (int)launch_strip<64>(a, epi, (hipStream_t)stream, tm);
#endif
  return (int)launch_strip<48>(a, epi, (hipStream_t)stream, tm);
}

int larva_conv3x3_fwd_strips(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                             const float* bias, const float* res0, const float* res1, const float* mask,
                             const float* base, float* out, int N, int cout, int H, int W, int pitch,
                             int relu, int mode, const unsigned* tile_tab, const unsigned* tile_tab_host,
                             int tiles_per_image, int plain_stores, void* stream) {
  return strips_dispatch(src, n_src, cin_per_src, wpk, bias, res0, res1, mask, base, out, N, cout, H, W, pitch, relu, mode,
                         tile_tab, tile_tab_host, tiles_per_image, plain_stores, stream, nullptr);
}

#if LARVA_DIAG & 32
int larva_diag_set_stamps(unsigned long long* buf) {
#if LARVA_DIAG & 512
  g_diag_host_base = buf;
#endif
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(larva::g_stamps), &buf, sizeof(buf));
}
#endif

}  // extern "C"
