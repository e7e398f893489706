// north_star: "MFMA used for the 32-64-channel im2col-as-GEMM contraction where rocprof shows it beats the direct path".
// The K = 27 head has its rocprof A/B (profiles/r02_b_head_*).  For the 48 -> 48 body the question is not rhetorical:
// MI355X's fp32 VALU peak (v_pk_fma_f32: 2 FMAs per lane per issue, 64 FLOP / clk / SIMD) EQUALS its fp32 MFMA peak
// (v_mfma_f32_16x16x4_f32: 2048 FLOP per 32 clk).  This probe times the INNER LOOP of a direct 48-cout tile the way
// tools/probe_mfma_rate.hip times the MFMA loop (one wave per SIMD, s_memtime ticks), reported per 16x16x4-EQUIVALENT
// (= 1024 MACs per wave = 8 v_pk_fma_f32), so that the figures stand beside the 32 (pipe rate) / 34.7 (the conv
// kernel's k-step with its LDS operand reads) of the MFMA loop:
//   direct tile of one wave: 64 lanes x PX pixels per lane x 48 output channels; per k (= one input channel x tap)
//   24 v_pk_fma_f32 per pixel: acc[2c:2c+1] += {w[2c], w[2c+1]} (an SGPR pair) * {x, x} (one VGPR, op_sel broadcast)
//   variants:  regs     weights held in SGPRs (no scalar loads in the loop), activations in registers: the issue rate
//              stream   weights streamed by s_load_dwordx16 from an 83 KB image (48 x 48 x 9 floats, as a layer's),
//                       activations by ds_read from LDS: the realistic loop
//              hybrid   the stream loop on waves 4-7 of a 512-thread workgroup whose waves 0-3 run the MFMA loop
//                       (7 MFMAs + 8 ds_read_b32 per step) on the SAME SIMDs: do the two pipes add up?
//   hipcc --offload-arch=gfx950 -O3 -o probe_valu_conv tools/probe_valu_conv.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// acc += {w.lo, w.hi} * {x, x}: src0 = SGPR pair, src1 = a VGPR pair of which only the LOW dword is used for both halves
__device__ __forceinline__ void pk_fma_sw(f32x2& acc, unsigned long long w_sgpr_pair, f32x2 x_pair) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(w_sgpr_pair), "v"(x_pair));
}
__device__ __forceinline__ void pk_fma_vv(f32x2& acc, f32x2 w, f32x2 x_pair) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(x_pair));
}

constexpr int kCout = 48, kPairs = kCout / 2;

// ---- regs: issue rate only ---------------------------------------------------------------------------------------
template <int PX, bool SGPR_W>
__global__ __launch_bounds__(256) void valu_regs(unsigned long long* out, float* sink, const float* wimg, int iters) {
  const int lane = threadIdx.x & 63;
  f32x2 acc[PX][kPairs];
  for (int p = 0; p < PX; ++p)
    for (int c = 0; c < kPairs; ++c) acc[p][c] = f32x2{0.f, 0.f};
  f32x2 x[PX];
  for (int p = 0; p < PX; ++p) x[p] = f32x2{lane * 1e-3f + p, 0.f};
  // 48 weights of one k: 24 SGPR pairs (loaded once) or 24 VGPR pairs
  unsigned long long ws[kPairs];
  f32x2 wv[kPairs];
  const unsigned long long* w64 = reinterpret_cast<const unsigned long long*>(wimg);
  for (int c = 0; c < kPairs; ++c) {
    ws[c] = __builtin_nontemporal_load(w64 + c);
    ws[c] = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(ws[c] >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)ws[c]);
    wv[c] = f32x2{wimg[2 * c], wimg[2 * c + 1]};
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int c = 0; c < kPairs; ++c) {
          if constexpr (SGPR_W) pk_fma_sw(acc[p][c], ws[c], x[p]);
          else pk_fma_vv(acc[p][c], wv[c], x[p]);
        }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int p = 0; p < PX; ++p)
    for (int c = 0; c < kPairs; ++c) s += acc[p][c][0] + acc[p][c][1];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// ---- stream: weights by s_load_dwordx16, activations by ds_read ----------------------------------------------------
// A layer's weights as the direct loop wants them: [k = (cin, tap)][48 couts], one k = three 16-dword blocks.  The blocks
// form one FIFO through a ring of FOUR SGPR blocks (64 SGPRs; two whole k, 96, spill): per block step "wait for what is
// in flight, request block i + 3 into the registers block i - 1 has just left, 8 x PX FMAs on block i".  Scalar loads
// return out of order, so every wait is lgkmcnt(0): a request has exactly one block step (32 x PX cycles) to land.
// Twelve block steps (4 k) are written out per loop trip so that ring slot and k part are compile-time.
__device__ __forceinline__ i32x16 sload16(const float* p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) i32x16*>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ unsigned long long pair16(const i32x16& w, int j) {
  return ((unsigned long long)(unsigned)w[2 * j + 1] << 32) | (unsigned)w[2 * j];
}

template <int PX>
__device__ __forceinline__ void valu_stream_body(const float* wimg, const float* lds, int lane, int iters, f32x2 (&acc)[PX][kPairs]) {
  constexpr int kBlocks = 48 * 9 * 3;             // 16-dword blocks of one 48 -> 48 layer
  i32x16 w[4];
  w[0] = sload16(wimg);
  w[1] = sload16(wimg + 16);
  w[2] = sload16(wimg + 32);
  const float* xp = lds + lane * PX;              // this lane's PX adjacent pixels
  int b = 0;                                      // block consumed next
  f32x2 x[PX];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int slot = s % 4, part = s % 3;
      int nb = b + 3;
      nb = nb >= kBlocks ? nb - kBlocks : nb;
      w[(slot + 3) % 4] = sload16(wimg + (size_t)nb * 16);
      if (part == 0) {
#pragma unroll
        for (int p = 0; p < PX; ++p) x[p] = f32x2{xp[((b / 3) & 63) * (64 * PX) + p], 0.f};   // (a ds_read per pixel and k)
      }
#pragma unroll
      for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) pk_fma_sw(acc[p][part * 8 + j], pair16(w[slot], j), x[p]);
      b = b + 1 == kBlocks ? 0 : b + 1;
    }
  }
}

template <int PX>
__global__ __launch_bounds__(256) void valu_stream(unsigned long long* out, float* sink, const float* wimg, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 64 * PX + 64];
  for (int i = threadIdx.x; i < 64 * 64 * PX + 64; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x2 acc[PX][kPairs];
  for (int p = 0; p < PX; ++p)
    for (int c = 0; c < kPairs; ++c) acc[p][c] = f32x2{0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  valu_stream_body<PX>(wimg, lds, lane, iters, acc);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int p = 0; p < PX; ++p)
    for (int c = 0; c < kPairs; ++c) s += acc[p][c][0] + acc[p][c][1];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// ---- hybrid: waves 0-3 the conv kernel's MFMA step, waves 4-7 the direct loop, one of each per SIMD ----------------
// Both run a FIXED number of steps; each reports its own ticks, so "do the pipes add up" reads off the two rates.
template <int PX, bool WITH_MFMA, bool WITH_VALU>
__global__ __launch_bounds__(512) void hybrid(unsigned long long* out, float* sink, const float* wimg, int iters_mfma, int iters_valu) {
  __shared__ __attribute__((aligned(16))) float lds[12288];
  for (int i = threadIdx.x; i < 12288; i += 512) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  unsigned long long dt = 0;
  if (wave < 4) {
    if constexpr (WITH_MFMA) {
      f32x4 acc[7];
      for (int i = 0; i < 7; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      float ring[3][8];
      for (int r = 0; r < 3; ++r)
        for (int i = 0; i < 8; ++i) ring[r][i] = lane * 1e-4f + i + r;
      const float* p = lds + (lane & 15) + (lane >> 4) * 304;
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
        for (int st = 0; st < 3; ++st) {
          __builtin_amdgcn_sched_barrier(0);
          float* dst = ring[(st + 2) % 3];
#pragma unroll
          for (int r = 0; r < 8; ++r) dst[r] = (p + ((it * 3 + st) & 7) * 64 + r * 608)[0];
          const float* cur = ring[st];
#pragma unroll
          for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[7], cur[i], acc[i], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      dt = __builtin_amdgcn_s_memtime() - t0;
      for (int i = 0; i < 7; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if constexpr (WITH_VALU) {
      f32x2 acc[PX][kPairs];
      for (int p = 0; p < PX; ++p)
        for (int c = 0; c < kPairs; ++c) acc[p][c] = f32x2{0.f, 0.f};
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      valu_stream_body<PX>(wimg, lds, lane, iters_valu, acc);
      dt = __builtin_amdgcn_s_memtime() - t0;
      for (int p = 0; p < PX; ++p)
        for (int c = 0; c < kPairs; ++c) s += acc[p][c][0] + acc[p][c][1];
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = dt;
}

static double mean(const unsigned long long* h, int n, int stride = 1, int first = 0, int count = 1) {
  double s = 0;
  int m = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < count; ++j) { s += (double)h[i * stride + first + j]; ++m; }
  return s / m;
}

int main() {
  const int blocks = 256;
  unsigned long long* d;
  float *sink, *wimg;
  (void)hipMalloc(&d, blocks * 8 * 8);
  (void)hipMalloc(&sink, blocks * 512 * 4);
  (void)hipMalloc(&wimg, 48 * 48 * 9 * 4 + 4096);
  float* hw = (float*)malloc(48 * 48 * 9 * 4 + 4096);
  for (int i = 0; i < 48 * 48 * 9 + 1024; ++i) hw[i] = 1e-3f * (float)((i * 7919) % 1000);
  (void)hipMemcpy(wimg, hw, 48 * 48 * 9 * 4 + 4096, hipMemcpyHostToDevice);
  unsigned long long h[blocks * 8];
  auto fetch = [&](int words) { (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, words * 8, hipMemcpyDeviceToHost); };
  const int iters = 2000;
  printf("ticks per 16x16x4-equivalent (1024 MACs per wave = 8 v_pk_fma_f32); the fp32 matrix pipe's own rate is 32, the conv "
         "kernel's MFMA k-step with its LDS operand reads 34.7 (profiles/r03_probe_mfma_rate.txt); %d workgroups, one wave per SIMD\n", blocks);
#define RUN_REGS(PX, SG, what)                                                                                          \
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((valu_regs<PX, SG>), dim3(blocks), dim3(256), 0, 0, d, sink, wimg, iters); \
  fetch(blocks);                                                                                                        \
  printf("%-96s %6.2f\n", what, mean(h, blocks) / iters / (4.0 * PX * kPairs / 8.0));
  RUN_REGS(1, false, "regs:   v_pk_fma_f32, weights in VGPR pairs, 1 pixel per lane (24 accumulator pairs)")
  RUN_REGS(1, true, "regs:   v_pk_fma_f32, weights in SGPR pairs (held), 1 pixel per lane")
  RUN_REGS(2, true, "regs:   v_pk_fma_f32, weights in SGPR pairs (held), 2 pixels per lane (48 accumulator pairs)")
#define RUN_STREAM(PX, what)                                                                                            \
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((valu_stream<PX>), dim3(blocks), dim3(256), 0, 0, d, sink, wimg, iters); \
  fetch(blocks);                                                                                                        \
  printf("%-96s %6.2f\n", what, mean(h, blocks) / iters / (4.0 * PX * kPairs / 8.0));
  RUN_STREAM(1, "stream: weights by 3 x s_load_dwordx16 per k (83 KB image), x by ds_read, 1 pixel per lane")
  RUN_STREAM(2, "stream: the same, 2 pixels per lane (every SGPR pair feeds two FMAs)")
  RUN_STREAM(4, "stream: the same, 4 pixels per lane (96 accumulator pairs)")
  // hybrid: MFMA steps: 3 x 7 MFMAs per iteration; VALU: 4 k x PX x 24 pk_fma per iteration = PX * 12 equivalents
  const int im = 1000, iv = 1000;
#define RUN_HYB(PX, M, V, what)                                                                                         \
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((hybrid<PX, M, V>), dim3(blocks), dim3(512), 0, 0, d, sink, wimg, im, iv); \
  fetch(blocks * 8);                                                                                                    \
  printf("%-96s MFMA waves %6.2f per MFMA   VALU waves %6.2f per equivalent\n", what,                                    \
         M ? mean(h, blocks, 8, 0, 4) / im / 21.0 : 0.0, V ? mean(h, blocks, 8, 4, 4) / iv / (4.0 * PX * kPairs / 8.0) : 0.0);
  RUN_HYB(2, true, false, "hybrid harness, MFMA waves only (7 MFMAs + 8 ds_read_b32 per step)")
  RUN_HYB(2, false, true, "hybrid harness, VALU waves only (stream loop, 2 pixels per lane)")
  RUN_HYB(2, true, true, "hybrid: both on the same SIMDs, each for its own fixed work (rates while they overlap are at least these)")
  return 0;
}
