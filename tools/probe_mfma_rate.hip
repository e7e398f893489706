// What do LDS operand reads cost a wave that streams v_mfma_f32_16x16x4_f32 (one wave per SIMD, gfx950)?
// Each step = 7 independent MFMAs (the conv kernel's K loop) + NR LDS reads of width WIDTH dwords whose
// results feed the MFMAs TWO steps later (register ring of 3, fully unrolled: no copies, counted waits).
// Reported: s_memtime ticks per MFMA (32 = the matrix pipe's own rate).
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma_rate tools/probe_mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NR, int WIDTH, bool SHADOW, int NM = 7, bool USE = true>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[12288];
  for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[7];
  for (int i = 0; i < 7; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ring[3][8];
  for (int r = 0; r < 3; ++r)
    for (int i = 0; i < 8; ++i) ring[r][i] = lane * 1e-4f + i + r;
  const float* p = lds + (lane & 15) * WIDTH + (lane >> 4) * 304;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {   // three steps per trip: the ring indices are compile-time
      __builtin_amdgcn_sched_barrier(0);
      constexpr int DW = NR * WIDTH;   // dwords fetched per step (<= 8 are consumed)
      float* dst = ring[(s + 2) % 3];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float* q = p + ((it * 3 + s) & 7) * 64 + r * 608;
        if constexpr (WIDTH == 1) dst[r % 8] = q[0];
        if constexpr (WIDTH == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(q); dst[(2 * r) % 8] = v[0]; dst[(2 * r + 1) % 8] = v[1]; }
        if constexpr (WIDTH == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(q); dst[(4 * r) % 8] = v[0]; dst[(4 * r + 1) % 8] = v[1]; dst[(4 * r + 2) % 8] = v[2]; dst[(4 * r + 3) % 8] = v[3]; }
      }
      (void)DW;
      const float* cur = USE ? ring[s] : ring[0];
      if constexpr (!USE) {   // keep the loaded values alive without feeding the MFMAs
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(ring[(s + 2) % 3][i]));
      }
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i % 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[7], cur[i % 7], acc[i % 7], 0, 0, 0);
      if constexpr (SHADOW && NR > 0 && NM == 14) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 6) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 5) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 4) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 3) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 2) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 1) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR / 7, 0);
      }
      if constexpr (SHADOW && NR > 0 && NM == 7) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 6) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 5) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 4) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 3) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 2) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + 1) / 7, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR / 7, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 7; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// Round 3: the same stream with v_mfma_f32_32x32x2_f32 (64 cycles per issue, 4096 FLOP = two 16x16x4) for part of
// the step: N32 of them + N16 v_mfma_f32_16x16x4_f32 per step, NR ds_read_b32 dealt out between them (one read or
// two behind each MFMA), operands two steps ahead as above.  Reported per 16x16x4-EQUIVALENT (= 2 N32 + N16 per step).
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NM, int NR, int M>
__device__ __forceinline__ void deal() {   // MFMA, its share of the NR reads, MFMA, ...
  if constexpr (M < NM) {
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
    constexpr int share = (NR + NM - 1 - M) / NM;
    if constexpr (share > 0) __builtin_amdgcn_sched_group_barrier(0x100, share, 0);
    deal<NM, NR, M + 1>();
  }
}
template <int N32, int N16, int NR>
__global__ __launch_bounds__(256) void probe32(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[12288];
  for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 big[N32 > 0 ? N32 : 1];
  f32x4 acc[N16 > 0 ? N16 : 1];
  for (int i = 0; i < (N32 > 0 ? N32 : 1); ++i)
    for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
  for (int i = 0; i < (N16 > 0 ? N16 : 1); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ring[3][8];
  for (int r = 0; r < 3; ++r)
    for (int i = 0; i < 8; ++i) ring[r][i] = lane * 1e-4f + i + r;
  const float* p = lds + (lane & 15) + (lane >> 4) * 304;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      float* dst = ring[(s + 2) % 3];
#pragma unroll
      for (int r = 0; r < NR; ++r) dst[r % 8] = (p + ((it * 3 + s) & 7) * 64 + r * 608)[0];
      const float* cur = ring[s];
#pragma unroll
      for (int i = 0; i < N32; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[7], cur[i % 7], big[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < N16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[6], cur[(i + 3) % 6], acc[i], 0, 0, 0);
      constexpr int NM = N32 + N16;
      if constexpr (NR > 0) deal<NM, NR, 0>();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < (N32 > 0 ? N32 : 1); ++i)
    for (int e = 0; e < 16; ++e) sum += big[i][e];
  for (int i = 0; i < (N16 > 0 ? N16 : 1); ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int N32, int N16, int NR>
void run32(const char* what, int blocks) {
  unsigned long long* d;
  float* sink;
  (void)hipMalloc(&d, blocks * 8);
  (void)hipMalloc(&sink, blocks * 256 * 4);
  const int iters = 700;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe32<N32, N16, NR>), dim3(blocks), dim3(256), 0, 0, d, sink, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[1024];
  (void)hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < blocks; ++i) sum += (double)h[i];
  printf("%-72s %3d workgroups: %6.2f ticks per 16x16x4-equivalent (%d per step)\n", what, blocks,
         sum / blocks / iters / (3 * (2 * N32 + N16)), 2 * N32 + N16);
  (void)hipFree(d);
  (void)hipFree(sink);
}

template <int NR, int WIDTH, bool SHADOW, int NM = 7, bool USE = true>
void run(const char* what, int blocks) {
  unsigned long long* d;
  float* sink;
  (void)hipMalloc(&d, blocks * 8);
  (void)hipMalloc(&sink, blocks * 256 * 4);
  const int iters = 700;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<NR, WIDTH, SHADOW, NM, USE>), dim3(blocks), dim3(256), 0, 0, d, sink, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[1024];
  (void)hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < blocks; ++i) sum += (double)h[i];
  printf("%-64s %3d workgroups: %6.2f ticks per MFMA\n", what, blocks, sum / blocks / iters / (3 * NM));
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  for (int blocks : {256}) {
    run<0, 1, false>("no LDS reads", blocks);
    run<2, 1, false>("2 x ds_read_b32 per step", blocks);
    run<4, 1, false>("4 x ds_read_b32 per step", blocks);
    run<8, 1, false>("8 x ds_read_b32 per step (compiler's placement)", blocks);
    run<8, 1, true>("8 x ds_read_b32 per step, dealt out between the MFMAs", blocks);
    run<4, 2, true>("4 x ds_read_b64 per step, dealt out", blocks);
    run<2, 4, true>("2 x ds_read_b128 per step, dealt out", blocks);
    run<8, 4, true>("8 x ds_read_b128 per step, dealt out", blocks);
    run<0, 1, false, 14>("14 MFMAs per step, no LDS reads", blocks);
    run<8, 1, true, 14>("14 MFMAs per step, 8 x ds_read_b32 dealt out", blocks);
    run<16, 1, true, 14>("14 MFMAs per step, 16 x ds_read_b32 dealt out (= two k-steps per region)", blocks);
    run<8, 2, true, 14>("14 MFMAs per step, 8 x ds_read_b64 dealt out", blocks);
    run<8, 1, false, 14>("14 MFMAs per step, 8 x ds_read_b32 compiler's placement", blocks);
    run<8, 1, false, 21>("21 MFMAs per step, 8 x ds_read_b32 compiler's placement", blocks);
    // 32x32x2: the same FLOPs per step in half the issue slots
    run32<4, 0, 0>("4 x 32x32x2 per step, no LDS reads", blocks);
    run32<4, 0, 6>("4 x 32x32x2 per step (cout 32 x 64 pixels, k = 4), 6 x ds_read_b32 dealt out", blocks);
    run32<2, 3, 0>("2 x 32x32x2 + 3 x 16x16x4 per step, no LDS reads", blocks);
    run32<2, 3, 7>("2 x 32x32x2 + 3 x 16x16x4 per step (7 equivalents), 7 x ds_read_b32 dealt out", blocks);
    run32<2, 2, 6>("2 x 32x32x2 + 2 x 16x16x4 per step (48 cout x 32 pixels, k = 4), 6 x ds_read_b32", blocks);
    run32<3, 1, 8>("3 x 32x32x2 + 1 x 16x16x4 per step (7 equivalents), 8 x ds_read_b32 dealt out", blocks);
    run32<0, 7, 8>("control: 7 x 16x16x4 per step, 8 x ds_read_b32 dealt out (this harness)", blocks);
  }
  return 0;
}
