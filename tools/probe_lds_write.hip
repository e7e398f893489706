// Hardware probe (gfx950): issue cost of LDS writes from ONE wave per SIMD, in shader cycles per
// instruction (s_memtime around 256 back-to-back writes, 4 waves per workgroup, one workgroup).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds_write.hip -o probe_w && ./probe_w
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(long long* out, int stride_floats, float seed) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = smem + wave * 8192 + lane * stride_floats;
  f32x4 v = {seed, seed + 1, seed + 2, seed + 3};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 256; ++i) {
    float* p = base + (i & 7) * 4;   // small rotating offset, stays inside the wave's 32 KiB
    if constexpr (MODE == 0) {        // 2 x b64 (what the compiler merges into ds_write2_b64)
      *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
      *reinterpret_cast<f32x2*>(p + 2) = f32x2{v[2], v[3]};
    } else if constexpr (MODE == 1) { // b128
      *reinterpret_cast<f32x4*>(p) = v;
    } else if constexpr (MODE == 2) { // one b64
      *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
    } else if constexpr (MODE == 3) { // one b32
      *p = v[0];
    } else if constexpr (MODE == 4) { // 4 x b32
      p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3];
    }
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[wave] = t1 - t0;
  if (smem[threadIdx.x] == 12345.f) out[5] = 1;
}

template <int MODE>
void run(const char* name, int stride) {
  long long* d;
  hipMalloc(&d, 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(256), 140 * 1024, 0, d, stride, 1.0f);
  long long h[4];
  hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("%-28s lane stride %3d floats: %6.1f cycles per write instruction group (wave 0), %6.1f (wave 3)\n", name, stride,
         h[0] / 256.0, h[3] / 256.0);
  hipFree(d);
}

int main() {
  // s_memtime counts at a fixed 100 MHz on gfx9 (not shader cycles): numbers are RELATIVE
  for (int stride : {4, 2, 1, 5, 36}) {
    run<0>("2 x b64 (16 B per lane)", stride);
    run<1>("b128 (16 B per lane)", stride);
    run<2>("b64 (8 B per lane)", stride);
    run<3>("b32 (4 B per lane)", stride);
    run<4>("4 x b32 (16 B per lane)", stride);
  }
  return 0;
}
