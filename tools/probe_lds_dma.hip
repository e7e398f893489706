// Hardware probe (gfx950): what does an out-of-range lane of `buffer_load_dwordx4 ... lds` write
// to LDS -- zeros, or nothing?  Decides whether halo zero-padding can come from the buffer range
// check instead of a zero page.  hipcc --offload-arch=gfx950 tools/probe_lds_dma.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void probe(const float* __restrict__ src, float* __restrict__ dst, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) smem[i] = -7.0f;  // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int voff = lane * 16;
  if (lane % 4 == 3) voff = 0x7ffffff0;  // out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = lane; i < 256; i += 64) dst[i] = smem[i];
}
int main() {
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  float *s, *d;
  hipMalloc(&s, 1024); hipMalloc(&d, 1024);
  hipMemcpy(s, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, s, d, 1024);
  hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  printf("in-range lane 0: %g %g %g %g\n", h[0], h[1], h[2], h[3]);
  printf("OOB lane 3 (sentinel -7 = not written, 0 = zero-filled): %g %g %g %g\n", h[12], h[13], h[14], h[15]);
  printf("in-range lane 4: %g %g %g %g\n", h[16], h[17], h[18], h[19]);
  return 0;
}
