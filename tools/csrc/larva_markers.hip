// MEASUREMENT ENTRY POINTS (declared in tools/larva_diag.h, NOT in include/larva_hip.h): marker / delay / clock-probe
// launches.  Built by tools/build_diag.sh into tools/_diag/<name>.so; never part of liblarva_hip.so.
#include <hip/hip_runtime.h>

extern "C" {
// Measurement only: one lane stores the 100 MHz wall clock (s_memrealtime) into *dst, in stream order -- a capturable
// marker between the launches of a hipGraph whose kernels must stay exactly the product's (tools/step_marks.py:
// when does each chain of the captured step start and end?).  Costs one launch slot (~2 us) on its stream.
__global__ void stamp_clock_kernel(unsigned long long* dst) { *dst = __builtin_amdgcn_s_memrealtime(); }

int larva_stamp_clock(unsigned long long* dst, void* stream) {
  if (!dst) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(stamp_clock_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
  return (int)hipGetLastError();
}

// Measurement only: one wave that sleeps until the 100 MHz wall clock has advanced by `ticks` (bounded: at most 4096
// naps of 64 clocks), in stream order -- a capturable, tunable delay in front of one chain of a two-chain graph
// (profiles/r04_ab_stagger.txt: how does the step depend on the phase the two chains start in?).
__global__ void delay_kernel(int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < 4096; ++i) {
    if ((long long)(__builtin_amdgcn_s_memrealtime() - t0) >= (long long)ticks) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

int larva_delay_ticks(int ticks, void* stream) {
  if (ticks < 0 || ticks > 100000) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
  return (int)hipGetLastError();
}

// Measurement only: the shader clock WHILE something else runs.  One wave naps until the 100 MHz wall clock has advanced
// by `ticks` (bounded: at most 1 << 20 naps) and stores {wall ticks, shader cycles (s_memtime)} it saw go by: launched on
// a side stream beside the replays of the captured step, cycles / ticks * 100 MHz is the clock the chip sustains under
// that load (bench.py `step.sustained_clock_ghz`; the guide's peaks are quoted at 2.4 GHz).
__global__ void clock_probe_kernel(int ticks, unsigned long long* out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  for (int i = 0; i < (1 << 20); ++i) {
    r1 = __builtin_amdgcn_s_memrealtime();
    if ((long long)(r1 - r0) >= (long long)ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = r1 - r0;
    out[1] = c1 - c0;
  }
}

int larva_clock_probe(int ticks, unsigned long long* out, void* stream) {
  if (!out || ticks <= 0 || ticks > 10000000) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks, out);
  return (int)hipGetLastError();
}
}  // extern "C"
