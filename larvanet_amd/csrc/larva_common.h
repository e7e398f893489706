// Shared definitions for the gfx950 (MI355X / CDNA4) LarvaNet kernels.
//
// Geometry used by every 3x3 kernel in this directory:
//   * a workgroup (256 threads = 4 wave64, one wave per SIMD) owns one OUTPUT TILE of
//     3 rows x 48 columns of one image = 9 "pixel groups" of 16 consecutive pixels of a row;
//   * the matching INPUT HALO TILE is 5 rows x 50 columns per channel, staged in LDS as
//     [channel][5][LDS_RS] with the first wanted column (x0-1) at index 3 so that column x0
//     sits on a 16-byte boundary;
//   * the conv kernel walks the input channels in K chunks of 8 (kCh in conv3x3_mfma.hip = 2
//     k-steps of v_mfma_f32_16x16x4_f32 per tap).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace larva {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileRows = 3;
constexpr int kTileCols = 48;
constexpr int kHaloRows = kTileRows + 2;
constexpr int kMaxSrc = 8;                                 // channel-concatenated input tensors

// LDS row stride of a staged halo row (floats): idx 3 = x0-1, idx 4..51 = x0..x0+47, idx 52 = x0+48.
constexpr int kRS = 56;

// Host-side launch helpers are PER DEVICE: hipFuncSetAttribute applies to the current device's copy of a kernel, and the
// number of workgroup slots is a property of the device a launch goes to.  A process that drives a second GPU must not
// reuse the first one's slot count or skip the second one's LDS-size attribute.
constexpr int kMaxDevices = 64;
inline int current_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d < kMaxDevices ? d : kMaxDevices - 1;   // (ids past the table share its last slot: they repeat the set-up, harmlessly)
}
struct PerDeviceOnce {
  bool done[kMaxDevices] = {};
};
// hipFuncSetAttribute(kernel, MaxDynamicSharedMemorySize, lds) once per device and kernel instantiation
inline hipError_t ensure_dynamic_lds(PerDeviceOnce& once, const void* kernel, size_t lds) {
  const int d = current_device_slot();
  if (once.done[d] && d != kMaxDevices - 1) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) once.done[d] = true;
  return e;
}
// compute units of the current device (cached per device; 256 = MI355X if the query fails)
inline int device_cu_count() {
  static int cus[kMaxDevices] = {};
  const int d = current_device_slot();
  if (!cus[d] || d == kMaxDevices - 1) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus[d] = n;
  }
  return cus[d];
}

__host__ __device__ constexpr int cout_stride(int cout) {
  // Row stride (floats) of one [k][cout] weight row in LDS.  The two 16-lane halves of a 32-lane ds_read_b32 group
  // read rows k and k + 1: they must land on disjoint banks.  cout == 16 (mod 32): the stride itself does it.
  // cout == 0 (mod 32) -- 32 and 64 channels: the row is stored UNPADDED and odd rows are swizzled instead (column
  // c ^ 16, see cout_swizzled): until round 3 these rows were padded by 16 floats, 13.5 instead of 9 KiB of weights per
  // K chunk at 32 channels and 22.5 instead of 18 at 64 -- what kept two strip workgroups of 64 channels off one CU.
  return (cout % 32 == 16 || cout % 32 == 0) ? cout : cout + 16;
}
__host__ __device__ constexpr bool cout_swizzled(int cout) { return cout % 32 == 0; }

// XCD-aware, bijective block remap (8 XCDs, blocks dealt round-robin): gives every XCD a
// contiguous run of tiles so that vertically adjacent tiles (which share halo rows) and the
// same tile of the next layer hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// LDS-DMA goes through inline asm on purpose: with the compiler builtin in a loop, hipcc (ROCm 7.2)
// turns every counted lgkmcnt(N) in front of the MFMAs into lgkmcnt(0), i.e. waits for the operand
// reads it has only just issued.  The statement saves/restores M0 (compiler-reserved) and pads the
// SALU-write-M0 -> LDS-DMA hazard itself; the DMA is invisible to hipcc's vmcnt bookkeeping, so
// callers count it by hand (s_waitcnt vmcnt(N)) and drain it with vmcnt(0) before the workgroup ends.
//
// Through a raw buffer descriptor (4 SGPRs: base, stride 0, num_records 2^31-1): lane l's
// 16 bytes at base + soffset + voff land at LDS byte address lds_base + 16 l.  A lane whose voff is
// out of range (>= 2^31, see kDmaZero) WRITES ZEROS (probed on gfx950: tools/probe_lds_dma.hip), so
// padding costs no address select and no zero page.  Nothing here is per-lane arithmetic: the wave
// issues 3 scalar moves and the load.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kDmaZero = 0x80000000u;

__device__ __forceinline__ i32x4 dma_rsrc(const void* base) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
  r[2] = 0x7fffffff;
  r[3] = 0x00020000;
  return r;
}

// SC1: the load bypasses this CU's vector L1 (served by the XCD's L2) -- for bytes another workgroup of the SAME launch
// has just written (the pair-chain and layer-pipeline measurement kernels; across XCDs: conv3x3_pipe.inc); every other caller
// reads what an earlier launch wrote.
template <bool SC1 = false>
__device__ __forceinline__ void lds_dma16_buf(i32x4 rsrc, unsigned voff, int soff_uniform, unsigned lds_addr_uniform) {
  unsigned keep;
  if constexpr (SC1) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr_uniform), "v"(voff), "s"(rsrc), "s"(soff_uniform)
                 : "memory");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr_uniform), "v"(voff), "s"(rsrc), "s"(soff_uniform)
                 : "memory");
  }
}

__device__ __forceinline__ unsigned lds_addr_of(const float* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p));
}

}  // namespace larva
