// out[0] = ( sum_i scale_i * (sum of the count_i floats at p_i) ) / divisor: the mean over the exits of their
// L1 terms straight from partial sums (count_i of them, scale_i = 1 / numel), or of ready scalars (count 1,
// scale 1).  Every term is reduced in a fixed order by ONE 256-thread block, the terms are added in index
// order: reproducible.  Shared by larva_pointwise.hip (loss_from_partials_kernel) and wgrad3x3_mfma.hip (the
// weight-gradient reduction launch can carry it as an extra block: one launch less per training step).
#pragma once
#include "larva_common.h"

namespace larva {

struct TermList {
  const float* p[8];
  int count[8];
  float scale[8];
  int n;
};

// Call with all 256 threads of a block.  host_cell (may be null): a {float value, uint32 sequence} pair in coherent
// pinned host memory that receives the value too -- the host polls it instead of synchronising with the stream
// (larva_host_cell_alloc).  Every store bumps the sequence number and both words leave as ONE 8-byte store, so the
// host can tell this launch's value from an earlier launch's (it counts its own launches).
// dev_seq (may be null): the cell's sequence number kept in DEVICE memory as well (round 4): the launch then need not
// read it back over PCIe first -- a ~3 us round trip in an 8.5 us launch that stands between the exits' forward and
// their backward.
__device__ __forceinline__ void loss_terms_block(const TermList& l, float divisor, float* __restrict__ out,
                                                 float* __restrict__ host_cell = nullptr, unsigned* __restrict__ dev_seq = nullptr) {
  // (without a device copy the sequence number comes over PCIe: asked for first, needed last)
  unsigned long long seq = 0ull;
  if (host_cell && threadIdx.x == 0) {
    if (dev_seq) seq = dev_seq[0];
    else seq = __hip_atomic_load(reinterpret_cast<unsigned long long*>(host_cell), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >> 32;
  }
  // every term's strided partial sums first (all their loads in flight together), then ONE pair of block barriers for
  // all terms: per term the same additions in the same order as a loop of "sum, barrier, combine" (round 4: that loop
  // took 8 us for four terms of 1024 partial sums, a dependent chain of four load latencies and eight barriers)
  float sv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float s_ = 0.f;
    if (i < l.n)
      for (int k = threadIdx.x; k < l.count[i]; k += 256) s_ += l.p[i][k];
    sv[i] = s_;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sv[i] += __shfl_xor(sv[i], o);
  __shared__ float wsa[8][4];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) wsa[i][threadIdx.x >> 6] = sv[i];
  }
  __syncthreads();
  float total = 0.f;
  for (int i = 0; i < l.n; ++i) total += ((wsa[i][0] + wsa[i][1]) + (wsa[i][2] + wsa[i][3])) * l.scale[i];
  if (threadIdx.x == 0) {
    const float v = total / divisor;
    out[0] = v;
    if (host_cell) {
      if (dev_seq) dev_seq[0] = (unsigned)(seq + 1ull);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_cell),
                         ((seq + 1ull) << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
      // (relaxed on purpose: the cell IS the message -- value and sequence number leave as one 8-byte store and the host
      // reads nothing else the device wrote.  A system-scope RELEASE store first writes the whole L2 back to memory,
      // megabytes of the exits' outputs, which is most of what this launch used to take.)
    }
  }
}

}  // namespace larva
