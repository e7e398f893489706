/* Measurement-only entry points (NOT part of the product ABI, include/larva_hip.h): they exist in the libraries
 * tools/build_diag.sh builds (the product's sources + -DLARVA_DIAG_API, into tools/_diag/<name>.so) and nowhere else.
 * bench.py's `launch_alone_ms` fields and the tools under tools/ bind them with ctypes (tools/diag_lib.py); the product
 * package larvanet_amd/ never does.  No reference counterpart.  Conventions as in include/larva_hip.h. */
#ifndef LARVA_DIAG_H
#define LARVA_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

/* The same launch as larva_conv3x3_fwd `iters` times with kernel-attached events (hipExtLaunchKernelGGL); mean / min
 * kernel duration in ms = what a profiler reports per dispatch.  Synchronises the stream; not capturable. */
int larva_conv3x3_fwd_timed(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                            const float* bias, const float* res0, const float* res1, const float* mask,
                            const float* base, float* out, int N, int cout, int H, int W, int relu,
                            int mode, void* stream, int iters, float* mean_ms, float* min_ms);
/* ... and for a strip-tile launch (larva_conv3x3_fwd_strips): one half-batch launch running alone, beside bench.py's time
 * per layer with two of them running concurrently. */
int larva_conv3x3_fwd_strips_timed(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                                   const float* bias, const float* res0, const float* res1, const float* mask,
                                   const float* base, float* out, int N, int cout, int H, int W, int pitch,
                                   int relu, int mode, const unsigned* tile_tab, const unsigned* tile_tab_host,
                                   int tiles_per_image, int plain_stores, void* stream, int iters, float* mean_ms,
                                   float* min_ms);
/* A one-lane launch that stores the 100 MHz wall clock into *dst in stream order: a capturable marker between the
 * launches of a graph (tools/step_marks.py). */
int larva_stamp_clock(unsigned long long* dst, void* stream);
/* A one-wave launch that sleeps until the 100 MHz wall clock has advanced by `ticks` (<= 100000; bounded), in stream
 * order: a tunable delay in front of one chain of a captured two-chain graph (tools/ab_stagger.sh). */
int larva_delay_ticks(int ticks, void* stream);
/* Round 5: both half-batch conv3x3 + ReLU chains of `layers` layers in ONE launch, the two strip tiles of a CU owned by
 * one 704-thread workgroup (csrc/conv3x3_pair_chain.inc has the protocol; tools/probe_pair_chain.py drives it;
 * profiles/r05_probe_pair_chain.txt has the result: 13.1-13.4 us per layer against 14.2-14.35 as launches, gate 12.0). */
int larva_conv3x3_pair_chain_probe(float* buf0, float* buf1, const float* wpk, const float* bias, int N, int H, int W,
                                   int pitch, const unsigned* tab0, const unsigned* tab1, const int* nbr, const int* deg,
                                   int tiles_per_image, unsigned* state, int* xcc_out, unsigned long long* trace, int layers,
                                   int lock, int naps, int prio_a, int prio_b, void* stream);

#ifdef __cplusplus
}
#endif
#endif
