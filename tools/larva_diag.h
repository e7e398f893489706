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
 * order: a tunable delay in front of one chain of a captured two-chain graph (profiles/r04_ab_stagger.txt; the script is in git history). */
int larva_delay_ticks(int ticks, void* stream);

/* One wave that naps for `ticks` of the 100 MHz wall clock and stores out[0] = wall ticks, out[1] = shader cycles
 * (s_memtime) that went by: beside another stream's work, out[1] / out[0] * 0.1 = the clock in GHz the chip sustains
 * under that load (bench.py `step.sustained_clock_ghz`). */
int larva_clock_probe(int ticks, unsigned long long* out, void* stream);
/* Round 5: both half-batch conv3x3 + ReLU chains of `layers` layers in ONE launch, the two strip tiles of a CU owned by
 * one 704-thread workgroup (csrc/conv3x3_pair_chain.inc has the protocol; tools/probe_pair_chain.py drives it;
 * profiles/r05_probe_pair_chain.txt has the result: 13.1-13.4 us per layer against 14.2-14.35 as launches, gate 12.0). */
int larva_conv3x3_pair_chain_probe(float* buf0, float* buf1, const float* wpk, const float* bias, int N, int H, int W,
                                   int pitch, const unsigned* tab0, const unsigned* tab1, const int* nbr, const int* deg,
                                   int tiles_per_image, unsigned* state, int* xcc_out, unsigned long long* trace, int layers,
                                   int lock, int naps, int prio_a, int prio_b, void* stream);

/* ---- layer pipeline (round 5 experiment; csrc/conv3x3_pipe.inc) -------------------------------
 * A CHAIN of 48 -> 48 convolutions over one [N][48][H][pitch] tensor shape in ONE launch: the body of a full-image
 * forward, LarvaNetModule.forward's `for i in range(self.len): fea = body_i(fea)` + the first conv of the last leg
 * (models/LarvaNet.py:205-220,236-248,256-257,283-293) as validate.py:94-102 / runtime.py call it.  Persistent workgroups walk
 * the (layer, tile) positions in order; a tile of layer i waits until the three tile rows around it of layer dep are
 * stored and visible (per-row counters, agent scope), so the layers overlap where the per-layer launches each paid a
 * launch boundary, a cold start and a tail.  Results are bit-identical to
 * larva_conv3x3_fwd_pitched layer by layer.
 *
 * Every layer writes a tensor of its OWN (nothing the launch still reads is overwritten).  dep: the layer whose output
 * this layer's src is, -1 when src was written before the launch; every other operand written inside the launch
 * (res0, res1, further src tensors) must be the output of dep or of a layer dep depends on.  relu / res0 / res1 as
 * larva_conv3x3_fwd (relu with residuals: unsupported).  16-byte path only (pitch % 4 == 0, aligned tensors):
 * hipErrorNotSupported otherwise.
 *
 * larva_conv3x3_pipeline_plan validates, writes the device-side layer table into `workspace`
 * (larva_conv3x3_pipeline_workspace_bytes(), 256-byte aligned; a SYNCHRONOUS copy: call it outside captured regions)
 * and fills `plan` (host memory, larva_conv3x3_pipeline_plan_bytes()).  larva_conv3x3_pipeline_run issues one memset
 * node (the counters) and the launch: stream-ordered, allocation-free, capturable.  error_word: one device-accessible
 * word (pinned host memory works), zeroed by the caller; every wait inside the kernel is bounded (spin_limit polls,
 * 0 = about two seconds) and a wait that expires sets the word to 1 and lets the launch drain -- its outputs are then
 * garbage and the caller must refuse them.
 * MEASURED AND NOT ADOPTED (profiles/r05_layer_pipeline.txt): bit-identical on every case of
 * tests/test_hip_kernels.py, but 72.7 us per layer against 62.8 for one persistent launch per layer; with every wait,
 * signal and coherent access switched off it is 63.2 -- the per-layer launches already run at the K loop's ceiling at the
 * clock the chip sustains (2.2-2.25 GHz), there is no launch-boundary time left to win. */
typedef struct larva_pipe_layer {
  const float* src[8];
  int n_src, cin_per_src;
  const float* wpk;
  const float* bias;
  const float* res0;
  const float* res1;
  float* out;
  int relu, dep;
} larva_pipe_layer;
long long larva_conv3x3_pipeline_workspace_bytes(int n_layers, int N, int H);
long long larva_conv3x3_pipeline_plan_bytes(void);
int larva_conv3x3_pipeline_plan(const void* layers /* larva_pipe_layer[n_layers] */, int n_layers, int N, int cout, int H,
                                int W, int pitch, void* workspace, unsigned* error_word, void* plan);
int larva_conv3x3_pipeline_run(const void* plan, int spin_limit, void* stream);

#ifdef __cplusplus
}
#endif
#endif
