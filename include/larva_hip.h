/* C ABI of liblarva_hip.so -- the gfx950 (MI355X) kernels behind the LarvaNet hot path.
 *
 * The reference (Geunwoo-Jeon/LarvaNet) has no FFI: its hot path is torch.nn call sites inside
 * models/LarvaNet.py / models/LarvaNetV2.py.  Every entry point below names the reference call
 * site(s) it replaces (file:line into the reference tree).  INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add.
 *
 * Conventions: all tensors are contiguous fp32 NCHW device memory owned by the caller; `stream`
 * is a hipStream_t passed as void*; every call is stream-ordered, never allocates, never
 * synchronises, keeps no global mutable state (safe for hipGraph capture).  Return value: 0 on
 * success, otherwise a hipError_t code (larva_error_string() gives the text).
 */
#ifndef LARVA_HIP_H
#define LARVA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int larva_abi_version(void);
const char* larva_error_string(int code);

/* ---- weight packing ------------------------------------------------------------------------
 * nn.Conv2d weights stay in PyTorch layout [cout][cin][3][3] (models/LarvaNet.py:210,212,227,
 * 256,258; models/LarvaNetV2.py:318).  The conv kernel consumes a packed image of 8-channel K
 * chunks, [cin/8][9][8][stride(cout)]; the input-gradient pass consumes the tap-mirrored,
 * channel-transposed image [cout/8][9][8][stride(cin)]; stride(c) = c if c % 32 is 0 or 16, else c + 16 floats
 * (LDS bank layout: the 16-lane halves of a read hit rows k and k + 1).  Where c % 32 == 0 (32 and 64 channels;
 * ABI version 3: these rows used to be padded by 16 floats) the rows are unpadded and every odd row (k odd) is stored
 * with column c ^ 16 -- an opaque kernel layout: only larva_pack_weights* writes it.  cin and cout are multiples of 8.
 * larva_packed_weight_floats(cout, cin) = (cin/8) * 9 * 8 * stride(cout).
 * `w_cin_total`/`w_cin_off` select a channel slice [w_cin_off, w_cin_off+cin) of a wider weight;
 * channels >= w_cin_total pack as 0 (the 3-channel head conv is packed for a 16-channel, i.e.
 * two-chunk, zero-padded input).  Either output may be NULL. */
long long larva_packed_weight_floats(int cout, int cin);
int larva_pack_weights(const float* w, float* wpk_fwd, float* wpk_bwd, int cout, int cin,
                       int w_cin_total, int w_cin_off, void* stream);

/* njobs (<= 64) packs in ONE launch; every argument is a host array of length njobs. */
int larva_pack_weights_batch(const float* const* w, float* const* wpk_fwd, float* const* wpk_bwd,
                             const int* cout, const int* cin, const int* w_cin_total,
                             const int* w_cin_off, int njobs, void* stream);

/* Everything train_step_larva does before its first convolution, in one launch: the packed images of
 * njobs (<= 64) weights (the optimizer has just changed them, models/LarvaNet.py:114), the head's
 * input zero-padded to 16 channels (x16 [N][16][H][W]: channels [0, C) are written, the others must
 * already be zero) and the bicubic x4 base image (models/LarvaNet.py:283-285; base [N][C][4H][4W]).
 * x16 and base may be NULL (base needs x16). */
int larva_step_prologue(const float* const* w, float* const* wpk_fwd, float* const* wpk_bwd, const int* cout,
                        const int* cin, const int* w_cin_total, const int* w_cin_off, int njobs, const float* x,
                        float* x16, float* base, int N, int C, int H, int W, void* stream);

/* ---- fused 3x3 convolution (forward and input-gradient) -----------------------------------
 * Replaces nn.Conv2d(k=3,s=1,p=1) plus its elementwise neighbours:
 *   conv+ReLU                 models/LarvaNet.py:210-211, 256-257
 *   conv + torch.add(x,res)   models/LarvaNet.py:212, 217-220      (res0)
 *   ... + outer body skip     models/LarvaNet.py:246-248            (res0, res1)
 *   conv->PixelShuffle(4)->+= base   models/LarvaNet.py:258,261,263-267   (mode 1)
 *   torch.cat(features)+merge_conv   models/LarvaNetV2.py:328-330  (n_src > 1)
 * and, fed with the `wpk_bwd` image, autograd's conv input-gradient with the ReLU-backward
 * mask (mask) and the skip-connection gradient adds (res0/res1) fused.
 * src: n_src (<= 8) tensors [N][cin_per_src][H][W] (cin_per_src % 8 == 0) read as one
 * channel-concatenated input.  cout in {32, 48, 64}.
 * Epilogue order: +bias -> relu -> (mask > 0 ? v : 0) -> +res0 -> +res1.
 * mode 0: out [N][cout][H][W]; supported fusions: none | relu | mask | res0 | res0+res1.
 * mode 1: out [N][cout/16][4H][4W] = PixelShuffle(4)(conv) (+ base, same shape, may be NULL). */
int larva_conv3x3_fwd(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                      const float* bias, const float* res0, const float* res1, const float* mask,
                      const float* base, float* out, int N, int cout, int H, int W, int relu,
                      int mode, void* stream);

/* Same, for operands whose rows are `pitch` >= W floats apart (src, res0, res1, mask, mode-0
 * out); columns [W, pitch) of the inputs must be zero and are written as zero.  Lets a width that
 * is not a multiple of 4 (DIV2K x4 LR images are 510 wide) take the 16-byte staging path. */
int larva_conv3x3_fwd_pitched(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                              const float* bias, const float* res0, const float* res1, const float* mask,
                              const float* base, float* out, int N, int cout, int H, int W, int pitch,
                              int relu, int mode, void* stream);

/* njobs (2..4) independent convolutions of one shape and one fusion in ONE launch: at the training
 * shape a conv launch is one workgroup per CU, the kernel fits two, and the jobs' workgroups fill
 * each other's prologue / epilogue bubbles.  src: njobs * n_src pointers (job-major); the other
 * operands arrays of njobs pointers, or NULL when unused by all jobs.  hipErrorNotSupported (801)
 * when the 16-byte staging path does not apply: issue the jobs one by one then. */
int larva_conv3x3_fwd_batch(int njobs, const float* const* src, int n_src, int cin_per_src,
                            const float* const* wpk, const float* const* bias, const float* const* res0,
                            const float* const* res1, const float* const* mask, const float* const* base,
                            float* const* out, int N, int cout, int H, int W, int pitch, int relu, int mode,
                            void* stream);

/* LarvaHead.forward (models/LarvaNet.py:223-233: nn.Conv2d(3, 48, 3, 1, 1), no activation) as a
 * direct convolution: x [N][3][H][W] unpadded, w [cout][3][3][3] in PyTorch layout (no packed image),
 * bias [cout] or NULL, out [N][cout][H][pitch] with columns [W, pitch) zeroed; cout % 16 == 0.
 * Bandwidth-bound (0.44 MB in, 7.08 MB out at 16x3x48x48): A/B against the same layer on the MFMA
 * kernel (image zero-padded to 16 channels) in profiles/. */
int larva_head_conv3_direct(const float* x, const float* w, const float* bias, float* out, int N, int cout,
                            int H, int W, int pitch, void* stream);

/* njobs (2..4) exits of the training step, each scored by nn.L1Loss inside the conv launch
 * (models/LarvaNet.py:104-109: out_i = leg(fea_i, base); loss += L1(out_i, truth)): the image
 * PixelShuffle(4)(conv) + base is compared with `truth` in the accumulators; partial[j] receives
 * larva_exit_l1_partials(N, H, pitch) partial sums of |out - truth| (added in index order by
 * larva_loss_from_partials: reproducible), grad[j] the gradient sign(out - truth) * gval
 * (sign(0) = 0) in the pixel-unshuffled layout [N][cout][H][pitch]; out[j] ([N][cout/16][4H][4W])
 * may be NULL when the image itself is not wanted.  Replaces larva_conv3x3_fwd_batch(mode 1) +
 * larva_l1_partial_grad_batch: one launch and four 14 MB sweeps less per step.  cout 48, 16-byte
 * staging path only (801 otherwise). */
int larva_exit_l1_partials(int N, int H, int pitch);
int larva_conv3x3_exit_l1_batch(int njobs, const float* const* src, int n_src, int cin_per_src,
                                const float* const* wpk, const float* const* bias, const float* const* base,
                                const float* const* truth, float* const* out, float* const* grad,
                                float* const* partial, float gval, int N, int cout, int H, int W, int pitch,
                                void* stream);

/* Strip tiles: the same convolution (bit-identical results) with the image cut into 5 x 16 and
 * 4 x 16 pixel tiles instead of 3 x 48.  Half a training batch (8 x 48 x 48) is then 256 workgroups
 * like the whole batch is with 3 x 48 tiles, so the two halves of a batch can run as two independent
 * layer chains on two streams -- two workgroups per CU, out of phase, each chain's launch boundary,
 * prologue and store burst hidden under the other's K loop (the reference's layer chain,
 * models/LarvaNet.py:205-220,236-248, has no other independent work to offer).
 * larva_strip_tile_table: host-side table of ONE H x W image's tiles (entry = y0 | x0 << 12 |
 * five_rows << 31; the two heights alternate along the table, `phase` 0 / 1 = it starts with a
 * 5-row / 4-row tile), returns the tile count (> cap: table truncated) or < 0 when H cannot be cut
 * into 5s and 4s.  larva_conv3x3_fwd_strips: larva_conv3x3_fwd_pitched with `tile_tab` = a DEVICE
 * copy of larva_strip_tile_table(H, pitch) and `tile_tab_host` = the HOST array it filled, or NULL (given, and small
 * enough -- <= 64 tiles per image, H <= 256, pitch <= 2048 --, the table travels inside the kernel arguments and a
 * workgroup finds its tile without a dependent memory round trip); plain_stores = write the output with plain instead
 * of non-temporal stores; cout 32, 48 or 64 and the 16-byte staging path only (hipErrorNotSupported otherwise).  An
 * image sub-range of a batch is addressed by offsetting the operand pointers and passing its image count as N.
 * (ABI version 5: the host table joined the signature; the `_mb` variants of version 4 are gone with the ReLU sign-bit
 * masks they carried -- built, exact and measured no faster in round 4, profiles/r04_ab_maskbits_*.txt.) */
int larva_strip_tile_table(int H, int W, int phase, unsigned* tab, int cap);
int larva_conv3x3_fwd_strips(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                             const float* bias, const float* res0, const float* res1, const float* mask,
                             const float* base, float* out, int N, int cout, int H, int W, int pitch,
                             int relu, int mode, const unsigned* tile_tab, const unsigned* tile_tab_host,
                             int tiles_per_image, int plain_stores, void* stream);

/* larva_conv3x3_fwd_pitched with the tile height of the whole-tensor launch chosen by the caller.  tile_rows 0 = the
 * library's choice, as every other entry point makes it: 3 x 48-pixel tiles; 4 x 48 for 32-channel launches of more
 * tiles than the chip has workgroup slots (339 x 510: 32.9 instead of 39.1 us per layer); and a launch of more tiles
 * than slots -- a whole validation image, validate.py:94-102 -- runs as one PERSISTENT workgroup per slot that walks its
 * tiles (round 5; LARVA_PERSIST=0 in the environment: one workgroup per tile).  3 / 4 = that height (4 exists on the
 * 16-byte path at 48 / 32 channels for the epilogues of an inference forward: hipErrorNotSupported otherwise).  The
 * results do not depend on the tiling, bit for bit. */
int larva_conv3x3_fwd_tiled(const float* const* src, int n_src, int cin_per_src, const float* wpk,
                            const float* bias, const float* res0, const float* res1, const float* mask,
                            const float* base, float* out, int N, int cout, int H, int W, int pitch,
                            int relu, int mode, int tile_rows, void* stream);

/* ---- weight / bias gradient ---------------------------------------------------------------
 * Replaces autograd's conv weight/bias gradient for the call sites above (loss.backward(),
 * models/LarvaNet.py:113).  njobs (<= 64) same-shape layers per call; job i reads dy[i]
 * [N][cout][H][W] and x[i] [N][cin][H][W], uses partial[i] (larva_wgrad_partial_floats()
 * floats) as workspace and OVERWRITES dw[i] ([cout][w_cin_total[i]][3][3], channels
 * [cin_off[i], cin_off[i]+cin_valid[i])) and db[i] ([cout], may be NULL).
 * (cout, cin) in {(48,48), (48,16), (32,32), (64,64)}.  Deterministic (no atomics). */
long long larva_wgrad_partial_floats(int cout, int cin, int splits);
/* Workgroups of the (cout, cin) weight-gradient kernel that share a CU (1 or 2: the small shapes -- (32,32), the (C,16)
 * heads -- fit twice and hide each other's staging phases): launch 256 * this many workgroups in total to fill the chip. */
int larva_wgrad_cu_share(int cout, int cin);
int larva_conv3x3_wgrad(const float* const* dy, const float* const* x, float* const* partial,
                        float* const* dw, float* const* db, const int* cin_off,
                        const int* cin_valid, const int* w_cin_total, int njobs, int splits,
                        int N, int cout, int cin, int H, int W, void* stream);

/* Phase 1 as ONE grid of nwg workgroups over all njobs (<= 64) layers: their tiles form one sequence
 * cut evenly over the workgroups (no CU idles at any layer count; a workgroup whose share crosses a
 * layer boundary contributes a partial image to both layers).  48 -> 48, 32 -> 32 or 64 -> 64 channels (the last
 * as two passes over 32 input channels each per workgroup and layer), W % 4 == 0, 16-byte
 * aligned tensors (hipErrorNotSupported otherwise).  splits_out[i] = partial images of layer i for
 * larva_wgrad_reduce; partial[i] holds larva_wgrad_flat_max_splits(njobs, nwg, tiles per layer) images
 * (tiles per layer = N * ceil(H/3) * ceil(W/48)). */
int larva_wgrad_flat_max_splits(int njobs, int nwg, int tiles_per_layer);
int larva_conv3x3_wgrad_partial_flat(const float* const* dy, const float* const* x, float* const* partial,
                                     int njobs, int nwg, int N, int cout, int cin, int H, int W,
                                     int* splits_out, void* stream);
/* The same grid with LarvaHead's weight gradient (models/LarvaNet.py:227-233; a (48, 16) layer: head_x16 = the
 * head's input zero-padded to 16 channels, [N][16][H][W]) as the tail of the sequence: the last workgroups take
 * its tiles (priced at 0.7 of a 48 -> 48 tile) instead of a launch of its own behind this one.  head_partial holds
 * larva_wgrad_flat_head_splits(njobs, nwg, tiles per layer) images of larva_wgrad_partial_floats(48, 16, 1)
 * floats; *head_splits_out = the images written (what larva_wgrad_reduce needs for that layer). */
int larva_wgrad_flat_head_splits(int njobs, int nwg, int tiles_per_layer);
int larva_conv3x3_wgrad_partial_flat_head(const float* const* dy, const float* const* x, float* const* partial,
                                          int njobs, const float* head_dy, const float* head_x16, float* head_partial,
                                          int nwg, int N, int H, int W, int* splits_out, int* head_splits_out,
                                          void* stream);

/* The two phases separately: partial images for njobs (<= 64) layers, and the fixed-order
 * reduction of up to 64 layers' partial images (each with its own split count and kernel shape)
 * in one launch. */
int larva_conv3x3_wgrad_partial(const float* const* dy, const float* const* x, float* const* partial,
                                int njobs, int splits, int N, int cout, int cin, int H, int W,
                                int* splits_used, void* stream);
int larva_wgrad_reduce(const float* const* partial, float* const* dw, float* const* db,
                       const int* cin_off, const int* cin_valid, const int* w_cin_total,
                       const int* splits, const int* cout, const int* cin, int njobs, void* stream);

/* larva_wgrad_reduce + larva_loss_from_partials in ONE launch (same arguments, same arithmetic as the two):
 * the loss of a training step (models/LarvaNet.py:104-109) is not needed before the step ends, so its
 * finishing block rides on the last launch of backward. */
int larva_wgrad_reduce_with_loss(const float* const* partial, float* const* dw, float* const* db,
                                 const int* cin_off, const int* cin_valid, const int* w_cin_total,
                                 const int* splits, const int* cout, const int* cin, int njobs,
                                 const float* const* terms, const int* count, const float* scale, int nterms,
                                 float divisor, float* loss_out, void* stream);

/* ---- base image -----------------------------------------------------------------------------
 * F.interpolate(x, scale_factor=4, mode='bicubic', align_corners=False), models/LarvaNet.py:283-285.
 * in [N][C][H][W] -> out [N][C][4H][4W]. */
int larva_bicubic4_fwd(const float* in, float* out, int N, int C, int H, int W, void* stream);
/* F.interpolate(x, scale_factor=4, mode, align_corners=False) for the modes with which the reference's call
 * (models/LarvaNet.py:57,283-285) does not raise: mode 0 bicubic (as above), 1 bilinear.  out 16-byte aligned. */
int larva_upsample4_fwd(const float* in, float* out, int N, int C, int H, int W, int mode, void* stream);

/* ---- L1 loss ---------------------------------------------------------------------------------
 * nn.L1Loss() forward/backward, models/LarvaNet.py:85,108,113.  Pointers 16-byte aligned. */
int larva_l1_workspace_floats(void);
int larva_l1_fwd(const float* a, const float* b, long long numel, float* partial, float* loss,
                 void* stream);
int larva_l1_bwd(const float* a, const float* b, const float* gout, long long numel, float* ga,
                 void* stream);
/* nn.L1Loss backward fused with the nn.PixelShuffle(4) backward of the exit it scores
 * (models/LarvaNet.py:108,113,261): a, b [N][C][4H][4W] -> ga [N][16C][H][W]. */
/* ga = sign(a - b) * (gout[0] * gscale) / numel; gscale carries the 1/num_modules of the mean over
 * exits (models/LarvaNet.py:109) so that no separate scaling pass is needed. */
int larva_l1_bwd_unshuffle4(const float* a, const float* b, const float* gout, float gscale, float* ga, int N,
                            int C, int H, int W, void* stream);
/* `loss += ...; loss / num_modules` (models/LarvaNet.py:104-109): out = (sum of n <= 8 device scalars) / divisor. */
int larva_sum_scalars(const float* const* terms, int n, float divisor, float* out, void* stream);
/* The same loss tail in two launches less per exit: larva_l1_partial leaves the block partial sums
 * of sum|a - b| (*blocks <= larva_l1_workspace_floats() floats), larva_loss_from_partials computes
 * ( sum_i scale[i] * sum(terms[i][0..count[i])) ) / divisor for n <= 8 terms (scale = 1 / numel for
 * partial sums; count 1, scale 1 for a ready scalar) -- bit-identical to larva_l1_fwd per exit
 * followed by larva_sum_scalars. */
int larva_l1_partial(const float* a, const float* b, long long numel, float* partial, int* blocks_out,
                     void* stream);
int larva_loss_from_partials(const float* const* terms, const int* count, const float* scale, int n,
                             float divisor, float* out, void* stream);
/* `return loss.item()` (models/LarvaNet.py:139) without waiting for the rest of the step: larva_host_cell_alloc
 * hands out 8 bytes {float value; uint32 sequence} of coherent, device-mapped pinned host memory (NaN, 0 on return);
 * the _to_host variant stores the loss there as well, together with sequence + 1, as ONE system-scope 8-byte release
 * store, and the host polls the cell instead of synchronising: it counts its launches and takes the value once the
 * sequence number is its own count (a late store of an earlier launch cannot be mistaken for this one's).
 * host_cell may be NULL (= larva_loss_from_partials).  larva_host_cell_free waits for the device before it frees. */
int larva_loss_from_partials_to_host(const float* const* terms, const int* count, const float* scale, int n,
                                     float divisor, float* out, float* host_cell, void* stream);
/* ... _seq: the same with the cell's sequence number ALSO kept in device memory (`dev_seq`, one zero-initialised
 * unsigned owned by the caller, used by no other cell): the launch takes the number from there instead of reading the
 * host cell back over PCIe before it can store (NULL = larva_loss_from_partials_to_host). */
int larva_loss_from_partials_to_host_seq(const float* const* terms, const int* count, const float* scale, int n,
                                         float divisor, float* out, float* host_cell, unsigned* dev_seq, void* stream);
int larva_host_cell_alloc(float** cell);
int larva_host_cell_free(float* cell);
/* larva_l1_partial and larva_l1_bwd_unshuffle4 in one pass over (a, b), for a gradient value known
 * on the host (training seeds loss.backward() with 1): grad [N][16C][H][W] = sign(a - b) * gvalue *
 * gscale / numel, a, b [N][C][4H][4W]; same partial sums and gradient values as the two calls. */
int larva_l1_partial_grad(const float* a, const float* b, float gvalue, float gscale, float* partial,
                          int* blocks_out, float* grad, int N, int C, int H, int W, void* stream);
/* The same for n <= 8 exit images a[i] scored against one truth image b, in one launch. */
int larva_l1_partial_grad_batch(const float* const* a, const float* b, int n, float gvalue, float gscale,
                                float* const* partial, int* blocks_out, float* const* grad, int N, int C, int H,
                                int W, void* stream);

/* ---- PixelShuffle(4) backward (models/LarvaNet.py:261): in [N][C][4H][4W] -> out [N][16C][H][W] */
int larva_pixel_unshuffle4(const float* in, float* out, int N, int C, int H, int W, void* stream);

/* ---- AdamW over a flat buffer (optim.AdamW, models/LarvaNet.py:86-88,114) ------------------
 * step_lr: device floats {step (1-based), lr}.  g is multiplied by grad_scale first
 * (1/world_size after a sum all-reduce).  Buffers that are all 16-byte aligned are walked 16 bytes per lane
 * (any n; the tail is element-wise), others element-wise; the values are the same either way. */
int larva_adamw_step(float* p, const float* g, float* m, float* v, const float* step_lr,
                     double beta1, double beta2, double eps, double weight_decay, float grad_scale,
                     long long n, void* stream);

/* The same update with step (1-based) and lr from the host.  The hyper-parameters are DOUBLES, as torch.optim.AdamW's
 * are (Python floats): 1 - beta^step, lr / (1 - beta1^step), 1 - lr * weight_decay, 1 - beta1 and 1 - beta2 are
 * computed in double and rounded to float once, exactly the scalars ATen's kernels receive (in float arithmetic
 * 1.f - 0.999f is 0.00100004673, not 0.001f).  The kernel follows torch's single-tensor step operation by operation:
 * against torch 2.10 on the CPU the moments are bit-identical, the parameters on all but ~0.1 % of the elements (1 ulp). */
int larva_adamw_step_host(float* p, const float* g, float* m, float* v, int step, double lr, double beta1,
                          double beta2, double eps, double weight_decay, float grad_scale, long long n,
                          void* stream);
/* larva_adamw_step_host that also copies one float, copy_dst[0] = copy_src[0] (both may be NULL): the step's
 * loss out of the captured graph's static buffer into a tensor the caller keeps (models/LarvaNet.py:139). */
int larva_adamw_step_host_copy(float* p, const float* g, float* m, float* v, int step, double lr, double beta1,
                               double beta2, double eps, double weight_decay, float grad_scale, long long n,
                               const float* copy_src, float* copy_dst, void* stream);

/* ---- device-resident patch sampler ----------------------------------------------------------
 * Crop + np.rot90(k) + horizontal flip + uint8->float of a training batch from a dataset held
 * in HBM (dataloaders/div2k_train_loader.py:72-98 and the H2D copy of train_larva.py:123-124).
 * data: uint8 CHW images at data + offsets[i], (H, W) = hw[2i], hw[2i+1]; draws[b] =
 * {image, x, y, k, flip}; crop origin (y*mult, x*mult), size P x P; out [B][3][P][P] float. */
int larva_gather_patches(const unsigned char* data, const long long* offsets, const int* hw,
                         const int* draws, float* out, int B, int P, int mult, void* stream);

/* ---- validation metric on the device (validate.py:17-27) --------------------------------------
 * acc[0] += sum (truth_u8 - clip(rint(out), 0, 255))^2 over out [C][H][W] float against truth
 * [C][TH][TW] uint8 cropped top-left; exact integer.  Caller zeroes acc. */
int larva_sqerr_u8(const float* out, const unsigned char* truth, int C, int H, int W, int TH, int TW,
                   unsigned long long* acc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LARVA_HIP_H */
