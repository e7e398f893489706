#!/usr/bin/env python3
"""Headline benchmark: HR Mpixels/s of the LarvaNet x4 multi-exit TRAINING step on MI355X.

  python bench.py --gpus N --steps K --warmup W [--extras]

N > 1 works both ways: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or as a plain
`python bench.py --gpus N`: the parent then starts one child process per GPU itself -- before it
has touched the GPU in any way -- relays rank 0's JSON line and exits non-zero if a rank fails.

A step = one pass of the hot path over one synthetic batch per GPU: head conv, 4 bodies x 4
residual blocks, 4 pixel-shuffle exits, 4 L1 losses, full backward (dgrad + wgrad + bias grads),
gradient all-reduce over RCCL when N > 1, AdamW -- i.e. train_step_larva (models/LarvaNet.py:98-114
of the reference) at the BASELINE configuration: 16 x 3 x 48 x 48 fp32 patches per GPU -> 16 x 3 x
192 x 192, `--num_modules=4 --num_blocks=4,4,4,4`, 48 channels (the only channel count the
reference can express).  value = N * 16 * 192 * 192 / t_step (pixels counted once).
The K-step loop is timed --rounds times (barrier + synchronize on both sides of each); value and
ms_per_step are the MEDIAN round, min / max are reported beside it.  The loop has the reference's
semantics: every step is handed fresh device tensors (train_larva.py:123-128) and returns loss.item()
(models/LarvaNet.py:139).

OUTPUT.  stdout carries exactly ONE line: a COMPACT JSON object (<= 4 KB, no prose fields; `compact_line`
below names every key it keeps).  The FULL record -- every block with its method notes -- goes to
`bench_full.json` (in $LARVA_BENCH_FULL, else gpurun_out/ when it exists, else beside this file) and, as one
line prefixed `bench_full: `, to stderr.

Blocks of the default run (all in the compact line):
  roofline            fp32-MFMA roofline of the dominant kernel (fused conv3x3+ReLU, 48->48, 16x48x48) the way
                      the step runs it: two concurrent half-batch strip-tile launches per layer; avg_ms = replay
                      of a captured 40-link graph / 40 (the step's own chain length), settled like the headline,
                      median of 7 sets with frac_min / frac_max; beside it the steady-state slope, the rocprofv3
                      lone-launch average and the PMC MFMA-busy fraction read from the committed profiles/
  roofline_wgrad      all weight gradients of the step, priced on what the step pays (captured forward+backward
                      with and without them)
  step                the whole step's 183.7 GFLOP over ms_per_step against the fp32 matrix peak
  infer               inference forward (LarvaNetModule.forward) on the batch
  infer_full_image    V1 on one 3 x 339 x 510 image (BASELINE config 5 at N = 1) + its dominant layer
  cpu_baseline        the same training step in the torch CPU restatement (oracle/, kind "port") on the host
                      cores of this box, bounded sample; configs[0] (EDSR-baseline CPU step) beside it
  rccl_ranks, allreduce_exposed_us, dp_schedule, ms_per_step_per_rank     N > 1 only
--extras adds (full record only): the whole-batch single chain, 32 / 64 channels, V2 full-image inference, the
data-parallel schedules on one GPU with and without a one-rank RCCL communicator, the async / resident loop, the
chains' in-step marker timing, the sustained clock.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dmabuf IPC for RCCL: in the environment before the HIP runtime comes up (children inherit it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

BATCH, PATCH, SCALE, CH = 16, 48, 4, 48
BLOCKS = [4, 4, 4, 4]
FLAGS = ["--num_modules=4", "--num_blocks=4,4,4,4"]
HR_PIX_PER_BATCH = BATCH * (PATCH * SCALE) ** 2          # 589 824
FP32_MFMA_PEAK_TFLOPS = 157.3                            # MI355X_MICROARCH.md: Peak FP32 (matrix)
FULL_IMAGE = (3, 339, 510)                               # DIV2K-val-like LR image
# HBM-side bytes per launch, the MFMA-busy fraction and the lone-launch duration come from the committed rocprofv3 passes
# (counters cannot be collected by this run itself); they are READ from the CSVs under profiles/ at run time -- newest
# round first -- and the file used is named in the record: a kernel change that is not followed by new passes shows as a
# stale file name, not as a silently wrong constant.  FETCH_SIZE is doubled per the guide's gfx950 correction for
# 16-byte-per-lane streams; both counters are in KB.
PROFILE_ROUNDS = ("r06", "r05")
PMC_WGRAD_LAYERS = 40
SIMDS = 256 * 4                                          # MFMA-busy cycles are summed over the chip's SIMDs


def profile_csv(stem):
    """profiles/<round>_<stem>.csv of the newest round that has one (tools/profile_r06.sh writes them)."""
    for r in PROFILE_ROUNDS:
        rel = "profiles/%s_%s.csv" % (r, stem)
        if os.path.exists(os.path.join(ROOT, rel)):
            return rel
    raise SystemExit("bench.py: no profiles/{%s}_%s.csv -- roofline.traffic is taken from the committed PMC passes"
                     % (",".join(PROFILE_ROUNDS), stem))


def pmc_mean(path, kernel_substr, counter):
    """mean_per_launch of `counter` for the first kernel whose name contains `kernel_substr` in a committed PMC
    summary (columns: pass,kernel,counter,launches,mean_per_launch,mean_duration_us)."""
    import csv
    with open(os.path.join(ROOT, path), newline="") as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["kernel"] and row["counter"] == counter:
                return float(row["mean_per_launch"])
    raise SystemExit("bench.py: no %s row for a kernel matching %r in %s" % (counter, kernel_substr, path))


def hbm_traffic_bytes(path, kernel_substr):
    return (2.0 * pmc_mean(path, kernel_substr, "FETCH_SIZE") + pmc_mean(path, kernel_substr, "WRITE_SIZE")) * 1024.0


def mfma_busy_cycles_per_simd(path, kernel_substr):
    """SQ_VALU_MFMA_BUSY_CYCLES per launch / the chip's 1024 SIMDs: the cycles one SIMD's matrix pipe is busy in a launch
    (= MFMAs per SIMD x 32 for v_mfma_f32_16x16x4_f32: the PMC check that the kernel issues the algorithmic minimum)."""
    return pmc_mean(path, kernel_substr, "SQ_VALU_MFMA_BUSY_CYCLES") / SIMDS


def rocprof_avg_us(stem, kernel_substr):
    """(AverageNs / 1e3, calls, file) of a kernel in a committed `rocprofv3 --kernel-trace --stats` summary."""
    import csv
    path = profile_csv(stem)
    with open(os.path.join(ROOT, path), newline="") as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Name"]:
                return float(row["AverageNs"]) / 1e3, int(row["Calls"]), path
    return None, 0, path


def conv_traffic(dual):
    """(bytes per LAYER, source) of the fused conv+ReLU layer: two half-batch strip launches, or one whole-batch launch."""
    p = profile_csv("pmc_conv")
    if dual:
        return 2 * hbm_traffic_bytes(p, "conv3x3_mfma_strip_kernel<48, 1>"), p
    return hbm_traffic_bytes(p, "conv3x3_mfma_kernel<48, true, 1>"), p


def wgrad_traffic_per_layer():
    """Weight-gradient kernel (the flat grid over 40 layers) + its reduction, per layer."""
    p = profile_csv("pmc_wgrad")
    return (hbm_traffic_bytes(p, "wgrad3x3_pipe_flat_kernel<48, 48>") +
            hbm_traffic_bytes(p, "wgrad_reduce_kernel")) / float(PMC_WGRAD_LAYERS), p


def conv_flop(c):
    return 2 * 9 * c * c * BATCH * PATCH * PATCH         # 1.5288 GFLOP per 48->48 layer


def infer_flop_per_lr_pixel(blocks, c, v2):
    """algorithmic FLOP per LR pixel of the inference forward (2 * 9 * cin * cout per 3x3 conv): the head
    3 -> c, every body's c -> c convs, and V1: the LAST leg (c -> c, c -> 48); V2: the merge conv (M * c -> c) and the
    tail (c -> c, c -> 48).  c = 48, M4B4, V1: 1 412 640."""
    f = 2 * 9 * 3 * c + 2 * sum(blocks) * 2 * 9 * c * c + 2 * 9 * c * c + 2 * 9 * c * 48
    if v2:
        f += len(blocks) * 2 * 9 * c * c
    return f


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """Start one child per rank (this process never touches the GPU), relay rank 0's stdout (the
    JSON line), fail if any rank fails.  Children are this same script under RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT, i.e. exactly what torch.distributed.run would set."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        # (rank 0 writes one short line: its pipe cannot fill up while we wait)
        time.sleep(0.2)
    for r, p in enumerate(procs):
        if failed is None and p.returncode != 0:
            failed = (r, p.returncode)
    if failed is not None:
        for p in procs:          # the exact children started above, nothing else
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: rank %d exited with code %s\n" % failed)
        return 1
    out = procs[0].stdout.read().decode()
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0 if out.strip() else 1


# ------------------------------------------------------------------------------------------------
class TinyValLoader:
    """train_step_larva validates once at global_step == 1 (models/LarvaNet.py:116-117); that
    happens inside the warm-up steps.  One small synthetic pair is enough."""

    def get_num_images(self):
        return 1

    def get_image_pair(self, image_index, scale):
        import numpy as np
        rng = np.random.RandomState(3)
        return (rng.randint(0, 256, (3, 24, 24)).astype(np.float32),
                rng.randint(0, 256, (3, 96, 96)).astype(np.float32), "synthetic")


def barrier_sync(dist_on):
    import torch
    import torch.distributed as td
    if dist_on:
        td.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def chain_operands(dev, c, chain=40, decaying=False):
    """Operands of the roofline chains: x0 = N(0,1) * 20 activations that STAY at that scale down the
    chain.  conv + ReLU with zero bias is positively homogeneous in the weights, so one eager pass over the chain
    measures its gain and the weights are rescaled by gain^(-1/chain): the RMS after `chain` layers equals the RMS
    of x0 (every layer in between within a few per cent of it).  Layer 0 always reads x0, which nothing overwrites,
    so every replay computes the same thing.  decaying=True: round 2's operands (ones in, weights x 0.05: the
    activations shrink ~30x per layer and are exactly zero from the first replay on) -- kept for one A/B figure.
    Returns (x0, packed weights, bias, [buf_a, buf_b], rms of the last layer's output)."""
    import torch
    from larvanet_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(BATCH, c, PATCH, PATCH, generator=g) * 20).to(dev)
    w = (torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev)
    b = torch.zeros(c, device=dev)
    bufs = [torch.empty_like(x0), torch.empty_like(x0)]
    if decaying:
        fwd, _ = K.pack_weights((torch.randn(c, c, 3, 3, generator=g) * 0.05).to(dev))
        return x0 * 0.0 + 1.0, fwd * 0.05, b, bufs, 0.0

    def run(wpk):
        src = x0
        for i in range(chain):
            K.conv3x3(src, wpk, c, bias=b, relu=True, out=bufs[i & 1])
            src = bufs[i & 1]
        return float(src.pow(2).mean().sqrt())

    fwd, _ = K.pack_weights(w)
    rms0 = float(x0.pow(2).mean().sqrt())
    gain = run(fwd) / rms0
    if not (gain > 0 and gain == gain and gain != float("inf")):
        raise SystemExit("bench.py: roofline chain calibration failed (gain %r)" % gain)
    fwd = fwd * gain ** (-1.0 / chain)
    return x0, fwd, b, bufs, run(fwd)


def replay_stats(graph, reps=10, sets=7, settle=True):
    """Event-timed sets of `reps` back-to-back replays -> {"median", "min", "max", "sets": [ms per replay ...],
    "settle_blocks"}.  settle: the rule of the headline loop in front of the timed sets -- untimed blocks of `reps`
    replays until two consecutive blocks agree within 0.5 % (at most 20): a chain replayed behind idle time runs its first
    few hundred microseconds on a ramping clock, which is what spread round 5's `roofline.frac` over 0.640-0.683 while
    the step time held +-0.4 %."""
    import torch

    def one():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    blocks, prev = 0, None
    for _ in range(20 if settle else 1):
        cur = one()
        blocks += 1
        if prev is not None and abs(cur - prev) <= 0.005 * prev:
            break
        prev = cur
    vals = [one() for _ in range(sets)]
    srt = sorted(vals)
    return {"median": srt[len(srt) // 2], "min": srt[0], "max": srt[-1], "sets": vals, "settle_blocks": blocks}


def replay_ms(graph, reps=10):
    """Median ms per replay of 3 event-timed sets (no settle phase): the extras' short form."""
    return replay_stats(graph, reps, sets=3, settle=False)["median"]


# Layers per captured chain.  The roofline fraction is priced on the SHORT chain (replay / 40: the chain length of the
# training step, fixed cost of a replay included -- graph launch, the chains ramping up, the last drain: ~55 us per
# replay in profiles/r03_dual_chain_overlap.txt).  The slope between the long and the short chain, (t(160) - t(40)) / 120
# = what one more layer costs in the steady state, is reported beside it as `*_steady_state`, never as the headline.
CHAIN_SHORT, CHAIN_LONG = 40, 160


def chain_graphs(dev, c, dual, chain=CHAIN_SHORT, decaying=False, lengths=(CHAIN_SHORT, CHAIN_LONG)):
    """Captured graphs of the fused conv3x3+ReLU layer chained `n` times (each launch reads the previous one's output),
    n in `lengths`.  dual=False: whole-batch launches (3 x 48 tiles, conv3x3_mfma_kernel) on one stream.  dual=True: the
    way the training step runs its layer chain (autograd.DualChain): two half-batch chains of strip-tile launches
    (5 x 16 / 4 x 16 pixel tiles, 256 workgroups per launch, plain stores) on two streams inside one graph.
    -> ({n: graph}, rms of the calibrated chain's last output), or None when the strip tiling does not apply."""
    import torch
    from larvanet_amd import kernels as K
    if dual:
        for phase in (0, 1):
            if K.strip_tile_table(PATCH, PATCH, dev, phase) is None:
                return None
    x0, wpk, b, bufs, rms = chain_operands(dev, c, chain, decaying)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    parts = ((0, BATCH // 2), (BATCH // 2, BATCH))

    def body(n):
        if not dual:
            for i in range(n):
                # (the calibrated chain is `chain` layers long: every `chain` layers it starts again from x0)
                K.conv3x3(x0 if i % chain == 0 else bufs[(i - 1) & 1], wpk, c, bias=b, relu=True, out=bufs[i & 1])
            return
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        for i in range(n):
            src = x0 if i % chain == 0 else bufs[(i - 1) & 1]
            for k, st in enumerate(streams):
                with torch.cuda.stream(st):
                    K.conv3x3(src, wpk, c, bias=b, relu=True, out=bufs[i & 1], images=parts[k],
                              strips=2 if k else True, plain_stores=True)
        for st in streams:
            cur.wait_stream(st)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body(2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs = {}
    for n in lengths:
        graphs[n] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graphs[n], capture_error_mode="thread_local"):
            body(n)
    return graphs, rms


def dual_chain_time_ms(dev, c=CH, chain=CHAIN_SHORT, reps=10, decaying=False, dual=True):
    """(steady-state slope in ms per layer, rms of the last output, replay / 40 in ms) of the two-chain graph, settled,
    median of 5 sets: the short form the tools under tools/ use (diag_overlap.py, probe_pair_chain.py)."""
    res = chain_graphs(dev, c, dual, chain, decaying)
    if res is None:
        return None
    graphs, rms = res
    t40 = replay_stats(graphs[CHAIN_SHORT], reps, sets=5)["median"]
    t160 = replay_stats(graphs[CHAIN_LONG], reps, sets=3, settle=False)["median"]
    return (t160 - t40) / (CHAIN_LONG - CHAIN_SHORT), rms, t40 / CHAIN_SHORT


def chain_time_ms(dev, c, chain=CHAIN_SHORT, reps=10, decaying=False):
    """The same for one chain of whole-batch launches."""
    return dual_chain_time_ms(dev, c, chain, reps, decaying, dual=False)


def diag_lib():
    """tools/diag_lib.py (the measurement library tools/build_diag.sh builds: kernel-attached launch timing lives there,
    not in the product's C ABI), or None when it has not been built."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("diag_lib", os.path.join(ROOT, "tools", "diag_lib.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod if mod.available() else None


def strip_launch_alone_ms(dev, c, iters=50):
    """(mean, min) ms of one half-batch strip-tile conv+ReLU launch running alone (kernel-attached events: the
    measurement library only; None without it)."""
    import torch
    from larvanet_amd import kernels as K
    D = diag_lib()
    if D is None:
        return None
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(BATCH, c, PATCH, PATCH, generator=g) * 20).to(dev)
    w = (torch.randn(c, c, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.zeros(c, device=dev)
    fwd, _ = K.pack_weights(w)
    out = torch.empty_like(x)
    try:
        D.conv3x3_strips_timed(x, fwd, c, b, out, 5, images=(0, BATCH // 2), relu=True)
        return D.conv3x3_strips_timed(x, fwd, c, b, out, iters, images=(0, BATCH // 2), relu=True)
    except RuntimeError:
        return None


def roofline_block(dev, c=CH, dual=True, quick=False, extras=False, sets=7):
    """fp32-MFMA roofline of the fused conv3x3+ReLU layer at 16 x c x 48 x 48.  dual=True: the layer as the training
    step runs it -- two concurrent half-batch launches of conv3x3_mfma_strip_kernel; dual=False: one chain of
    whole-batch launches of conv3x3_mfma_kernel.  `avg_ms` is the time per FULL-BATCH layer either way.

    achieved = algorithmic FLOP per layer (2 * 9 * c * c per LR pixel x 36 864 pixels) / avg_ms, with
    avg_ms = MEDIAN over `sets` event-timed sets of 10 back-to-back replays of a captured 40-link graph / 40, taken
    after the settle rule of replay_stats; frac_min / frac_max are the slowest / fastest set.  `frac_steady_state`:
    the slope (t160 - t40) / 120.  quick: a short run for rocprofv3 --pmc passes (every dispatch is serialised there)."""
    res = chain_graphs(dev, c, dual)
    if res is None:
        return None
    graphs, rms = res
    reps = 2 if quick else 10
    short = replay_stats(graphs[CHAIN_SHORT], reps, sets=1 if quick else sets, settle=not quick)
    long_ = replay_stats(graphs[CHAIN_LONG], reps, sets=1 if quick else 3, settle=False)
    ms40 = short["median"] / CHAIN_SHORT
    slope = (long_["median"] - short["median"]) / (CHAIN_LONG - CHAIN_SHORT)
    flop = conv_flop(c)
    tf = lambda ms: flop / (ms * 1e-3) / 1e12
    kname = ("conv3x3_mfma_strip_kernel<%d, 1>" if dual else "conv3x3_mfma_kernel<%d, true, 1>") % c
    blk = {"bound": "mfma", "achieved": tf(ms40), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": tf(ms40) / FP32_MFMA_PEAK_TFLOPS,
           "frac_min": tf(short["max"] / CHAIN_SHORT) / FP32_MFMA_PEAK_TFLOPS,
           "frac_max": tf(short["min"] / CHAIN_SHORT) / FP32_MFMA_PEAK_TFLOPS,
           "traffic": None, "avg_ms": ms40, "sets": len(short["sets"]), "settle_blocks": short["settle_blocks"],
           "avg_ms_sets": [v / CHAIN_SHORT for v in short["sets"]],
           "avg_ms_steady_state": slope, "frac_steady_state": tf(slope) / FP32_MFMA_PEAK_TFLOPS,
           "kernel": kname + (" x2 concurrent half-batch strip launches = one 16x%dx48x48 conv3x3+bias+ReLU layer" % c if dual
                              else " one whole-batch launch, 16x%dx48x48 conv3x3+bias+ReLU" % c),
           "launches_per_layer": 2 if dual else 1, "flop_per_launch": flop // 2 if dual else flop, "flop_per_layer": flop,
           "algorithmic_bytes_per_layer": 2 * BATCH * c * PATCH * PATCH * 4 + 4 * (9 * c * c + c),
           "chain_links": CHAIN_SHORT, "last_output_rms": rms,
           "timing": "HIP event pairs around 10 back-to-back replays of a captured %d-link chain, / %d; settle blocks until two "
                     "agree within 0.5 %%, then median of %d sets (min / max -> frac_max / frac_min); steady state = "
                     "(t%d - t%d) / %d" % (CHAIN_SHORT, CHAIN_SHORT, sets, CHAIN_LONG, CHAIN_SHORT, CHAIN_LONG - CHAIN_SHORT)}
    if c == CH:
        blk["traffic"], src = conv_traffic(dual)
        blk["traffic_source"] = src
        blk["traffic_is"] = ("HBM-side bytes per LAYER from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH doubled per "
                             "the gfx950 correction), read at run time from the committed summary; not measured by this run")
        us, calls, stats = rocprof_avg_us("bench_kernel_stats", kname)
        if us is not None:
            # rocprofv3 serialises the two chains: its per-dispatch average is ONE launch running alone (half a layer when
            # dual).  mfma_busy_frac = the PMC pass's matrix-pipe cycles per SIMD over that duration at the 2.4 GHz the peak
            # is quoted at: equal to frac_lone_launch exactly when the kernel issues the algorithmic minimum of MFMAs
            lone_flop = flop // 2 if dual else flop
            cyc = mfma_busy_cycles_per_simd(src, kname)
            blk["rocprof"] = {"kernel_stats": stats, "pmc": src, "lone_launch_us": us, "calls": calls,
                              "frac_lone_launch": lone_flop / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                              "mfma_busy_cycles_per_simd": cyc, "mfma_busy_frac": cyc / 2.4e3 / us}
    if extras and not quick:
        if dual:
            alone = strip_launch_alone_ms(dev, c)
            if alone is not None:
                blk["launch_alone_ms"], blk["launch_alone_min_ms"] = alone
        if c == CH:
            g2, _ = chain_graphs(dev, c, dual, decaying=True, lengths=(CHAIN_SHORT,))
            blk["avg_ms_decaying_inputs"] = replay_ms(g2[CHAIN_SHORT]) / CHAIN_SHORT   # round 2's operands, for comparison
    return blk


def wgrad_block(dev, c=CH, jobs=40, iters=10):
    """Second kernel of the step (28 % of it): the weight-gradient launch as the step issues it -- ONE flat grid of 256
    workgroups over the tiles of all 40 C -> C layers (partial images) + the fixed-order reduction -- replayed from a
    captured graph and timed with an event pair.  c = 48: the pipelined kernel (the 3 -> 48 head, which the step
    appends to the same grid, is not part of this block: its FLOPs are not in `flop_per_layer` either).  Other channel
    counts (BASELINE configs 2 / 5 read at 32 / 64 channels): the launch the library has for them, 32 layers x 8
    workgroups."""
    import torch
    from larvanet_amd import kernels as K
    g = torch.Generator().manual_seed(6)
    dy = (torch.randn(BATCH, c, PATCH, PATCH, generator=g) * 1e-3).to(dev)
    xs = (torch.randn(BATCH, c, PATCH, PATCH, generator=g) * 20).to(dev)
    js = [{"dy": dy + 0, "x": xs + 0, "dw": torch.empty(c, c, 3, 3, device=dev), "db": torch.empty(c, device=dev)}
          for _ in range(jobs)]

    def pair():
        res = K.conv3x3_wgrad_partial_flat(js, c, c, 256)
        if res is None:   # (the flat grid does not apply: the per-layer launch, one or two workgroups per CU in total)
            K.conv3x3_wgrad(js[:32], c, c, 8 * K.wgrad_cu_share(c, c))
            return False
        K.wgrad_reduce([dict(j, partial=p, splits=s, cout=c, cin=c) for j, p, s in zip(js, *res)])
        return True

    flat = pair()
    nlayers = jobs if flat else 32
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        pair()
    ms = replay_stats(graph, iters, sets=3, settle=iters >= 10)["median"]
    achieved = conv_flop(c) * nlayers / (ms * 1e-3) / 1e12
    kernel = (("wgrad3x3_pipe_flat_kernel<64, 32>, two passes over 32 input channels each per workgroup and layer" if c == 64 else
               "wgrad3x3_pipe_flat_kernel<%d, %d>" % (c, c)) + " (one grid of 256 workgroups over %d layers) + wgrad_reduce_kernel" % nlayers
              if flat else "wgrad3x3 kernel for (%d, %d) + wgrad_reduce_kernel, 32 layers x %d workgroups" % (c, c, 8 * K.wgrad_cu_share(c, c)))
    blk = {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "kernel": kernel + ", 16x%dx48x48 fp32" % c, "layers": nlayers,
           "ms_per_launch_pair": ms, "ms_per_layer": ms / nlayers, "flop_per_layer": conv_flop(c), "traffic": None,
           "algorithmic_bytes_per_layer": 2 * BATCH * c * PATCH * PATCH * 4 + 4 * (9 * c * c + c),
           "timing": "HIP event pair around %d replays of a captured graph of the launch pair, back to back (median of 3)" % iters}
    if c == CH:
        per_layer, blk["traffic_source"] = wgrad_traffic_per_layer()
        blk["traffic"] = per_layer * nlayers   # HBM-side bytes of the launch pair (dy + x read once, partial images written and read once)
    return blk


def wgrad_in_step(model, x, truth, reps=10):
    """What the weight gradients cost INSIDE the training step: the captured forward+backward replayed
    with and without its deferred weight-gradient launches (32 + 8 layers, the 3 -> 48 head, one
    reduction); the difference is their time at the clocks and cache state the step gives them (an
    isolated back-to-back loop of the same launch runs ~10 % slower: sustained fp32-MFMA load pulls
    the clock down)."""
    import torch
    from larvanet_amd.autograd import DeferredWgrad

    def body(with_wgrad):
        model._zero_grad()
        with model._scope():
            loss, _ = model._exit_losses(x, truth)
            loss.backward(model._grad_one(loss))
            if not with_wgrad:
                DeferredWgrad.drop()

    times = {}
    for with_wgrad in (True, False):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body(with_wgrad)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            body(with_wgrad)
        times[with_wgrad] = replay_stats(graph, reps, sets=5, settle=True)["median"]
    layers = sum(BLOCKS) * 2 + 2 * len(BLOCKS)                      # 40 C->C layers
    flop = layers * conv_flop(CH) + 2 * 9 * 3 * CH * BATCH * PATCH * PATCH   # + the 3 -> 48 head
    ms = times[True] - times[False]
    achieved = flop / (ms * 1e-3) / 1e12
    return {"ms_all_weight_gradients": ms, "fwd_bwd_ms": times[True], "fwd_bwd_without_wgrad_ms": times[False],
            "layers": layers, "flop": flop, "achieved": achieved, "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
            "what": "captured forward+backward replayed with and without its deferred weight-gradient launches "
                    "(HIP events, settled, median of 5 x %d replays)" % reps}


def chains_in_step():
    """The layer chains INSIDE the captured training step, product kernels: one-lane marker launches (larva_stamp_clock)
    at the fork and at the end of each chain, forward and backward (tools/step_marks.py; four markers per phase, ~2 us each
    on their stream).  The forward chain is head + 32 conv(+ReLU / +residual) layers; fork -> last chain's end over its
    algorithmic FLOPs is the fraction of the matrix peak the step's own chain runs at -- the figure `roofline.frac`
    (a captured 40-link conv+ReLU chain, replay / 40) stands in for."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("step_marks", os.path.join(ROOT, "tools", "step_marks.py"))
    sm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sm)
    names, queued, lone, event_us, host_us = sm.measure(link_marks=())
    ph = sm.chain_phases(names, queued)
    # (the markers perturb the two chains' phase relation: the backward chain of a marked graph often sits in the slow mode
    # -- 580-600 us against ~485 in the unmarked step, profiles/r06_step_timeline_stamped.txt -- so these are upper bounds)
    out = {"graph_replay_us_with_markers": event_us, "host_us_per_graph_launch": host_us, "perturbed_by_markers": True,
           "what": "tools/step_marks.py: markers between the product kernels of the captured forward+backward, last of 10 "
                   "back-to-back replays; fork -> the later chain's end"}
    head_flop = 2 * 9 * 3 * CH * BATCH * PATCH * PATCH
    if "fwd" in ph:
        us, links = ph["fwd"]
        flop = (links - 1) * conv_flop(CH) + head_flop
        out["forward_chain"] = {"us": us, "links": links, "us_per_layer": us / links, "flop": flop,
                                "achieved": flop / us / 1e6, "frac": flop / us / 1e6 / FP32_MFMA_PEAK_TFLOPS}
    if "bwd" in ph:
        us, links = ph["bwd"]
        joint = len(BLOCKS) - 1                      # K = 96 joint input gradients: two layers' work in one link
        flop = (links + joint) * conv_flop(CH)
        out["backward_chain"] = {"us": us, "links": links, "joint_k96_links": joint, "us_per_layer": us / (links + joint),
                                 "flop": flop, "achieved": flop / us / 1e6, "frac": flop / us / 1e6 / FP32_MFMA_PEAK_TFLOPS}
    return out


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota (a GPU
    box hands each job a share of a large host; 256 threads on a 16-CPU share thrash)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    cores = min(cores, max(1, q // period))
        except Exception:
            continue
    return min(cores, int(os.environ.get("LARVA_CPU_BASELINE_THREADS", "32")))


def cpu_baseline(budget_s=10.0):
    """The reference CPU path (torch CPU operators, all host cores) on the same workload."""
    import numpy as np
    import torch
    from oracle import larva_torch as T
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = T.init_state_dict(BLOCKS, seed=0)
    x = torch.rand(BATCH, 3, PATCH, PATCH, generator=torch.Generator().manual_seed(0)) * 255
    truth = torch.rand(BATCH, 3, PATCH * SCALE, PATCH * SCALE, generator=torch.Generator().manual_seed(1)) * 255
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=4e-4)

    def step():
        loss = T.multi_exit_loss(params, x, truth, BLOCKS)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss.item()

    for _ in range(3):   # warm-ups
        step()
    times = []
    t_start = time.perf_counter()
    while len(times) < 10 or (time.perf_counter() - t_start < budget_s and len(times) < 50):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    # forward-only leg (LarvaNetModule.forward under no_grad, models/LarvaNet.py:287-293), same protocol
    fwd_sd = {k: v.detach() for k, v in params.items()}
    with torch.no_grad():
        for _ in range(3):
            T.forward(fwd_sd, x, BLOCKS)
        ftimes = []
        t_start = time.perf_counter()
        while len(ftimes) < 10 or (time.perf_counter() - t_start < budget_s / 3 and len(ftimes) < 50):
            t0 = time.perf_counter()
            T.forward(fwd_sd, x, BLOCKS)
            ftimes.append(time.perf_counter() - t0)
    fmed = float(np.median(ftimes))
    # BASELINE configs[0]: EDSR-baseline x4 (64 features, 16 residual blocks), batch 4 of 48x48 LR patches, the
    # reference's own CPU case (models/edsr.py:75-108 through train.py:83-105) -- oracle/edsr_torch.py, pinned by F15
    from oracle import edsr_torch as E
    estep = E.make_trainer(E.init_state_dict(seed=0), 16)
    ex = torch.rand(4, 3, PATCH, PATCH, generator=torch.Generator().manual_seed(2)) * 255
    et = torch.rand(4, 3, PATCH * SCALE, PATCH * SCALE, generator=torch.Generator().manual_seed(3)) * 255
    for _ in range(3):
        estep(ex, et)
    etimes = []
    t_start = time.perf_counter()
    while len(etimes) < 10 or (time.perf_counter() - t_start < budget_s / 3 and len(etimes) < 30):
        t0 = time.perf_counter()
        estep(ex, et)
        etimes.append(time.perf_counter() - t0)
    emed = float(np.median(etimes))
    edsr = {"value": 4 * (PATCH * SCALE) ** 2 / emed / 1e6, "unit": "HR Mpixels/s", "ms_per_step": emed * 1e3,
            "sample": "%d EDSR-baseline x4 train steps (64f/16RB, batch 4 of 3x48x48), BASELINE configs[0], CPU only" % len(etimes)}
    return {"value": HR_PIX_PER_BATCH / med / 1e6, "unit": "HR Mpixels/s", "cores": cores, "kind": "port",
            "ms_per_step": med * 1e3, "edsr_train_step": edsr,
            "sample": "%d train steps after 3 warm-ups, same M4B4 batch-16 workload, torch %s CPU ops, %d threads"
                      % (len(times), torch.__version__.split("+")[0], cores),
            "forward_only": {"value": HR_PIX_PER_BATCH / fmed / 1e6, "unit": "HR Mpixels/s", "ms_per_batch": fmed * 1e3,
                             "sample": "%d forwards after 3 warm-ups" % len(ftimes)}}


def full_image_block(dev, extras=False):
    """BASELINE config 5 at N = 1: whole-network inference of one 3 x 339 x 510 LR image (x4 ->
    1356 x 2040), device tensor in, device tensor out (upscale()'s H2D / D2H copies of the reference API are
    not in the timed region).  Default: V1; extras: V2 and V2 with 64 filters as well."""
    import importlib
    import torch
    out = {"lr_image": list(FULL_IMAGE), "hr_pixels": 16 * FULL_IMAGE[1] * FULL_IMAGE[2]}
    x = (torch.rand(1, *FULL_IMAGE, generator=torch.Generator().manual_seed(2)) * 255).to(dev)
    nets = [("LarvaNet", [])] + ([("LarvaNetV2", []), ("LarvaNetV2", ["--num_filters=64"])] if extras else [])
    for name, extra in nets:
        m = importlib.import_module("larvanet_amd.models." + name).create_model()
        m.parse_args(list(FLAGS) + extra)
        torch.manual_seed(0)
        m.prepare(is_training=False, scales=[SCALE])
        with torch.no_grad():
            for _ in range(3):
                m.fwd_runtime(x)
            reps, runs = 10, []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    m.fwd_runtime(x)
                torch.cuda.synchronize()
                runs.append((time.perf_counter() - t0) / reps * 1e3)
            ms = sorted(runs)[len(runs) // 2]
        key = name + ("_64ch" if extra else "")   # (BASELINE configs[4] reads "LarvaNetV2, 64ch body": the --num_filters extension)
        flop = infer_flop_per_lr_pixel(BLOCKS, 64 if extra else CH, v2=name.endswith("V2")) * FULL_IMAGE[1] * FULL_IMAGE[2]
        out[key] = {"ms_per_image": ms, "ms_min": min(runs), "ms_max": max(runs),
                    "value": out["hr_pixels"] / (ms * 1e-3) / 1e6, "unit": "HR Mpixels/s", "flop": flop,
                    "frac_of_peak": flop / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
        del m
    out["roofline"] = full_image_layer_block(dev, extras=extras)
    return out


def full_image_layer_block(dev, c=CH, extras=False):
    """The dominant kernel of the full-image forward the way `roofline` names the training step's: one 48 -> 48 conv3x3 of the
    339 x 510 image (pitch 512) -- what validate.py's forward issues 34 times per image -- as a captured graph of 20
    dependent launches on the persistent tiles the library picks for it (conv + ReLU; extras: per epilogue, and as one
    workgroup per tile under LARVA_PERSIST=0).  Algorithmic work 2 * 9 * c * c FLOP per LR pixel."""
    import torch
    from larvanet_amd import kernels as K
    H, W = FULL_IMAGE[1], FULL_IMAGE[2]
    P = (W + 3) // 4 * 4
    g = torch.Generator().manual_seed(0)
    x = torch.zeros(1, c, H, P, device=dev)
    x[..., :W] = (torch.randn(1, c, H, W, generator=g) * 20).to(dev)
    r0, r1 = x.flip(1).contiguous(), x.flip(2).contiguous()
    w = (torch.randn(c, c, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.zeros(c, device=dev)
    fwd, _ = K.pack_weights(w)
    bufs = [torch.empty_like(x) for _ in range(2)]
    flop = 2 * 9 * c * c * H * W
    kinds = {"relu": dict(relu=True)}
    modes = [("1", "persistent")]
    if extras:
        kinds.update({"res1": dict(res0=r0), "res2": dict(res0=r0, res1=r1)})
        modes.append(("0", "one_workgroup_per_tile"))
    res = {}
    keep = os.environ.get("LARVA_PERSIST")
    try:
        for mode, tag in modes:
            os.environ["LARVA_PERSIST"] = mode
            res[tag] = {}
            for name, kw in kinds.items():
                def chain():
                    src = x
                    for i in range(20):
                        K.conv3x3(src, fwd, c, bias=b, out=bufs[i & 1], logical_w=W, tile_rows=3, **kw)
                        src = bufs[i & 1]
                chain()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    chain()
                st = replay_stats(graph, 5, sets=5, settle=True)
                us = st["median"] * 1e3 / 20
                res[tag][name] = {"us_per_layer": us, "frac": flop / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                  "us_min": st["min"] * 1e3 / 20, "us_max": st["max"] * 1e3 / 20}
    finally:
        if keep is None:
            os.environ.pop("LARVA_PERSIST", None)
        else:
            os.environ["LARVA_PERSIST"] = keep
    tiles = ((H + 2) // 3) * ((P + 47) // 48)
    best = res["persistent"]["relu"]
    kname = "conv3x3_mfma_persist_kernel<%d, 1>" % c
    blk = {"bound": "mfma", "achieved": best["frac"] * FP32_MFMA_PEAK_TFLOPS, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": best["frac"], "traffic": None, "algorithmic_bytes": 2 * 4 * c * H * W,
           "kernel": kname + " conv3x3+bias+ReLU on a 1x%dx%dx%d image: %d tiles of 3x48 px, persistent workgroups" % (c, H, W, tiles),
           "flop_per_launch": flop, "avg_us": best["us_per_layer"], "by_epilogue": res,
           "timing": "HIP event pairs around 5 replays of a captured graph of 20 dependent launches (settled, median of 5)"}
    if c == CH:
        p = profile_csv("pmc_infer")
        blk["traffic"], blk["traffic_source"] = hbm_traffic_bytes(p, kname), p
        us, calls, stats = rocprof_avg_us("infer_LarvaNet_kernel_stats", kname)
        if us is not None:
            blk["rocprof"] = {"kernel_stats": stats, "launch_us": us, "calls": calls,
                              "frac": flop / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    return blk


# ------------------------------------------------------------------------------------------------
# the ONE stdout line
# ------------------------------------------------------------------------------------------------
COMPACT_MAX_BYTES = 4096
STR_MAX = 120


def _num(v):
    """Floats to 6 significant digits (the line is a scoreboard, not an archive: bench_full.json keeps every digit)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float("%.6g" % v)


def _pick(d, keys):
    """The named keys of a block: strings cut to STR_MAX characters, floats to 6 digits, containers dropped unless they are
    short lists of numbers.  A missing key is skipped; None is kept (traffic / vs_baseline may be null by contract)."""
    out = {}
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            continue
        v = d[k]
        if isinstance(v, str):
            out[k] = v[:STR_MAX]
        elif isinstance(v, (list, tuple)):
            if len(v) <= 8 and all(isinstance(e, (int, float, bool)) for e in v):
                out[k] = [_num(e) for e in v]
        elif isinstance(v, dict):
            continue
        else:
            out[k] = _num(v)
    return out


ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_min", "frac_max", "sets", "traffic", "avg_ms",
                 "frac_steady_state", "kernel", "flop_per_layer", "flop_per_launch", "launches_per_layer", "error")


def compact_line(full):
    """The stdout line: the contract's keys + `roofline` + `cpu_baseline` + one-number summaries of the other blocks, no
    prose.  Everything else stays in the full record.  Raises if the result is not <= COMPACT_MAX_BYTES (tests/test_bench_line.py)."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_min",
                        "ms_per_step_max", "settle_steps", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                        "bench_wall_s", "gpu_fault", "dry_run", "max_rank_plus_1", "backend"))
    line["config"] = _pick(full.get("config", {}), ("workload", "global_batch", "parallelism", "loss_sync_per_step",
                                                    "hip_graph", "hip_graph_fell_back", "final_loss"))
    if "roofline" in full:
        r = full["roofline"] or {}
        line["roofline"] = _pick(r, ROOFLINE_KEYS)
        if isinstance(r.get("rocprof"), dict):
            line["roofline"]["rocprof"] = _pick(r["rocprof"], ("kernel_stats", "lone_launch_us", "frac_lone_launch", "mfma_busy_frac"))
        if "traffic_source" in r:
            line["roofline"]["traffic_source"] = str(r["traffic_source"])[:STR_MAX]
    if "cpu_baseline" in full:
        c = full["cpu_baseline"]
        line["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind", "ms_per_step", "sample", "error"))
        if isinstance(c.get("edsr_train_step"), dict):
            line["cpu_baseline"]["edsr_ms_per_step"] = _num(c["edsr_train_step"].get("ms_per_step"))
        if isinstance(c.get("forward_only"), dict):
            line["cpu_baseline"]["forward_ms_per_batch"] = _num(c["forward_only"].get("ms_per_batch"))
    if "step" in full:
        line["step"] = _pick(full["step"], ("frac_of_peak", "achieved", "flop_per_step", "sustained_clock_ghz"))
    if "roofline_wgrad" in full:
        line["roofline_wgrad"] = _pick(full["roofline_wgrad"], ("bound", "frac", "achieved", "ms_all_weight_gradients",
                                                                "ms_per_launch_pair", "traffic", "kernel", "error"))
    if "infer" in full:
        line["infer"] = _pick(full["infer"], ("ms_per_batch", "value", "frac_of_peak"))
    fi = full.get("infer_full_image")
    if isinstance(fi, dict):
        blk = _pick(fi, ("lr_image", "error"))
        if isinstance(fi.get("LarvaNet"), dict):
            blk.update(_pick(fi["LarvaNet"], ("ms_per_image", "ms_min", "ms_max", "value", "frac_of_peak")))
        if isinstance(fi.get("roofline"), dict):
            blk["roofline"] = _pick(fi["roofline"], ("bound", "frac", "achieved", "peak", "unit", "avg_us", "traffic", "kernel"))
        line["infer_full_image"] = blk
    # N > 1: the data-parallel fields
    line.update(_pick(full, ("rccl_ranks", "dist_backend")))
    if isinstance(full.get("allreduce_exposed_us"), dict):
        line["allreduce_exposed_us"] = _pick(full["allreduce_exposed_us"], ("median", "min", "max", "steps", "overlap"))
    if isinstance(full.get("dp_schedule"), dict):
        line["dp_schedule"] = _pick(full["dp_schedule"], ("choice", "allreduce_isolated_us", "bucket_bytes", "ranks", "forced"))
    if isinstance(full.get("ms_per_step_per_rank"), dict):
        line["ms_per_step_per_rank"] = _pick(full["ms_per_step_per_rank"], ("min", "max", "ranks"))
    if "full_record" in full:
        line["full_record"] = str(full["full_record"])[-STR_MAX:]
    text = json.dumps(line, separators=(",", ":"))
    if len(text.encode()) > COMPACT_MAX_BYTES:
        raise AssertionError("bench.py: the compact line is %d bytes (> %d)" % (len(text.encode()), COMPACT_MAX_BYTES))
    return text


def full_record_path():
    d = os.environ.get("LARVA_BENCH_FULL")
    if d:
        return d if d.endswith(".json") else os.path.join(d, "bench_full.json")
    scratch = os.path.join(ROOT, "gpurun_out")
    return os.path.join(scratch if os.path.isdir(scratch) else ROOT, "bench_full.json")


class GpuFault(Exception):
    """The HIP context is poisoned (a fault inside an extra): no further GPU work, non-zero exit."""


def guarded(line, key, fn):
    """An EXTRA of the JSON line: a pure-Python failure inside it is recorded under its key instead of costing the run
    its headline, `roofline` and `cpu_baseline`.  A failure that leaves the HIP context unusable (an illegal access or a
    launch failure surfaces as a RuntimeError and is sticky) is NOT downgraded: the device is synchronised after every
    failed extra, and if that raises too the line gets `gpu_fault`, the remaining GPU extras are skipped (GpuFault) and
    the process exits non-zero after emitting what it has."""
    if line.get("gpu_fault") or (guarded.fault is not None):
        line[key] = {"skipped": "earlier GPU fault"}
        return
    try:
        line[key] = fn()
    except Exception as e:   # noqa: BLE001 (deliberately broad: extras must not take the line down)
        line[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        sys.stderr.write("bench.py: extra %r failed: %s: %s\n" % (key, type(e).__name__, e))
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception as e2:   # noqa: BLE001
            guarded.fault = "%s in extra %r, then synchronize: %s: %s" % (type(e).__name__, key, type(e2).__name__, e2)
            sys.stderr.write("bench.py: the HIP context is unusable after extra %r: %s\n" % (key, guarded.fault))


guarded.fault = None


def timed_rounds(model, args, val, x, truth, steps, rounds, dist_on):
    """`rounds` x (barrier + sync, `steps` steps, barrier + sync) -> seconds per round, max over ranks."""
    import torch
    import torch.distributed as td
    secs = []
    loss = None
    for _ in range(rounds):
        barrier_sync(dist_on)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = model.train_step_larva(args, val, x, truth)
        barrier_sync(dist_on)
        secs.append(time.perf_counter() - t0)
    per_rank = None
    if dist_on:
        t = torch.tensor(secs, dtype=torch.float64, device=model.device)
        # each rank's own wall time of every round: a sum all-reduce of a [world, rounds] table with one row filled
        # (all-reduce is the one collective every backend has for device tensors; gloo has no device all-gather)
        table = torch.zeros(td.get_world_size(), len(secs), dtype=torch.float64, device=model.device)
        table[td.get_rank()] = t
        td.all_reduce(table, op=td.ReduceOp.SUM)
        per_rank = [[float(v) for v in r] for r in table.cpu()]
        td.all_reduce(t, op=td.ReduceOp.MAX)
        secs = [float(v) for v in t.cpu()]
    timed_rounds.per_rank = per_rank
    return secs, loss


def dp_probe_child(a, emit):
    """`bench.py --dp-probe`: one GPU, the data-parallel step's two weight-gradient schedules side by side -- wall ms per
    step (reference semantics: fresh tensors + loss.item()), HOST microseconds per step (async loss, the host never waits
    for the GPU: what one rank's Python thread spends enqueueing a step -- eight ranks share one host), and, under
    LARVA_DIST_FORCE=1, everything the N > 1 line carries (`rccl_ranks`, the timed all-reduce of prepare(), the exposed
    all-reduce time) measured through a ONE-rank RCCL communicator.  A rehearsal of the code path, not a scaling number."""
    import importlib
    import numpy as np
    import torch
    import torch.distributed as td
    from larvanet_amd import dist as ldist
    rank, world = ldist.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(1000)
    x = (torch.rand(BATCH, 3, PATCH, PATCH, generator=g) * 255).to(dev)
    truth = (torch.rand(BATCH, 3, PATCH * SCALE, PATCH * SCALE, generator=g) * 255).to(dev)
    args = types.SimpleNamespace(train_path="/tmp")
    val = TinyValLoader()
    out = {"communicator": ("%s, %d rank(s)" % (td.get_backend(), world)) if ldist.active() else None,
           "rccl_ranks": world if ldist.active() and td.get_backend() == "nccl" else 0,
           "what": dp_probe_child.__doc__.split("\n\n")[0].replace("\n    ", " ")}
    for name in ("flat", "split"):
        os.environ["LARVA_OVERLAP_ALLREDUCE"] = "1" if name == "split" else "0"
        os.environ["LARVA_FORCE_SPLIT"] = "1" if name == "split" else "0"
        try:
            m = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
        finally:
            os.environ.pop("LARVA_OVERLAP_ALLREDUCE", None)
            os.environ.pop("LARVA_FORCE_SPLIT", None)
        m.parse_args(list(FLAGS))
        torch.manual_seed(0)
        m.volume_per_step = PATCH * PATCH * BATCH * 3
        m.prepare(is_training=True, scales=[SCALE])
        m.time_allreduce = ldist.active()
        for _ in range(max(a.warmup, 3)):
            m.train_step_larva(args, val, x, truth)
        if hasattr(m, "allreduce_events"):
            m.allreduce_events.clear()
        secs, _ = timed_rounds(m, args, val, x, truth, a.steps, 3, False)
        ms = float(np.median([s / a.steps * 1e3 for s in secs]))
        res = {"ms_per_step": ms, "schedule_ran": m.dp_schedule.get("choice"), "late_graph": getattr(m, "_graph_late", None) is not None,
               "split_at_float": getattr(m, "_early_lo", None)}
        if getattr(m, "allreduce_events", None):
            torch.cuda.synchronize()
            gaps = sorted(s.elapsed_time(e) * 1e3 for s, e in m.allreduce_events)
            res["allreduce_exposed_us"] = {"median": gaps[len(gaps) // 2], "min": gaps[0], "max": gaps[-1], "steps": len(gaps)}
        m.time_allreduce = False
        # host time: the loss stays on the device, nothing synchronises inside the loop
        m.sync_loss = False
        for _ in range(3):
            m.train_step_larva(args, val, x, truth)
        host = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m.train_step_larva(args, val, x, truth)
            host.append((time.perf_counter() - t0) / 10 * 1e6)
            torch.cuda.synchronize()
        res["host_us_per_step"] = float(np.median(host))
        out[name] = res
    if ldist.active():
        # the bucket's isolated all-reduce as prepare() times it when the schedule is left on "auto"
        m = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
        m.parse_args(list(FLAGS))
        m.prepare(is_training=True, scales=[SCALE])
        out["dp_schedule_auto"] = m.dp_schedule
    torch.cuda.synchronize()
    emit(out)
    if ldist.is_initialized():
        td.destroy_process_group()


def run_dp_probe(a, force_dist):
    """Start `bench.py --dp-probe` as a child (its communicator must exist before its first GPU call) and parse its line."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "LARVA_DIST_FORCE"):
        env.pop(k, None)
    if force_dist:
        env["LARVA_DIST_FORCE"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dp-probe", "--steps", str(a.steps), "--warmup", str(a.warmup)],
                       env=env, stdout=subprocess.PIPE, timeout=600)
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("bench.py --dp-probe%s exited with code %d" % (" under LARVA_DIST_FORCE=1" if force_dist else "", r.returncode))
    return json.loads(lines[-1])


def dry_run(a, rank, world, emit):
    """LARVA_BENCH_DRY=1: the launcher / rendezvous / collective / JSON plumbing of a multi-rank run
    with NO kernels (CPU, gloo) -- what the CPU tests drive with 2 and 8 ranks.  Not a measurement: the record has the
    SHAPE of a real one (every N > 1 field present, placeholder numbers) so that the compact line's size and keys are
    tested at N = 8 without hardware."""
    import torch
    import torch.distributed as td
    if os.environ.get("LARVA_BENCH_DRY_FAIL_RANK") == str(rank):   # test hook: a rank that dies
        sys.exit(7)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        td.barrier()
        td.all_reduce(t, op=td.ReduceOp.MAX)
    if rank == 0:
        full = headline_record(a, world, ms_rounds=[1.0 + 0.001 * i for i in range(max(1, a.rounds))], settle_steps=20,
                               final_loss=0.0, ref_semantics=True, hip_graph=True, dual_chain=True, fell_back=None, resident=False)
        full.update({"dry_run": True, "max_rank_plus_1": float(t.item()), "backend": td.get_backend() if world > 1 else None,
                     "roofline": {"bound": "mfma", "achieved": 0.0, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 0.0,
                                  "frac_min": 0.0, "frac_max": 0.0, "sets": 7, "traffic": None, "avg_ms": 0.0,
                                  "kernel": "dry run: no kernel was launched", "flop_per_layer": conv_flop(CH)}})
        if world > 1:
            full.update({"rccl_ranks": 0, "dist_backend": td.get_backend(),
                         "allreduce_exposed_us": {"median": 0.0, "min": 0.0, "max": 0.0, "steps": a.steps * a.rounds, "overlap": False},
                         "dp_schedule": {"choice": "flat", "allreduce_isolated_us": 0.0, "bucket_bytes": 3330816, "ranks": world,
                                         "rule": "dry run"},
                         "ms_per_step_per_rank": {"min": 1.0, "max": 1.0, "ranks": [1.0] * world}})
        emit(full)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


def headline_record(a, world, ms_rounds, settle_steps, final_loss, ref_semantics, hip_graph, dual_chain, fell_back, resident):
    """The contract's keys of the record from the timed rounds (ms per step of every round, in the order they ran)."""
    import numpy as np
    per_step = sorted(ms_rounds)
    ms_per_step = float(np.median(per_step))
    return {
        "metric": "HR Mpixels/s (LarvaNet x4 multi-exit train step, 48x48 LR patches)",
        "value": world * HR_PIX_PER_BATCH / (ms_per_step * 1e-3) / 1e6, "unit": "HR Mpixels/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup,
        "settle_steps": settle_steps,   # untimed steps behind the W warm-up steps until the step time had stopped moving (clock ramp)
        "ms_per_step": ms_per_step, "ms_per_step_min": per_step[0], "ms_per_step_max": per_step[-1],
        # every timed round in the order it ran, and which of them sat > 1 % over the fastest round
        "ms_per_step_rounds": list(ms_rounds), "slow_rounds": [bool(v > 1.01 * per_step[0]) for v in ms_rounds],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LarvaNet x4 train_step_larva M4 B4,4,4,4 48ch, batch 16 of 3x48x48 -> 3x192x192 fp32 per GPU (BASELINE config 2)",
                   "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                   "inputs": ("resident in the captured step's input buffers" if resident else
                              "fresh device tensors handed to train_step_larva every step (copied into the captured step's inputs)"),
                   "loss_sync_per_step": bool(ref_semantics), "hip_graph": bool(hip_graph), "dual_chain": bool(dual_chain),
                   "hip_graph_fell_back": fell_back, "final_loss": final_loss},
        "rounds": {"n": len(ms_rounds), "steps_each": a.steps, "value_is": "median round (max over ranks of each round's wall time)"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5, help="the --steps loop is timed this many times; value = median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true",
                    help="the extra single-GPU measurements (other widths, V2 inference, one-GPU data-parallel rehearsals, ...): "
                         "full record only, + ~40 s")
    ap.add_argument("--no-extras", action="store_true", help="headline + `roofline` only (rocprofv3 --kernel-trace runs)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only time the dominant kernel (short run for rocprofv3 --pmc passes)")
    ap.add_argument("--wgrad-only", action="store_true",
                    help="only the flat weight-gradient launch + reduction (short run for rocprofv3 --pmc passes)")
    ap.add_argument("--dp-probe", action="store_true",
                    help="child mode of the `dp_schedule_1gpu` / `rccl_world1` extras (see dp_probe_child)")
    ap.add_argument("--sync-loss", action="store_true", help="(the default since round 3; kept for old command lines)")
    ap.add_argument("--async-loss", action="store_true",
                    help="headline loop without the reference's per-step loss.item() and with the batch already in the "
                         "captured step's input buffers (round 2's headline; now the `value_async_resident` extra)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher BEFORE anything touches the GPU
        sys.exit(self_launch(a.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the compact JSON: everything else any library prints there -- the
    # plugin's progress lines, Gloo's / RCCL's C-level connection messages -- goes to stderr (fd level)
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    t_start = time.perf_counter()

    def emit_raw(obj):   # child modes / PMC helper runs: one plain JSON line, no size rule
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    def emit(full):
        """Full record -> bench_full.json + stderr; compact line -> stdout (the only stdout line)."""
        full["bench_wall_s"] = time.perf_counter() - t_start
        path = full_record_path()
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            full["full_record"] = os.path.relpath(path, ROOT)
        except OSError as e:
            full["full_record"] = "not written: %s" % e
        sys.stderr.write("bench_full: " + json.dumps(full) + "\n")
        sys.stderr.flush()
        os.write(json_fd, (compact_line(full) + "\n").encode())

    if a.dp_probe:
        return dp_probe_child(a, emit_raw)

    import numpy as np
    import torch
    from larvanet_amd import dist as ldist
    dry = os.environ.get("LARVA_BENCH_DRY", "0") != "0"
    rank, world = ldist.init_from_env(backend="gloo" if dry else None)
    if a.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE is %d" % (a.gpus, world))
    if dry:
        return dry_run(a, rank, world, emit)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    dev = torch.device("cuda", torch.cuda.current_device())
    if a.roofline_only:
        emit_raw({"roofline": roofline_block(dev, dual=False, quick=True), "roofline_dual": roofline_block(dev, dual=True, quick=True)})
        return
    if a.wgrad_only:
        emit_raw({"roofline_wgrad_isolated": wgrad_block(dev, iters=3)})
        return

    import importlib
    import torch.distributed as td
    sections = {}   # wall seconds of each part of this run (full record: `sections_s`)

    def section(name, t0):
        torch.cuda.synchronize()
        sections[name] = round(time.perf_counter() - t0, 3)

    t0 = time.perf_counter()
    model = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
    model.parse_args(list(FLAGS))
    torch.manual_seed(0)
    model.volume_per_step = PATCH * PATCH * BATCH * 3 * world
    model.prepare(is_training=True, scales=[SCALE])
    # The headline loop has the reference's semantics (models/LarvaNet.py:98-139, train_larva.py:123-128): a batch of
    # fresh device tensors is handed to train_step_larva, which returns loss.item() -- a host round trip -- every step.
    ref_semantics = not a.async_loss
    model.sync_loss = ref_semantics
    model.time_allreduce = world > 1

    g = torch.Generator().manual_seed(1000 + rank)
    x_fresh = (torch.rand(BATCH, 3, PATCH, PATCH, generator=g) * 255).to(dev)
    truth_fresh = (torch.rand(BATCH, 3, PATCH * SCALE, PATCH * SCALE, generator=g) * 255).to(dev)
    args = types.SimpleNamespace(train_path="/tmp")
    val = TinyValLoader()

    for _ in range(max(a.warmup, 1)):
        model.train_step_larva(args, val, x_fresh, truth_fresh)
    if model.hip_graph_fell_back:
        # a 2.4x slower step must not pass for the captured one: single GPU = an error; under data
        # parallelism the run goes on (a flagged slow point of the scaling curve says more than no curve)
        sys.stderr.write("bench.py: hipGraph capture FAILED, the step runs as eager launches (%s)\n" % model.hip_graph_fell_back)
        if world == 1 or os.environ.get("LARVA_BENCH_STRICT_GRAPH", "0") != "0":
            sys.exit(3)
    # --async-loss: the batch sits where a device-side producer (dataloaders/device_patch_loader, `out=`) puts it: in
    # the input buffers of the captured step, so the step does not copy it again
    x, truth = x_fresh, truth_fresh
    bufs = model.input_buffers(x.shape, truth.shape)
    if bufs is not None:
        bufs[0].copy_(x)
        bufs[1].copy_(truth)
    if bufs is not None and not ref_semantics:
        x, truth = bufs
    section("prepare_and_warmup", t0)
    # Settle: the W warm-up steps above are ~10 ms of GPU work behind the idle time of start-up, and the chip needs longer
    # than that under load to reach the clocks it then holds -- the first 10 steps after ANY idle gap run ~5 % slow
    # (tools/step_ramp.py: 1.72 ms, then 1.64 flat; again after a 0.5 s pause).  So untimed steps go on in blocks of 10
    # until two consecutive blocks agree within 0.5 % (at most 10 blocks); every timed round then starts on a warm chip.
    t0 = time.perf_counter()
    settle_steps, prev = 0, None
    for _ in range(10):
        barrier_sync(world > 1)
        t1 = time.perf_counter()
        for _ in range(10):
            model.train_step_larva(args, val, x, truth)
        barrier_sync(world > 1)
        cur = time.perf_counter() - t1
        settle_steps += 10
        ok = prev is not None and abs(cur - prev) <= 0.005 * prev
        if world > 1:   # every rank must take the same decision
            flag = torch.tensor([1.0 if ok else 0.0], device=dev)
            td.all_reduce(flag, op=td.ReduceOp.MIN)
            ok = bool(flag.item() > 0.5)
        prev = cur
        if ok:
            break
    if hasattr(model, "allreduce_events"):
        model.allreduce_events.clear()
    rounds = max(1, a.rounds)
    secs, loss = timed_rounds(model, args, val, x, truth, a.steps, rounds, world > 1)
    section("settle_and_timed_rounds", t0)
    exposed = None
    if world > 1 and getattr(model, "allreduce_events", None):
        torch.cuda.synchronize()
        gaps = sorted(s.elapsed_time(e) * 1e3 for s, e in model.allreduce_events)
        # HIP events on the compute stream: after the last weight-gradient kernel was issued -> after the stream has joined
        # the collectives (AdamW may start)
        exposed = {"median": gaps[len(gaps) // 2], "min": gaps[0], "max": gaps[-1], "steps": len(gaps),
                   "overlap": bool(model.overlap_allreduce and getattr(model, "_early_lo", None))}
    model.time_allreduce = False

    if rank != 0:
        if world > 1:
            td.barrier()  # rank 0 finishes its single-GPU measurements, then everybody leaves together
            td.destroy_process_group()
        return

    line = headline_record(a, world, [s / a.steps * 1e3 for s in secs], settle_steps, float(loss), ref_semantics,
                           model.use_hip_graph, model.dual_chain, model.hip_graph_fell_back, resident=bufs is not None and not ref_semantics)
    ms_per_step = line["ms_per_step"]
    line["sections_s"] = sections
    if world > 1:
        line["rccl_ranks"] = td.get_world_size() if td.get_backend() == "nccl" else 0
        line["dist_backend"] = td.get_backend()
        line["allreduce_exposed_us"] = exposed
        line["dp_schedule"] = model.dp_schedule   # isolated all-reduce time measured at prepare() and the schedule it chose
        if getattr(timed_rounds, "per_rank", None):
            # every rank's own ms_per_step (median round): a straggler or an exposed collective shows here
            per = [float(np.median([s / a.steps * 1e3 for s in r])) for r in timed_rounds.per_rank]
            line["ms_per_step_per_rank"] = {"min": min(per), "max": max(per), "ranks": per}

    # the whole step against the fp32 matrix peak: the only fraction tied to the driver-timed number.  SURVEY 8(d):
    # forward 40 C->C convs + head, backward dgrad + wgrad per C->C conv + the head's wgrad (elementwise work excluded)
    flop_step = (2 * sum(BLOCKS) + 2 * len(BLOCKS)) * 3 * conv_flop(CH) + 2 * (2 * 9 * 3 * CH * BATCH * PATCH * PATCH)
    line["step"] = {"flop_per_step": flop_step, "achieved": flop_step / (ms_per_step * 1e-3) / 1e12, "unit": "TFLOP/s",
                    "peak": FP32_MFMA_PEAK_TFLOPS, "frac_of_peak": flop_step / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    extras = world == 1 and a.extras and not a.no_extras
    basic = world == 1 and not a.no_extras

    # The clock the chip sustains under this load (the guide's 157.3 TFLOP/s is 2.4 GHz): one napping wave on a side stream
    # counts shader cycles against the 100 MHz wall clock while 40 more steps run (measurement library; untimed).
    if extras:
        D = diag_lib()
        if D is not None:
            try:
                side = torch.cuda.Stream()
                cells = torch.zeros(2, dtype=torch.int64, device=dev)
                for _ in range(5):
                    model.train_step_larva(args, val, x, truth)
                with torch.cuda.stream(side):
                    hip_check = D.load().larva_clock_probe(3000000, cells.data_ptr(), side.cuda_stream)   # 30 ms
                for _ in range(40):
                    model.train_step_larva(args, val, x, truth)
                torch.cuda.synchronize()
                ticks, cycles = (int(v) for v in cells.tolist())
                if hip_check == 0 and ticks > 0:
                    ghz = cycles / ticks * 0.1
                    line["step"]["sustained_clock_ghz"] = ghz
                    line["step"]["frac_of_peak_at_sustained_clock"] = line["step"]["frac_of_peak"] * 2.4 / ghz
            except Exception as e:   # an extra: never costs the line
                line["step"]["sustained_clock_error"] = "%s: %s" % (type(e).__name__, e)

    # the REQUIRED block first (`roofline`): a fault inside a later block then cannot cost the line its roofline
    t0 = time.perf_counter()
    dual = roofline_block(dev, dual=True, extras=extras) if model.dual_chain else None
    # the dominant kernel as the step runs it: the pair of strip-tile launches when the layer chain
    # runs as two half-batch chains, else the whole-batch launch
    line["roofline"] = dual if dual is not None else roofline_block(dev, dual=False, extras=extras)
    section("roofline", t0)
    if basic:
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(5):
                model.fwd_runtime(x)
            runs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    model.fwd_runtime(x)
                torch.cuda.synchronize()
                runs.append((time.perf_counter() - t1) / 20 * 1e3)
            infer_ms = sorted(runs)[1]
        infer_flop = infer_flop_per_lr_pixel(BLOCKS, CH, v2=False) * BATCH * PATCH * PATCH     # 52.08 GFLOP
        line["infer"] = {"ms_per_batch": infer_ms, "value": HR_PIX_PER_BATCH / (infer_ms * 1e-3) / 1e6,
                         "unit": "HR Mpixels/s", "flop": infer_flop,
                         "frac_of_peak": infer_flop / (infer_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
        section("infer", t0)

        # priced, like `roofline`, on what the STEP pays: the captured forward+backward with and without its
        # weight-gradient launches (the back-to-back loop of the launch pair alone -- clock pulled down by sustained
        # fp32-MFMA load, operands streamed cold from HBM every replay -- is the `isolated_loop` extra)
        def wgrad_in_the_step():
            ins = wgrad_in_step(model, x, truth)
            per_layer, tsrc = wgrad_traffic_per_layer()
            blk = {"bound": "mfma", "achieved": ins["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ins["frac"],
                   "kernel": "wgrad3x3_pipe_flat_kernel<48, 48> (one grid of 256 workgroups over the step's 40 layers + the 3->48 head) "
                             "+ wgrad_reduce_kernel, 16x48x48x48 fp32",
                   "ms_all_weight_gradients": ins["ms_all_weight_gradients"], "flop": ins["flop"], "layers": ins["layers"],
                   "traffic": per_layer * ins["layers"], "traffic_source": tsrc, "timing": ins["what"], "in_step": ins}
            if extras:
                blk["isolated_loop"] = wgrad_block(dev)
            return blk
        t0 = time.perf_counter()
        guarded(line, "roofline_wgrad", wgrad_in_the_step)
        section("roofline_wgrad", t0)
    elif world > 1:
        line["roofline_wgrad"] = wgrad_block(dev)

    if extras:
        t0 = time.perf_counter()
        line["roofline_single_chain"] = roofline_block(dev, dual=False, extras=True, sets=3)
        if ref_semantics and bufs is not None:
            # round 2's headline, now an extra: no per-step loss.item() (the loss comes back as a device scalar) and the
            # batch already resident in the captured step's input buffers (what dataloaders/device_patch_loader does)
            def async_resident():
                model.sync_loss = False
                try:
                    secs2, _ = timed_rounds(model, args, val, bufs[0], bufs[1], a.steps, min(rounds, 3), False)
                finally:
                    model.sync_loss = True
                ms2 = float(np.median([s / a.steps * 1e3 for s in secs2]))
                return {"value": HR_PIX_PER_BATCH / (ms2 * 1e-3) / 1e6, "unit": "HR Mpixels/s", "ms_per_step": ms2}
            guarded(line, "value_async_resident", async_resident)

        # The weight-gradient schedule a data-parallel rank runs (two launch groups instead of one flat grid, DESIGN
        # section 5) on this one GPU: wall and HOST time per step of both schedules without a communicator
        # (`dp_schedule_1gpu`), and the same under a one-rank RCCL communicator (`rccl_world1`: the collectives really
        # run -- librccl, async all-reduce + Work.wait() between the two captured graphs, the timed all-reduce of
        # prepare()).  Each in a child process: a communicator must exist before the process's first GPU call.
        def dp_schedule():
            r = run_dp_probe(a, False)
            r.update(ms_per_step=r["split"]["ms_per_step"], value=HR_PIX_PER_BATCH / (r["split"]["ms_per_step"] * 1e-3) / 1e6,
                     unit="HR Mpixels/s", late_graph=r["split"]["late_graph"],
                     split_costs_us=(r["split"]["ms_per_step"] - r["flat"]["ms_per_step"]) * 1e3)
            return r
        guarded(line, "dp_schedule_1gpu", dp_schedule)
        # (ONE rank: a rehearsal of the backend="nccl" code path on the one GPU of this box, NOT a scaling measurement)
        guarded(line, "rccl_world1", lambda: run_dp_probe(a, True))

        # BASELINE configs 2 / 5 name 32- and 64-channel bodies, which the reference cannot express:
        # the same M4B4 network built with --num_filters (every leg's last conv kept at 48 outputs), same batch,
        # same loop semantics.  Extras: the headline stays the reference's 48-channel network.
        def width(nf):
            mw = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
            mw.parse_args(list(FLAGS) + ["--num_filters=%d" % nf])
            torch.manual_seed(0)
            mw.volume_per_step = PATCH * PATCH * BATCH * 3
            mw.prepare(is_training=True, scales=[SCALE])
            mw.sync_loss = ref_semantics
            for _ in range(max(a.warmup, 1)):
                mw.train_step_larva(args, val, x_fresh, truth_fresh)
            secs_w, _ = timed_rounds(mw, args, val, x_fresh, truth_fresh, a.steps, min(rounds, 3), False)
            ms_w = float(np.median([s / a.steps * 1e3 for s in secs_w]))
            with torch.no_grad():
                for _ in range(5):
                    mw.fwd_runtime(x_fresh)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    mw.fwd_runtime(x_fresh)
                torch.cuda.synchronize()
                inf_w = (time.perf_counter() - t1) / 20 * 1e3
            cc = (2 * sum(BLOCKS) + len(BLOCKS)) * conv_flop(nf) + len(BLOCKS) * 2 * 9 * nf * 48 * BATCH * PATCH * PATCH
            flop_w = 3 * cc + 2 * (2 * 9 * 3 * nf * BATCH * PATCH * PATCH)
            return {"train_ms_per_step": ms_w, "value": HR_PIX_PER_BATCH / (ms_w * 1e-3) / 1e6, "unit": "HR Mpixels/s",
                    "flop_per_step": flop_w, "frac_of_peak": flop_w / (ms_w * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                    "infer_ms_per_batch": inf_w, "hip_graph_fell_back": mw.hip_graph_fell_back}
        line["other_widths"] = {}
        for nf in (32, 64):
            guarded(line["other_widths"], "num_filters_%d" % nf, lambda nf=nf: width(nf))
        guarded(line["roofline"], "in_step", chains_in_step)

        def two_chain(c):   # two half-batch strip chains, like the 48-channel layer; else one chain
            blk = roofline_block(dev, c, dual=True, sets=3)
            return blk if blk is not None else roofline_block(dev, c, dual=False, sets=3)
        for c in (32, 64):
            guarded(line, "roofline_c%d" % c, lambda c=c: two_chain(c))
            guarded(line, "roofline_c%d_single_chain" % c, lambda c=c: roofline_block(dev, c, dual=False, sets=3))
            guarded(line, "roofline_wgrad_c%d" % c, lambda c=c: wgrad_block(dev, c))
        section("extras", t0)
    if basic:
        del model
        t0 = time.perf_counter()
        guarded(line, "infer_full_image", lambda: full_image_block(dev, extras=extras))
        section("infer_full_image", t0)
    if world == 1 and not a.no_cpu_baseline:
        t0 = time.perf_counter()
        line["cpu_baseline"] = cpu_baseline()
        sections["cpu_baseline"] = round(time.perf_counter() - t0, 3)
    if guarded.fault is not None:
        line["gpu_fault"] = guarded.fault
    emit(line)
    if guarded.fault is not None:
        os._exit(4)   # (the context is unusable: no orderly teardown of it)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
