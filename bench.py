#!/usr/bin/env python3
"""Headline benchmark: HR Mpixels/s of the LarvaNet x4 multi-exit TRAINING step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A step = one pass of the hot path over one synthetic batch per GPU: head conv, 4 bodies x 4
residual blocks, 4 pixel-shuffle exits, 4 L1 losses, full backward (dgrad + wgrad + bias grads),
gradient all-reduce over RCCL when N > 1, AdamW -- i.e. train_step_larva (models/LarvaNet.py:98-114
of the reference) at the BASELINE configuration: 16 x 3 x 48 x 48 fp32 patches per GPU -> 16 x 3 x
192 x 192, `--num_modules=4 --num_blocks=4,4,4,4`, 48 channels (the only channel count the
reference can express).  value = N * 16 * 192 * 192 / t_step (pixels counted once).

The JSON line also carries
  roofline      fp32-MFMA roofline of the dominant kernel (fused conv3x3+ReLU, 48->48, 16x48x48),
                timed live with events on the launch stream
  cpu_baseline  the same training step in the torch CPU restatement (oracle/, kind "port") on the
                host cores of this box, bounded sample
  infer         inference-forward throughput (LarvaNetModule.forward) as extra information
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

BATCH, PATCH, SCALE, CH = 16, 48, 4, 48
BLOCKS = [4, 4, 4, 4]
HR_PIX_PER_BATCH = BATCH * (PATCH * SCALE) ** 2          # 589 824
CONV_FLOP = 2 * 9 * CH * CH * BATCH * PATCH * PATCH      # 1.5288 GFLOP per 48->48 layer
FP32_MFMA_PEAK_TFLOPS = 157.3                            # MI355X_MICROARCH.md: Peak FP32 (matrix)
# HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE doubled per
# the guide's gfx950 correction + WRITE_SIZE), see profiles/README.md; None until measured.
HBM_TRAFFIC_PER_LAUNCH = (2 * 3919.0 + 7149.6) * 1024   # profiles/r01_h_pmc_conv3x3_relu_final.csv (KB)


class TinyValLoader:
    """train_step_larva validates once at global_step == 1 (models/LarvaNet.py:116-117); that
    happens inside the warm-up steps.  One small synthetic pair is enough."""

    def get_num_images(self):
        return 1

    def get_image_pair(self, image_index, scale):
        rng = np.random.RandomState(3)
        return (rng.randint(0, 256, (3, 24, 24)).astype(np.float32),
                rng.randint(0, 256, (3, 96, 96)).astype(np.float32), "synthetic")


def barrier_sync(dist_on):
    import torch.distributed as td
    if dist_on:
        td.barrier()
    torch.cuda.synchronize()


def time_dominant_kernel(dev, iters=50):
    """Duration of the fused conv3x3+ReLU kernel at 16x48x48x48 on the launch stream, two ways:
    kernel-attached HIP events (hipExtLaunchKernelGGL start/stop = the kernel's own begin/end, what
    rocprofv3 --kernel-trace reports) and a plain event pair around each launch (includes the
    launch gap).  roofline.achieved uses the former."""
    from larvanet_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(BATCH, CH, PATCH, PATCH, generator=g) * 20).to(dev)
    w = (torch.randn(CH, CH, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.zeros(CH, device=dev)
    fwd, _ = K.pack_weights(w)
    out = torch.empty_like(x)
    for _ in range(5):
        K.conv3x3(x, fwd, CH, bias=b, relu=True, out=out)
    torch.cuda.synchronize()
    k_mean, k_min = K.conv3x3_relu_timed(x, fwd, CH, b, out, iters)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        K.conv3x3(x, fwd, CH, bias=b, relu=True, out=out)
        e.record()
    torch.cuda.synchronize()
    pair = sorted(s.elapsed_time(e) for s, e in evs)
    # ... and the way the kernel runs inside the training step: a captured chain of dependent launches
    # (each reads the previous one's output), back to back; time per launch = replay time / length.
    chain, reps = 40, 10
    bufs = [x.clone() * 0.0 + 1.0, torch.empty_like(x)]
    wsmall = fwd * 0.05  # keeps the activations finite down the chain
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        K.conv3x3(bufs[0], wsmall, CH, bias=b, relu=True, out=bufs[1])
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for i in range(chain):
            K.conv3x3(bufs[i & 1], wsmall, CH, bias=b, relu=True, out=bufs[(i + 1) & 1])
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    in_graph = s.elapsed_time(e) / (reps * chain)
    return k_mean, k_min, float(np.mean(pair)), in_graph


def roofline_block(dev):
    k_mean_ms, k_min_ms, pair_ms, graph_ms = time_dominant_kernel(dev)
    # priced on the in-graph time per launch (what the step pays, boundaries included), which is
    # also what rocprofv3 reports for this kernel inside the captured step
    achieved = CONV_FLOP / (graph_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": HBM_TRAFFIC_PER_LAUNCH,
            "kernel": "conv3x3_mfma_kernel<48, true, 1> (fused conv3x3+bias+ReLU), 16x48x48x48 fp32",
            "flop_per_launch": CONV_FLOP, "avg_ms": graph_ms,
            "timing": "HIP event pair around 10 replays of a captured chain of 40 dependent launches, per launch",
            "isolated_kernel_attached_ms": k_mean_ms, "isolated_min_ms": k_min_ms,
            "event_pair_ms_incl_launch_gap": pair_ms,
            "algorithmic_bytes_per_launch": 2 * BATCH * CH * PATCH * PATCH * 4 + 4 * (9 * CH * CH + CH)}


def wgrad_block(dev, jobs=32, iters=20):
    """Second kernel of the step (28 % of it): the weight-gradient launch as the step issues it
    (32 layers x 8 workgroups, partial images + fixed-order reduction), timed with an event pair."""
    from larvanet_amd import kernels as K
    g = torch.Generator().manual_seed(6)
    dy = (torch.randn(BATCH, CH, PATCH, PATCH, generator=g) * 1e-3).to(dev)
    xs = (torch.randn(BATCH, CH, PATCH, PATCH, generator=g) * 20).to(dev)
    js = [{"dy": dy + 0, "x": xs + 0, "dw": torch.empty(CH, CH, 3, 3, device=dev), "db": torch.empty(CH, device=dev)}
          for _ in range(jobs)]
    parts = K.conv3x3_wgrad(js, CH, CH, 256 // jobs)
    for j, p in zip(js, parts):
        j["partial"] = p
    for _ in range(5):  # the chip needs a few hundred microseconds of load to settle its clock
        K.conv3x3_wgrad(js, CH, CH, 256 // jobs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        K.conv3x3_wgrad(js, CH, CH, 256 // jobs)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    achieved = CONV_FLOP * jobs / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "kernel": "wgrad3x3_pipe_kernel<48, 48> + wgrad_reduce_kernel, "
            "%d layers x %d workgroups, 16x48x48x48 fp32" % (jobs, 256 // jobs), "ms_per_launch_pair": ms,
            "flop_per_layer": CONV_FLOP}


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


def cpu_baseline(budget_s=15.0):
    """The reference CPU path (torch CPU operators, all host cores) on the same workload."""
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

    step()  # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 50):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": HR_PIX_PER_BATCH / med / 1e6, "unit": "HR Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%d train steps (median %.1f ms) of the same M4B4 batch-16 workload, torch %s CPU ops, %d threads"
                      % (len(times), med * 1e3, torch.__version__, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only time the dominant kernel (short run for rocprofv3 --pmc passes)")
    ap.add_argument("--sync-loss", action="store_true",
                    help="return loss.item() every step like the reference (host sync per step)")
    a = ap.parse_args()

    from larvanet_amd import dist as ldist
    rank, world = ldist.init_from_env()
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % a.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    dev = torch.device("cuda", torch.cuda.current_device())
    if a.roofline_only:
        print(json.dumps({"roofline": roofline_block(dev)}))
        return

    # stdout carries exactly ONE line, the JSON; the plugin's progress prints go to stderr
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    import importlib
    model = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
    model.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"])
    torch.manual_seed(0)
    model.volume_per_step = PATCH * PATCH * BATCH * 3 * world
    model.prepare(is_training=True, scales=[SCALE])
    model.sync_loss = bool(a.sync_loss)

    g = torch.Generator().manual_seed(1000 + rank)
    x = (torch.rand(BATCH, 3, PATCH, PATCH, generator=g) * 255).to(dev)
    truth = (torch.rand(BATCH, 3, PATCH * SCALE, PATCH * SCALE, generator=g) * 255).to(dev)
    args = types.SimpleNamespace(train_path="/tmp")
    val = TinyValLoader()

    for _ in range(max(a.warmup, 1)):
        model.train_step_larva(args, val, x, truth)
    # the batch sits where a device-side producer (dataloaders/device_patch_loader, `out=`) puts it:
    # in the input buffers of the captured step, so the step does not copy it again
    bufs = model.input_buffers(x.shape, truth.shape)
    if bufs is not None:
        bufs[0].copy_(x)
        bufs[1].copy_(truth)
        x, truth = bufs
    barrier_sync(world > 1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = model.train_step_larva(args, val, x, truth)
    barrier_sync(world > 1)
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as td
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    value = world * HR_PIX_PER_BATCH / (ms_per_step * 1e-3) / 1e6

    if rank != 0:
        if world > 1:
            import torch.distributed as td
            td.barrier()  # rank 0 finishes its extra single-GPU measurements, then everybody leaves together
            td.destroy_process_group()
        return

    # inference forward (extra information)
    with torch.no_grad():
        for _ in range(5):
            model.model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model.model(x)
        torch.cuda.synchronize()
        infer_ms = (time.perf_counter() - t0) / 20 * 1e3

    line = {
        "metric": "HR Mpixels/s (LarvaNet x4 multi-exit train step, 48x48 LR patches)",
        "value": value, "unit": "HR Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LarvaNet x4 train_step_larva, num_modules=4 num_blocks=4,4,4,4, 48 channels "
                               "(BASELINE config 2 at the reference's only channel count), batch 16 x 3x48x48 "
                               "-> 3x192x192 per GPU, fp32",
                   "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                   "inputs": "resident in the captured step's input buffers" if bufs is not None else "resident in HBM",
                   "loss_sync_per_step": bool(a.sync_loss), "hip_graph": bool(model.use_hip_graph),
                   "final_loss": float(loss)},
        "roofline": roofline_block(dev),
        "roofline_wgrad": wgrad_block(dev),
        "infer": {"ms_per_batch": infer_ms, "value": HR_PIX_PER_BATCH / (infer_ms * 1e-3) / 1e6,
                  "unit": "HR Mpixels/s"},
    }
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    sys.stdout = real_stdout
    print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
