#!/usr/bin/env python3
"""Un-profiled wall time (graph replays, back to back) of the step's regions, dual chain on / off:
forward only (head + 32 body convs + exits + loss) and forward + backward."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x = (torch.rand(16, 3, 48, 48, generator=g) * 255).to(dev)
t = (torch.rand(16, 3, 192, 192, generator=g) * 255).to(dev)


def build(dual):
    m = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
    m.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"])
    torch.manual_seed(0)
    m.prepare(is_training=True, scales=[4])
    m.dual_chain = dual
    return m


def timed(replay, reps=50):
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            replay()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / reps * 1e6)
    return sorted(out)[1]


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, capture_error_mode="thread_local"):
        fn()
    return gph


for dual in (False, True):
    m = build(dual)

    def fwd():
        with m._scope():
            loss, _ = m._exit_losses(x, t)
        return loss

    def fwd_bwd():
        from larvanet_amd.autograd import DeferredWgrad
        with m._scope():
            loss, _ = m._exit_losses(x, t)
            loss.backward(m._grad_one(loss))

    def bodies_only():
        from larvanet_amd.autograd import DualChain
        with m._scope():
            net = m.model
            fea = net.head(x)
            for i in range(4):
                fea = getattr(net, "body_%d" % i)(fea)
            DualChain.join()

    with torch.no_grad():
        pass
    gb = capture(bodies_only)
    tb = timed(gb.replay)
    gf = capture(fwd)
    tf = timed(gf.replay)
    gfb = capture(fwd_bwd)
    tfb = timed(gfb.replay)
    print("dual_chain=%-5s  head+32 body convs %7.1f us   forward (all exits, loss) %7.1f us   forward+backward %7.1f us"
          % (dual, tb, tf, tfb))
