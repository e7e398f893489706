#!/usr/bin/env python3
"""When do the two half-batch chains of the captured training step start and end -- in the PRODUCT build?

tools/diag_step.py stamps every workgroup but needs a -DLARVA_DIAG build whose stamps cost ~0.9 us per layer and may
change which phase the two chains settle in.  Here the kernels are the product's: one-lane marker launches
(larva_stamp_clock: s_memrealtime -> memory) are captured BETWEEN them, in stream order -- on each chain stream before
its first link and after links 0, 8, 16, 24 and the last one (forward and backward), and on the main stream after the
prologue, the exits' launches, the weight-gradient grid and the reduction.  A marker costs its stream one launch slot
(~2 us).  --marks-every-link: a marker after EVERY link (the link periods, at ~2 us per link).

  python tools/step_marks.py [out.txt] [--marks-every-link]      (GPU box; the plugin's environment switches apply)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(every=False, link_marks=(0, 8, 16, 24)):
    """-> (names, queued us, lone us, event us per replay, host us per graph launch).  link_marks: chain links behind which
    a marker is captured on each chain stream (() = only the chains' starts and ends: four markers per phase)."""
    import importlib
    import numpy as np
    import torch
    from larvanet_amd import autograd as A, hip_lib, kernels as K
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import diag_lib   # (larva_stamp_clock is a measurement entry point: tools/build_diag.sh, tools/larva_diag.h)
    lib = diag_lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(16, 3, 48, 48, generator=g) * 255).to(dev)
    t = (torch.rand(16, 3, 192, 192, generator=g) * 255).to(dev)
    m = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
    m.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"])
    torch.manual_seed(0)
    m.prepare(is_training=True, scales=[4])

    def body():
        m._zero_grad()
        with m._scope():
            loss, _ = m._exit_losses(x, t)
            loss.backward(m._grad_one(loss))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    marks = torch.zeros(1024, device=dev, dtype=torch.int64)
    names = []
    on = [False]

    def mark(name, stream=None):
        if not on[0]:
            return
        st = stream if stream is not None else torch.cuda.current_stream()
        hip_lib.check(lib.larva_stamp_clock(marks.data_ptr() + 8 * len(names), st.cuda_stream), "larva_stamp_clock")
        names.append(name)

    phase = {"name": "fwd", "link": 0}
    real_conv = A.DualChain.conv.__func__
    real_join = A.DualChain.join.__func__

    def conv(cls, srcs, wpk, cout, forward=False, **kw):
        first = srcs if isinstance(srcs, torch.Tensor) else srcs[0]
        n, _, h, p = (int(v) for v in first.shape)
        splits = not (kw.get("logical_w") is not None or kw.get("shuffle") or not cls.wants(n, h, p))
        if on[0] and splits and not cls._forked:
            cur = torch.cuda.current_stream()
            phase["name"], phase["link"] = ("fwd" if forward else "bwd"), 0
            mark("%s: main stream at the fork" % phase["name"])
            cls._main = cur     # (DualChain.conv's own fork: chain 0 may run on the current stream itself)
            for k in range(2):
                if cls._stream(k) != cur:
                    cls._stream(k).wait_stream(cur)
                mark("%s chain %d: start (before link 0)" % (phase["name"], k), cls._stream(k))
            cls._forked = True
        out = real_conv(cls, srcs, wpk, cout, forward=forward, **kw)
        if on[0] and splits:
            i = phase["link"]
            if every or i in link_marks:
                for k in range(2):
                    mark("%s chain %d: after link %d" % (phase["name"], k, i), cls._stream(k))
            phase["link"] = i + 1
        return out

    def join(cls):
        if on[0] and cls._forked:
            for k in range(2):
                mark("%s chain %d: end (after link %d)" % (phase["name"], k, phase["link"] - 1), cls._stream(k))
            real_join(cls)
            mark("%s: main stream after the join" % phase["name"])
            return
        real_join(cls)

    A.DualChain.conv = classmethod(conv)
    A.DualChain.join = classmethod(join)
    restore = []

    def wrap(name, label):
        real = getattr(K, name)
        restore.append((name, real))

        def fn(*a, **kw):
            out = real(*a, **kw)
            mark("main: after " + label)
            return out
        setattr(K, name, fn)

    wrap("step_prologue", "the prologue launch")
    wrap("conv3x3_batch", "a batched exits launch")
    wrap("conv3x3_exit_l1_batch", "the exits' pixel-shuffle + L1 launch")
    wrap("conv3x3_wgrad_partial_flat", "the flat weight-gradient grid")
    wrap("wgrad_reduce", "the weight-gradient reduction (+ loss)")

    on[0] = True
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        mark("main: graph start")
        body()
        mark("main: graph end")
    on[0] = False
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        runs.append(s.elapsed_time(e) / 20 * 1e3)
    event_us = sorted(runs)[1]
    # the LAST of 10 back-to-back replays (the host is far ahead: every packet was enqueued long before it ran) ...
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        graph.replay()
    host_us = (time.perf_counter() - t0) / 10 * 1e6     # host time per graph launch
    torch.cuda.synchronize()
    ahead = marks.cpu().numpy()[:len(names)].astype(np.float64) * 0.01
    # ... and a LONE replay after a device synchronisation (the host enqueues while the GPU already runs)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    lone = marks.cpu().numpy()[:len(names)].astype(np.float64) * 0.01
    A.DualChain.conv = classmethod(real_conv)
    A.DualChain.join = classmethod(real_join)
    for name, real in restore:
        setattr(K, name, real)
    del graph
    return names, ahead, lone, event_us, host_us


def chain_phases(names, arr):
    """{phase: (fork -> last chain end in us, links per chain)} from one column of measure()."""
    d = dict(zip(names, arr))
    out = {}
    for ph in ("fwd", "bwd"):
        fork = d.get("%s: main stream at the fork" % ph)
        ends = [(v, n_) for n_, v in d.items() if n_.startswith("%s chain" % ph) and ": end" in n_]
        if fork is None or not ends:
            continue
        links = int(ends[0][1].rsplit("link ", 1)[1].rstrip(")")) + 1
        out[ph] = (max(v for v, _ in ends) - fork, links)
    return out


def main():
    import numpy as np
    every = "--marks-every-link" in sys.argv
    names, ahead, lone, event_us, host_us = measure(every)
    out = []
    w = out.append
    w("markers between the PRODUCT kernels of the captured forward+backward (M4B4, 48 channels, 16 x 3 x 48 x 48); %d markers%s"
      % (len(names), ", one after every chain link" if every else ""))
    w("environment: " + " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("LARVA_")))
    w("HIP event pair around 20 back-to-back replays (median of 3): %.1f us per forward+backward (markers included); the host "
      "spends %.0f us in one graph launch" % (event_us, host_us))
    w("%10s %10s  %s" % ("queued", "lone", "marker (us after the graph's first marker; queued = last of 10 back-to-back replays, lone = one replay after a sync)"))
    order = np.argsort(ahead)
    for i in order:
        w("%10.1f %10.1f  %s" % (ahead[i] - ahead.min(), lone[i] - lone.min(), names[i]))
    w("")
    for col, arr in (("queued", ahead), ("lone", lone)):
        d = {n_: v for n_, v in zip(names, arr)}
        for ph in ("fwd", "bwd"):
            try:
                s0, s1 = d["%s chain 0: start (before link 0)" % ph], d["%s chain 1: start (before link 0)" % ph]
                ends = [v for n_, v in d.items() if n_.startswith("%s chain" % ph) and ": end" in n_]
                fork = d["%s: main stream at the fork" % ph]
                w("%s, %s: fork -> chain 0 starts %.1f us, chain 1 starts %.1f us (%.1f us after chain 0); chains end %.1f / %.1f us "
                  "after the fork" % (col, ph, s0 - fork, s1 - fork, s1 - s0, ends[0] - fork, ends[1] - fork))
            except KeyError:
                pass
    text = "\n".join(out)
    print(text)
    outs = [a_ for a_ in sys.argv[1:] if not a_.startswith("--")]
    if outs:
        with open(outs[0], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
