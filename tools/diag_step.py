#!/usr/bin/env python3
"""Un-profiled timeline of the REAL captured training step: where do its ~1.66 ms go, what do the
forks / joins of the two half-batch chains cost, do the weight gradients wait for anything?

rocprofv3 cannot answer this (its per-dispatch overhead serialises the two chains), so the kernels stamp themselves: a
-DLARVA_DIAG=544 build gives every conv launch and every flat weight-gradient launch of the captured forward+backward
its own stamp area; wave 0 of every workgroup writes the 100 MHz wall clock at entry / first K chunk landed / K loop
done / stores drained (conv) or entry / exit (wgrad).  The graph is the plugin's own (`_scope()` + `_exit_losses` +
`backward`, M4B4, 16 x 3 x 48 x 48), replayed un-profiled; the stamps of the LAST of six back-to-back replays are read.

  python tools/diag_step.py --build            (build container)
  python tools/diag_step.py [out.txt]          (GPU box; environment switches of the plugin apply, e.g. LARVA_WGRAD_EARLY_WG)
"""
import ctypes as ct
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "_diag", "liblarva_step.so")
WG, WORDS, WWORDS = 256, 16, 4


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    csrc = os.path.join(ROOT, "larvanet_amd", "csrc")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLARVA_DIAG=544",
                           "-I" + csrc, os.path.join(csrc, "conv3x3_mfma.hip"), os.path.join(csrc, "wgrad3x3_mfma.hip"),
                           os.path.join(csrc, "larva_pointwise.hip"), "-o", LIB])
    print(LIB)


def main():
    os.environ["LARVA_HIP_LIB"] = LIB
    import importlib
    import numpy as np
    import torch
    from larvanet_amd import autograd as A, hip_lib, kernels as K
    hip_lib.load()
    raw = ct.CDLL(LIB)
    raw.larva_diag_set_stamps.argtypes = [ct.c_void_p]
    raw.larva_diag_set_wgrad_stamps.argtypes = [ct.c_void_p]
    dev = torch.device("cuda", 0)
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

    CAP, WCAP = 256, 8
    stamps = torch.zeros(CAP * WG * WORDS, device=dev, dtype=torch.int64)
    wstamps = torch.zeros(WCAP * WG * WWORDS, device=dev, dtype=torch.int64)
    assert raw.larva_diag_set_stamps(stamps.data_ptr()) == 0 and raw.larva_diag_set_wgrad_stamps(wstamps.data_ptr()) == 0

    log = []     # (kind, description, first slot, slots, stream handle)

    def wrap(name, describe, wgrad=False):
        real = getattr(K, name)

        def fn(*a, **kw):
            nxt = raw.larva_diag_arm_wgrad_slots if wgrad else None
            s0 = raw.larva_diag_next_slot() if not wgrad else wslot[0]
            out = real(*a, **kw)
            if wgrad:
                if out is not None:
                    log.append(("wgrad", describe(*a, **kw), wslot[0], 1, torch.cuda.current_stream().cuda_stream))
                    wslot[0] += 1
            else:
                s1 = raw.larva_diag_next_slot()
                if s1 > s0:
                    log.append(("conv", describe(*a, **kw), s0, s1 - s0, torch.cuda.current_stream().cuda_stream))
            return out
        setattr(K, name, fn)
        return real

    wslot = [0]

    def d_conv(srcs, wpk, cout, **kw):
        epi = "+".join(k for k in ("relu", "mask", "res0", "res1", "shuffle", "base") if kw.get(k) is not None and kw.get(k) is not False) or "plain"
        first = srcs if isinstance(srcs, torch.Tensor) else srcs[0]
        nsrc = 1 if isinstance(srcs, torch.Tensor) else len(srcs)
        return "%s%s images=%s %s" % ("strips " if kw.get("strips") else "wide ", epi, kw.get("images"), "K=%d" % (nsrc * int(first.shape[1])))

    saved = [("conv3x3", wrap("conv3x3", d_conv)),
             ("conv3x3_batch", wrap("conv3x3_batch", lambda jobs, cout, **kw: "batch x%d %s" % (len(jobs), "+".join(
                 k for k in ("relu", "shuffle") if kw.get(k)) + ("+mask" if jobs[0].get("mask") is not None else "")))),
             ("conv3x3_exit_l1_batch", wrap("conv3x3_exit_l1_batch", lambda jobs, *a, **kw: "exits x%d shuffle+base+L1" % len(jobs))),
             ("conv3x3_wgrad_partial_flat", wrap("conv3x3_wgrad_partial_flat",
                                                 lambda jobs, cout, cin, nwg, head=None: "flat wgrad %d layers%s on %d workgroups" % (len(jobs), " + head" if head is not None else "", nwg), wgrad=True))]
    raw.larva_diag_arm_slots(0, CAP)
    raw.larva_diag_arm_wgrad_slots(0, WCAP)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        body()
    raw.larva_diag_arm_slots(-1, 0)
    raw.larva_diag_arm_wgrad_slots(-1, 0)
    for name, real in saved:
        setattr(K, name, real)
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
    # the stamps read are those of the LAST of six back-to-back replays: the host is then far ahead of the GPU, as it is
    # in a training loop (a lone replay after a synchronisation starts its second chain ~100 us late: the host is still
    # enqueueing the graph's ~140 nodes, 460 us of host time per launch, while the first ones already run)
    for _ in range(6):
        graph.replay()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(CAP, WG, WORDS).astype(np.float64) * 0.01
    ws = wstamps.cpu().numpy().reshape(WCAP, WG, WWORDS).astype(np.float64) * 0.01

    wg_notes = []
    rows = []    # (first entry, last entry, first drain, last drain, median life, median K, kind, description, stream)
    streams = {}
    for kind, desc, s0, n, stream in log:
        sid = streams.setdefault(stream, len(streams))
        if kind == "conv":
            sel = st[s0:s0 + n].reshape(-1, WORDS)
            sel = sel[sel[:, 0] > 0]
            if not len(sel):
                continue
            rows.append((sel[:, 0].min(), sel[:, 0].max(), sel[:, 3].min(), sel[:, 3].max(), np.median(sel[:, 3] - sel[:, 0]),
                         np.median(sel[:, 2] - sel[:, 1]), len(sel), desc, sid))
            if n > 1:   # a batched launch: its jobs enter in rounds -- do later rounds (warm instruction / scalar caches) start faster?
                per_job = []
                for jslot in range(s0, s0 + n):
                    sj = st[jslot]
                    sj = sj[sj[:, 0] > 0]
                    per_job.append("job %d: enters %.1f..%.1f us, entry -> first chunk landed %.2f, K loop %.2f" % (
                        jslot - s0, sj[:, 0].min() - st[s0:s0 + n][st[s0:s0 + n][:, :, 0] > 0][:, 0].min(), sj[:, 0].max() - st[s0:s0 + n][st[s0:s0 + n][:, :, 0] > 0][:, 0].min(),
                        np.median(sj[:, 1] - sj[:, 0]), np.median(sj[:, 2] - sj[:, 1])))
                wg_notes.append("%s: %s" % (desc, "; ".join(per_job)))
        else:
            sel = ws[s0]
            sel = sel[sel[:, 0] > 0]
            if not len(sel):
                continue
            rows.append((sel[:, 0].min(), sel[:, 0].max(), sel[:, 1].min(), sel[:, 1].max(), np.median(sel[:, 1] - sel[:, 0]),
                         float("nan"), len(sel), desc, sid))
            # how evenly do the workgroups of the flat grid finish?  (a workgroup = a CU for the whole launch)
            full = ws[s0]
            idx = np.nonzero(full[:, 0] > 0)[0]
            life = full[idx, 1] - full[idx, 0]
            hwid = wstamps.cpu().numpy().reshape(WCAP, WG, WWORDS)[s0, idx, 2].astype(np.uint64)
            xcc = ((hwid >> np.uint64(32)) & np.uint64(0xF)).astype(int)
            wg_notes.append("%s: workgroup lifetimes min %.1f / p10 %.1f / median %.1f / p90 %.1f / max %.1f us; mean %.1f = %.1f %% of "
                            "the longest (the CUs idle for the rest)" % (desc, life.min(), np.percentile(life, 10), np.median(life),
                                                                        np.percentile(life, 90), life.max(), life.mean(), 100 * life.mean() / life.max()))
            wg_notes.append("   by XCD (XCC_ID: mean lifetime): " + "  ".join("%d: %.1f" % (k, life[xcc == k].mean()) for k in sorted(set(xcc))))
            order_ = np.argsort(idx)
            nb = 8
            wg_notes.append("   by workgroup index (mean lifetime of each eighth of the grid): " + "  ".join(
                "%.1f" % life[order_][i * len(idx) // nb:(i + 1) * len(idx) // nb].mean() for i in range(nb)))
            wg_notes.append("   the last four workgroups (the head's tiles ride there): " + "  ".join("%.1f" % v for v in life[order_][-4:]))
    rows.sort()
    t0 = rows[0][0]
    out = []
    w = out.append
    w("captured forward+backward of the plugin's training step (M4B4, 48 channels, 16 x 3 x 48 x 48), one graph, un-profiled "
      "replay; in-kernel 100 MHz stamps of the last of six back-to-back replays; build -DLARVA_DIAG=544 (the launch carries its stamp "
      "area's address: a stamp is one s_memrealtime + one store)")
    w("environment: " + " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("LARVA_") and k != "LARVA_HIP_LIB"))
    w("HIP event pair around 20 replays (median of 3): %.1f us per forward+backward" % event_us)
    w("stamps: first kernel entry -> last stamped exit %.1f us (the reduction / loss launches behind it carry no stamps)"
      % (max(r[3] for r in rows) - t0))
    w("")
    w("%8s %8s %8s %8s %7s %6s %4s %3s  %s" % ("entry0", "entryN", "drain0", "drainN", "life", "K", "wgs", "st", "launch"))
    for r in rows:
        w("%8.1f %8.1f %8.1f %8.1f %7.2f %6.2f %4d %3d  %s" % (r[0] - t0, r[1] - t0, r[2] - t0, r[3] - t0, r[4], r[5], r[6], r[8], r[7]))
    w("")
    # phases: the chain links are the strip launches; forward ones carry relu / res0, backward ones mask / res
    strips = [r for r in rows if r[7].startswith("strips")]
    per_stream = {}
    for r in strips:
        per_stream.setdefault(r[8], []).append(r)
    exits_f = [r for r in rows if r[7].startswith("batch") and "mask" not in r[7] or r[7].startswith("exits")]
    exits_b = [r for r in rows if r[7].startswith("batch") and "mask" in r[7]]
    wg = [r for r in rows if "wgrad" in r[7]]
    if exits_f and exits_b and strips:
        f_end = min(r[0] for r in exits_f)
        b_start = max(r[3] for r in exits_b)
        fwd = [r for r in strips if r[3] <= f_end + 1e-6]
        bwd = [r for r in strips if r[0] >= b_start - 1e-6]
        for name, part in (("forward chain", fwd), ("backward chain", bwd)):
            if not part:
                continue
            a, b = min(r[0] for r in part), max(r[3] for r in part)
            links = {}
            for r in part:
                links.setdefault(r[8], []).append(r)
            n_links = max(len(v) for v in links.values())
            w("%s: %.1f -> %.1f us = %.1f us for %d links per chain = %.2f us per full-batch layer" % (name, a - t0, b - t0, b - a, n_links, (b - a) / n_links))
            for sid, v in sorted(links.items()):
                v.sort()
                per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
                gaps = [v[i + 1][0] - v[i][3] for i in range(len(v) - 1)]
                w("   stream %d: first entry %.1f, last drain %.1f; link period first 4: %s  median of the rest %.2f; last drain -> next entry "
                  "median %.2f" % (sid, v[0][0] - t0, v[-1][3] - t0, " ".join("%.1f" % p for p in per[:4]), np.median(per[4:]) if len(per) > 4 else float("nan"), np.median(gaps)))
        w("forward chain's last drain -> exits' first entry %.2f us; exits (forward) %.1f us; exits' last drain -> backward exits' "
          "first entry %.2f us" % (min(r[0] for r in exits_f) - max(r[3] for r in fwd), max(r[3] for r in exits_f) - min(r[0] for r in exits_f),
                                  min(r[0] for r in exits_b) - max(r[3] for r in exits_f)))
        w("backward exits %.1f us; their last drain -> backward chain's first entry %.2f us" % (
            max(r[3] for r in exits_b) - min(r[0] for r in exits_b), min(r[0] for r in bwd) - max(r[3] for r in exits_b)))
        for r in wg:
            w("%s: entry %.1f -> exit %.1f (%.1f us; workgroups enter within %.1f us, leave within %.1f us); backward chain's last drain -> its "
              "first entry %.2f us" % (r[7], r[0] - t0, r[3] - t0, r[3] - r[0], r[1] - r[0], r[3] - r[2], r[0] - max(x_[3] for x_ in bwd)))
        for note in wg_notes:
            w(note)
        last = max(r[3] for r in rows)
        w("un-stamped remainder (prologue launch in front, reduction + loss behind, graph launch): %.1f us of the %.1f us replay"
          % (event_us - (last - t0), event_us))
    text = "\n".join(out)
    print(text)
    outs = [a_ for a_ in sys.argv[1:] if not a_.startswith("--")]
    if outs:
        with open(outs[0], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
