#!/usr/bin/env python3
"""Do the two half-batch chains really overlap on the chip?  rocprofv3 cannot show it (its per-dispatch overhead
makes the multi-stream graph launch host-bound and the chains serialise), so the kernels stamp themselves: a
-DLARVA_DIAG=544 build gives every launch of the captured graph its own stamp area, wave 0 of every workgroup
writes the 100 MHz wall clock at kernel entry / first chunk landed / K loop done / stores drained plus its HW_ID and
XCC_ID.  The graph is bench.py's `roofline` graph (two chains of 40 strip-tile conv+ReLU launches, N(0,1)*20
activations) and is replayed UN-PROFILED, timed by the same HIP event pair as bench.py.

  python tools/diag_overlap.py --build          (build container)
  python tools/diag_overlap.py [out.txt]        (GPU box)
"""
import ctypes as ct
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = "--prologue-steps" in sys.argv     # the heavier build that also stamps the prologue's steps
LIB = os.path.join(ROOT, "tools", "_diag", "liblarva_overlap%s.so" % ("_steps" if STEPS else ""))
if not STEPS and not os.path.exists(LIB) and os.path.exists(os.path.join(ROOT, "tools", "_diag", "liblarva_step.so")):
    LIB = os.path.join(ROOT, "tools", "_diag", "liblarva_step.so")   # tools/diag_step.py's build: the same -DLARVA_DIAG=544
WG, SLOT_WORDS = 256, 16


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    csrc = os.path.join(ROOT, "larvanet_amd", "csrc")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLARVA_DIAG=%d" % (544 + (2048 if STEPS else 0)), "-I" + csrc,
           os.path.join(csrc, "conv3x3_mfma.hip"), os.path.join(csrc, "wgrad3x3_mfma.hip"),
           os.path.join(csrc, "larva_pointwise.hip"), "-o", LIB]
    subprocess.check_call(cmd)
    print(LIB)


def union_len(iv):
    iv = sorted(iv)
    total, cur_s, cur_e = 0.0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        total += cur_e - cur_s
    return total


def main():
    os.environ["LARVA_HIP_LIB"] = LIB
    import numpy as np
    import torch
    import bench
    from larvanet_amd import hip_lib, kernels as K
    lib = hip_lib.load()
    raw = ct.CDLL(LIB)
    raw.larva_diag_set_stamps.argtypes = [ct.c_void_p]
    raw.larva_diag_arm_slots.argtypes = [ct.c_int, ct.c_int]
    dev = torch.device("cuda", 0)
    chain, c, nb = 40, bench.CH, bench.BATCH
    nslots = 2 * chain
    stamps = torch.zeros(nslots * WG * SLOT_WORDS, device=dev, dtype=torch.int64)
    assert raw.larva_diag_set_stamps(stamps.data_ptr()) == 0
    x0, wpk, b, bufs, rms = bench.chain_operands(dev, c, chain)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    parts = ((0, nb // 2), (nb // 2, nb))

    def body():
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        src = x0
        for i in range(chain):
            for k, st in enumerate(streams):
                with torch.cuda.stream(st):
                    K.conv3x3(src, wpk, c, bias=b, relu=True, out=bufs[i & 1], images=parts[k], strips=2 if k else True,
                              plain_stores=True)
            src = bufs[i & 1]
        for st in streams:
            cur.wait_stream(st)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    raw.larva_diag_arm_slots(0, nslots)       # launch (layer i, chain k) of the capture takes slot 2 i + k
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        body()
    raw.larva_diag_arm_slots(-1, 0)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        runs.append(s.elapsed_time(e) / (10 * chain) * 1e3)
    event_us = sorted(runs)[1]
    # the same build with the stamps switched off at run time (no slot armed): bench.py's figures in this process
    slope_ms, _, ms40 = bench.dual_chain_time_ms(dev, c)
    t = stamps.cpu().numpy().reshape(nslots, WG, SLOT_WORDS)
    hw = t[:, :, 4].astype(np.uint64)
    tt = t[:, :, :4].astype(np.float64) * 0.01      # us
    t0 = tt[:, :, 0].min()
    tt -= t0
    # CU identity: XCC_ID[3:0] (bits 32..35), SE_ID (HW_ID[15:13]), SH_ID (HW_ID[12]), CU_ID (HW_ID[11:8])
    cu_key = ((hw >> np.uint64(32)) & np.uint64(0xF)) * np.uint64(256) + ((hw >> np.uint64(8)) & np.uint64(0xFF))
    out = []
    w = out.append
    w("two half-batch chains of %d conv3x3+ReLU strip launches (8x48x48x48 each, 256 workgroups), one captured graph, "
      "un-profiled replay; in-kernel 100 MHz stamps of the LAST of 33 replays" % chain)
    w("activations N(0,1)*20 kept at that scale (RMS after the last layer %.1f); build -DLARVA_DIAG=%d" % (rms, 544 + (2048 if STEPS else 0)))
    w("HIP event pair around 10 replays (median of 3): %.2f us per full-batch layer = replay / 40 with the stamps ARMED  <- the "
      "definition of bench.py's roofline.avg_ms / roofline.frac since round 4 (replay of a captured 40-link chain / 40)" % event_us)
    w("same process, stamps not armed, bench.py's own functions: roofline.avg_ms = t40 / 40 = %.2f us per full-batch layer (%.3f of "
      "the fp32 matrix peak); roofline.avg_ms_steady_state = (t160 - t40) / 120 = %.2f us (%.3f)"
      % (ms40 * 1e3, bench.conv_flop(c) / (ms40 * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS,
         slope_ms * 1e3, bench.conv_flop(c) / (slope_ms * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS))
    span = tt[:, :, 3].max() - tt[:, :, 0].min()
    w("stamps: first kernel entry -> last store drained %.1f us = %.2f us per full-batch layer (a replay's kernels only: the "
      "event figure also holds the graph launch gap between two replays, %.1f us)" % (span, span / chain, event_us * chain - span))
    starts = [min(tt[2 * i, :, 0].min(), tt[2 * i + 1, :, 0].min()) for i in range(chain)]
    steady = (starts[chain - 1] - starts[8]) / (chain - 1 - 8)
    w("stamps, steady state: layer 8's first entry -> layer %d's first entry = %.2f us per full-batch layer  <- what bench.py's "
      "roofline.avg_ms_steady_state (slope between a 160- and a 40-layer chain) measures; the replay's fixed cost (graph launch gap, the second "
      "chain starting %.1f us after the first, the first layers' cold operands) is %.1f us per replay by the event pair"
      % (chain - 1, steady, tt[1, :, 0].min() - tt[0, :, 0].min(), event_us * chain - steady * chain))
    w("")
    lay = [(tt[s, :, 0].min(), tt[s, :, 3].max()) for s in range(nslots)]   # launch = [first WG entry, last WG drained]
    w("per layer: launch interval of chain 0 / chain 1 (us after the first entry), time BOTH chains have a launch resident, "
      "workgroup lifetime, K loop (medians over the 256 workgroups)")
    w("%5s %19s %19s %9s %9s %9s" % ("layer", "chain 0", "chain 1", "both us", "life us", "K us"))
    both_total = 0.0
    for i in range(chain):
        a0, a1 = lay[2 * i], lay[2 * i + 1]
        # overlap of chain 0's layer-i launch with ANY launch of chain 1
        ov = sum(max(0.0, min(a0[1], lay[2 * j + 1][1]) - max(a0[0], lay[2 * j + 1][0])) for j in range(chain))
        both_total += ov
        life = np.median(np.concatenate([tt[2 * i + k, :, 3] - tt[2 * i + k, :, 0] for k in (0, 1)]))
        kl = np.median(np.concatenate([tt[2 * i + k, :, 2] - tt[2 * i + k, :, 1] for k in (0, 1)]))
        if i < 6 or i >= chain - 3 or i % 8 == 0:
            w("%5d %8.1f -%8.1f  %8.1f -%8.1f  %9.2f %9.2f %9.2f" % (i, a0[0], a0[1], a1[0], a1[1], ov, life, kl))
    steady = slice(16, nslots)
    pro = np.median(tt[steady, :, 1] - tt[steady, :, 0])
    kl = np.median(tt[steady, :, 2] - tt[steady, :, 1])
    epi = np.median(tt[steady, :, 3] - tt[steady, :, 2])
    skew_in = np.median([tt[s_, :, 0].max() - tt[s_, :, 0].min() for s_ in range(16, nslots)])
    skew_out = np.median([tt[s_, :, 3].max() - tt[s_, :, 3].min() for s_ in range(16, nslots)])
    gap = np.median([tt[s_ + 2, :, 0].min() - tt[s_, :, 3].max() for s_ in range(16, nslots - 2)])
    w("a workgroup's life (medians, layers 8+): entry -> first chunk landed %.2f us, K loop %.2f us, K loop done -> stores drained "
      "%.2f us; per launch: first to last workgroup entry %.2f us, first to last drain %.2f us; last drain of a launch -> first "
      "entry of the chain's next launch %.2f us" % (pro, kl, epi, skew_in, skew_out, gap))
    if STEPS:
        raw = t[16:, :, :].astype(np.float64) * 0.01
        ph = [np.median(raw[:, :, b_] - raw[:, :, a_]) for a_, b_ in ((0, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 1))]
        w("the prologue, step by step (medians; this build's five extra stamps per workgroup cost ~1 us of its life): arguments + "
          "tile decode %.2f us, chunk 0's weight pieces issued +%.2f, bias requested +%.2f, input plan + chunk 0's input pieces "
          "+%.2f, chunk 1's pieces +%.2f, first chunk landed (wait + barrier) +%.2f" % tuple(ph))
    busy0 = union_len([lay[2 * i] for i in range(chain)])
    busy1 = union_len([lay[2 * i + 1] for i in range(chain)])
    w("chain 0 has a launch resident %.1f us, chain 1 %.1f us, both at once %.1f us of the %.1f us span (%.0f %%)"
      % (busy0, busy1, both_total, span, 100 * both_total / span))
    w("")
    # per-CU co-residency: how long does a CU hold 0 / 1 / 2 workgroups?
    keys = np.unique(cu_key)
    hist = np.zeros(4)
    pair_kinds = {"chain0+chain1": 0.0, "same chain": 0.0}
    for key in keys:
        sel = np.argwhere(cu_key == key)
        ev = []
        for s_, g_ in sel:
            ev.append((tt[s_, g_, 0], 1, s_ & 1))
            ev.append((tt[s_, g_, 3], -1, s_ & 1))
        ev.sort()
        n, last, per_chain = 0, ev[0][0], [0, 0]
        for when, d, ch in ev:
            hist[min(n, 3)] += when - last
            if n == 2:
                pair_kinds["chain0+chain1" if per_chain[0] == 1 else "same chain"] += when - last
            last = when
            n += d
            per_chain[ch] += d
    tot = hist.sum()
    w("%d distinct CUs seen (XCC_ID, SE/SH/CU of HW_ID).  Share of a CU's time between its first entry and its last drain with"
      % len(keys))
    w("  0 workgroups %.1f %%   1 workgroup %.1f %%   2 workgroups %.1f %%   3+ %.1f %%" % tuple(100 * hist / tot))
    two = pair_kinds["chain0+chain1"] + pair_kinds["same chain"]
    if two > 0:
        w("  of the 2-workgroup time: one of each chain %.1f %%, two of the same chain %.1f %%"
          % (100 * pair_kinds["chain0+chain1"] / two, 100 * pair_kinds["same chain"] / two))
    wg_per_cu = np.array([(cu_key == k_).sum() for k_ in keys]) / float(nslots)
    w("  workgroups per CU per launch: min %.2f  median %.2f  max %.2f" % (wg_per_cu.min(), np.median(wg_per_cu), wg_per_cu.max()))
    # do workgroup i of both chains' launches land on the same CU?  and which tile heights meet on a CU?
    same = np.mean([np.mean(cu_key[2 * i] == cu_key[2 * i + 1]) for i in range(chain)])
    w("  workgroup index i of chain 0's and chain 1's launch of a layer sit on the same CU in %.0f %% of the cases" % (100 * same))
    tabs = [K.strip_tile_table(48, 48, dev, ph)[0].cpu().numpy().astype(np.int64) for ph in (0, 1)]
    per_img = len(tabs[0])

    def rows_of(block, ph):   # tile height of workgroup `block` of a launch with table phase ph (xcd_remap as in the kernel)
        nwg = WG
        q, r, xcd = nwg >> 3, nwg & 7, block & 7
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        tile = base + (block >> 3)
        return 5 if (tabs[ph][tile % per_img] >> 31) & 1 else 4

    heights = np.array([[rows_of(g_, s_ & 1) for g_ in range(WG)] for s_ in range(nslots)])
    # a workgroup's phases by tile height (layers 8+): does the 4-row tile, with 3 / 4 of the MFMAs, finish earlier?
    for hgt in (5, 4):
        sel_h = heights[16:] == hgt
        seg = tt[16:]
        w("  %d-row tiles (layers 8+, medians): entry -> first chunk landed %.2f us, K loop %.2f us (p10 %.2f, p90 %.2f), K loop done -> "
          "drained %.2f us, lifetime %.2f us; enter %.2f us and drain %.2f us after their launch's first workgroup"
          % (hgt, np.median((seg[:, :, 1] - seg[:, :, 0])[sel_h]), np.median((seg[:, :, 2] - seg[:, :, 1])[sel_h]),
             np.percentile((seg[:, :, 2] - seg[:, :, 1])[sel_h], 10), np.percentile((seg[:, :, 2] - seg[:, :, 1])[sel_h], 90),
             np.median((seg[:, :, 3] - seg[:, :, 2])[sel_h]), np.median((seg[:, :, 3] - seg[:, :, 0])[sel_h]),
             np.median((seg[:, :, 0] - seg[:, :, 0].min(axis=1, keepdims=True))[sel_h]),
             np.median((seg[:, :, 3] - seg[:, :, 0].min(axis=1, keepdims=True))[sel_h])))
    pairs = {"5+4": 0.0, "5+5": 0.0, "4+4": 0.0}
    for key in keys:
        sel = [(s_, g_) for s_, g_ in np.argwhere(cu_key == key)]
        ev = sorted([(tt[s_, g_, 0], 1, heights[s_, g_]) for s_, g_ in sel] + [(tt[s_, g_, 3], -1, heights[s_, g_]) for s_, g_ in sel])
        live, last = [], ev[0][0]
        for when, d, h in ev:
            if len(live) == 2:
                pairs["%d+%d" % (max(live), min(live))] += when - last
            last = when
            if d > 0:
                live.append(h)
            else:
                live.remove(h)
    tp = sum(pairs.values())
    if tp > 0:
        w("  tile heights of the two workgroups that share a CU (share of the 2-workgroup time): 5 + 4 rows %.0f %%, 5 + 5 %.0f %%, "
          "4 + 4 %.0f %%" % (100 * pairs["5+4"] / tp, 100 * pairs["5+5"] / tp, 100 * pairs["4+4"] / tp))
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
