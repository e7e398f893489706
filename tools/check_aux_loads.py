#!/usr/bin/env python3
"""The conv kernel's loader-wave path issues its epilogue operands (mask / residuals / base / truth / sign bits) as the
YOUNGEST vector loads of an MFMA wave's prologue and every counted wait of the LDS-DMA ring leaves `kAuxLoads` of them
in flight (run_role, kAuxLate).  That count is a compile-time product; the loads are the compiler's.  If it
emitted FEWER vector loads than counted, `s_waitcnt vmcnt(NPW + kAuxLoads)` could pass with a piece of chunk 0 still
in flight.  This script reads the device assembly (hipcc -S) and, for every ring wait of every kernel, counts the
global loads between the last LDS-DMA piece in front of it and the wait:

  first wait of a role   vmcnt(N):  N - (global loads since the last piece) must equal the role's pieces per wave (NPW)
  later waits            vmcnt(N):  N must equal the first wait's load count

Since round 6 every ring wait carries its constants into the assembly as a comment in front of it
("; LARVA_RING pieces=P aux=A", wait_and_barrier; "; LARVA_RING loader chunk_pieces=P ahead=K", run_loader): where the tag
is present the check is EXACT -- N == P + A, and for a role's first wait the loads emitted since the last piece == A,
for the loader N == (K - 1) * P -- and the older range heuristics (3..9 pieces per wave, >= 12 per chunk) only serve
assembly without tags.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/conv.s larvanet_amd/csrc/conv3x3_mfma.hip
  python tools/check_aux_loads.py /tmp/conv.s
larvanet_amd/build.py runs it on the device assembly of every build of conv3x3_mfma.hip (-save-temps) and fails the
build on a mismatch (exit status 1).
"""
import re
import sys

path = sys.argv[1]
kernel = None
bad = 0
rows = {}
state = None
with open(path) as f:
    lines = f.read().split("\n")
i = 0
while i < len(lines):
    ln = lines[i].strip()
    m = re.match(r"^(_ZN5larva\w+):", lines[i])
    if m:
        kernel = m.group(1)
        state = {"loads_since_dma": 0, "dma_run": 0, "first": None, "pieces_last_run": 0}
        rows[kernel] = []
    elif kernel and state is not None:
        if ln.startswith("buffer_load_dwordx4") and " lds" in ln:
            if state["loads_since_dma"] or state["dma_run"] == 0:
                state["pieces_last_run"] = 0
            state["dma_run"] += 1
            state["pieces_last_run"] += 1
            state["loads_since_dma"] = 0
        elif re.match(r"^(global_load_|buffer_load_(?!dwordx4.* lds))", ln):
            state["loads_since_dma"] += 1
        elif ln.startswith("s_waitcnt vmcnt(") and i + 1 < len(lines) and lines[i + 1].strip() == "s_barrier":
            n = int(re.search(r"vmcnt\((\d+)\)", ln).group(1))
            tag = None
            prev = lines[i - 1].strip() if i > 0 else ""
            mt = re.match(r"^; LARVA_RING pieces=(\d+) aux=(\d+)", prev)
            ml = re.match(r"^; LARVA_RING loader chunk_pieces=(\d+) ahead=(\d+)", prev)
            if mt:
                tag = ("role", int(mt.group(1)), int(mt.group(2)))
            elif ml:
                tag = ("loader", int(ml.group(1)), int(ml.group(2)))
            if n > 0 or tag:   # (an untagged vmcnt(0) waits for everything: always safe -- the persistent loader's last hand-overs)
                rows[kernel].append((n, state["loads_since_dma"], state["dma_run"], tag))
            state["dma_run"] = 0
        elif ln.startswith(".Lfunc_end"):
            kernel, state = None, None
    i += 1

import subprocess
def demangle(n):
    try:
        return subprocess.check_output(["c++filt", n]).decode().strip().replace("larva::", "").replace("(larva::ConvArgs)", "").replace("(larva::ConvBatch)", "")
    except Exception:
        return n

tagged = 0
for k, r in rows.items():
    if not r:
        continue
    # a role's first wait follows a run of DMA pieces (dma_run > 0); later waits of the same role have dma_run == 0
    out, first_loads = [], None
    j = 0
    while j < len(r):
        n, loads, run, tag = r[j]
        if tag is not None:
            tagged += 1
            if tag[0] == "loader":
                ok = n == (tag[2] - 1) * tag[1]
                out.append("loader wave: vmcnt(%d) = (%d - 1) x %d pieces%s" % (n, tag[2], tag[1], "" if ok else "  <-- MISMATCH"))
            elif tag[1] > 0 or run > 0:      # a role's first wait: the next chunk's pieces + the operand loads behind them
                ok = n == tag[1] + tag[2] and (run == 0 or loads == tag[2])
                out.append("first vmcnt(%d) = %d pieces + %d operand loads (emitted since the last piece: %d)%s"
                           % (n, tag[1], tag[2], loads, "" if ok else "  <-- MISMATCH"))
            else:
                ok = n == tag[2]
                out.append("later vmcnt(%d) = %d operand loads%s" % (n, tag[2], "" if ok else "  <-- MISMATCH"))
            bad += 0 if ok else 1
            j += 1
            continue
        if run > 0 and loads == 0 and n >= 12:
            out.append("loader wave: vmcnt(%d) = one chunk's pieces" % n)     # (run_loader: a whole chunk in flight)
            first_loads = None
        elif run > 0:
            npw = n - loads
            if loads <= n and 3 <= npw <= 9:
                first_loads = loads
                out.append("first vmcnt(%d) = %d pieces + %d operand loads" % (n, npw, loads))
            elif n == loads and j + 1 < len(r) and r[j + 1][2] == 0 and r[j + 1][3] is None and 3 <= r[j + 1][0] - loads <= 9:
                # the loop was laid out in front of its entry: this is the LATER wait, the next one the first
                first_loads = loads
                out.append("first vmcnt(%d) = %d pieces + %d operand loads; later vmcnt(%d)   (loop rotated in the listing)"
                           % (r[j + 1][0], r[j + 1][0] - loads, loads, n))
                j += 1
            else:
                out.append("first vmcnt(%d) with %d operand loads in front of it  <-- CHECK" % (n, loads))
                bad += 1
        elif first_loads is None:
            out.append("loader wave: vmcnt(%d)" % n)
        else:
            ok = n == first_loads
            out.append("later vmcnt(%d)%s" % (n, "" if ok else "  <-- expected %s" % first_loads))
            bad += 0 if ok else 1
        j += 1
    uniq = []
    for o in out:
        if not uniq or uniq[-1][0] != o:
            uniq.append([o, 1])
        else:
            uniq[-1][1] += 1
    print(demangle(k))
    for o, c in uniq:
        print("    %s%s" % (o, " (x%d)" % c if c > 1 else ""))
print("%d ring waits checked against their LARVA_RING constants; %d suspicious waits" % (tagged, bad))
sys.exit(1 if bad else 0)
