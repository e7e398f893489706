"""The C-ABI library loads on a machine without a GPU and exports exactly what include/larva_hip.h
declares (no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "larva_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(larva_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("larva_conv3x3_fwd", "larva_conv3x3_wgrad", "larva_pack_weights", "larva_bicubic4_fwd",
                 "larva_l1_fwd", "larva_l1_bwd", "larva_pixel_unshuffle4", "larva_adamw_step"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from larvanet_amd import hip_lib
    if not os.path.exists(hip_lib.LIB_PATH):
        from larvanet_amd.build import build_extension
        build_extension(verbose=False)
    lib = hip_lib.load()
    declared = _declared()
    assert sorted(hip_lib.SIGNATURES) == declared  # binding table and header agree
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.larva_abi_version() == 5
    # product entry points only: the measurement entry points (tools/larva_diag.h) exist in tools/_diag/*.so builds alone
    for name in ("larva_stamp_clock", "larva_delay_ticks", "larva_conv3x3_fwd_timed", "larva_conv3x3_fwd_strips_timed",
                 "larva_conv3x3_pair_chain_probe", "larva_conv3x3_chain_probe", "larva_maskbits_bytes"):
        assert name not in declared and not hasattr(lib, name), name
    # pure host-side size helpers need no device
    def stride(c):   # cout_stride() of csrc/larva_common.h: c itself where c % 32 is 16 (or 0: swizzled rows), else c + 16
        return c if c % 32 in (0, 16) else c + 16

    for cout, cin in ((48, 48), (64, 64), (32, 32), (48, 16), (48, 192), (48, 8)):
        # [cin/8] chunks x 9 taps x 8 channels x stride(cout) -- the layout include/larva_hip.h documents
        assert lib.larva_packed_weight_floats(cout, cin) == (cin // 8) * 9 * 8 * stride(cout), (cout, cin)
    assert lib.larva_wgrad_partial_floats(48, 48, 2) == 2 * (27 * 3 * 256 + 48)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from larvanet_amd import hip_lib
    monkeypatch.setattr(hip_lib, "_lib", None)
    monkeypatch.setattr(hip_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        hip_lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "larvanet_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)


def test_flat_wgrad_partition_covers_every_tile_once_within_the_advertised_image_counts():
    """The flat weight-gradient grid (larva_conv3x3_wgrad_partial_flat[_head]): workgroup w owns
    [G w / nwg, G (w + 1) / nwg) of the sequence of njobs x tiles tiles followed by the head's pseudo-tiles.  Replayed
    here in Python for seeded (njobs, nwg, tiles): every tile of every layer and of the head falls into exactly one
    share, and the workgroups touching a layer / the head never exceed what the two host-side size helpers
    (pure host code) tell the caller to allocate."""
    import random
    from larvanet_amd import hip_lib
    lib = hip_lib.load()
    rnd = random.Random(7)
    cases = [(40, 256, 256), (8, 256, 256), (1, 3, 96), (5, 7, 6), (64, 1024, 30), (3, 64, 2)]
    cases += [(rnd.randint(1, 64), rnd.randint(1, 600), rnd.randint(1, 400)) for _ in range(60)]
    for njobs, nwg, tiles in cases:
        for with_head in (False, True):
            hu = (tiles * 6 + 9) // 10 if with_head else 0   # (LARVA_HEAD_COST10 = 6: a head tile priced at 0.6 of a 48 -> 48 tile)
            T = njobs * tiles
            G = T + hu
            n = min(nwg, G)
            owners = [[] for _ in range(njobs)]
            covered = [0] * njobs
            head_owner, head_cov = [], 0
            for w in range(n):
                g0, g1 = G * w // n, G * (w + 1) // n
                g, g_end = g0, min(g1, T)
                while g < g_end:
                    jb = g // tiles
                    seg_end = min(g_end, (jb + 1) * tiles)
                    owners[jb].append(w)
                    covered[jb] += seg_end - g
                    g = seg_end
                if hu and g1 > T:
                    p0, p1 = max(g0, T) - T, g1 - T
                    head_owner.append(w)
                    head_cov += p1 * tiles // hu - p0 * tiles // hu
            assert covered == [tiles] * njobs, (njobs, nwg, tiles, with_head)
            cap = lib.larva_wgrad_flat_max_splits(njobs, nwg, tiles)
            assert all(o == list(range(o[0], o[0] + len(o))) and len(o) <= cap for o in owners), (njobs, nwg, tiles)
            if with_head:
                assert head_cov == tiles and head_owner == list(range(head_owner[0], n)), (njobs, nwg, tiles)
                assert len(head_owner) == lib.larva_wgrad_flat_head_splits(njobs, nwg, tiles), (njobs, nwg, tiles)

