"""Data-parallel training step with the split backward: two ranks share the one GPU of the test
box (gloo carries the collectives, the kernels are the HIP ones), and the step whose all-reduce
overlaps the second half of the weight-gradient kernels must train exactly like the step that
reduces the whole bucket at the end -- eager launches and hipGraph replay alike."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Val:
    def get_num_images(self):
        return 1

    def get_image_pair(self, image_index, scale):
        rng = np.random.RandomState(3)
        return (rng.randint(0, 256, size=(3, 8, 12)).astype(np.float32),
                rng.randint(0, 256, size=(3, 32, 48)).astype(np.float32), "v")


def _worker(rank, world, port, q, backend="gloo"):
    # gloo: both ranks share GPU 0 (the one-GPU test box); nccl (= RCCL): one GPU per rank
    local = rank if backend == "nccl" else 0
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(local), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LARVA_DIST_BACKEND": backend})
    try:
        import torch.distributed as td
        from larvanet_amd import dist as ldist
        from larvanet_amd.autograd import DeferredWgrad
        from larvanet_amd.models import LarvaNet as L
        ldist.init_from_env(backend=backend)
        assert td.get_backend() == backend
        dev = torch.device("cuda", local)
        DeferredWgrad.jobs_per_launch = 4  # a small network still ends backward in several launches
        g = torch.Generator().manual_seed(50 + rank)  # every rank trains on its own patches
        x = (torch.rand(2, 3, 12, 16, generator=g) * 255).to(dev)
        t = (torch.rand(2, 3, 48, 64, generator=g) * 255).to(dev)
        args = types.SimpleNamespace(train_path="/tmp")
        out = {}
        for mode in ("whole", "overlap_eager", "overlap_graph"):
            m = L.create_model()
            m.parse_args(["--num_modules=2", "--num_blocks=2,1"])
            torch.manual_seed(7)
            m.prepare(is_training=True, scales=[4])
            # prepare() timed the bucket's isolated all-reduce and chose a schedule from it (LARVA_OVERLAP_ALLREDUCE=auto)
            assert m.dp_schedule["choice"] in ("split", "flat") and m.dp_schedule["allreduce_isolated_us"] > 0
            assert isinstance(m.overlap_allreduce, bool) and m.overlap_allreduce == (m.dp_schedule["choice"] == "split")
            assert m.overlap_allreduce == (m.dp_schedule["allreduce_isolated_us"] > m.DP_SPLIT_ABOVE_US)
            out.setdefault("choices", []).append(m.dp_schedule["choice"])
            m.overlap_allreduce = mode != "whole"
            m.use_hip_graph = mode == "overlap_graph"
            losses = [m.train_step_larva(args, _Val(), x, t) for _ in range(3)]
            out[mode] = {"losses": losses, "split_at": getattr(m, "_early_lo", None),
                         "graph": bool(m.use_hip_graph),
                         "sd": {k: v.cpu().numpy().copy() for k, v in m.model.state_dict().items()}}
        # round 6: rank 0 decodes the resident dataset, rank 1 receives it (gloo here: device tensors staged through the host)
        from larvanet_amd.dataloaders import device_patch_loader as D
        calls = []
        real = D.build_host_tables
        D.build_host_tables = lambda *a_, **k_: (calls.append(rank), real(*a_, **k_))[1]
        out_loader = _resident_loader_digest()
        D.build_host_tables = real
        out_loader["decoded_here"] = len(calls)
        torch.cuda.synchronize()
        q.put((rank, None, out, out_loader))
        td.destroy_process_group()
    except Exception as e:  # hand the failure to the parent instead of hanging its queue
        import traceback
        q.put((rank, "%s\n%s" % (e, traceback.format_exc()), None, None))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_overlapped_allreduce_trains_like_one_collective(hip_device, backend):
    """backend nccl = RCCL over xGMI, one GPU per rank: needs two devices (skipped on the one-GPU
    test box, where the gloo variant carries the same step with both ranks on GPU 0)."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has %d" % torch.cuda.device_count())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, err, _, _ in res:
        assert err is None, "rank %d: %s" % (rank, err)
    (_, _, a, la), (_, _, b, lb) = res
    # the resident dataset: decoded once (rank 0), identical tables on both ranks, different draws (seed + 1000 rank)
    assert (la.pop("decoded_here"), lb.pop("decoded_here")) == (1, 0)
    xa, xb = la.pop("x"), lb.pop("x")
    ya, yb = la.pop("y_sum"), lb.pop("y_sum")
    assert la == lb and not np.array_equal(xa, xb) and ya != yb
    assert a.pop("choices") == b.pop("choices")    # both ranks chose the same schedule every time (MAX over ranks)
    assert a["whole"]["split_at"] is None
    for mode in ("overlap_eager", "overlap_graph"):
        assert a[mode]["split_at"] and a[mode]["split_at"] > 0, "backward was not split: nothing overlapped"
    assert a["overlap_graph"]["graph"], "hipGraph capture of the split backward fell back to eager launches"
    for mode in a:
        # ranks hold identical weights after training on different patches (mean of the gradients) ...
        for k in a[mode]["sd"]:
            assert np.array_equal(a[mode]["sd"][k], b[mode]["sd"][k]), (mode, k)
        # ... and the overlapped schedule is the same arithmetic as the single collective
        for k in a[mode]["sd"]:
            assert np.array_equal(a[mode]["sd"][k], a["whole"]["sd"][k]), (mode, k)
    assert a["whole"]["losses"] != b["whole"]["losses"]  # the ranks did see different data


def _band_worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LARVA_DIST_BACKEND": "gloo"})
    try:
        import torch.distributed as td
        from larvanet_amd import validate
        torch.manual_seed(0)
        res = validate.main(_VALIDATE_ARGS + ["--band_gpus"])
        q.put((rank, None, res))
        td.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, "%s\n%s" % (e, traceback.format_exc()), None))


_VALIDATE_ARGS = ["--model=LarvaNet", "--dataloader=synthetic_loader", "--num_modules=2", "--num_blocks=1,1",
                  "--synthetic_images=2", "--synthetic_lr_size=33", "--synthetic_uint8"]


@pytest.mark.timeout(600)
def test_validate_with_one_row_band_per_rank_scores_like_one_gpu(hip_device):
    """validate.py --band_gpus on 2 ranks (each computes half of every image + halo) gives the PSNR
    of the plain single-process run exactly."""
    from larvanet_amd import validate
    torch.manual_seed(0)
    plain = validate.main(list(_VALIDATE_ARGS))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for rank, err, _ in res:
        assert err is None, "rank %d: %s" % (rank, err)
    banded = res[0][2]
    assert [r[:2] for r in banded[4]["per_image"]] == [r[:2] for r in plain[4]["per_image"]]
    assert res[1][2][4]["per_image"] == banded[4]["per_image"]  # gathered on every rank


# ---------------------------------------------------------------------------------------------------------------------
# RCCL on the ONE GPU of the test box (round 5): LARVA_DIST_FORCE=1 builds a one-rank "nccl" communicator and every helper
# of larvanet_amd.dist runs its real collective through it.  Not a scaling measurement: a rehearsal of the code path the
# driver's multi-GPU launch takes (librccl beside liblarva_hip.so, Work.wait() between the two captured graphs of the
# split schedule, the timed all-reduce, broadcast, PSNR all-reduce, the banded all-gather).
# ---------------------------------------------------------------------------------------------------------------------
def _train_three_steps(dev, mode):
    from larvanet_amd.autograd import DeferredWgrad
    from larvanet_amd.models import LarvaNet as L
    keep, DeferredWgrad.jobs_per_launch = DeferredWgrad.jobs_per_launch, 4   # several launches even for a small network
    try:
        g = torch.Generator().manual_seed(50)
        x = (torch.rand(2, 3, 12, 16, generator=g) * 255).to(dev)
        t = (torch.rand(2, 3, 48, 64, generator=g) * 255).to(dev)
        args = types.SimpleNamespace(train_path="/tmp")
        m = L.create_model()
        m.parse_args(["--num_modules=2", "--num_blocks=2,1"])
        torch.manual_seed(7)
        m.prepare(is_training=True, scales=[4])
        if mode != "plain":
            m.overlap_allreduce = mode != "whole"
            m.use_hip_graph = mode == "overlap_graph"
        losses = [m.train_step_larva(args, _Val(), x, t) for _ in range(3)]
        torch.cuda.synchronize()
    finally:
        DeferredWgrad.jobs_per_launch = keep
    return m, {"losses": losses, "split_at": getattr(m, "_early_lo", None), "graph": bool(m.use_hip_graph),
               "sd": {k: v.cpu().numpy().copy() for k, v in m.model.state_dict().items()}}


def _resident_loader_digest():
    """device_patch_loader over the synthetic source: its tables and one batch drawn from a fixed seed."""
    from larvanet_amd.dataloaders import device_patch_loader as D
    ld = D.create_loader()
    ld.parse_args(["--device_source=synthetic_loader", "--synthetic_images=5", "--synthetic_lr_size=40", "--data_seed=3"])
    ld.prepare([4])
    t = ld.tables[4]
    x, y = ld.get_device_batch(4, 4, 16)
    torch.cuda.synchronize()
    return {"shapes": [tuple(s) for s in ld.shapes], "lr_sum": int(t["lr"].to(torch.int64).sum()), "hr_sum": int(t["hr"].to(torch.int64).sum()),
            "lr_off": t["lr_off"].tolist(), "hr_hw": t["hr_hw"].tolist(), "dtypes": [str(v.dtype) for v in t.values()],
            "x": x.cpu().numpy().copy(), "y_sum": float(y.double().sum())}


def _rccl_world1_worker(q):
    os.environ.update({"LARVA_DIST_FORCE": "1", "LARVA_DIST_BACKEND": "nccl"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        os.environ.pop(k, None)
    try:
        import torch.distributed as td
        from larvanet_amd import dist as ldist, image_utils, validate
        rank, world = ldist.init_from_env()
        assert (rank, world) == (0, 1) and td.get_backend() == "nccl" and ldist.active()
        dev = torch.device("cuda", 0)
        out = {}
        for mode in ("whole", "overlap_eager", "overlap_graph"):
            m, out[mode] = _train_three_steps(dev, mode)
            # prepare() timed a device tensor through RCCL and chose from it
            out[mode]["dp_schedule"] = dict(m.dp_schedule)
        # the banded all-gather: one band = the whole image, through all_gather_into_tensor on the device
        lr = np.random.RandomState(5).randint(0, 256, size=(3, 9, 12)).astype(np.float32)
        m.model.eval()
        with torch.no_grad():
            banded = image_utils.upscale_banded_device(m, lr, 4, 0, 1, ldist.all_gather_tensor)
            whole = m.upscale_tensor(input_list=[lr])[0]
        out["band_equal"] = bool(torch.equal(banded, whole))
        out["psnr_sum"] = ldist.allreduce_scalar_sum(12.5, dev)
        # round 6: the resident dataset is decoded on rank 0 and BROADCAST (device_patch_loader.share_tables): the size
        # announcement (broadcast_object_list) and one device broadcast per table go through RCCL here
        out["loader"] = _resident_loader_digest()
        torch.manual_seed(0)
        out["validate"] = validate.main(_VALIDATE_ARGS + ["--band_gpus"])[4]["per_image"]
        torch.cuda.synchronize()
        q.put((None, out))
        td.destroy_process_group()
    except Exception as e:
        import traceback
        q.put(("%s\n%s" % (e, traceback.format_exc()), None))


@pytest.mark.timeout(600)
def test_one_rank_rccl_communicator_runs_every_collective_of_the_data_parallel_step(hip_device):
    """The "nccl" backend has no other GPU to talk to on this box, but a one-rank communicator still goes through
    librccl: the split schedule (async all-reduce of the bucket's upper slice + Work.wait() between the two captured
    graphs), the single collective, broadcast_parameters, the timed all-reduce of prepare(), the PSNR all-reduce and the
    banded all-gather all execute, and train exactly like the step without a communicator (a sum over one rank and a
    mean scale of 1/1 are the identity, bit for bit)."""
    from larvanet_amd import validate
    _, plain = _train_three_steps(hip_device, "plain")
    torch.manual_seed(0)
    plain_val = validate.main(list(_VALIDATE_ARGS))[4]["per_image"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(q,))
    p.start()
    err, out = q.get(timeout=500)
    p.join(timeout=60)
    assert err is None, err
    assert out["whole"]["split_at"] is None
    for mode in ("overlap_eager", "overlap_graph"):
        assert out[mode]["split_at"] and out[mode]["split_at"] > 0, "backward was not split: nothing overlapped"
    assert out["overlap_graph"]["graph"], "hipGraph capture of the split backward fell back to eager launches"
    for mode in ("whole", "overlap_eager", "overlap_graph"):
        sched = out[mode]["dp_schedule"]
        assert sched["ranks"] == 1 and sched["allreduce_isolated_us"] > 0 and sched["choice"] in ("flat", "split")
        assert out[mode]["losses"] == plain["losses"], mode
        for k in plain["sd"]:
            assert np.array_equal(out[mode]["sd"][k], plain["sd"][k]), (mode, k)
    assert out["band_equal"] and out["psnr_sum"] == 12.5
    ref = _resident_loader_digest()      # the same loader without a communicator: a plain upload
    got = out["loader"]
    assert np.array_equal(got.pop("x"), ref.pop("x")) and got == ref
    assert [r[:2] for r in out["validate"]] == [r[:2] for r in plain_val]
