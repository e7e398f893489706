"""The data-parallel host logic on 2 CPU processes over gloo: flat-bucket gradient mean, weight
broadcast, PSNR all-reduce, per-rank patch streams, validation sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch.distributed as td
    from larvanet_amd import dist as ldist
    r, w = ldist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and ldist.is_main() == (rank == 0)
    torch.manual_seed(100 + rank)  # ranks start from different weights
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 3))
    ldist.broadcast_parameters(net)
    w0 = torch.cat([p.detach().flatten() for p in net.parameters()])
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    ldist.allreduce_gradients(net)
    grads = [float(p.grad.flatten()[0]) for p in net.parameters()]
    psnr_sum = ldist.allreduce_scalar_sum(10.0 + rank, torch.device("cpu"))
    gathered = ldist.gather_objects([(rank, rank * 2.0)])

    # validation sharding: rank r scores images r, r+world, ...; every rank steps its scheduler with the same mean
    from larvanet_amd.models import LarvaNet as L
    m = L.create_model()
    m.parse_args(["--num_modules=1", "--num_blocks=1"])
    m.prepare(is_training=True, scales=[4])
    seen = []

    def fake_upscale(input_list, scale):
        seen.append(int(input_list[0][0, 0, 0]))
        return np.repeat(np.repeat(np.asarray(input_list, np.float32), 4, axis=2), 4, axis=3)

    m.upscale = fake_upscale

    class Val:
        def get_num_images(self):
            return 5

        def get_image_pair(self, image_index, scale):
            lr = np.full((3, 4, 4), image_index, np.float32)
            hr = np.full((3, 16, 16), image_index + 1 + image_index % 2, np.float32)
            return lr, hr, str(image_index)

    avg = m.validate_for_train(None, Val())

    # one image as one row band per rank, moved by a tensor all-gather (validate.py --band_gpus)
    from larvanet_amd import image_utils

    class Nearest:   # a 1-row-halo "network": x4 nearest upscale of a vertical 3-tap sum (zero padded)
        device = torch.device("cpu")

        def receptive_halo(self):
            return 1

        def upscale_tensor(self, input_list):
            a = torch.from_numpy(np.asarray(input_list, np.float32))
            p = torch.nn.functional.pad(a, (0, 0, 1, 1))
            s = p[:, :, :-2] + p[:, :, 1:-1] + p[:, :, 2:]
            return s.repeat_interleave(4, 2).repeat_interleave(4, 3)

    img = np.random.RandomState(5).randint(0, 256, size=(3, 7, 5)).astype(np.float32)
    banded = image_utils.upscale_banded_device(Nearest(), img, 4, rank, world, ldist.all_gather_tensor)
    band_ok = bool(torch.equal(banded, Nearest().upscale_tensor([img])[0]))
    gathered_t = ldist.all_gather_tensor(torch.full((2, 3), float(rank)))
    band_ok = band_ok and gathered_t.shape == (world, 2, 3) and float(gathered_t[1, 0, 0]) == 1.0

    # a data-parallel step whose gradient bucket gets detached mid-run (somebody called
    # zero_grad(set_to_none=True)): the mean over ranks must be applied once, not 1/world twice
    from larvanet_amd.optim import FlatAdamW
    lin = torch.nn.Linear(2, 1, bias=False)
    with torch.no_grad():
        lin.weight.fill_(1.0)

    class Bucket:
        flat = torch.zeros(2)

        def intact(self, module=None):
            return False

    opt = FlatAdamW(list(lin.parameters()), lin.weight.data.view(-1), Bucket(), lr=0.1, weight_decay=0.0)
    m.optim, m.model, m.grad_bucket = opt, lin, Bucket()
    opt.mean_scale = 1.0 / world   # left over from the steps that ran on the intact bucket
    lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))
    m._finish_backward()
    mean_grad = float(lin.weight.grad.flatten()[0])
    opt.step()
    scale_after = opt.mean_scale
    from larvanet_amd.dataloaders import synthetic_loader
    ld = synthetic_loader.create_loader()
    ld.parse_args(["--synthetic_images=4", "--synthetic_lr_size=24"])
    ld.prepare([4])
    first_patch = ld.get_patch_batch(1, 4, 12)[0][0]
    q.put((rank, w0.numpy(), grads, psnr_sum, gathered, seen, float(avg), float(np.asarray(first_patch).sum()),
           band_ok, mean_grad, scale_after, float(lin.weight.grad.flatten()[0])))
    td.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, g0, s0, ga0, seen0, avg0, patch0, *x0), (r1, w1, g1, s1, ga1, seen1, avg1, patch1, *x1) = out
    assert x0[0] and x1[0]                               # row bands through the tensor all-gather == full image
    assert x0[1] == x1[1] == 1.5 and x0[2] == 1.0        # mean of (1, 2) once; the stale 1/world is reset
    assert x0[3] == 1.5                                  # ... and the optimizer did not scale it again
    assert np.array_equal(w0, w1)                       # broadcast from rank 0
    assert g0 == g1 == [1.5 * (i + 1) for i in range(4)]  # mean of (1, 2) x (i+1)
    assert s0 == s1 == 21.0
    assert ga0 == ga1 == [[(0, 0.0)], [(1, 2.0)]]
    assert seen0 == [0, 2, 4] and seen1 == [1, 3]        # image i -> rank i mod world
    assert avg0 == avg1                                  # identical scheduler input on every rank
    assert patch0 != patch1                              # per-rank patch streams


def _forced_worker(q):
    os.environ.update({"LARVA_DIST_FORCE": "1", "LARVA_DIST_BACKEND": "gloo"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        os.environ.pop(k, None)
    import torch.distributed as td
    from larvanet_amd import dist as ldist
    assert not ldist.active()
    r, w = ldist.init_from_env()
    ok = (r, w) == (0, 1) and ldist.is_initialized() and ldist.active() and td.get_backend() == "gloo"
    calls = []
    real = td.all_reduce
    td.all_reduce = lambda *a, **k: (calls.append("all_reduce"), real(*a, **k))[1]
    net = torch.nn.Conv2d(3, 4, 3)
    ldist.broadcast_parameters(net)
    for p in net.parameters():
        p.grad = torch.full_like(p, 3.0)
    ldist.allreduce_gradients(net)
    grads = [float(p.grad.flatten()[0]) for p in net.parameters()]
    flat = torch.arange(8, dtype=torch.float32)
    work = ldist.allreduce_sum(flat[4:], async_op=True)
    ldist.allreduce_sum(flat[:4])
    work.wait()
    t_us = ldist.time_allreduce_us(flat)
    s = ldist.allreduce_scalar_sum(2.5, torch.device("cpu"))
    gathered = ldist.all_gather_tensor(torch.full((2, 3), 7.0))
    objs = ldist.gather_objects("x")
    q.put((ok, grads, flat.tolist(), t_us > 0, s, tuple(gathered.shape), objs, len(calls)))
    td.destroy_process_group()


@pytest.mark.timeout(120)
def test_forced_one_rank_communicator_issues_real_collectives():
    """LARVA_DIST_FORCE=1 (round 5: the rehearsal of the RCCL path on a one-GPU box, here over gloo): a process
    without torchrun's environment builds a one-rank group and the helpers stop short-circuiting on "one rank"."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(q,))
    p.start()
    ok, grads, flat, timed, s, shape, objs, ncalls = q.get(timeout=100)
    p.join(timeout=30)
    assert p.exitcode == 0
    assert ok and grads == [3.0, 3.0] and flat == list(map(float, range(8))) and timed and s == 2.5
    assert shape == (1, 2, 3) and objs == ["x"]
    assert ncalls >= 5   # the gradients, two slices, the timed loop, the scalar: none was skipped


def test_helpers_stay_inert_without_a_communicator():
    from larvanet_amd import dist as ldist
    assert not ldist.active() and ldist.world_size() == 1
    t = torch.ones(3)
    assert ldist.allreduce_sum(t) is None and ldist.time_allreduce_us(t) == 0.0
    assert ldist.all_gather_tensor(t).shape == (1, 3) and ldist.gather_objects(1) == [1]


# ------------------------------------------------------------------------------------------------
# round 6: decode once on rank 0, broadcast the resident tables (dataloaders/device_patch_loader.py)
# ------------------------------------------------------------------------------------------------
class _CountingSource:
    """synthetic_loader with a decode counter: only rank 0 may ever decode."""

    def __init__(self):
        from larvanet_amd.dataloaders import synthetic_loader
        self.inner = synthetic_loader.create_loader()
        self.inner.parse_args(["--synthetic_images=6", "--synthetic_lr_size=20"])
        self.inner.prepare([4])
        self.decodes = 0

    def get_num_images(self):
        return self.inner.get_num_images()

    def get_image_pair(self, i, scale):
        self.decodes += 1
        return self.inner.get_image_pair(i, scale)


def _tables_worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch.distributed as td
    from larvanet_amd import dist as ldist
    from larvanet_amd.dataloaders import device_patch_loader as D
    ldist.init_from_env(backend="gloo")
    threads = ldist.limit_host_threads() if "OMP_NUM_THREADS" not in os.environ else None
    src = _CountingSource()
    host, shapes = D.build_host_tables(src, [4], workers=3) if ldist.is_main() else (None, None)
    tables, shapes = D.share_tables(host, shapes, [4], torch.device("cpu"))
    # the rank's own draw stream over the shared shapes (seed + 1000 rank)
    rng = np.random.RandomState(ldist.seed_for_rank(7))
    draws = D.draw_batch(rng, shapes, 4, 12)
    t = tables[4]
    digest = (int(t["lr"].to(torch.int64).sum()), int(t["hr"].to(torch.int64).sum()), t["lr_off"].tolist(), t["hr_off"].tolist(),
              t["lr_hw"].tolist(), t["hr_hw"].tolist(), [str(v.dtype) for v in t.values()])
    q.put((rank, src.decodes, digest, shapes, draws.tolist(), threads, ldist.host_threads()))
    td.destroy_process_group()


@pytest.mark.timeout(300)
def test_resident_tables_are_decoded_once_and_broadcast():
    """start-up O(1) in the world size.  Rank 0 decodes (thread pool), the uint8 tables and the
    offset / size tables reach rank 1 by broadcast (rank 1 never calls get_image_pair), both ranks hold identical tables
    equal to a serial single-process build, and their draw streams differ."""
    from larvanet_amd.dataloaders import device_patch_loader as D
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tables_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, dec0, dig0, sh0, dr0, th0, ht0), (_, dec1, dig1, sh1, dr1, th1, ht1) = out
    assert dec0 == 6 and dec1 == 0
    assert dig0 == dig1 and sh0 == sh1 and dr0 != dr1
    # ... and equal to the serial build without a communicator (this process)
    src = _CountingSource()
    host, shapes = D.build_host_tables(src, [4], workers=1)
    tables, shapes2 = D.share_tables(host, shapes, [4], torch.device("cpu"))
    t = tables[4]
    assert dig0[:2] == (int(t["lr"].to(torch.int64).sum()), int(t["hr"].to(torch.int64).sum()))
    assert dig0[2] == t["lr_off"].tolist() and dig0[5] == t["hr_hw"].tolist() and [tuple(s) for s in sh0] == shapes2
    assert t["lr_off"][1] == 3 * shapes[0][0] * shapes[0][1]            # offsets count bytes of the CHW uint8 images
    assert dig0[6] == ["torch.uint8", "torch.uint8", "torch.int64", "torch.int64", "torch.int32", "torch.int32"]
    # each rank took half of the host's cores for itself
    assert ht0 == ht1 >= 1
    if th0 is not None:
        assert th0 == ht0


def test_host_threads_divides_the_cores_among_the_local_ranks(monkeypatch):
    from larvanet_amd import dist as ldist
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    one = ldist.host_threads()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert ldist.host_threads() == max(1, one // 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "100000")
    assert ldist.host_threads() == 1
