"""Data parallelism for the training path: one process per GPU, torch.distributed with the
"nccl" backend (= RCCL over xGMI on ROCm), gloo for CPU tests.

The reference is single-process (no collective anywhere); patches are independent and the loss
is a mean, so the only exchange a step needs is ONE sum all-reduce of all gradients (832 704
fp32 = 3.33 MB for M4B4).  At that size the collective is latency-bound on the xGMI mesh, so the
gradients travel as a single flat bucket, not per tensor.  Validation PSNR is all-reduced so that
every rank's ReduceLROnPlateau takes the same decision.
"""
import os

# dmabuf IPC (the only mode the host driver supports): must be in the environment BEFORE the
# HIP/HSA runtime comes up, i.e. before the first torch.cuda call of the process -- importing this
# module is early enough, init_from_env() after torch.cuda.set_device() would not be.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


def rank():
    return td.get_rank() if is_initialized() else 0


def world_size():
    return td.get_world_size() if is_initialized() else 1


def is_main():
    return rank() == 0


def forced():
    """LARVA_DIST_FORCE=1: a process without torchrun's environment still builds a ONE-rank communicator (nccl = RCCL on
    a GPU box) and every helper below runs its real collective through it instead of short-circuiting on "one rank".
    A rehearsal of the code the driver's multi-GPU launch takes -- librccl loaded beside liblarva_hip.so, Work.wait()
    ordering a collective against captured graphs, device tensors through all_reduce / broadcast / all_gather --, not
    a scaling measurement."""
    return os.environ.get("LARVA_DIST_FORCE", "0") not in ("", "0")


def active():
    """Do the helpers issue collectives?  More than one rank, or a forced one-rank communicator."""
    return is_initialized() and (world_size() > 1 or forced())


def init_from_env(backend=None):
    """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (no-op if absent or 1)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if is_initialized() or (ws <= 1 and not forced()):
        return rank(), world_size()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    ndev = torch.cuda.device_count()   # (counting devices does not initialise the runtime)
    use_gpu = ndev > 0
    # LARVA_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = backend or os.environ.get("LARVA_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if use_gpu:
        if backend == "nccl" and local >= ndev:
            # RCCL refuses two ranks on one device; mapping them there silently would only hang later
            raise RuntimeError("larvanet_amd: LOCAL_RANK %d needs its own GPU but only %d device(s) are visible "
                               "(one process per GPU; LARVA_DIST_BACKEND=gloo rehearses more ranks than GPUs)"
                               % (local, ndev))
        torch.cuda.set_device(local % ndev)
    if ws <= 1:
        # forced one-rank communicator (see forced()): a rendezvous of its own through a FileStore created in-process --
        # no port to race for, and nothing exported into os.environ (a child started later must not look like a torchrun rank)
        import tempfile
        fd, path = tempfile.mkstemp(prefix="larva_rdzv_")
        os.close(fd)
        os.unlink(path)    # (FileStore creates the file itself and removes it when the last reference goes)
        td.init_process_group(backend=backend, store=td.FileStore(path, 1), rank=0, world_size=1)
    else:
        td.init_process_group(backend=backend)
    return rank(), world_size()


def broadcast_parameters(module, src=0):
    """Every rank starts from rank `src`'s weights (the reference seeds nothing)."""
    if not active():
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    if not tensors:
        return
    flat = torch._utils._flatten_dense_tensors(tensors)
    td.broadcast(flat, src=src)
    for t, synced in zip(tensors, torch._utils._unflatten_dense_tensors(flat, tensors)):
        t.copy_(synced)


def broadcast_object(obj, src=0):
    """Rank `src`'s (small, picklable) python object on every rank."""
    if not active():
        return obj
    box = [obj if rank() == src else None]
    td.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_tensor(t, src=0):
    """In-place broadcast of rank `src`'s tensor (same shape / dtype allocated on every rank) -> t.  One collective over
    RCCL for device tensors; under gloo (the CPU rehearsal, or LARVA_DIST_BACKEND=gloo with HIP tensors) device tensors
    are staged through the host."""
    if not active():
        return t
    if t.is_cuda and td.get_backend() != "nccl":
        host = t.cpu()
        td.broadcast(host, src=src)
        t.copy_(host)
        return t
    td.broadcast(t, src=src)
    return t


def host_threads():
    """This rank's share of the host's cores: usable cores (affinity mask, cgroup quota) // ranks on this host
    (LOCAL_WORLD_SIZE, torchrun sets it), at least 1.  Eight ranks of one node each defaulting to `all cores` intra-op
    threads oversubscribe the host 8x and stretch every rank's launch loop."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    return max(1, cores // max(1, local_world))


def limit_host_threads():
    """Called by the drivers right after init_from_env(): torch's intra-op pool (CPU restatement of nothing on the product
    path, but numpy / PIL / torch host ops of loaders and validation use it) is capped at this rank's share unless the
    user set OMP_NUM_THREADS.  No CPU affinity is set here: under torchrun the expected launch is one rank per GPU with
    the default (unpinned) affinity; pin with `numactl` / `taskset` per rank outside if the node's NUMA layout calls for
    it.  -> the thread count in use."""
    if os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    n = host_threads()
    torch.set_num_threads(n)
    return n


def allreduce_sum(t, async_op=False):
    """In-place sum over ranks of a (slice of a) flat buffer; async_op -> the Work handle."""
    if not active():
        return None
    return td.all_reduce(t, op=td.ReduceOp.SUM, async_op=async_op)


def time_allreduce_us(flat, warmup=3, iters=10):
    """Median wall time (us) of an ISOLATED sum all-reduce of `flat` (the gradient bucket), agreed on by all ranks
    (MAX over ranks): what the training step would wait for if it issued ONE collective after backward.  The buffer's
    contents are restored.  Used once at prepare() to choose the data-parallel weight-gradient schedule."""
    if not active():
        return 0.0
    import time
    keep = flat.clone()
    cuda = flat.is_cuda
    times = []
    for i in range(warmup + iters):
        if cuda:
            torch.cuda.synchronize()
        td.barrier()
        t0 = time.perf_counter()
        td.all_reduce(flat, op=td.ReduceOp.SUM)
        if cuda:
            torch.cuda.synchronize()
        if i >= warmup:
            times.append((time.perf_counter() - t0) * 1e6)
        flat.copy_(keep)
    med = sorted(times)[len(times) // 2]
    t = torch.tensor([med], dtype=torch.float64, device=flat.device if td.get_backend() == "nccl" else "cpu")
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(module, bucket=None):
    """Mean of the gradients over ranks as ONE collective.  With a GradBucket the gradients already
    live in one flat buffer (no flatten / unflatten copies); otherwise they are flattened here."""
    ws = world_size()
    if not active():
        return
    if bucket is not None and bucket.intact(module):
        td.all_reduce(bucket.flat, op=td.ReduceOp.SUM)
        bucket.flat.mul_(1.0 / ws)
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch._utils._flatten_dense_tensors(grads)
    td.all_reduce(flat, op=td.ReduceOp.SUM)
    flat.mul_(1.0 / ws)
    for g, synced in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(synced)


def allreduce_scalar_sum(value, device):
    if not active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if td.get_backend() == "nccl" else "cpu")
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())


def all_gather_tensor(t):
    """[world][*t.shape]: every rank's tensor `t` (same shape and dtype everywhere), in rank order,
    as ONE device collective (RCCL all-gather over xGMI; gloo for the CPU rehearsal) -- no pickling,
    no host round trip."""
    ws = world_size()
    t = t.contiguous()
    if not active():
        return t.unsqueeze(0)
    out = torch.empty((ws,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    try:
        td.all_gather_into_tensor(out, t)
    except (RuntimeError, NotImplementedError):   # a backend without the flat form
        td.all_gather(list(out.unbind(0)), t)
    return out


def seed_for_rank(base_seed):
    """Distinct patch streams per rank (the reference draws from the un-seeded global numpy RNG,
    dataloaders/div2k_train_loader.py:63,79-80,87,92: identical streams under naive replication)."""
    return int(base_seed) + 1000 * rank()


def gather_objects(obj):
    """All ranks' python objects, in rank order, on every rank."""
    if not active():
        return [obj]
    out = [None] * world_size()
    td.all_gather_object(out, obj)
    return out
