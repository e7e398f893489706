"""larvanet_amd: MI355X-native (gfx950) implementation of the LarvaNet data-parallel hot path --
the cascaded conv3x3/ReLU residual bodies and the per-exit conv -> pixel-shuffle heads of
Geunwoo-Jeon/LarvaNet -- behind the reference's models/base.py plugin surface.

  csrc/        hand-written HIP kernels + the C ABI (include/larva_hip.h)
  hip_lib.py   ctypes binding of that ABI
  kernels.py   tensor-level launch wrappers (validation, allocation, stream)
  autograd.py  torch.autograd.Function glue so that loss.backward() runs the HIP backward
  models/      mirror of the reference plugin interface (create_model(), BaseModel, LarvaNet, LarvaNetV2)
  dist.py      one-process-per-GPU data parallelism over RCCL (flat gradient bucket)

There is no CPU or PyTorch fallback for the compute path.
"""
__version__ = "0.1.0"
