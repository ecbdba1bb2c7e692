"""autoprog_amd -- MI355X-native hot path of AutoProg (VOLO/DeiT progressive training).

Layout (only what the hot path needs):
  csrc/            hand-written gfx950 HIP kernels + the C ABI (include/autoprog_hip.h)
  _lib.py          ctypes binding of libautoprog_hip.so (fails loudly when missing)
  ops.py           thin tensor-level wrappers + autograd Functions over the C ABI
  models/, loss/, prog/   host-side mirror of the reference's model / loss / index-map API
  dist.py          data-parallel gradient exchange (RCCL through torch.distributed)
"""
__version__ = "0.1.0"
