"""TEST INFRASTRUCTURE: point the product's ctypes binding at the host-emulator build of the same kernel sources.

The shipped binding (rqvae/_native.py) has no host-pointer path and no switch for one; the emulator tests swap two
module attributes for their own duration -- the loaded library and the pointer marshalling functions -- and restore them."""
import ctypes as C


def install(native, path):
    saved = (native._lib, native.ptr)

    def host_ptr(t, dtype=None):
        if t is None:
            return None
        if not t.is_contiguous():
            raise ValueError('non-contiguous tensor passed to librqamd')
        if dtype is not None and t.dtype != dtype:
            raise ValueError(f'expected {dtype}, got {t.dtype}')
        return C.c_void_p(t.data_ptr())
    def host_view_f32(address, shape, device):
        import numpy as np
        import torch
        n = int(np.prod(shape))
        buf = (C.c_float * n).from_address(int(address))
        return torch.from_numpy(np.frombuffer(buf, dtype=np.float32)).view(*shape)
    saved = saved + (native._view_f32,)
    native._lib = native._bind(path)
    native.ptr = host_ptr
    native._view_f32 = host_view_f32
    return saved


def restore(native, saved):
    native._lib, native.ptr, native._view_f32 = saved
