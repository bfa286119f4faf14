"""ctypes loader of tools/probes/libbd_probes.so (diagnostic kernels that are NOT part of the product library)."""
import ctypes as C
import os

_here = os.path.dirname(os.path.abspath(__file__))
_path = os.path.join(_here, "libbd_probes.so")
if not os.path.exists(_path):
    raise ImportError(f"{_path} missing: run `make -C tools/probes`")
lib = C.CDLL(_path)
P, I32 = C.c_void_p, C.c_int
lib.bd_dense_ws_supported.restype, lib.bd_dense_ws_supported.argtypes = I32, [I32, I32, I32]
lib.bd_dense_ws.restype, lib.bd_dense_ws.argtypes = I32, [P, I32, P, P, P, I32, I32, I32, I32, P, I32, P]
lib.bd_mfma_probe.restype, lib.bd_mfma_probe.argtypes = I32, [I32, I32, P, P]
lib.bd_probe_last_error.restype = C.c_char_p


def check(rc):
    if rc != 0:
        raise RuntimeError(lib.bd_probe_last_error().decode())
