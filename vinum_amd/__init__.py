"""vinum_amd -- MI355X (gfx950) native operators behind Vinum's native operator boundary.

The package mirrors the reference's hot-path interfaces:
  vinum_amd.vinum_lib        <- the pybind11 module `vinum_lib` (vinum/core/vinum_lib.cpp)
  vinum_amd.core.*           <- the Python physical operators (vinum/core/{algebra,aggregate,expressions}.py)
  vinum_amd.ops / .device    <- device-level building blocks over the C ABI (include/vinum_hip.h)
There is no CPU fallback: without libvinum_hip.so or without a GPU the operators raise.
"""
__version__ = "0.1.0"

_batch_size = 1 << 24  # rows per RecordBatch handed to the GPU (reference default: 10 000, vinum/__init__.py:52)


def get_batch_size():
    return _batch_size


def set_batch_size(batch_size: int):
    global _batch_size
    _batch_size = int(batch_size)
