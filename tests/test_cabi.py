"""The C ABI: libvinum_hip.so loads here (no GPU) and exports every symbol include/vinum_hip.h declares;
the ctypes prototypes cover exactly that set; initialisation without a GPU fails loudly (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vinum_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vnm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vinum_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vinum_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes out of sync with the header"


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vinum_amd import _lib
    lib = _lib.load()
    assert lib.vnm_init(-1) != 0
    assert b"no CPU fallback" in lib.vnm_last_error()
    with pytest.raises(_lib.VinumHipError):
        _lib.lib()
