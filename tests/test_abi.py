"""The C-ABI library loads without a GPU and exports every symbol include/smcp_amd.h declares;
compute entry points refuse to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import torch

from smcp_amd import _lib, problems
from smcp_amd.symbolic import Symbolic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "smcp_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:csp|kkt|dense)_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        return
    symb = Symbolic(problems.band_pattern(10, 2))
    x = np.ones(symb.blklen)
    rc = _lib.lib().csp_cholesky(symb.handle, x.ctypes.data, None)
    assert rc == -2                                   # SMCP_ENODEV: context has no device
    assert _lib.lib().csp_device_init(symb.handle, 0, 1) == -2
