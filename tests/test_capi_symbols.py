"""The C-ABI library must load and export every symbol include/vgk.h declares
(no compute calls here: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from util import ENGINE_LIB, ORACLE_LIB, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vgk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vgk_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("vgk_create", "vgk_destroy", "vgk_gssw_pack", "vgk_gssw_run", "vgk_gssw_fetch", "vgk_gssw_align",
              "vgk_batch_free", "vgk_batch_kernel_ms", "vgk_batch_alg_bytes"):
        assert s in syms


@pytest.mark.parametrize("lib", [ENGINE_LIB, ORACLE_LIB])
def test_library_exports_every_declared_symbol(lib):
    if lib == ENGINE_LIB and not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-s", "lib"], cwd=ROOT)     # hipcc cross-compiles without a GPU
    h = ctypes.CDLL(lib)
    for s in declared_symbols():
        assert hasattr(h, s), "%s does not export %s" % (lib, s)
    h.vgk_abi_version.restype = ctypes.c_int
    assert h.vgk_abi_version() == 6


def test_engine_refuses_to_run_without_a_gpu_instead_of_falling_back():
    """On a box without a HIP device vgk_create must fail with VGK_ENODEV (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from vg_amd import capi
    with pytest.raises(capi.VgkError):
        capi.Engine(lib=ENGINE_LIB)
