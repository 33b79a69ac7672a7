"""vg_amd — MI355X-native engine for vg's per-read alignment hot path.

The product is the C-ABI shared library vg_amd/libvgamd.so (HIP kernels for
gfx950, see include/vgk.h) plus the C++ host shim vg_amd/libvgamd_host.so that
mirrors vg's Aligner interface.  This Python package is plumbing for tests and
bench.py: a ctypes binding (capi) and synthetic workload generators (workloads).
"""
from . import capi  # noqa: F401
