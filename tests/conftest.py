import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the CPU-side libraries (oracle + host shim) once per session.
    The HIP library is built by __graft_entry__.build(); GPU tests require it."""
    need = [os.path.join(ROOT, "oracle", "libvgoracle.so"), os.path.join(ROOT, "vg_amd", "libvgamd_host.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-s", "oracle", "host"], cwd=ROOT)
