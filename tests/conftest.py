import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference under oracle/_ref (build container only)")


@pytest.fixture(scope="session")
def nv():
    """The product binding; builds the library first when hipcc is present and the .so is missing."""
    lib = ROOT / "navtex_amd" / "libnavtex_amd.so"
    if not lib.exists():
        import importlib.util
        spec = importlib.util.spec_from_file_location("nvx_build", ROOT / "navtex_amd" / "build.py")
        build = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(build)
        build.build_lib()
    import navtex_amd
    return navtex_amd


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding


def have_ref() -> bool:
    return (ROOT / "oracle" / "_ref" / "ref_full").exists()


def have_gpu() -> bool:
    try:
        import navtex_amd
        return navtex_amd.device_count() > 0
    except Exception:
        return False
