import importlib.util
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def load_pkg():
    """Import the package directory `h264-fer_amd` under the importable name h264_fer_amd."""
    if "h264_fer_amd" in sys.modules:
        return sys.modules["h264_fer_amd"]
    d = ROOT / "h264-fer_amd"
    spec = importlib.util.spec_from_file_location("h264_fer_amd", d / "__init__.py", submodule_search_locations=[str(d)])
    m = importlib.util.module_from_spec(spec)
    sys.modules["h264_fer_amd"] = m
    spec.loader.exec_module(m)
    return m


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def fo():
    import fo_py
    fo_py.lib()
    return fo_py
