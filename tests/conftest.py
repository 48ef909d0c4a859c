import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _rtc_disk_cache_of_this_session(tmp_path_factory):
    """The run-time compiler's disk cache (csrc/teb_rtc.hpp) lives in a directory of this test session: the tests neither read nor write
    the user's ~/.cache/teb_amd, and every session starts with real compilations (the library reads the variable at its first request)."""
    os.environ["TEB_AMD_RTC_CACHE"] = str(tmp_path_factory.mktemp("teb_amd_rtc_cache"))
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py
