import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import oracle_c

    oracle_c.build()
    return oracle_c


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if needed) and loads libsoftrod_hip.so; never falls back."""
    from gym_softrobot_amd import _capi

    if not _capi.library_path().exists():
        import __graft_entry__

        __graft_entry__.build()
    lib = _capi.load_library()
    # The .so is git-ignored and travels prebuilt to the GPU box: a binary older than csrc/ or
    # include/ would be tested against the wrong kernels.  Fail loudly, here and on the GPU box.
    built, disk = _capi.library_source_hash(), _capi.source_hash()
    if built != disk:
        pytest.fail(f"libsoftrod_hip.so was built from sources {built}, the tree holds {disk}: "
                    "run `python __graft_entry__.py build`", pytrace=False)
    return lib
