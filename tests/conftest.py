import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def ctx():
    """One device context per test session (gpu tests only)."""
    from uq_amd.device import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _stale_lds_is_garbage(request):
    """LDS keeps its bytes between launches, so a kernel that reads LDS it never wrote usually finds what its previous
    launch left there -- the right answer.  Before every GPU test the LDS of all CUs is overwritten with a pattern
    that changes from test to test, so that such a read shows up as a wrong result instead of hiding."""
    if 'ctx' in request.fixturenames:
        import zlib
        from uq_amd import ops
        ops.scribble_lds(request.getfixturevalue('ctx'), zlib.crc32(request.node.nodeid.encode()) | 0x80808080)
    yield
