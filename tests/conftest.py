import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import loader
    loader.lib()
    return loader


@pytest.fixture(scope="session")
def sfb():
    """The product package; importing it loads libsfb.so (built by __graft_entry__.build())."""
    lib_path = os.path.join(ROOT, "smooth_feedback_amd", "libsfb.so")
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build()
    import smooth_feedback_amd
    return smooth_feedback_amd


@pytest.fixture
def knobs(sfb):
    """Debug knobs of libsfb.so (sfb_debug_set, csrc/knobs.h) for one test: knobs.set(SFB_SP_GRID=4), knobs.clear("SFB_SP_GRID");
    everything the test set is cleared afterwards.  The library reads no environment variable."""
    class K:
        def __init__(self):
            self.touched = set()

        def set(self, **kw):
            for k, v in kw.items():
                sfb.debug_set(k, v)
                self.touched.add(k)

        def clear(self, *names):
            for k in names:
                sfb.debug_set(k, None)

    k = K()
    yield k
    for name in k.touched:
        sfb.debug_set(name, None)
