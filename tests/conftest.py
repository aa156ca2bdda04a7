# -*- coding: utf-8 -*-
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """CPU tests need the in-tree liblookahead_hip.so (hipcc cross-compiles without a GPU)."""
    so = os.path.join(ROOT, "painlessinferenceacceleration_amd", "liblookahead_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    yield


def pytest_sessionstart(session):
    """LA_LAB_SET="k=v,k=v": apply kernel-lab knobs (include/lookahead_hip_lab.h) for the whole session — how scripts/ re-run a
    suite under a variant (e.g. another K-split count) before it becomes a default."""
    import os
    spec = os.environ.get('LA_LAB_SET')
    if not spec:
        return
    from painlessinferenceacceleration_amd._lib import check, lab_set
    for kv in spec.split(','):
        k, v = kv.split('=')
        check(lab_set(int(k), int(v)), 'la_lab_set')          # every loaded build, and the fp16 build when it loads later
