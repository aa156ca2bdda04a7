# -*- coding: utf-8 -*-
import os
import sys

import pytest

if os.environ.get('LA_LAB_SET'):
    os.environ.setdefault('LA_LAB_BUILD', '1')      # a session re-run under a kernel-lab variant takes the lab build as the process library
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


@pytest.fixture
def lab_build(request, monkeypatch):
    """The "variant == default" tests run on the LAB build of the same sources (liblookahead_hip_lab.so: csrc/la_knobs.h knobs mutable,
    la_lab_* exported) — the product library has neither the knobs nor their entry points.  For the duration of the test the module-level
    `lib` handle of the test module (and of tests.gpu_utils) is the lab library and every engine created takes it."""
    import sys
    from painlessinferenceacceleration_amd import _lib, llama_engine
    lab = _lib.lab_lib_for('bfloat16')
    for mod in (request.module, sys.modules.get('tests.gpu_utils')):
        if mod is not None and hasattr(mod, 'lib'):
            monkeypatch.setattr(mod, 'lib', lab)
    monkeypatch.setattr(llama_engine, 'DEFAULT_LAB', True)
    yield lab


def pytest_sessionstart(session):
    """LA_LAB_SET="k=v,k=v" (with LA_LAB_BUILD=1: the whole session on the lab build): apply kernel-lab knobs
    (include/lookahead_hip_lab.h) for the whole session — how scripts/ re-run a suite under a variant (e.g. another K-split count)
    before it becomes a default."""
    import os
    spec = os.environ.get('LA_LAB_SET')
    if not spec:
        return
    assert os.environ.get('LA_LAB_BUILD', '') not in ('', '0'), 'LA_LAB_SET needs LA_LAB_BUILD=1: the product libraries have no knobs'

    from painlessinferenceacceleration_amd._lib import check, lab_set
    for kv in spec.split(','):
        k, v = kv.split('=')
        check(lab_set(int(k), int(v)), 'la_lab_set')          # every loaded build, and the fp16 build when it loads later
