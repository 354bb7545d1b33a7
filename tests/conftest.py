import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The GPU tests are about the specialised kernels: a plan shape the code-object cache has not seen compiles them before the first
# execution instead of starting on the pre-compiled generic pipeline (tests/test_gpu_generic_pipeline.py turns that back on).
os.environ.setdefault("RSQ_GENERIC", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """engine context on GPU 0; the engine has no CPU fallback, so a missing GPU is an error here"""
    from resql_amd import engine
    ctx = engine.Context(device=0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def compile_ctx(tmp_path_factory):
    """compile-only engine context (no GPU): describe + codegen + hiprtc for gfx950"""
    from resql_amd import engine
    ctx = engine.Context(device=-1, cache_dir=str(tmp_path_factory.mktemp("kcache")))
    yield ctx
    ctx.close()
