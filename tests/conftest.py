import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The HNSW traversal's int8 rejection test is on by default only for launches that fill the chip; the parity tests use
# small batches, so new handles in the test processes get it on EVERY launch (mode 2): every bit-for-bit comparison
# against the oracle below then also proves that the test rejects nothing the reference would admit.
# test_rejection_test_modes_agree covers modes 0 and 1.
os.environ.setdefault("HNSWGPU_PREFILTER", "2")
# The boundary between the GEMV-order IVF searches (survivor stream) and the MFMA tile scan is 48 (query, list) pairs per
# list on a handle with int8 rows, 12 without: the parity tests' small indexes sit on both sides of 12, so the test
# processes pin it there (every regime then runs at test size); test_ivf_production_boundary covers the default.
os.environ.setdefault("HNSWGPU_TILE_PAIRS", "12")
# Small HNSW launches spread one query over several CUs (solo_kernels.hpp) from ef 200 by default; the parity tests search
# with every ef, mostly small ones, so the test processes take that path at EVERY ef (hnsw-clj_amd/_native.py applies
# HNSWGPU_TUNE through hnswgpu_set_tuning when the library is loaded); test_helpers_evaluate_small_launches also runs the
# round-2 helper kernel (SOLO = 0) and the default rule.
os.environ.setdefault("HNSWGPU_TUNE", "SOLO=2")
# (the first two are among the six names the library reads from the environment, once, when it is loaded; every other switch a
# test flips goes through hnswgpu_set_tuning: the `tune` fixture below)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def native_lib():
    """libhnswgpu.so, built in-tree (cross-compiles for gfx950 without a GPU)."""
    from hnsw_clj_amd import _native

    _native.build()
    return _native


class _Tune:
    """hnswgpu_set_tuning for one test: set(name, value) / unset(name); every key goes back to what it was afterwards."""

    def __init__(self, native):
        self._n, self._saved = native, {}

    def set(self, name, value):
        self._saved.setdefault(name, self._n.get_tuning(name))
        self._n.set_tuning(name, int(value))

    def unset(self, name):
        self._saved.setdefault(name, self._n.get_tuning(name))
        self._n.set_tuning(name, None)

    def restore(self):
        for name, v in self._saved.items():
            self._n.set_tuning(name, v)
        self._saved.clear()


@pytest.fixture
def tune(native_lib):
    t = _Tune(native_lib)
    yield t
    t.restore()
