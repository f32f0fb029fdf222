"""Import alias: the package directory is named ``hnsw-clj_amd`` (after damesek/hnsw-clj), which is
not a valid Python identifier, so ``import hnsw_clj_amd`` resolves its submodules from that
directory.  All code lives in ``hnsw-clj_amd/``; nothing lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "hnsw-clj_amd")
__path__.insert(0, _real)
PACKAGE_DIR = _real
