import os
import sys
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build everything once per session (HIP library cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def small_index(built, tmp_path_factory):
    """1 Mb three-contig genome with repeat families, tandem repeats and N holes + its index."""
    import common
    d = tmp_path_factory.mktemp("idx")
    fa = str(d / "g1.fa")
    common.bw.make_genome(fa, 11, [600000, 300000, 100000], repeats=True)
    common.bw.make_index(fa, str(d / "g1"))
    return {"dir": str(d), "fa": fa, "prefix": str(d / "g1")}
