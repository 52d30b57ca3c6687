import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "k1_variants: runs against the variants build of the library (the wave "
                                       "reduce-scatter / re-seating / cooperative deposit paths of the 2-D tiled kernel)")


@pytest.fixture(autouse=True)
def _k1_variant_library(request):
    """The product library has one deposit path in its 2-D tiled kernel; the three measured-slower alternatives live in
    csrc/build/liblambdapic_amd_variants.so (-DLPA_K1_VARIANTS=1).  Tests that pin those paths -- marked ``k1_variants``,
    or parametrised with ``order`` = CELL_MAJOR (0: wave reduce-scatter) / PADDED (2: cooperative deposit) -- run with
    ``lib()`` switched to that build; everything else runs against the product."""
    params = getattr(getattr(request.node, "callspec", None), "params", {})
    if request.node.get_closest_marker("k1_variants") is None and params.get("order") not in (0, 2):
        yield
        return
    from lambdapic_amd import _lib
    with _lib.use_variants():
        yield


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / f"{name}.npz")
    return load
