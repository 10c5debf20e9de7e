import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, f"{name}.npz")
    if not os.path.exists(path):
        # a missing fixture is a broken checkout, not a reason to shrink the parity surface silently
        pytest.fail(f"golden fixture tests/golden/{name}.npz is missing (regenerate it with tools/make_golden*.py "
                    "in the build container)")
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_names(pred=lambda n: True):
    if not os.path.isdir(GOLDEN):
        return []
    # model fixtures only: `<cfg>_x.npz` (round-2 extras on <cfg>'s model state), `compat_<cfg>.npz` (round-3 small
    # operators) and `g7.npz` (the C1 trace) are loaded by name where they are used
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and not f.startswith(("_", "compat_", "multistart_")) and not f.endswith("_x.npz") and f not in ("g7.npz", "tgn.npz")
                  and pred(f[:-4]))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
