import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
ALL_FIXTURES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith("aux_"))
# training-mode fixtures (dropout active, the keep decisions recorded): their own tests; the generic ones run the network in eval mode
DROP_GOLDEN = [n for n in ALL_FIXTURES if "_drop_" in n]
GOLDEN_NAMES = [n for n in ALL_FIXTURES if n not in DROP_GOLDEN]
SMALL_GOLDEN = [n for n in GOLDEN_NAMES if n.endswith("_small") or n in ("short_fg_nohier", "short_fg_s40")]
# fixtures the HIP path does not implement yet (the oracle and the host mirror do): per-point FiLM
NOT_ON_GPU_YET = set()
NO_BACKWARD_YET = set()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """One golden fixture (tests/golden/<name>.npz, written by tests/golden/make_golden.py)."""

    def __init__(self, name):
        self.name = name
        self.d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(bytes(self.d["meta_json"]).decode())

    def __contains__(self, k):
        return k in self.d.files

    def __getitem__(self, k):
        return self.d[k]

    def get(self, k):
        return self.d[k] if k in self.d.files else None

    def volumes(self):
        """Feature volume(s): one array, or the list of pyramid levels."""
        extra = sorted(k for k in self.d.files if k.startswith("feature_volume_l"))
        return [self.d["feature_volume"]] + [self.d[k] for k in extra] if extra else self.d["feature_volume"]

    def params(self, prefix="param/siren."):
        return {k[len(prefix):]: self.d[k] for k in self.d.files if k.startswith(prefix)}

    def grads(self, prefix="grad/siren."):
        return {k[len(prefix):]: self.d[k] for k in self.d.files if k.startswith(prefix)}

    def oracle_dropout(self, T):
        """Keyword arguments of oracle.render_oracle.render that put the network in this fixture's training mode."""
        p = self.meta.get("drop_out", 0)
        return dict(drop_p=p, drop_coarse=T(self.get("drop_coarse")), drop_fine=T(self.get("drop_fine"))) if p else {}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


def scaled_err(a, b):
    """max |a-b| / max(|b|, rms(b)): relative error with the tensor's own rms as the floor.

    This is the metric behind every 'rel' tolerance in the parity tests: pure |a-b|/|b| is
    meaningless for densities that cross zero, and an absolute floor like 1e-3 is below what
    fp32 itself resolves once the head is scaled (the reference's own fp32-vs-fp64 error on
    sigma is ~7e-6 of its range).
    """
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    floor = max(float(np.sqrt(np.mean(b * b))), 1e-30)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
