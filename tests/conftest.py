import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "efficient-speech-codec_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The oracle (torch on the host cores) is what most of the GPU suite's wall time goes to.  A GPU box has 256 hardware threads and torch takes 128 of them by default: the oracle's
# many small operators then run 4.5x (ESC-Base) to 40x (the tiny configuration) SLOWER than on 16 threads (measured on an MI355X box: four oracle-heavy tests 49 s on 16 threads,
# 67 s on 32, 210-232 s on the default) - the same sweep bench.py's cpu_baseline does.  The fixtures were generated on 8 threads; the oracle is pinned to them at any thread count.
try:
    import torch as _torch
    _torch.set_num_threads(min(16, os.cpu_count() or 1))
except Exception:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


def load_manifest(name):
    return json.load(open(os.path.join(GOLDEN, f"{name}_manifest.json")))


def synth_state(name):
    """torch state_dict (reference key names) filled by esc.synth from the fixture manifest."""
    import torch
    from esc import synth
    man = load_manifest(name)
    sd = synth.synth_state_dict(man)
    out = {}
    for k, v in sd.items():
        if k.endswith(".window"):
            v = torch.hann_window(v.shape[0]).numpy()
        out[k] = torch.from_numpy(np.ascontiguousarray(v))
    return out


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """libescx.so is git-ignored: build it in-tree if this checkout does not have it yet (hipcc cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("escx_build", os.path.join(ROOT, "efficient-speech-codec_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=False, verbose=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
