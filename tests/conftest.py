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
