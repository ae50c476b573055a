import os
import subprocess
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first: torch's bundled HIP runtime must be the one libyhair.so binds to)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "yocto-hair_amd", "python"))
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = os.environ.get("YHAIR_SCENES", "/tmp/yhair_test_scenes")
# The kernel-trial record on disk is opt-in (yh_set_trial_cache_dir / YHAIR_CACHE_DIR) and bench.py opts in: keep every process this
# suite starts off it, so that no test depends on what an earlier run left behind and the suite leaves nothing behind for the bench
# (ADVICE r04). test_kernel_trials_persist_on_disk drops the variable for its own subprocesses and names its own directory.
os.environ.setdefault("YHAIR_NO_DISK_CACHE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The in-tree shared libraries travel with the snapshot; build whatever is missing."""
    lib = os.path.join(ROOT, "yocto-hair_amd", "libyhair.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "yocto-hair_amd"), "libyhair.so"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libyh_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])


@pytest.fixture(scope="session")
def built():
    _ensure_built()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_capi as oc
    return oc.Oracle()


@pytest.fixture(scope="session")
def yh(built):
    import yhair_capi
    yhair_capi.load()
    return yhair_capi


@pytest.fixture(scope="session")
def ctx(yh):
    """The HIP context. Fails loudly (never skips, never falls back) when there is no GPU."""
    c = yh.Context(0)
    yield c
    c.close()


def golden(name):
    return np.load(os.path.join(GOLD, name))


def scene_path(name, **kw):
    import make_scenes
    return make_scenes.ensure_scene(name, SCENES, **kw)


# the scene variants tests/golden/scene_*.npz were rendered on (oracle/make_golden.py)
GOLDEN_SCENES = [
    ("sphere-hairblock", dict(scale=0.02)),
    ("sphere-hairblock", dict(scale=0.05, zoom=True)),
    ("straight-hair", dict(scale=0.05)),
    ("straight-hair", dict(scale=0.05, beta_m=0.1)),
    ("curly-hair", dict(scale=0.05)),
    ("hair-curls", dict(scale=0.05)),
    ("lobes", dict(scale=0.05)),
    ("volumes", dict(scale=0.05)),
    ("sphere-hairblock", dict(scale=0.05, dof=True)),
    ("textured", dict(scale=0.05)),
    ("crowd", dict(scale=0.05)),
    ("straight-hair", dict(scale=0.05, beta_m=0.25)),
    ("straight-hair", dict(scale=0.05, beta_m=0.6)),
]


def scene_tag(name, kw):
    return os.path.basename(os.path.dirname(scene_path(name, **kw)))
