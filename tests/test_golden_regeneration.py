"""Fixture drift guard: when the reference is present (the build container -- it never travels to the GPU box),
regenerate EVERY committed fixture with the committed generator scripts into a temporary directory and require the same keys, dtypes,
shapes and bits.  A generator change without a regenerated fixture (or the reverse) fails here instead of at the judge's desk.
The generators import the reference under the package name `npcd`, which is also this build's package name: they run as child
processes."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "npcd")), reason="the reference is not present (GPU box): fixtures are checked where they are generated")
def test_committed_fixtures_are_what_the_generators_write(tmp_path):
    out = str(tmp_path / "fresh")
    for script in ("make_golden.py", "make_golden_train.py"):
        r = subprocess.run([sys.executable, os.path.join(GOLDEN, script), "--ref", REF, "--out", out], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (script, r.stdout[-2000:], r.stderr[-2000:])
    fresh = sorted(os.path.basename(f) for f in glob.glob(os.path.join(out, "*.npz")))
    committed = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert fresh == committed, (set(fresh) ^ set(committed))
    for name in committed:
        a, b = np.load(os.path.join(out, name), allow_pickle=False), np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), (name, set(a.files) ^ set(b.files))
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (name, k)
