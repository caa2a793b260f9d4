"""Pin the pin: the committed fixtures must be what tools/make_golden.py produces
from the UNMODIFIED reference today - arrays AND metadata - so that generator and fixtures cannot drift apart
unnoticed.  Runs only where the reference checkout exists (the build container); the GPU box has none and skips."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "hmvec")), reason="reference checkout not present (GPU box)")
def test_generator_reproduces_the_committed_fixtures(tmp_path):
    names = ["unit_pins", "case_a", "case_c"]
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_golden.py"), "--out", str(tmp_path),
                        "--only", ",".join(names)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for n in names:
        with np.load(os.path.join(tmp_path, n + ".npz"), allow_pickle=False) as new, \
                np.load(os.path.join(GOLDEN, n + ".npz"), allow_pickle=False) as old:
            assert sorted(new.files) == sorted(old.files), n
            for k in new.files:
                a, b = new[k], old[k]
                assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), f"{n}:{k} drifted"
