"""The committed recipe for tests/golden/ is self-checking: where /root/reference exists (the build container), every target of
oracle/gen_golden.py (+ the SSIM known-answer script) is run into a scratch directory, exactly as its docstring says
(`python oracle/gen_golden.py`), and every array / JSON document it writes must equal the committed fixture BIT FOR BIT.
A fixture that went stale, a target that no longer runs, or a new fixture nobody generates fails here.  Skipped on the GPU box
(no reference there; the fixtures are data)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
HAVE_REF = os.path.isdir("/root/reference/core/networks")


def _same_npz(a_path, b_path):
    with np.load(a_path, allow_pickle=False) as a, np.load(b_path, allow_pickle=False) as b:
        assert sorted(a.files) == sorted(b.files), (os.path.basename(a_path), sorted(set(a.files) ^ set(b.files)))
        for k in a.files:
            x, y = a[k], b[k]
            assert x.dtype == y.dtype and x.shape == y.shape, (os.path.basename(a_path), k, x.dtype, y.dtype, x.shape, y.shape)
            # bitwise: NaNs (SURREAL's missed rays) compare by their bytes
            assert x.tobytes() == y.tobytes(), f"{os.path.basename(a_path)}[{k}] differs from the committed fixture"


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference (build container only)")
def test_every_golden_target_regenerates_the_committed_fixture_bitwise(tmp_path):
    out = str(tmp_path / "golden")
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--out", out], capture_output=True, text=True,
                       env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    made = sorted(os.listdir(out))
    committed = sorted(f for f in os.listdir(GOLDEN) if f != "ssim_known_answer.npz")
    assert made == committed, (sorted(set(made) ^ set(committed)), "every committed fixture has a generator and vice versa")
    for f in made:
        if f.endswith(".npz"):
            _same_npz(os.path.join(out, f), os.path.join(GOLDEN, f))
        else:
            with open(os.path.join(out, f)) as fa, open(os.path.join(GOLDEN, f)) as fb:
                a, b = json.load(fa), json.load(fb)
            if f == "PROVENANCE.json":      # a statement about the generating host: the build the fixtures were made with is this one
                assert a["torch"] == b["torch"] and a["torch_norm_equals_fma_chain"] == b["torch_norm_equals_fma_chain"], (a, b)
            else:
                assert a == b, f


def test_torch_norm_of_this_build_is_the_fma_chain_the_bounds_restate():
    """near / far are asserted BIT-equal between kernel bodies, oracle and the reference's tensors; that rests on torch.norm's
    reduction step being one fma on the host that generated the goldens (compiler / ISA dependent).  The committed
    PROVENANCE.json records it for the generating host; on any other host this probe says whether REGENERATED goldens could differ
    (by <= 1 ulp) -- the committed fixtures themselves are data and do not depend on it."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    with open(os.path.join(GOLDEN, "PROVENANCE.json")) as f:
        prov = json.load(f)
    assert prov["torch_norm_equals_fma_chain"]["rows_equal"] == prov["torch_norm_equals_fma_chain"]["rows"]
    import danbo_oracle as o
    import torch
    rng = np.random.default_rng(3)
    x = (rng.normal(size=(50000, 3)) * np.exp(rng.uniform(-6, 6, size=(50000, 1)))).astype(np.float32)
    a = torch.norm(torch.tensor(x), dim=-1).numpy()
    b = o.torch_norm(x)
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, "torch.norm is further than one ulp from the restated chain"
    if ulp.max() != 0:
        pytest.xfail(f"torch.norm of this build ({torch.__version__}) is not the fma chain on {int((ulp != 0).sum())} of 50000 rows: goldens "
                     "regenerated HERE would move near / far by <= 1 ulp against the committed ones")


def test_ssim_known_answer_regenerates_bitwise(tmp_path):
    """oracle/gen_ssim_golden.py needs no reference (an independent float64 scipy evaluation of the published algorithm)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_ssim_golden", os.path.join(ROOT, "oracle", "gen_ssim_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.OUT = str(tmp_path / "ssim_known_answer.npz")
    m.main()
    _same_npz(m.OUT, os.path.join(GOLDEN, "ssim_known_answer.npz"))


def test_unknown_target_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "no_such_target"], capture_output=True, text=True)
    assert r.returncode != 0
