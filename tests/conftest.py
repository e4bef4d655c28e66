import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the native pieces once (cross-compiles on a CPU-only box, instant when up to date)."""
    from smcounter_amd import build
    build.build_hip()
    build.build_synth()
    import oracle_lib
    oracle_lib.build()


def golden_files():
    import glob
    # (the bam_*.npz fixtures hold BAM bytes and tools_*.npz the down-samplers' files, not pileups: test_bam_golden.py, test_tools.py)
    return sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(f).startswith(("bam_", "tools_")))


def load_golden(path):
    from smcounter_amd import features, pileup, synth
    from smcounter_amd.params import VcParams
    pb, extra = pileup.load_npz(path)
    P = VcParams(**extra["params"])
    db = features.extract_features(pb, P)
    return pb, db, P, synth.StringRef(extra["chroms"]), extra["expected"]


@pytest.fixture(scope="session")
def engine0():
    from smcounter_amd import engine
    eng = engine.Engine(0)
    yield eng
    eng.close()
