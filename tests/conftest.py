import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """Path of the CPU oracle binary (test infrastructure only); built on demand with g++."""
    d = os.path.join(ROOT, "oracle")
    exe = os.path.join(d, "modkit_oracle")
    srcs = [os.path.join(d, f) for f in ("modkit_oracle.cpp", "oracle_core.hpp", "oracle_pileup.hpp")]
    # (under pytest-xdist several workers come through here at once: one builds, the others wait — never run a binary that is being written)
    import fcntl
    with open(os.path.join(d, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in srcs):
            subprocess.check_call(["make", "-C", d, "modkit_oracle"])
    return exe


# MKP_SOAK_SHIFT=n moves the seeds of the fuzzed suites (hemi, extract, ingest, duplicate names, summary) by 1000 n for one-off soak runs
# on the GPU box (`profiles/*_soak.txt`); 0 = the seeds the suite pins.  test_gpu_parity_fuzz.py has MKP_FUZZ_SEEDS / MKP_FUZZ_PROFILE_SEEDS.
SOAK = 1000 * int(os.environ.get("MKP_SOAK_SHIFT", "0"))
