"""AddressSanitizer + UndefinedBehaviorSanitizer on the host builds of the optimiser: GPU sanitizers are
not available on the pool, the host restatements compile the very headers the kernels run (csrc/neo_lbfgs*.hpp,
neo_linesearch.hpp) -- round 5 found two latent memory-ordering / stale-buffer defects by luck.  The work happens in a child
process started with libasan preloaded (tests/sanitize/run_sanitized.py)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_builds_are_clean_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan.so next to gcc")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "sanitize", "run_sanitized.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "SANITIZERS CLEAN" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-3000:]
