"""CPU sanitizer tier: the NIC_HD kernel bodies (env step forward / backward, policy heads, whole-horizon small
rollout — the code the HIP kernels run per scenario) built for the host with AddressSanitizer + UndefinedBehaviorSanitizer and
driven through the same parity checks as the plain host build.  Out-of-bounds pipeline slots, reads past a table, signed
overflow in an index computation or a misaligned access abort the child process.  (GPU ASan is not available on the pool.)"""
import os
import subprocess
import sys

import pytest

import hostsim_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = ["tests/test_hostsim_kernel_bodies.py", "tests/test_host_rollout.py", "tests/test_hostsim_small_rollout.py"]


def test_kernel_bodies_under_asan_ubsan():
    rt = hostsim_util.asan_runtime()
    if rt is None:
        pytest.skip("gcc's libasan.so not found")
    env = dict(os.environ)
    env.update(NIC_HOSTSIM_SANITIZE="1", LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    tests = [t for t in TESTS if os.path.isfile(os.path.join(ROOT, t))]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stdout + r.stderr and "runtime error" not in r.stdout + r.stderr, tail
    assert " passed" in r.stdout
    assert os.path.isfile(os.path.join(ROOT, "tests", "hostsim", "_build", "libhostsim_asan.so"))
