"""CPU-side sanitizers (GPU AddressSanitizer is not available on the pool, so the CPU builds are what can be
sanitized): (a) the C oracle built with gcc -fsanitize=address,undefined runs the golden-vector tests of
tests/test_oracle_golden.py in a child process (libasan preloaded into python); (b) the exact scan's host-side segment planner
(lemon_amd/csrc/scan_plan.hpp) is fuzzed by tests/native/plan_fuzz.cpp under the same sanitizers."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GCC = shutil.which("gcc")


def _runtime(name):
    p = subprocess.run([GCC, f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(GCC is None, reason="gcc not available")
def test_oracle_golden_vectors_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("libasan.so not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "liblemon_oracle_san.so"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "liblemon_oracle_san.so")
    env = dict(os.environ, LEMON_ORACLE_SO=so, LD_PRELOAD=":".join(p for p in (asan, ubsan) if p),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=87",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_reference_loop.py"), "-x", "-q", "-s", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)      # (-s below: a sanitizer report must reach stderr, not pytest's capture)
    tail = (r.stdout[-1500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert r.returncode == 0 and " passed" in r.stdout, tail
    # the sanitized library really was the one loaded
    chk = subprocess.run([sys.executable, "-c", "from oracle import oracle as o; o.lib(); print(any('liblemon_oracle_san' in l for l in open('/proc/self/maps')))"],
                         env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert chk.stdout.strip().endswith("True"), (chk.stdout, chk.stderr[-1000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_scan_planner_fuzz_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "plan_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "native", "plan_fuzz.cpp"), "-o", exe])
    for seed in ("1", "20261004"):
        r = subprocess.run([exe, "1500", seed], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "plan_fuzz: ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
