"""The CPU-side checker under AddressSanitizer + UndefinedBehaviorSanitizer: the known-answer
tests of the C oracle and the tests of the CPU baseline run again, in a child Python that has the sanitizer runtime preloaded and
loads `oracle/libcpd_oracle_asan.so` / `libcpd_baseline_asan.so` (oracle/Makefile: target asan; -fno-sanitize-recover, so any
finding ends the child with a non-zero status), and the plain-C drivers of the C ABI compile cleanly under -Wall -Wextra -fanalyzer.
No GPU sanitizer and no XNACK on this pool: the HIP side is not covered here."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_and_cpu_baseline_known_answers_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update({"GINGR_ORACLE_SANITIZED": "1", "LD_PRELOAD": asan,
                # CPython itself is not leak-clean; everything else the sanitizers find is fatal
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
                "OMP_NUM_THREADS": "4", "PYTHONPATH": ROOT})
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_cpu_baseline.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in p.stdout, tail


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
@pytest.mark.parametrize("src", ["cabi_driver.c", "cabi_fitter_driver.c", "cabi_rccl_rank.c"])
def test_plain_c_drivers_are_clean_under_the_static_analyzer(src, tmp_path):
    """The programs that call the C ABI from plain C (tests/test_gpu_cabi_from_c.py runs them on the GPU box): no warning under
    -Wall -Wextra -fanalyzer, compiled to an object only (linking needs the HIP library's dependencies)."""
    p = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-fanalyzer", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                        "-isystem", "/opt/rocm/include", "-c",
                        os.path.join(ROOT, "tests", "c", src), "-o", str(tmp_path / (src + ".o"))], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
