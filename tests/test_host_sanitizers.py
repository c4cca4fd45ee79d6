"""The host layer (rt_host.cpp) and the hierarchy builders (rt_tree.cpp) under AddressSanitizer and UBSan:
tests/_bin/host_sanitized is tests/host_sanitizer_exercise.cpp linked with those sources, host code instrumented
(make -C tests, part of __graft_entry__.build).  It needs no GPU: every handle is created without a device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_bin", "host_sanitized")


def test_host_code_is_clean_under_the_sanitizers():
    if not os.path.isfile(EXE):
        pytest.skip("tests/_bin/host_sanitized is not built (make -C tests)")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([EXE], capture_output=True, text=True, timeout=300, env=env)
    report = run.stdout[-2000:] + run.stderr[-4000:]
    assert run.returncode == 0, report
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, report
    assert run.stdout.strip().endswith("done")
