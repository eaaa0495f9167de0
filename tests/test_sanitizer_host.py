"""Host-build sanitizer job: the host side of libssc.so -- every entry point's argument validation,
descriptor copies, the thread-local error buffer -- compiled with -fsanitize=address,undefined and driven by
tests/sanitizer/abi_args.c with invalid, empty and boundary arguments.  Everything returns before the first HIP call,
so this needs no GPU (GPU AddressSanitizer is not available on this pool: the device code is built unsanitized)."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="hipcc not available")


def test_abi_argument_validation_under_asan_ubsan():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ssc_sanitizer_build", os.path.join(os.path.dirname(__file__), "sanitizer", "build.py"))
    sb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sb)
    exe = sb.build()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "all expectations hold" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
