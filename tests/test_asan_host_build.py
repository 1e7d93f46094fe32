"""Host-side AddressSanitizer / UBSan run of the CPU ABI tests: `make asan`
instruments the host code of every translation unit (device code is compiled normally; GPU ASan is not available on
this pool), tools/run_asan_abi_tests.sh loads that build into pytest through SVGP_LIB_PATH."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_tests_pass_under_host_asan():
    if not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"):
        pytest.skip("clang ASan runtime not present")
    if os.environ.get("SVGP_LIB_PATH"):
        pytest.skip("already running inside the ASan harness")
    r = subprocess.run([os.path.join(ROOT, "tools", "run_asan_abi_tests.sh")], capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in tail
