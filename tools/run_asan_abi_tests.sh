#!/bin/bash
# CPU-side ABI tests against the host-ASan/UBSan build of the library (make asan).  No GPU needed: the tests exercise
# argument validation, layout computation, error strings and symbol export.  GPU ASan is not available on this pool.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$ROOT/svgp-vae_amd/csrc" -j8 asan
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
cd "$ROOT"
# python itself is not instrumented: leak detection would report the interpreter's own arenas
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
SVGP_LIB_PATH="$ROOT/build/asan/libsvgpvae_hip_asan.so" \
python -m pytest tests/test_abi_cpu.py -x -q -p no:cacheprovider "$@"
