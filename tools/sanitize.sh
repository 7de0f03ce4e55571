#!/usr/bin/env bash
# Sanitizer pass over the host-side code, no GPU needed:
#   1. the CPU restatement (oracle/*.c) built with -fsanitize=address,undefined, driven by the whole CPU test suite;
#   2. the bounds-checked decoders of the drop-in boundary (include/xwb_simulator.hpp StatePacket::decode,
#      include/xwb_endpoint.hpp wire::Message) built the same way, fed the truncated / oversized-count inputs of
#      tests/cpp/test_cpp_interface.cpp ("packet") and tests/cpp/test_endpoint.cpp ("wire", "buffer").
# Usage: tools/sanitize.sh [log file]     (default profiles/r4/sanitizers.txt)
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LOG="${1:-$ROOT/profiles/r4/sanitizers.txt}"
mkdir -p "$(dirname "$LOG")"
ASAN="$(gcc -print-file-name=libasan.so)"
UBSAN="$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"      # (CPython itself is not leak-clean)
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
fail=0
{
  echo "# sanitizer pass, $(date -u +%Y-%m-%dT%H:%M:%SZ), commit $(git -C "$ROOT" rev-parse --short HEAD 2>/dev/null || echo unknown), $(gcc --version | head -1)"
  echo "## 1. oracle/*.c with -fsanitize=address,undefined under the CPU suite"
  make -s -C "$ROOT/oracle" asan || fail=1
  ( cd "$ROOT" && LD_PRELOAD="$ASAN:$UBSAN" XWB_ORACLE_LIB="$ROOT/oracle/_asan/liboracle.so" \
      python -m pytest tests -q -m "not gpu" -p no:cacheprovider \
      --deselect tests/test_sharding_gloo.py 2>&1 | tail -15 ) || fail=1
  echo "## 1b. canary: a deliberate 4-byte destination handed to orc_cv_resize_linear_8u must trip the sanitizer"
  ( cd "$ROOT" && LD_PRELOAD="$ASAN:$UBSAN" XWB_ORACLE_LIB="$ROOT/oracle/_asan/liboracle.so" python - <<'PY' 2>&1 | grep -E "SUMMARY|canary" | head -3
import os, sys
sys.path.insert(0, "tests")
import numpy as np
import _oracle as O
L = O.lib()
pid = os.fork()
if pid == 0:
    src, dst = np.zeros((64, 64, 3), np.uint8), np.zeros(4, np.uint8)
    L.orc_cv_resize_linear_8u(src.ctypes.data_as(O.u8p), 64, 64, 3, dst.ctypes.data_as(O.u8p), 12, 12)
    os._exit(0)
st = os.waitpid(pid, 0)[1]
print("canary", "caught" if st != 0 else "MISSED (sanitizer not active)")
sys.exit(0 if st != 0 else 1)
PY
  ) || fail=1
  echo "## 2. the boundary's decoders (xwb_simulator.hpp, xwb_endpoint.hpp) with -fsanitize=address,undefined"
  SAN="-std=c++11 -O1 -g -Wall -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"
  mkdir -p "$ROOT/tests/cpp/_asan"
  g++ $SAN "$ROOT/tests/cpp/test_cpp_interface.cpp" -o "$ROOT/tests/cpp/_asan/test_cpp_interface" -L"$ROOT/xworld_amd" -lxwb \
      -Wl,-rpath,"$ROOT/xworld_amd" || fail=1
  g++ $SAN "$ROOT/tests/cpp/test_endpoint.cpp" -o "$ROOT/tests/cpp/_asan/test_endpoint" -L"$ROOT/xworld_amd" -lxwb -lpthread \
      -Wl,-rpath,"$ROOT/xworld_amd" || fail=1
  for run in "test_cpp_interface packet" "test_endpoint wire" "test_endpoint buffer"; do
    set -- $run
    echo "-- $1 $2"
    "$ROOT/tests/cpp/_asan/$1" "$2" 2>&1 | tail -5
    [ "${PIPESTATUS[0]}" -eq 0 ] || fail=1
  done
  echo "## result: $([ $fail -eq 0 ] && echo CLEAN || echo FINDINGS)"
} 2>&1 | tee "$LOG"
grep -q "result: CLEAN" "$LOG"
