#!/bin/bash
# Runs the sanitized demos on the reference's own data (tests/golden/refdata_full.npz) on a GPU box.
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
T=$(mktemp -d)
python3 - "$T" <<'PY'
import sys, numpy as np
g = np.load("tests/golden/refdata_full.npz")
print(list(g.keys()))
np.savetxt(sys.argv[1] + "/train_x", g["X"], fmt="%.17g")
np.savetxt(sys.argv[1] + "/train_y", g["y"], fmt="%.17g")
np.savetxt(sys.argv[1] + "/test_x", g["Xs"], fmt="%.17g")
PY
cd "$T"
export ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
for cmd in "gp_asan 2000" "gp_asan 8000" "fitc_asan 2000 100" "vfe_asan 2000 100"; do
  echo "== $cmd"
  timeout -k 10 300 $ROOT/tools/asan/$cmd > out.log 2> err.log || { echo "FAILED rc=$?"; tail -40 err.log; exit 1; }
  tail -2 out.log | head -1; grep -c "runtime error" err.log || true; tail -2 err.log
done
echo "== facade_check_asan"
timeout -k 10 300 $ROOT/tools/asan/facade_check_asan 500 50 0 > "$T/fc.log" 2> "$T/fc.err" || { echo "FAILED rc=$?"; tail -40 "$T/fc.err"; exit 1; }
tail -2 "$T/fc.log"; grep -c "runtime error" "$T/fc.err" || true
# the same programs against libgpx.so built with the host side sanitized too (tools/asan/lib, clang runtime)
if [ -x "$ROOT/tools/asan/gp_lib_asan" ]; then
  cd "$T"
  export ASAN_OPTIONS=$ASAN_OPTIONS:protect_shadow_gap=0
  for cmd in "gp_lib_asan 2000" "fitc_lib_asan 2000 100" "vfe_lib_asan 2000 100"; do
    echo "== $cmd"
    timeout -k 10 300 $ROOT/tools/asan/$cmd > out.log 2> err.log || { echo "FAILED rc=$?"; tail -60 err.log; exit 1; }
    tail -2 out.log | head -1; grep -c "ERROR: AddressSanitizer" err.log || true; tail -2 err.log
  done
fi
