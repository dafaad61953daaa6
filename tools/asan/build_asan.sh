#!/bin/bash
# Host-side AddressSanitizer / UBSan build of the C++ facade and the three demos (the GPU library itself is the normal
# libgpx.so: sanitizers run on host code only on this pool).  usage: tools/asan/build_asan.sh ; then on a GPU box
# tools/asan/run_asan.sh
set -e
cd "$(dirname "$0")/../.."
H=gp_amd/host
FLAGS="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -std=c++17 -Wall -Wextra -Wno-unused-parameter"
for m in 0:gp 1:fitc 2:vfe; do
  g++ $FLAGS -DDEMO_MODEL=${m%%:*} -o tools/asan/${m##*:}_asan $H/demos/demo.cpp $H/gp.cpp $H/cov.cpp $H/matrix_solver.cpp $H/optimizer.cpp \
      -Lgp_amd -lgpx -Wl,-rpath,'$ORIGIN/../../gp_amd'
done
g++ $FLAGS -o tools/asan/facade_check_asan tests/cpp/facade_check.cpp $H/gp.cpp $H/cov.cpp $H/matrix_solver.cpp $H/optimizer.cpp \
    -Lgp_amd -lgpx -Wl,-rpath,'$ORIGIN/../../gp_amd'
echo "built tools/asan/*_asan"
