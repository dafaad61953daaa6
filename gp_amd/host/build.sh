#!/bin/bash
# Builds the C++ host facade (libgpxhost.so), the three demos and the facade test program.
set -e
cd "$(dirname "$0")"
CXX=${CXX:-g++}
FLAGS="-O2 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter"
$CXX $FLAGS -shared -o ../libgpxhost.so gp.cpp cov.cpp matrix_solver.cpp optimizer.cpp -L.. -lgpx -Wl,-rpath,'$ORIGIN'
mkdir -p bin
LINK="-L.. -lgpxhost -lgpx -Wl,-rpath,\$ORIGIN/../.."
for m in 0:gp 1:fitc 2:vfe; do
  $CXX $FLAGS -DDEMO_MODEL=${m%%:*} -o bin/${m##*:} demos/demo.cpp -L.. -lgpxhost -lgpx -Wl,-rpath,'$ORIGIN/../..'
done
$CXX $FLAGS -o bin/facade_check ../../tests/cpp/facade_check.cpp -L.. -lgpxhost -lgpx -Wl,-rpath,'$ORIGIN/../..'
echo "built host facade"
