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
# programs that own an RCCL communicator (the library itself resolves RCCL at run time and does not link it)
RCCL="-D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,/opt/rocm/lib"
$CXX $FLAGS -fopenmp -o bin/gpx_dist demos/dist_runner.cpp -L.. -lgpxhost -lgpx -Wl,-rpath,'$ORIGIN/../..' $RCCL -lpthread
$CXX $FLAGS -fopenmp -o bin/rccl_check ../../tests/cpp/rccl_check.cpp -L.. -lgpxhost -lgpx -Wl,-rpath,'$ORIGIN/../..' $RCCL
echo "built host facade"
