#!/bin/bash
# Builds tools/asan_check with AddressSanitizer on the HOST side of every .hip source (device code untouched)
# into gpurun_out-independent build dir tools/build_asan/, then the binary can be run on the GPU box:
#   bash tools/asan_check.sh && gpurun -- 'ASAN_OPTIONS=detect_leaks=0 timeout -k 10 300 tools/build_asan/asan_check'
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/build_asan
SRC=iterative_solvers_amd/csrc
FLAGS="-O1 -g -std=c++17 --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
for f in context spmv vsweep convert krylov bicgstab gmres generate dist; do
  if [ ! tools/build_asan/$f.o -nt $SRC/$f.hip ]; then /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o tools/build_asan/$f.o & fi
done
wait
/opt/rocm/bin/hipcc $FLAGS -c tools/asan_check.cpp -x hip -o tools/build_asan/main.o 2>/dev/null || /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=address -c tools/asan_check.cpp -o tools/build_asan/main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -o tools/build_asan/asan_check tools/build_asan/*.o -ldl
ls -la tools/build_asan/asan_check
