#!/bin/bash
# Developer: host code of the library (engine host side, symbolic analysis, planner) under ASan + UBSan, CPU only.
set -e
cd "$(dirname "$0")/.."
B=/tmp/ba_asan; mkdir -p $B
HIPCC=/opt/rocm/bin/hipcc; CXX=/opt/rocm/lib/llvm/bin/clang++
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
$HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC $SAN -c structure_from_motion_amd/csrc/ba_engine.hip -o $B/engine.o
$CXX -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -c structure_from_motion_amd/csrc/ba_symbolic.cpp -o $B/symbolic.o
$CXX -O1 -g -std=c++17 -fsanitize=address,undefined -c tools/host_asan.cpp -o $B/driver.o
$HIPCC --offload-arch=gfx950 -fsanitize=address,undefined $B/driver.o $B/engine.o $B/symbolic.o -lpthread -o $B/host_asan
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $B/host_asan ${1:-200}
