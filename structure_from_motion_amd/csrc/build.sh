#!/bin/bash
# Builds libba_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libba_hip.so
mkdir -p build
$HIPCC -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fPIC -Wall -Wno-unused-function ${BA_EXTRA_FLAGS} \
    -c ba_engine.hip -o build/ba_engine.o
g++ -O2 -std=c++17 -fPIC -Wall -pthread -c ba_symbolic.cpp -o build/ba_symbolic.o
$HIPCC --offload-arch=gfx950 -shared -fPIC build/ba_engine.o build/ba_symbolic.o -lpthread -o $OUT
echo "built $(realpath $OUT)"
# host-side marshalling helper of the drop-in module (CPython extension, plain C, no GPU code)
gcc -O2 -fPIC -shared -Wall $(python3-config --includes) ba_pyhelp.c -o ../_ba_pyhelp.so
echo "built $(realpath ../_ba_pyhelp.so)"
