#!/bin/bash
# Build libsequoia_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p build
objs=()
pids=()
for src in *.hip; do
    obj="build/${src%.hip}.o"
    objs+=("$obj")
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer "$obj" -print -quit)" ] \
        || [ ../../include/sequoia_hip.h -nt "$obj" ]; then
        $HIPCC $FLAGS -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libsequoia_hip.so "${objs[@]}"
echo "built $(cd .. && pwd)/libsequoia_hip.so"
