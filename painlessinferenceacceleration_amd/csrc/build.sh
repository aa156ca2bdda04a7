#!/bin/bash
# Build liblookahead_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/../liblookahead_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p "$HERE/_obj"
pids=()
for f in la_kernels.hip la_mblock.hip la_trie_dev.hip la_engine.cpp la_abi.cpp la_trie.cpp la_comm.cpp; do
  o="$HERE/_obj/${f%.*}.o"
  if [ ! -f "$o" ] || [ "$HERE/$f" -nt "$o" ] || [ "$HERE/la_common.h" -nt "$o" ] || [ "$HERE/la_kernels.h" -nt "$o" ] || [ "$HERE/la_mblock.h" -nt "$o" ] \
     || [ "$HERE/../../include/lookahead_hip.h" -nt "$o" ]; then
    ( $HIPCC $FLAGS -x hip -c "$HERE/$f" -o "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/_obj/la_kernels.o "$HERE"/_obj/la_mblock.o "$HERE"/_obj/la_trie_dev.o "$HERE"/_obj/la_engine.o "$HERE"/_obj/la_abi.o "$HERE"/_obj/la_trie.o "$HERE"/_obj/la_comm.o -ldl
echo "built $OUT"
