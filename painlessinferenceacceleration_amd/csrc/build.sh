#!/bin/bash
# Build liblookahead_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/../liblookahead_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 ${LA_EXTRA_HIPCC_FLAGS:-}"
OBJ="${LA_OBJ_DIR:-$HERE/_obj}"
mkdir -p "$OBJ"
pids=()
for f in la_kernels.hip la_attn1.hip la_mblock.hip la_trie_dev.hip la_engine.cpp la_abi.cpp la_lab.cpp la_trie.cpp la_comm.cpp; do
  o="$OBJ/${f%.*}.o"
  if [ ! -f "$o" ] || [ "$HERE/$f" -nt "$o" ] || [ "$HERE/la_common.h" -nt "$o" ] || [ "$HERE/la_kernels.h" -nt "$o" ] || [ "$HERE/la_mblock.h" -nt "$o" ] \
     || [ "$HERE/../../include/lookahead_hip.h" -nt "$o" ] || [ "$HERE/../../include/lookahead_hip_lab.h" -nt "$o" ]; then
    ( $HIPCC $FLAGS -x hip -c "$HERE/$f" -o "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/la_kernels.o "$OBJ"/la_attn1.o "$OBJ"/la_mblock.o "$OBJ"/la_trie_dev.o "$OBJ"/la_engine.o "$OBJ"/la_abi.o "$OBJ"/la_lab.o "$OBJ"/la_trie.o "$OBJ"/la_comm.o -ldl
echo "built $OUT"
