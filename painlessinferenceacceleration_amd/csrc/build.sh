#!/bin/bash
# Build liblookahead_hip.so (bf16) and liblookahead_hip_f16.so (fp16: the same sources with -DLA_DTYPE=1) for gfx950 (MI355X) in-tree — the
# PRODUCT libraries: every lab knob a constexpr default (la_knobs.h), no la_lab_* entry point — and the LAB builds of the same sources
# (-DLA_LAB=1 + la_lab.cpp: liblookahead_hip_lab.so, liblookahead_hip_lab_f16.so) that the A/B scripts and the variant tests load.
# hipcc cross-compiles without a GPU.  build.sh [out.so] builds the bf16 product library only when an output path is given;
# LA_SKIP_LAB=1 skips the lab builds.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
BASEFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 ${LA_EXTRA_HIPCC_FLAGS:-}"
SRCS_PRODUCT="la_kernels.hip la_attn1.hip la_mblock.hip la_trie_dev.hip la_trie_wg.hip la_engine.cpp la_abi.cpp la_trie.cpp la_comm.cpp"
SRCS="$SRCS_PRODUCT"
build_one() {     # out.so, object dir, extra flags, extra sources
  local OUT="$1" OBJ="$2" FLAGS="$BASEFLAGS $3" SRCS="$SRCS_PRODUCT ${4:-}"
  mkdir -p "$OBJ"
  local pids=() objs=()
  for f in $SRCS; do
    local b="${f##*/}"; local o="$OBJ/${b%.*}.o"
    objs+=("$o")
    if [ ! -f "$o" ] || [ "$HERE/$f" -nt "$o" ] || [ "$HERE/la_common.h" -nt "$o" ] || [ "$HERE/la_kernels.h" -nt "$o" ] || [ "$HERE/la_mblock.h" -nt "$o" ] || [ "$HERE/la_trie_dev.h" -nt "$o" ] || [ "$HERE/la_knobs.h" -nt "$o" ] || [ -n "$(find "$HERE/lab" -newer "$o" -name '*.inc' 2>/dev/null | head -1)" ] \
       || [ "$HERE/../../include/lookahead_hip.h" -nt "$o" ] || [ "$HERE/../../include/lookahead_hip_lab.h" -nt "$o" ] || [ "$HERE/build.sh" -nt "$o" ]; then
      ( $HIPCC $FLAGS -x hip -c "$HERE/$f" -o "$o" ) &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o "$OUT" "${objs[@]}" -ldl
  echo "built $OUT"
}
if [ $# -ge 1 ]; then
  build_one "$1" "${LA_OBJ_DIR:-$HERE/_obj}" ""
else
  build_one "$HERE/../liblookahead_hip.so" "${LA_OBJ_DIR:-$HERE/_obj}" "" &
  b1=$!
  build_one "$HERE/../liblookahead_hip_f16.so" "${LA_OBJ_DIR_F16:-$HERE/_obj_f16}" "-DLA_DTYPE=1" &
  b2=$!
  wait $b1; wait $b2
  if [ -z "${LA_SKIP_LAB:-}" ]; then
    build_one "$HERE/../liblookahead_hip_lab.so" "$HERE/_obj_lab" "-DLA_LAB=1" "la_lab.cpp lab/la_oproj_merge.hip" &
    b3=$!
    build_one "$HERE/../liblookahead_hip_lab_f16.so" "$HERE/_obj_lab_f16" "-DLA_LAB=1 -DLA_DTYPE=1" "la_lab.cpp lab/la_oproj_merge.hip" &
    b4=$!
    wait $b3; wait $b4
  fi
fi
