#!/bin/bash
# A/B of differently built liblookahead_hip.so files (variants/*.so, built here with LA_EXTRA_HIPCC_FLAGS / source switches):
# each one is copied over the in-tree library and timed with the default bench leg (no CPU baseline, no secondary lines).
# usage: gpu_ab_variants.sh name1 name2 ...   ("base" = the in-tree build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
LIB=painlessinferenceacceleration_amd/liblookahead_hip.so
cp $LIB /tmp/lib_base.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/lib_base.so $LIB; else cp variants/lib_$v.so $LIB; fi
  timeout 300 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --secondary "" --profile-iters 1 ${BENCH_ARGS:-} > $OUT/ab_${v}_$rep.json 2> $OUT/ab_${v}_$rep.err
  python - "$v" "$rep" <<'PY'
import json, sys
v, rep = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f'gpurun_out/ab_{v}_{rep}.json'))
    print(f"{v:16s} rep {rep}: {d['ms_per_step']:.4f} ms/step  {d['value']:.1f} tok/s  equal_greedy={d['config'].get('lookahead_equals_greedy')}")
except Exception as e:
    print(v, rep, 'FAILED', e)
PY
done; done | tee -a $OUT/ab_summary.txt
cp /tmp/lib_base.so $LIB
