#!/bin/bash
# round 5, call 1: whole GPU suite + smoke at the round-5 head, A/B of the attention's early first-tile request (la_lab_set 18 bit 0 =
# the round-4 order), A/B of the overlapped batch trie update (mstep_async) at Mistral bs=8, the harness table at the 7B shape
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --durations=8 -x > $OUT/r5_pytest_full.log 2>&1
echo "pytest exit $?" >> $OUT/r5_pytest_full.log
tail -25 $OUT/r5_pytest_full.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5_smoke.log 2>&1; echo "smoke exit $?" >> $OUT/r5_smoke.log; tail -3 $OUT/r5_smoke.log | cut -c1-300
B1="python bench.py --steps 64 --warmup 8 --no-cpu-baseline --secondary \"\""
for tag in spec_a nospec_a spec_b nospec_b; do
  case $tag in nospec*) export LA_DEBUG="18=1";; *) unset LA_DEBUG;; esac
  timeout 400 bash -c "$B1" > $OUT/r5_ab_$tag.json 2> $OUT/r5_ab_$tag.err
done
unset LA_DEBUG
for tag in overlap_a strict_a overlap_b strict_b; do
  case $tag in strict*) X="--strict-trie-order";; *) X="";; esac
  timeout 400 python bench.py --model mistral --batch 8 --steps 24 --warmup 4 --no-cpu-baseline $X > $OUT/r5_mistral_$tag.json 2> $OUT/r5_mistral_$tag.err
done
timeout 600 python scripts/bench_harness.py --queries 8 --table $OUT/r5_harness_table.md > $OUT/r5_harness.log 2>&1; echo "harness exit $?" >> $OUT/r5_harness.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r5_ab_*.json')) + sorted(glob.glob('gpurun_out/r5_mistral_*.json')):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            vs = d['roofline'].get('verify_step', {})
            print(f.split('/')[-1], d['value'], d['ms_per_step'], 'accept', d['config']['mean_accept_len'], 'eq', d['config']['lookahead_equals_greedy'],
                  'attn_ms', vs.get('ms_by_class_events', {}).get('attn'), 'frac', vs.get('frac'), d['config'].get('trie_update'))
PY
tail -5 $OUT/r5_harness.log | cut -c1-400
cat $OUT/r5_harness_table.md 2>/dev/null | cut -c1-200
