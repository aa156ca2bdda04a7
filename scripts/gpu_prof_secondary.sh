#!/bin/bash
# The batch configurations (secondary legs of bench.py) under rocprofv3: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own passes.
# Output: gpurun_out/kernel_stats_<model>_b<B>.txt (per-kernel durations, bytes per launch, TB/s) and gpurun_out/pmc_secondary.json
# (HBM bytes of ONE steady verify step per configuration = the `traffic` figure of the secondary roofline objects of bench.py).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
STEPS=${STEPS:-12}
for leg in "mistral 8" "13b 4" "mixtral 4"; do
  set -- $leg
  RAW=/tmp/la_sec_$1; rm -rf $RAW; mkdir -p $RAW
  CMD="cd $REPO && BENCH_IS_SECONDARY=1 python bench.py --model $1 --batch $2 --steps $STEPS --warmup 2 --no-cpu-baseline --profile-iters 1"
  ( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o run -- bash -c "$CMD" > $OUT/sec_$1_stats.log 2>&1 )
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $RAW/pmc_$C -o run -- bash -c "$CMD" > $OUT/sec_$1_$C.log 2>&1 )
  done
done
python - "$STEPS" <<'PY'
import collections, csv, glob, json, os, sys
steps = int(sys.argv[1])
out = {'source': 'scripts/gpu_prof_secondary.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of '
                 'bench.py --model M --batch B --steps %d --warmup 2 (BENCH_IS_SECONDARY=1)' % steps,
       'correction': 'gfx950: read bytes = 2 x FETCH_SIZE x 1024 (64 B counted per 128-B request), write bytes = WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM)',
       'step_rule': 'a verify step = the dispatches from one k_build_inputs_mb to the next; median over the last %d steps of the run' % steps,
       'configs': {}}
for model, B in (('mistral', 8), ('13b', 4), ('mixtral', 4)):
    raw = '/tmp/la_sec_%s' % model
    kern = collections.OrderedDict()
    for f in glob.glob(os.path.join(raw, 'stats', '**', '*kernel_stats*.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Name'].startswith(('k_', 'void k_')):
                kern[r['Name'][:70]] = {'dispatches': int(r['Calls']), 'avg_us': round(float(r['AverageNs']) / 1e3, 2), 'total_ms': round(int(r['TotalDurationNs']) / 1e6, 2)}
    per_step = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        rows = []
        for f in glob.glob(os.path.join(raw, 'pmc_' + c, '**', '*counter_collection*.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get('Counter_Name') == c:
                    rows.append((int(r['Dispatch_Id']), r['Kernel_Name'][:70], float(r['Counter_Value'])))
        rows.sort()
        agg = collections.defaultdict(lambda: [0, 0.0])
        for _, k, v in rows:
            agg[k][0] += 1; agg[k][1] += v
        for k, (n, v) in agg.items():
            if k in kern:
                kern[k]['fetch_size_kb' if c == 'FETCH_SIZE' else 'write_size_kb'] = round(v / n, 1)
        starts = [i for i, r in enumerate(rows) if r[1].startswith('k_build_inputs_mb')]
        segs = [sum(v for _, _, v in rows[a:b]) for a, b in zip(starts, starts[1:] + [len(rows)])]
        last = sorted(segs[-steps:]) if len(segs) >= steps else sorted(segs)
        per_step[c] = last[len(last) // 2] if last else None
    for k, d in kern.items():
        if 'fetch_size_kb' in d:
            d['hbm_bytes_per_launch'] = int(2 * d['fetch_size_kb'] * 1024 + d.get('write_size_kb', 0.0) * 1024)
            d['TBps'] = round(d['hbm_bytes_per_launch'] / (d['avg_us'] * 1e-6) / 1e12, 2) if d['avg_us'] > 0 else None
    traffic = None
    if per_step.get('FETCH_SIZE') is not None and per_step.get('WRITE_SIZE') is not None:
        traffic = int(2 * per_step['FETCH_SIZE'] * 1024 + per_step['WRITE_SIZE'] * 1024)
    out['configs']['%s_b%d' % (model, B)] = {'hbm_bytes_per_step': traffic, 'fetch_size_kb_per_step': per_step.get('FETCH_SIZE'),
                                             'write_size_kb_per_step': per_step.get('WRITE_SIZE'), 'kernels': kern}
    with open('gpurun_out/kernel_stats_%s_b%d.txt' % (model, B), 'w') as fo:
        for k, d in sorted(kern.items(), key=lambda kv: -kv[1]['total_ms']):
            fo.write(f"{k:72s} calls {d['dispatches']:6d} avg {d['avg_us']:8.2f} us total {d['total_ms']:9.2f} ms  bytes/launch {d.get('hbm_bytes_per_launch', 0) / 1e6:9.2f} MB  {d.get('TBps') or 0:5.2f} TB/s\n")
    print(model, B, 'HBM bytes per step', traffic)
json.dump(out, open('gpurun_out/pmc_secondary.json', 'w'), indent=1)
PY
head -12 $OUT/kernel_stats_mistral_b8.txt | cut -c1-200
head -12 $OUT/kernel_stats_13b_b4.txt | cut -c1-200
tail -3 $OUT/sec_mistral_stats.log | cut -c1-200
