#!/bin/bash
# round 5, third final record: whole GPU suite + smoke at HEAD (k_gemm_fat with the re-mappable workgroup ids), then FETCH_SIZE / WRITE_SIZE of ONE
# launch per form of scripts/gpu_mb_gemm.py at the Mistral shape (wide, fat, fat + XCD K map; gate/up and down; 256 and 512 rows) — the counter check of
# the "W + 8 x x + outputs" traffic model and of what the XCD mapping takes off it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 --durations=3 > $OUT/r5_pytest_final3.log 2>&1
echo "pytest exit $?" >> $OUT/r5_pytest_final3.log
tail -7 $OUT/r5_pytest_final3.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r5_smoke3.log 2>&1; echo "smoke exit $?" >> $OUT/r5_smoke3.log; tail -2 $OUT/r5_smoke3.log | cut -c1-300
for C in FETCH_SIZE WRITE_SIZE; do
  RAW=/tmp/la_f3_$C; rm -rf $RAW; mkdir -p $RAW
  ( cd /tmp && MB_F=14336 MB_K=4096 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $RAW -o run -- python $REPO/scripts/gpu_mb_gemm.py once > $OUT/r5_f3_$C.log 2>&1 )
done
python - <<'PY'
import csv, glob
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = []
    for f in glob.glob('/tmp/la_f3_%s/**/*counter_collection*.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == c and 'k_gemm' in r['Kernel_Name']:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'][:60], float(r['Counter_Value'])))
    rows.sort()
    res[c] = rows
with open('gpurun_out/r5_f3_traffic.txt', 'w') as o:
    o.write('# one launch per form, dispatch order = rows {256, 512} x form {wide, fat, fat + XCD K map} x {gate/up, down}; Mistral shape (F 14336, K 4096)\n')
    o.write('# read MB = 2 x FETCH_SIZE x 1024 / 1e6 (gfx950: 64 B counted per 128-B request), write MB = WRITE_SIZE x 1024 / 1e6\n')
    for (d, k, f), (_, _, w) in zip(res['FETCH_SIZE'], res['WRITE_SIZE']):
        line = '%5d %-60s read %8.1f MB  write %7.1f MB' % (d, k, 2 * f * 1024 / 1e6, w * 1024 / 1e6)
        o.write(line + '\n'); print(line)
PY
