cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; REPO=$PWD
rm -rf /tmp/pw; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o run -- bash -c "cd $REPO && python bench.py --steps 12 --warmup 2 --no-cpu-baseline --secondary '' --decoding-length 128 --branch-length 32" > /dev/null 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/pw/**/*kernel_stats*.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if r['Name'].startswith(('k_','void k_'))]
    rows.sort(key=lambda r:-int(r['TotalDurationNs']))
    for r in rows[:14]:
        print(f"{r['Name'][:64]:66s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us total {int(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
