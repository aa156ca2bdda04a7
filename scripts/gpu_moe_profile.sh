#!/bin/bash
# rocprofv3 kernel stats of the Mixtral-shaped step (8 layers, full expert sizes): per-kernel time of the merged expert launches
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
rm -rf /tmp/la_moe; mkdir -p /tmp/la_moe gpurun_out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/la_moe -o run -- bash -c "cd $REPO && python scripts/bench_moe.py --layers 8 --steps 10" > $REPO/gpurun_out/moe_prof.log 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/la_moe/**/*kernel_stats*.csv', recursive=True):
    with open('gpurun_out/moe_kernel_stats.txt', 'w') as fo:
        for r in csv.DictReader(open(f)):
            if r['Name'].startswith(('k_', 'void k_')) and 'pack' not in r['Name']:
                line = f"{r['Name'][:60]:62s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.2f} us total {int(r['TotalDurationNs']) / 1e6:8.2f} ms"
                print(line); fo.write(line + '\n')
PY
grep "^{" gpurun_out/moe_prof.log | cut -c1-300
