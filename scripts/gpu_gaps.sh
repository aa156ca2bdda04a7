#!/bin/bash
# where does a verify step spend time that is not inside a kernel?  rocprofv3 kernel trace of a short bench run,
# per-step span vs sum of kernel durations, and the largest inter-kernel gaps by (previous kernel -> next kernel).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
rm -rf /tmp/la_gaps; mkdir -p /tmp/la_gaps gpurun_out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/la_gaps -o run -- bash -c "cd $REPO && python bench.py --steps 12 --warmup 2 --no-cpu-baseline --profile-iters 1" > $REPO/gpurun_out/gaps_run.log 2>&1 )
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/la_gaps/**/*kernel_trace*.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]) for r in csv.DictReader(open(f))]
rows.sort()
steps, cur = [], []
for r in rows:
    if r[2].startswith('k_build_tree_inputs') and cur:
        steps.append(cur); cur = []
    cur.append(r)
steps.append(cur)
full = [s for s in steps if 250 < len(s) < 280 and any(k[2].startswith('k_accept_scan') for k in s)]
print('steps with a full kernel list:', len(full))
gaps = collections.defaultdict(lambda: [0, 0.0])
for s in full[-8:]:
    span = (s[-1][1] - s[0][0]) / 1e3
    busy = sum(e - b for b, e, _ in s) / 1e3
    print(f'step: {len(s)} kernels, span {span:.1f} us, in kernels {busy:.1f} us, gaps {span - busy:.1f} us')
    for a, b in zip(s, s[1:]):
        g = gaps[(a[2][:28], b[2][:28])]
        g[0] += 1; g[1] += (b[0] - a[1]) / 1e3
print('mean gap by kernel pair (us):')
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f'  {k[0]:30s} -> {k[1]:30s} n={n:4d} mean {t / n:6.2f} total/step {t / 8:7.1f}')
# time between the end of one step and the start of the next (host turnaround)
turn = [(b[0][0] - a[-1][1]) / 1e3 for a, b in zip(full, full[1:])]
print('step-to-step turnaround us (end of kv_commit -> next build_tree_inputs):', [round(x, 1) for x in turn[-8:]])
PY
