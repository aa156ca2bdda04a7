#!/bin/bash
# round 6, session 3, call 19: the SQ decomposition of GPU call 13 repeated at HEAD — k_gemm_fatd (gate/up with direct weights) beside the unchanged fat slab / QKV launches
# (2) the secondary legs' kernel stats + FETCH_SIZE / WRITE_SIZE passes at HEAD defaults (scripts/gpu_prof_secondary.sh -> pmc_secondary.json)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out; RAW=/tmp/la_fatpmc
rm -rf $RAW; mkdir -p $OUT $RAW
CMD="cd $REPO && BENCH_IS_SECONDARY=1 python bench.py --model mistral --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --profile-iters 1"
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ( cd /tmp && timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $RAW/p$i -o run -- bash -c "$CMD" > $OUT/r6b19_fatd_pmc_$i.log 2>&1 )
  tail -1 $OUT/r6b19_fatd_pmc_$i.log | cut -c1-160
done
python - <<'PY'
import csv, glob, collections
acc = collections.OrderedDict(); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in sorted(glob.glob('/tmp/la_fatpmc/**/*counter_collection*.csv', recursive=True)):
    p = f.split('/')[3]
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if not ('k_gemm_fat' in n or 'k_gemm_wide' in n or 'k_tree_attn_mb' in n or 'k_row_norm_mb' in n): continue
        k = (p, n[:44])
        acc.setdefault(k, collections.defaultdict(float))[r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for f in sorted(glob.glob('/tmp/la_fatpmc/**/*kernel_trace*.csv', recursive=True)):
    p = f.split('/')[3]
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_gemm_fat' in n or 'k_gemm_wide' in n or 'k_tree_attn_mb' in n or 'k_row_norm_mb' in n:
            dur[(p, n[:44])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open('gpurun_out/r6b19_fatd_pmc.txt', 'w') as fo:
    for k, v in acc.items():
        d = dur.get(k, [0.0]); d = sum(d) / len(d)
        line = f'{k[0]} {k[1]:46s} avg_dur={d:8.2f} us  ' + ' '.join(f'{a}={b / cnt[(k, a)]:.5g}' for a, b in v.items())
        print(line); fo.write(line + '\n')
PY
