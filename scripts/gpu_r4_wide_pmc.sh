#!/bin/bash
# round 4, call 12: where the wave cycles of the 512-row gate/up launch go (SQ wait / active buckets, effective clock), for the full
# kernel and its measurement builds (dbg 1..5); summaries -> gpurun_out/r4_wide_pmc.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
RAW=/tmp/la_wpmc
rm -rf $RAW; mkdir -p $OUT $RAW
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u > $OUT/r4_counter_names.txt )
wc -l $OUT/r4_counter_names.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $RAW/p$i -o run -- bash -c "cd $REPO && python scripts/gpu_mb_gemm.py onceparts" > $OUT/r4_wide_pmc_$i.log 2>&1 )
  tail -2 $OUT/r4_wide_pmc_$i.log | cut -c1-200
done
python - <<'PY'
import csv, glob, collections
acc = collections.OrderedDict()
dur = {}
for f in sorted(glob.glob('/tmp/la_wpmc/**/*counter_collection*.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_gemm_wide' not in n: continue
        acc.setdefault((f.split('/')[3], n[:40], int(r['Dispatch_Id'])), {})[r['Counter_Name']] = float(r['Counter_Value'])
for f in sorted(glob.glob('/tmp/la_wpmc/**/*kernel_trace*.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'k_gemm_wide' in r['Kernel_Name']:
            dur[(f.split('/')[3], int(r['Dispatch_Id']))] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
with open('gpurun_out/r4_wide_pmc.txt', 'w') as fo:
    for k, v in acc.items():
        d = dur.get((k[0], k[2]))
        line = f'{k[0]} {k[1]:42s} d{k[2]:<4d} dur={d} us  ' + ' '.join(f'{a}={b:.5g}' for a, b in v.items())
        print(line); fo.write(line + '\n')
PY
