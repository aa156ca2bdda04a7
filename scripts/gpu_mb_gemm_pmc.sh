#!/bin/bash
# counters of the multi-block GEMMs (scripts/gpu_mb_gemm.py once): L2 hit rate, LDS, MFMA busy; summaries -> gpurun_out/mb_gemm_pmc.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
RAW=/tmp/la_mbpmc
rm -rf $RAW; mkdir -p $OUT $RAW
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $RAW/p$i -o run -- bash -c "cd $REPO && python scripts/gpu_mb_gemm.py once" > $OUT/mb_gemm_pmc_$i.log 2>&1 )
done
python - <<'PY'
import csv, glob, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob('/tmp/la_mbpmc/**/*counter_collection*.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_gemm' not in n: continue
        key = (n[:40], r.get('Grid_Size', ''), r['Dispatch_Id'])
        acc.setdefault((f.split('/')[3], n[:44], int(r['Dispatch_Id'])), {})[r['Counter_Name']] = float(r['Counter_Value'])
with open('gpurun_out/mb_gemm_pmc.txt', 'w') as fo:
    for k, v in acc.items():
        line = f'{k[0]} {k[1]:46s} d{k[2]:<4d} ' + ' '.join(f'{a}={b:.4g}' for a, b in v.items())
        print(line); fo.write(line + '\n')
PY
