#!/bin/bash
# per-launch means of a few hardware counters for the two GEMMs of tools/gemm_pmc.py (launch order: 12 x FFN1, 12 x FFN2)
root=$(pwd); export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"; do
  # (a pass with the TCP_*_LATENCY_sum / TCP_UTCL1_* counters aborted rocprofv3 with SIGABRT on this image and hung
  #  until the gpurun limit: wrap every pass in `timeout 120` before adding counters)
  out=$root/gpurun_out/pmc_gemm_x; rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python $root/tools/gemm_pmc.py > $out/run.out 2>&1)
  f=$(find $out -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "== $c : no output"; tail -3 $out/run.out; continue; fi
  python - "$f" <<'P'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'kk_glds' in r['Kernel_Name']]
ids = sorted({int(r['Dispatch_Id']) for r in rows})
first = set(ids[:len(ids) // 2])
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = ('FFN1 K=512 N=2048' if int(r['Dispatch_Id']) in first else 'FFN2 K=2048 N=512', r['Counter_Name'])
    acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items()):
    print('%-18s %-34s mean per launch %.5g' % (k[0], k[1], v / n))
P
  rm -rf $out
done
