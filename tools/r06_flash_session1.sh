#!/bin/bash
# round 6, GPU session 1: micro-probes, flash-attention baseline vs the batched-asm-read dK/dV kernel, its ablations, PMC
root=$(pwd); export TMPDIR=/tmp
out=$root/gpurun_out/r06s1; mkdir -p $out
./tools/probe/r06_probes > $out/probes.log 2>&1
python tools/flash_bench.py 128 10 > $out/flash_tree.log 2>&1
NSP_LIB_OVERRIDE=$root/tools/variants/libnsp_hip_v1.so python tools/flash_bench.py 128 10 > $out/flash_v1.log 2>&1
kstats() {  # $1 = tag, $2 = lib ('' = tree)
  d=$out/prof_$1; rm -rf $d
  (cd /tmp && NSP_LIB_OVERRIDE=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $root/tools/flash_bench.py 128 6 800 > $d.out 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $1" >> $out/kernel_split.log
  if [ -n "$f" ]; then python - "$f" >> $out/kernel_split.log <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'flash' in n:
        print('%-40s calls %4s avg %9.1f us' % (n.split('(')[0][-40:], r['Calls'], float(r['AverageNs']) / 1e3))
P
  else tail -3 $d.out >> $out/kernel_split.log; fi
  rm -rf $d
}
kstats tree ""
kstats v1 $root/tools/variants/libnsp_hip_v1.so
for a in 1 2 4 8 3 7 15; do kstats v1abl$a $root/tools/variants/libnsp_hip_v1abl$a.so; done
# PMC: wave-level stall accounting of the three kernels (tree build), T = 800 only
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$out/pmc_x; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python $root/tools/flash_bench.py 128 3 800 > $d.out 2>&1)
  f=$(find $d -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "== $c : no output" >> $out/pmc.log; tail -3 $d.out >> $out/pmc.log; continue; fi
  python - "$f" >> $out/pmc.log <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: [set(), 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'flash' not in n: continue
    k = ('fwd' if 'fwd' in n else 'dkv' if 'dkv' in n else 'dq', r['Counter_Name'])
    acc[k][0].add(r['Dispatch_Id']); acc[k][1] += float(r['Counter_Value'])
for k, (ids, v) in sorted(acc.items()):
    print('%-4s %-28s mean per launch %.6g' % (k[0], k[1], v / max(1, len(ids))))
P
  rm -rf $d
done
echo done > $out/done
