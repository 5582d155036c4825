#!/bin/bash
# HBM traffic of the bench step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; kernel trace only, as the guide
# prescribes) -> gpurun_out/<tag>_pmc_hbm_traffic.txt and profiles-ready pmc_gemm_traffic.json
# usage (GPU box, repo root): tools/profile_pmc_traffic.sh <tag>
tag=$1
root=$(pwd)
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_$c
  rm -rf $out; mkdir -p $out
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-b16 > $out/bench.out 2> $out/bench.err)
done
# third pass, no profiler: the algorithmic bytes of every GEMM launch of the same command
NSP_GEMM_DEBUG=1 python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-b16 > /dev/null 2> /tmp/gemm_debug_$tag.err
grep nsp_gemm_bf16 /tmp/gemm_debug_$tag.err > /tmp/gemm_shapes_$tag.txt
f=$(find $root/gpurun_out/pmc_${tag}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find $root/gpurun_out/pmc_${tag}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
{
  echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-b16  (per-GPU batch 128; 2 steps) at revision ${NSP_REV:-$(cat $root/.git_rev 2>/dev/null)}"
  echo "# aggregated by tools/pmc_traffic.py: FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md, HBM section); bytes per launch"
  python $root/tools/pmc_traffic.py $f $w $root/gpurun_out/${tag}_pmc_gemm_traffic.json "${NSP_REV:-$(cat $root/.git_rev 2>/dev/null)}" 128 /tmp/gemm_shapes_$tag.txt
} > $root/gpurun_out/${tag}_pmc_hbm_traffic.txt 2>&1
rm -rf $root/gpurun_out/pmc_${tag}_FETCH_SIZE $root/gpurun_out/pmc_${tag}_WRITE_SIZE
head -14 $root/gpurun_out/${tag}_pmc_hbm_traffic.txt | cut -c1-140
