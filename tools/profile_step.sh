#!/bin/bash
# rocprofv3 kernel trace of the default bench workload (or "$@" extra bench flags) -> gpurun_out/<tag>_kernel_stats.txt
# usage (on the GPU box, from the repo root): tools/profile_step.sh <tag> [bench flags]
tag=$1; shift
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out -- python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-b16 --no-kernel-events "$@" > $out/bench.out 2> $out/bench.err)
db=$(find $out -name '*.db' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-b16 --no-kernel-events $@   (6 steps in the trace) at revision ${NSP_REV:-$(cat $root/.git_rev 2>/dev/null)}"
  echo "# per-kernel durations from the rocpd database (tools/rocpd_stats.py); divide totals by 6 for one step"
  python $root/tools/rocpd_stats.py $db 70
} > $root/gpurun_out/${tag}_kernel_stats.txt
python $root/tools/trace_overlap.py $db > $root/gpurun_out/${tag}_overlap.txt 2>&1
tail -1 $out/bench.out | cut -c1-400
python $root/tools/rocpd_lstm_shadow.py $db > $root/gpurun_out/${tag}_lstm_shadow.txt 2>&1
[ -n "$AROUND" ] && python $root/tools/rocpd_around.py $db "$AROUND" > $root/gpurun_out/${tag}_around.txt 2>&1
rm -rf $out   # the database is tens of MB
