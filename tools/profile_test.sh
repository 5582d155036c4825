#!/bin/bash
# rocprofv3 kernel trace of one pytest selection -> gpurun_out/<tag>_kernel_stats.txt (all kernels, by total time)
# usage (GPU box, repo root): tools/profile_test.sh <tag> <pytest args...>
tag=$1; shift
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
args=()
for a in "$@"; do case "$a" in tests/*) args+=("$root/$a");; *) args+=("$a");; esac; done
(cd /tmp && PYTHONPATH=$root rocprofv3 --kernel-trace --stats -d $out -- python -m pytest -q -p no:cacheprovider --rootdir $root "${args[@]}" > $out/test.out 2> $out/test.err)
db=$(find $out -name '*.db' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python -m pytest $@   at revision $(cat $root/.git_rev 2>/dev/null)"
  python $root/tools/rocpd_stats.py $db 200
} > $root/gpurun_out/${tag}_kernel_stats.txt
tail -2 $out/test.out
rm -rf $out
