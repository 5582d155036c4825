#!/bin/bash
# round 6, closing sequence at the shipping commit (one gpurun call): the whole device suite, the default bench line, kernel stats of the same
# command, the PMC traffic passes, the 16-utterance kernel stats and the host profile.  Everything lands under gpurun_out/r06final/.
root=$(pwd); out=$root/gpurun_out/r06final; mkdir -p $out; export TMPDIR=/tmp
echo "revision $(cat .git_rev)" | tee $out/revision.txt
python -m pytest tests -m gpu -q > $out/pytest_gpu_full.log 2>&1; tail -3 $out/pytest_gpu_full.log
python bench.py > $out/bench_line.json 2> $out/bench.err || tail -5 $out/bench.err
tail -c 600 $out/bench_line.json | head -c 400; echo
bash tools/r06_gpu_check.sh r06final_check "" stats > $out/gpu_check.txt 2>&1; head -3 $out/gpu_check.txt
cp $root/gpurun_out/r06final_check/kernel_stats.txt $out/kernel_stats.txt 2>/dev/null
NSP_REV=$(cat .git_rev) bash tools/profile_pmc_traffic.sh r06final > $out/pmc.txt 2>&1; tail -3 $out/pmc.txt
cp $root/gpurun_out/r06final_pmc_hbm_traffic.txt $root/gpurun_out/r06final_pmc_gemm_traffic.json $out/ 2>/dev/null
bash tools/r06_b16_stats.sh > $out/b16.txt 2>&1; head -3 $out/b16.txt; cp $root/gpurun_out/r06b16/kernel_stats_b16.txt $out/ 2>/dev/null
python tools/host_profile_step.py > $out/host_profile.log 2>&1; tail -5 $out/host_profile.log
