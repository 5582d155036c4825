#!/bin/bash
# round 5: the shipping revision on a fresh box -- device suite, PMC traffic passes, kernel trace, bench line
# usage: tools/r05_final.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-r05z}
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${tag}_pytest_gpu_full.log 2>&1
echo "rc=$?" >> gpurun_out/${tag}_pytest_gpu_full.log
tail -3 gpurun_out/${tag}_pytest_gpu_full.log
bash tools/profile_pmc_traffic.sh $tag > gpurun_out/${tag}_pmc_head.txt 2>&1
[ -s gpurun_out/${tag}_pmc_gemm_traffic.json ] && cp gpurun_out/${tag}_pmc_gemm_traffic.json profiles/pmc_gemm_traffic.json
bash tools/profile_step.sh $tag > gpurun_out/${tag}_profile_step.txt 2>&1
timeout 900 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_stderr.log
echo "bench rc=$?"
python - <<'PY'
import json,sys
tag=sys.argv[1] if len(sys.argv)>1 else 'r05z'
PY
tail -c 1500 gpurun_out/${tag}_bench_line.json | head -c 600
