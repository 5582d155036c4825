#!/bin/bash
# round 5: the environment-sensitive device tests N times over (two ranks on one device, second tenant, hostile streams)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
N=${1:-8}
L=gpurun_out/r05_soak.log
: > $L
fails=0
for i in $(seq 1 $N); do
  timeout 600 python -m pytest tests/test_hostile_neighbour_gpu.py tests/test_ddp_hip_gpu.py tests/test_fullsize_parity_gpu.py -q -x -m gpu > gpurun_out/r05_soak_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/r05_soak_$i.log)" >> $L
  if [ $rc -ne 0 ]; then fails=$((fails+1)); else rm -f gpurun_out/r05_soak_$i.log; fi
done
echo "runs $N failures $fails" >> $L
cat $L
