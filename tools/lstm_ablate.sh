#!/bin/bash
# forward-kernel ablations of the persistent LSTM (tools/probe/ab/libnsp_hip_lstmabl<bits>.so built with
# tools/make_variant_lib.sh lstmabl<bits> lstm.hip -DNSP_LSTM_ABL=<bits>):
#   1 no operand loads, 2 no MFMAs (and so no loads), 4 no state stores, 8 no grid barrier
for b in ${BATCHES:-64 16}; do
  echo "B=$b product: $(LB=$b python tools/lstm_stack_bench.py 2>&1 | grep persistent=1)"
  for v in "$@"; do
    echo "B=$b abl=$v: $(LB=$b NSP_LIB_OVERRIDE=tools/probe/ab/libnsp_hip_lstmabl$v.so python tools/lstm_stack_bench.py 2>&1 | grep persistent=1)"
  done
done
