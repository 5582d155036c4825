for v in xcd0 xcd1; do echo "$v: $(LB=64 NSP_LIB_OVERRIDE=tools/probe/ab/libnsp_hip_lstm$v.so python tools/lstm_stack_bench.py 2>&1 | grep persistent=1)"; done
NSP_LIB_OVERRIDE=tools/probe/ab/libnsp_hip_lstmxcd1.so timeout 300 python -m pytest tests/test_kernels_conv_loss_gpu.py -q -k "lstm" -x 2>&1 | tail -2
BATCHES=64 tools/lstm_ablate.sh 1 4 8 | grep abl
