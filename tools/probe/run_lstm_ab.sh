# usage: run_lstm_ab.sh <variant names...> : timing at B=64 (+ the LSTM tests on the last variant)
for v in "$@"; do echo "$v: $(LB=${LB:-64} NSP_LIB_OVERRIDE=tools/probe/ab/libnsp_hip_lstm$v.so python tools/lstm_stack_bench.py 2>&1 | grep persistent=1)"; done
NSP_LIB_OVERRIDE=tools/probe/ab/libnsp_hip_lstm$v.so timeout 300 python -m pytest tests/test_kernels_conv_loss_gpu.py -q -k "lstm" -x 2>&1 | tail -2
