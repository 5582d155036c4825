for lw in 0 1; do for b in 64 128; do echo "layerwise=$lw B=$b: $(NSP_LSTM_LAYERWISE=$lw LB=$b python tools/lstm_stack_bench.py 2>&1 | grep persistent=1)"; done; done
timeout 300 python -m pytest tests/test_kernels_conv_loss_gpu.py tests/test_fullsize_parity_gpu.py -q -k "lstm" -x 2>&1 | tail -2
for lw in 0 1; do echo "bench layerwise=$lw: $(NSP_LSTM_LAYERWISE=$lw python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-events --no-b16 2>/dev/null | tail -1 | cut -c1-260)"; done
