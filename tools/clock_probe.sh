#!/bin/bash
# engine clock and package power while the bench step runs vs idle vs a back-to-back library-sized GEMM loop
# usage (GPU box): tools/clock_probe.sh > gpurun_out/<tag>_clocks.log
sample() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done; }
echo "== idle"; sample 3
python bench.py --steps 200 --warmup 4 --no-cpu-baseline --no-b16 --no-kernel-events > /tmp/bench_clk.out 2>/dev/null &
pid=$!
sleep 20
echo "== during the training step (B = 128)"; sample 12
wait $pid
tail -1 /tmp/bench_clk.out | cut -c1-220
python - <<'P' &
import torch, time
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(50): a @ b
    torch.cuda.synchronize()
P
pid=$!
sleep 5
echo "== during a back-to-back 8192^3 bf16 library GEMM loop"; sample 8
wait $pid
