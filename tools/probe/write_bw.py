import torch, time
dev = torch.device('cuda:0')
for mb in (256, 840, 2048):
    x = torch.empty(mb * 1024 * 1024 // 4, device=dev)
    y = torch.empty_like(x)
    for name, fn, nb in (('fill', lambda: x.fill_(1.0), 1), ('copy', lambda: y.copy_(x), 2), ('mul (r+w)', lambda: torch.mul(x, 2.0, out=y), 2)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%5d MB %-10s %8.1f us  %6.2f TB/s (bytes moved: %dx)' % (mb, name, us, nb * mb * 1.048576 / us, nb))
