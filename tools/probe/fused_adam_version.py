import torch
p = torch.nn.Parameter(torch.randn(64, 64, device='cuda'))
for fused in (True, False):
    opt = torch.optim.Adam([p], lr=1e-3, fused=fused)
    p.grad = torch.randn_like(p)
    v0 = p._version; d0 = p.detach().clone()
    opt.step()
    print('fused=%s: _version %d -> %d, values changed: %s' % (fused, v0, p._version, bool((p.detach() != d0).any())))
