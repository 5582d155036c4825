import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
torch.zeros(1, device='cuda')
hip = ctypes.CDLL('libamdhip64.so.7') if False else ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
lib = ctypes.CDLL(os.path.join(here, 'tr_probe.so'))
class dim3(ctypes.Structure):
    _fields_ = [('x', ctypes.c_uint), ('y', ctypes.c_uint), ('z', ctypes.c_uint)]
hip.hipLaunchKernel.argtypes = [ctypes.c_void_p, dim3, dim3, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
def launch(fn, args, block):
    arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    rc = hip.hipLaunchKernel(ctypes.cast(fn, ctypes.c_void_p), dim3(1, 1, 1), dim3(block, 1, 1), arr, 0, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
for mode in (0, 1, 2):
    out = torch.zeros(256, dtype=torch.int16, device='cuda')
    launch(lib.tr_probe, [ctypes.c_void_p(out.data_ptr()), ctypes.c_int(mode)], 64)
    o = out.cpu().view(64, 4).tolist()
    print('mode', mode)
    for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 32, 48, 63):
        print('  lane %2d ->' % l, o[l])
src = torch.arange(1024, dtype=torch.int32, device='cuda')
out = torch.zeros(1024, dtype=torch.int32, device='cuda')
launch(lib.glds_probe, [ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr())], 64)
o = out.cpu().tolist()
print('glds: lds[0:8] =', o[0:8], ' lds[252:256] =', o[252:256], 'lds[256:260]=', [hex(x & 0xffffffff) for x in o[256:260]])
