"""Where does the runtime put pinned host memory, and what device address does it get?  (the NaN flags of the loss live there)"""
import ctypes as C, io, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from pqdet_amd import abi
l = abi.lib()
torch.cuda.init()
x = torch.zeros(4, device='cuda')
for n in (3, 3, 1024, 1 << 20):
    h = torch.zeros(n, dtype=torch.int32).pin_memory()
    dp = C.c_void_p()
    rc = l.pq_host_device_pointer(C.c_void_p(h.data_ptr()), C.byref(dp))
    print('pinned %8d ints: host 0x%x  device 0x%x  rc %d' % (n, h.data_ptr(), dp.value or 0, rc))
print('device tensor 0x%x' % x.data_ptr())
