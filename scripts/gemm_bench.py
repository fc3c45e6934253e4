"""GEMM micro-benchmark: time pq_conv1x1_{fwd,dgrad,wgrad} on the layer shapes of mobilenetv2-fpn (HIP events, many reps)."""
import ctypes as C, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi

SHAPES = [(1048576, 16, 96), (1048576, 32, 16), (262144, 24, 144), (262144, 144, 24), (65536, 32, 192), (65536, 192, 32),
          (65536, 128, 256), (65536, 256, 128), (16384, 64, 384), (16384, 384, 64), (16384, 96, 576), (16384, 576, 96),
          (16384, 256, 512), (16384, 512, 256), (4096, 160, 960), (4096, 960, 160), (4096, 320, 1280), (4096, 1280, 512),
          (4096, 512, 1024), (4096, 1024, 512)]
if os.environ.get('NARROW'):
    SHAPES = [(1048576, 16, 96), (1048576, 32, 16), (262144, 24, 144), (262144, 144, 24), (262144, 96, 24), (65536, 32, 192),
              (65536, 192, 32), (65536, 144, 32)]
if os.environ.get('SWEEP'):
    SHAPES = [(m, k, n) for (m, n) in ((4096, 512), (4096, 1024), (16384, 256), (65536, 128), (262144, 128)) for k in (32, 64, 128, 256, 512, 1024, 2048)]
if os.environ.get('SHAPES'):      # "M,I,O;M,I,O;..."
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd,dgrad,wgrad'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
l = abi.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = 'cuda'

def vec(n, fill=None):
    t = torch.rand(n + 32, device=dev) + 0.5 if fill is None else torch.full((n + 32,), fill, device=dev)
    return t

def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

tot = {}
for (M, I, O) in SHAPES:
    x = torch.randn(M, I, device=dev); w = torch.randn(O, I, device=dev) / math.sqrt(I)
    ldo = (O + 3) // 4 * 4
    y = torch.randn(M, ldo, device=dev); g = torch.randn(M, ldo, device=dev); gin = torch.empty(M, I, device=dev)
    sc, sh = vec(I), vec(I)
    p, q, mu, r = vec(O), vec(O), vec(O), vec(O)
    a2, b2, mn, iv = vec(I), vec(I), vec(I), vec(I)
    flops = 2.0 * M * I * O; byts = 4.0 * (M * I + M * O + I * O)
    line = 'M=%8d I=%4d O=%4d ' % (M, I, O)
    if 'fwd' in which:
        nb = l.pq_conv1x1_blocks_k(M, O, I, 0); ldp = (O + 15) // 16 * 16
        part = torch.empty(nb * 2 * ldp, device=dev)
        a = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), I, 2), w.data_ptr(), None, y.data_ptr(), part.data_ptr(),
                           M, I, O, ldo, ldp, 1, 0, 0, 0, 0)
        ms = timeit(lambda: abi.check(l.pq_conv1x1_fwd(C.byref(a), st)))
        tot['fwd'] = tot.get('fwd', 0) + ms
        line += '| fwd %.3f ms %5.1f TF %6.0f GB/s ' % (ms, flops / ms / 1e9, byts / ms / 1e6)
    if 'dgrad' in which:
        nb = l.pq_conv1x1_blocks_k(M, I, O, 1); ldp = (I + 15) // 16 * 16
        part = torch.empty(nb * 2 * ldp, device=dev)
        two = os.environ.get('DG_ONE') is None
        fin = 0 if os.environ.get('DG_NOFIN') else 1
        a = abi.Conv1x1Dgrad(abi.GSrc(g.data_ptr(), y.data_ptr() if two else None, p.data_ptr(), q.data_ptr(), mu.data_ptr(), r.data_ptr(), ldo, ldo), w.data_ptr(),
                             abi.GSink(gin.data_ptr(), x.data_ptr(), a2.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), part.data_ptr(), I, I, 2, 0, fin, ldp),
                             M, I, O)
        if os.environ.get('DG_WT'):
            wt = torch.zeros(I, ldo, device=dev); wt[:, :O] = w.t()
            a.wt = wt.data_ptr()
        ms = timeit(lambda: abi.check(l.pq_conv1x1_dgrad(C.byref(a), st)))
        tot['dgrad'] = tot.get('dgrad', 0) + ms
        line += '| dgrad %.3f ms %5.1f TF %6.0f GB/s ' % (ms, flops / ms / 1e9, byts / ms / 1e6)
    if 'wgrad' in which:
        scratch = torch.empty(l.pq_conv1x1_wgrad_scratch_bytes(M, I, O) // 4 + 16, device=dev)
        dw = torch.empty(O, I, device=dev)
        a = abi.Conv1x1Wgrad(abi.GSrc(g.data_ptr(), y.data_ptr(), p.data_ptr(), q.data_ptr(), mu.data_ptr(), r.data_ptr(), ldo, ldo),
                             abi.Src(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), I, 2), dw.data_ptr(), None, scratch.data_ptr(), M, I, O)
        ms = timeit(lambda: abi.check(l.pq_conv1x1_wgrad(C.byref(a), st)))
        tot['wgrad'] = tot.get('wgrad', 0) + ms
        line += '| wgrad %.3f ms %5.1f TF %6.0f GB/s' % (ms, flops / ms / 1e9, byts / ms / 1e6)
    if 'bwd' in which and l.pq_conv1x1_bwd_blocks(M, I, O) > 0:
        nb = l.pq_conv1x1_bwd_blocks(M, I, O); ldp = (I + 15) // 16 * 16
        part = torch.empty(nb * 2 * ldp, device=dev)
        scratch = torch.empty(l.pq_conv1x1_bwd_scratch_bytes(M, I, O) // 4 + 16, device=dev)
        dw = torch.empty(O, I, device=dev)
        a = abi.Conv1x1Bwd(abi.GSrc(g.data_ptr(), y.data_ptr(), p.data_ptr(), q.data_ptr(), mu.data_ptr(), r.data_ptr(), ldo, ldo),
                           w.data_ptr(), abi.Src(x.data_ptr(), a2.data_ptr(), b2.data_ptr(), I, 2),
                           abi.GSink(gin.data_ptr(), x.data_ptr(), a2.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), part.data_ptr(), I, I, 2, 0, 1, ldp),
                           dw.data_ptr(), scratch.data_ptr(), M, I, O)
        ms = timeit(lambda: abi.check(l.pq_conv1x1_bwd(C.byref(a), st)))
        tot['bwd'] = tot.get('bwd', 0) + ms
        tb = 4.0 * M * (2 * O + 3 * I)
        line += '| fused bwd %.3f ms %6.0f GB/s(real) grid %d' % (ms, tb / ms / 1e6, nb)
    print(line, flush=True)
print('sum', {k: round(v, 3) for k, v in tot.items()})
