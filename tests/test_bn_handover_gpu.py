"""BatchNorm statistics handed from the producing kernel to its first consumer without a finalize launch (pq_bn_stat,
include/pqdet_hip.h): the producer's fixed-point accumulators hold the EXACT sum of its fp32 partials (order independent), the
consumer derives the same coefficients pq_bn_fwd_finalize would and its first workgroup stores the vectors / running statistics
-- checked per kernel pair against nn.BatchNorm2d (model/parser.py:404-405) and, for the whole step, against the plan that keeps
the finalize launches (PQ_BN_ATOMIC=0)."""
import ctypes as C
import io
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import test_kernels_gpu as K  # noqa: E402
from pqdet_amd import abi  # noqa: E402
from pqdet_amd.model import zoo  # noqa: E402
from pqdet_amd.model.interpreter import DetectionModel  # noqa: E402
from seedfill import seeded_fill  # noqa: E402

DEV = 'cuda'


class Stat:
    """Accumulators + BatchNorm state of one layer, as the engine's hand-over pass sets them up."""

    def __init__(self, Cc, nblk, count, g):
        l = abi.lib()
        self.C, self.ldp, self.count = Cc, (Cc + 15) // 16 * 16, float(count)
        self.slots = l.pq_bn_stat_slots(nblk)
        self.acc = torch.zeros(l.pq_bn_stat_bytes(self.ldp, self.slots) // 8, dtype=torch.int64, device=DEV)
        self.bad = torch.zeros(16, dtype=torch.int32, device=DEV)
        self.bn = torch.nn.BatchNorm2d(Cc)
        with torch.no_grad():
            self.bn.weight.copy_(torch.rand(Cc, generator=g) + 0.5)
            self.bn.bias.copy_(K.rnd(Cc, g=g, scale=0.5))
            self.bn.running_mean.copy_(K.rnd(Cc, g=g))
            self.bn.running_var.copy_(torch.rand(Cc, generator=g) + 0.5)
        self.gamma, self.beta = K.chan_vec(self.bn.weight.detach()), K.chan_vec(self.bn.bias.detach())
        self.rm, self.rv = K.chan_vec(self.bn.running_mean), K.chan_vec(self.bn.running_var)
        self.nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
        self.mean, self.inv, self.scale, self.shift = (torch.full((self.ldp + 16,), float('nan'), device=DEV) for _ in range(4))

    def struct(self):
        return abi.BnStat(self.acc.data_ptr(), self.bad.data_ptr(), self.gamma.data_ptr(), self.beta.data_ptr(), self.rm.data_ptr(),
                          self.rv.data_ptr(), self.nbt.data_ptr(), self.mean.data_ptr(), self.inv.data_ptr(), self.scale.data_ptr(),
                          self.shift.data_ptr(), self.slots, self.ldp, self.count, 1e-5, 0.1, 1.0 / self.count)

    def sums(self):
        a = self.acc.cpu().view(self.slots, 2, self.ldp, 2).double()
        v = (a[..., 1] + a[..., 0] * 2.0 ** -50).sum(0)
        return v[0, :self.C], v[1, :self.C]

    def check_against(self, y):
        """y: the raw (M, C) tensor the statistics describe (CPU).  Sums, vectors and running statistics vs BatchNorm2d."""
        s1, s2 = self.sums()
        yd = y.double()
        K.close(s1.float(), yd.sum(0).float(), rt=1e-5, at=1e-3 * math.sqrt(y.shape[0] / 4096 + 1), what='sum x')
        K.close(s2.float(), (yd ** 2).sum(0).float(), rt=1e-5, at=1e-3 * math.sqrt(y.shape[0] / 4096 + 1), what='sum x^2')
        self.bn.train()
        out = self.bn(y.t().reshape(1, self.C, -1, 1))            # updates the running statistics like one train step
        mean, var = yd.mean(0), yd.var(0, unbiased=False)
        inv = 1 / torch.sqrt(var + 1e-5)
        K.close(self.mean[:self.C], mean.float(), rt=1e-5, at=1e-6, what='mean')
        K.close(self.inv[:self.C], inv.float(), rt=1e-5, at=1e-6, what='invstd')
        K.close(self.scale[:self.C], (self.bn.weight.detach().double() * inv).float(), rt=1e-5, at=1e-6, what='scale')
        K.close(self.shift[:self.C], (self.bn.bias.detach().double() - mean * self.bn.weight.detach().double() * inv).float(), rt=1e-4, at=1e-5, what='shift')
        K.close(self.rm[:self.C], self.bn.running_mean, rt=1e-5, at=1e-6, what='running_mean')
        K.close(self.rv[:self.C], self.bn.running_var, rt=1e-5, at=1e-6, what='running_var')
        assert self.nbt.item() == 1 and int(self.bad[0]) == 0
        return out.reshape(self.C, -1).t()                         # BatchNorm2d's normalised tensor (M, C)


@pytest.mark.parametrize('M,I,O,C2', [(4096, 160, 960, 160),      # ring kernel -> ring kernel (project conv), 64 row blocks, 1 slot
                                      (16384, 64, 384, 64),       # register-staged producer (K = 64), 256 row blocks: 4 slots
                                      (65536, 128, 256, 128),     # 512 row blocks of 128: 16 slots
                                      (4000, 144, 208, 75)])      # ragged rows / columns
def test_conv1x1_to_conv1x1_handover(M, I, O, C2, stats_b=False, bias_b=False, expect_b=None):
    """conv A (stats -> accumulators) -> BatchNorm + ReLU6 -> conv B (coefficients derived in its prologue).  stats_b / bias_b:
    conv B itself emits statistic rows / adds a bias (the epilogue variants the nets use); expect_b pins its instantiation."""
    l = abi.lib()
    g = torch.Generator().manual_seed(M + I + O)
    x = K.rnd(M, I, g=g)
    wA = K.rnd(O, I, g=g, scale=1 / math.sqrt(I))
    wB = K.rnd(C2, O, g=g, scale=1 / math.sqrt(O))
    yA = x @ wA.t()
    nblk = l.pq_conv1x1_blocks_k(M, O, I, 0)
    st = Stat(O, nblk, M, g)
    xd, wAd, wBd = x.to(DEV), wA.to(DEV), wB.to(DEV)
    ya = torch.full((M, O), float('nan'), device=DEV)
    a = abi.Conv1x1Fwd(abi.Src(xd.data_ptr(), None, None, I, 0), wAd.data_ptr(), None, ya.data_ptr(), None, M, I, O, O, st.ldp,
                       1, 0, 0, 0, 0)
    a.stat_acc, a.stat_bad, a.stat_slots = st.acc.data_ptr(), st.bad.data_ptr(), st.slots
    assert l.pq_bn_stat_ok(abi.OP_CONV1X1_FWD, C.byref(a), 0) == 1
    abi.check(l.pq_conv1x1_fwd(C.byref(a), K.stream()))
    ldo = (C2 + 3) // 4 * 4
    yb = torch.full((M, ldo), float('nan'), device=DEV)
    biasB = K.rnd(C2, g=g).to(DEV) if bias_b else None
    nblkB, ldpB = l.pq_conv1x1_blocks_k(M, C2, O, 0), (C2 + 15) // 16 * 16
    partB = torch.zeros(nblkB * 2 * ldpB, device=DEV) if stats_b else None
    b = abi.Conv1x1Fwd(abi.Src(ya.data_ptr(), None, None, O, 2), wBd.data_ptr(), abi.ptr(biasB), yb.data_ptr(), abi.ptr(partB), M, O, C2,
                       ldo, ldpB if stats_b else 0, 1, 0, 0, 0, 0)
    b.inp.stat = st.struct()
    assert l.pq_bn_stat_ok(abi.OP_CONV1X1_FWD, C.byref(b), 1) == 1
    plan = K.gemm_plan('fwd', b, expect_b)
    assert plan[0] == 2 and plan[4] == 3, plan                     # LDS-DMA ring kernel, AMODE 3
    abi.check(l.pq_conv1x1_fwd(C.byref(b), K.stream()))
    torch.cuda.synchronize()
    K.close(ya, yA, what='conv A')
    norm = st.check_against(ya.cpu())
    want = norm.detach().clamp(0, 6) @ wB.t() + (biasB.cpu() if bias_b else 0)
    K.close(yb[:, :C2], want, rt=1e-3, at=2e-4, what='conv B on the handed-over BatchNorm')
    if stats_b:
        s1, s2 = K.reduce_partial(partB, nblkB, ldpB, C2)
        got = yb[:, :C2].cpu().double()
        K.close(s1.float(), got.sum(0).float(), rt=1e-3, at=1e-2, what='conv B sum')
        K.close(s2.float(), (got ** 2).sum(0).float(), rt=1e-3, at=1e-2, what='conv B sumsq')
    # a later consumer reads the stored vectors: same result through the ordinary path
    yb2 = torch.full((M, ldo), float('nan'), device=DEV)
    b2 = abi.Conv1x1Fwd(abi.Src(ya.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), O, 2), wBd.data_ptr(), abi.ptr(biasB),
                        yb2.data_ptr(), None, M, O, C2, ldo, 0, 1, 0, 0, 0, 0)
    abi.check(l.pq_conv1x1_fwd(C.byref(b2), K.stream()))
    torch.cuda.synchronize()
    assert torch.equal(yb2[:, :C2], yb[:, :C2])                    # same coefficients, same kernel arithmetic


def test_unsupported_consumers_say_so():
    l = abi.lib()
    x = torch.zeros(1024, 32, device=DEV)
    w = torch.zeros(64, 32, device=DEV)
    out = torch.zeros(1024, 64, device=DEV)
    b = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), None, None, 32, 2), w.data_ptr(), None, out.data_ptr(), None, 1024, 32, 64, 64, 0,
                       1, 0, 0, 0, 0)
    assert l.pq_bn_stat_ok(abi.OP_CONV1X1_FWD, C.byref(b), 1) == 0          # K = 32: register-staged kernel
    b.inp.stat.acc = x.data_ptr()
    assert l.pq_conv1x1_fwd(C.byref(b), K.stream()) == -3                    # PQ_EUNSUPPORTED, loudly
    assert l.pq_bn_stat_ok(abi.OP_DW_DGRAD, C.byref(b), 1) == 0


@pytest.mark.parametrize('B,H,W,Cc,stride', [(2, 32, 32, 96, 1), (4, 64, 64, 32, 2), (3, 16, 16, 200, 1)])
def test_conv1x1_to_depthwise_to_conv1x1_handover(B, H, W, Cc, stride):
    """expand conv -> BN + ReLU6 -> depthwise 3x3 (consumer AND producer of a hand-over) -> BN + ReLU6 -> project conv."""
    l = abi.lib()
    g = torch.Generator().manual_seed(B * H * W + Cc)
    I, O2 = 128, 64
    M = B * H * W
    x = K.rnd(M, I, g=g)
    wA = K.rnd(Cc, I, g=g, scale=1 / math.sqrt(I))
    wD = K.rnd(Cc, 1, 3, 3, g=g, scale=0.4)
    wB = K.rnd(O2, Cc, g=g, scale=1 / math.sqrt(Cc))
    P = (Cc + 3) // 4 * 4
    xd, wAd, wDd, wBd = x.to(DEV), wA.to(DEV), wD.to(DEV), wB.to(DEV)
    st1 = Stat(Cc, l.pq_conv1x1_blocks_k(M, Cc, I, 0), M, g)
    ya = torch.zeros(M, P, device=DEV)
    a = abi.Conv1x1Fwd(abi.Src(xd.data_ptr(), None, None, I, 0), wAd.data_ptr(), None, ya.data_ptr(), None, M, I, Cc, P, st1.ldp,
                       1, 0, 0, 0, 0)
    a.stat_acc, a.stat_bad, a.stat_slots = st1.acc.data_ptr(), st1.bad.data_ptr(), st1.slots
    abi.check(l.pq_conv1x1_fwd(C.byref(a), K.stream()))
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    M2 = B * OH * OW
    st2 = Stat(Cc, l.pq_dw3x3_fwd_blocks(B, OH, OW, Cc, stride), M2, g)
    yd_ = torch.zeros(M2, P, device=DEV)
    d = abi.DwFwd(abi.Src(ya.data_ptr(), None, None, P, 2), wDd.data_ptr(), yd_.data_ptr(), None, B, H, W, OH, OW, Cc, stride,
                  P, st2.ldp)
    d.inp.stat = st1.struct()
    d.stat_acc, d.stat_bad, d.stat_slots = st2.acc.data_ptr(), st2.bad.data_ptr(), st2.slots
    assert l.pq_bn_stat_ok(abi.OP_DW_FWD, C.byref(d), 0) == 1 and l.pq_bn_stat_ok(abi.OP_DW_FWD, C.byref(d), 1) == 1
    abi.check(l.pq_dw3x3_fwd(C.byref(d), K.stream()))
    yb = torch.zeros(M2, O2, device=DEV)
    b = abi.Conv1x1Fwd(abi.Src(yd_.data_ptr(), None, None, P, 2), wBd.data_ptr(), None, yb.data_ptr(), None, M2, Cc, O2, O2, 0,
                       1, 0, 0, 0, 0)
    b.inp.stat = st2.struct()
    ok_b = l.pq_bn_stat_ok(abi.OP_CONV1X1_FWD, C.byref(b), 1)
    if not ok_b:                                       # e.g. K = 200 (not a multiple of 16): finalize-free hand-over not offered
        pytest.skip('project conv shape is outside the ring kernels')
    abi.check(l.pq_conv1x1_fwd(C.byref(b), K.stream()))
    torch.cuda.synchronize()
    n1 = st1.check_against(ya[:, :Cc].cpu())
    act1 = n1.detach().clamp(0, 6).reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    yD = F.conv2d(act1, wD, stride=stride, padding=1, groups=Cc).permute(0, 2, 3, 1).reshape(M2, Cc)
    K.close(yd_[:, :Cc], yD, rt=1e-3, at=2e-4, what='depthwise on the handed-over BatchNorm')
    n2 = st2.check_against(yd_[:, :Cc].cpu())
    K.close(yb, n2.detach().clamp(0, 6) @ wB.t(), rt=1e-3, at=2e-4, what='project conv')


@pytest.mark.parametrize('B,H,W,Cc,gw,stride', [(2, 32, 32, 240, 24, 1), (2, 64, 64, 96, 24, 2), (3, 16, 16, 176, 16, 1)])
def test_conv1x1_to_grouped3x3_to_conv1x1_handover(B, H, W, Cc, gw, stride):
    """RegNetX block: 1x1 conv -> BN + ReLU -> grouped 3x3 (consumer AND producer of a hand-over) -> BN + ReLU -> 1x1 conv."""
    l = abi.lib()
    g = torch.Generator().manual_seed(B * H * W + Cc + gw)
    I, O2 = 128, 64
    M = B * H * W
    x = K.rnd(M, I, g=g)
    wA = K.rnd(Cc, I, g=g, scale=1 / math.sqrt(I))
    wG = K.rnd(Cc, gw, 3, 3, g=g, scale=1 / math.sqrt(9 * gw))
    wB = K.rnd(O2, Cc, g=g, scale=1 / math.sqrt(Cc))
    xd, wAd, wGd, wBd = x.to(DEV), wA.to(DEV), wG.to(DEV), wB.to(DEV)
    st1 = Stat(Cc, l.pq_conv1x1_blocks_k(M, Cc, I, 0), M, g)
    ya = torch.zeros(M, Cc, device=DEV)
    a = abi.Conv1x1Fwd(abi.Src(xd.data_ptr(), None, None, I, 0), wAd.data_ptr(), None, ya.data_ptr(), None, M, I, Cc, Cc, st1.ldp,
                       1, 0, 0, 0, 0)
    a.stat_acc, a.stat_bad, a.stat_slots = st1.acc.data_ptr(), st1.bad.data_ptr(), st1.slots
    abi.check(l.pq_conv1x1_fwd(C.byref(a), K.stream()))
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    M2 = B * OH * OW
    st2 = Stat(Cc, l.pq_gconv3x3_blocks(B, OH, OW), M2, g)
    yg = torch.zeros(M2, Cc, device=DEV)
    d = abi.GconvFwd(abi.Src(ya.data_ptr(), None, None, Cc, 1), wGd.data_ptr(), yg.data_ptr(), None, B, H, W, OH, OW, Cc, gw, stride,
                     Cc, st2.ldp)
    d.inp.stat = st1.struct()
    d.stat_acc, d.stat_bad, d.stat_slots = st2.acc.data_ptr(), st2.bad.data_ptr(), st2.slots
    assert l.pq_bn_stat_ok(abi.OP_GCONV_FWD, C.byref(d), 0) == 1 and l.pq_bn_stat_ok(abi.OP_GCONV_FWD, C.byref(d), 1) == 1
    abi.check(l.pq_gconv3x3_fwd(C.byref(d), K.stream()))
    yb = torch.zeros(M2, O2, device=DEV)
    b = abi.Conv1x1Fwd(abi.Src(yg.data_ptr(), None, None, Cc, 1), wBd.data_ptr(), None, yb.data_ptr(), None, M2, Cc, O2, O2, 0,
                       1, 0, 0, 0, 0)
    b.inp.stat = st2.struct()
    if not l.pq_bn_stat_ok(abi.OP_CONV1X1_FWD, C.byref(b), 1):
        pytest.skip('second 1x1 conv shape is outside the ring kernels')
    abi.check(l.pq_conv1x1_fwd(C.byref(b), K.stream()))
    torch.cuda.synchronize()
    n1 = st1.check_against(ya.cpu())
    act1 = n1.detach().clamp(min=0).reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    yG = F.conv2d(act1, wG, stride=stride, padding=1, groups=Cc // gw).permute(0, 2, 3, 1).reshape(M2, Cc)
    K.close(yg, yG, rt=1e-3, at=2e-4, what='grouped 3x3 on the handed-over BatchNorm')
    n2 = st2.check_against(yg.cpu())
    K.close(yb, n2.detach().clamp(min=0) @ wB.t(), rt=1e-3, at=2e-4, what='second 1x1 conv')


def test_shortcut_handover_and_nan_poison():
    """project conv -> BN (linear) -> shortcut add (copy_fwd derives the coefficients); then a NaN in the producer's output
    must reach the statistics (the step's NaN trap depends on it)."""
    l = abi.lib()
    g = torch.Generator().manual_seed(3)
    B, H, W, I, Cc = 2, 32, 32, 192, 32
    M = B * H * W
    x, other = K.rnd(M, I, g=g), K.rnd(M, Cc, g=g)
    wA = K.rnd(Cc, I, g=g, scale=1 / math.sqrt(I))
    xd, od, wAd = x.to(DEV), other.to(DEV), wA.to(DEV)
    for poison in (False, True):
        if poison:
            xd = xd.clone()
            xd[5, 7] = float('nan')
        st = Stat(Cc, l.pq_conv1x1_blocks_k(M, Cc, I, 0), M, g)
        ya = torch.zeros(M, Cc, device=DEV)
        a = abi.Conv1x1Fwd(abi.Src(xd.data_ptr(), None, None, I, 0), wAd.data_ptr(), None, ya.data_ptr(), None, M, I, Cc, Cc, st.ldp,
                           1, 0, 0, 0, 0)
        a.stat_acc, a.stat_bad, a.stat_slots = st.acc.data_ptr(), st.bad.data_ptr(), st.slots
        abi.check(l.pq_conv1x1_fwd(C.byref(a), K.stream()))
        out = torch.zeros(M, Cc, device=DEV)
        c = abi.Copy(abi.Src(ya.data_ptr(), None, None, Cc, 0), abi.Src(od.data_ptr(), None, None, Cc, 0), out.data_ptr(), B, H, W, Cc,
                     Cc, 0, 1)
        c.a.stat = st.struct()
        assert l.pq_bn_stat_ok(abi.OP_COPY_FWD, C.byref(c), 1) == 1
        abi.check(l.pq_copy_fwd(C.byref(c), K.stream()))
        torch.cuda.synchronize()
        if poison:
            # (the activation's fmin / fmax swallow a NaN operand, with or without the hand-over: the statistics carry it)
            assert int(st.bad[0]) == 1 and torch.isnan(st.mean[:Cc]).all() and torch.isnan(st.scale[:Cc]).all()
        else:
            norm = st.check_against(ya.cpu())
            K.close(out, norm.detach() + other, rt=1e-3, at=2e-4, what='shortcut on the handed-over BatchNorm')


@pytest.mark.parametrize('name', ['mobilenetv2-fpn', 'regnetx-600m-fpn'])
def test_step_with_handover_equals_step_with_finalize_launches(name, monkeypatch):
    """Whole train step: PQ_BN_ATOMIC=0 (a finalize launch per BatchNorm) against the default hand-over.  The statistics are the
    same numbers up to the last bit of an fp64 sum (exact integer accumulation vs a fixed-order fp64 sum), so losses, running
    statistics and gradients agree far inside the fp32 noise of the step; the hand-over plan replays bit-identically."""
    from test_model_gpu import _targets
    from seedfill import seeded_images
    runs = {}
    for mode in ('1', '0', '1'):
        monkeypatch.setenv('PQ_BN_ATOMIC', mode)
        net = DetectionModel(io.StringIO(zoo.cfg_text(name)))
        seeded_fill(net, 4)
        net = net.to(DEV)
        net.train()
        x = seeded_images(4, 256, 256, 3).to(DEV)
        tgt = tuple(t.to(DEV) for t in _targets(4, 256, 256, 20, seed=9))
        out = net(x, tgt)
        out['loss'].mean().backward()
        net.check_nan()
        plan = list(net._plans.values())[0]
        rec = (torch.stack([out[k] for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]).detach().clone(), net._flat.grad.clone(),
               torch.cat([b_.detach().flatten().float() for b_ in net.buffers()]), plan.bn_handover, net)
        runs.setdefault(mode, []).append(rec)
    on, off = runs['1'][0], runs['0'][0]
    assert on[3] >= 20 and off[3] == 0, (on[3], off[3])
    assert torch.equal(runs['1'][0][0], runs['1'][1][0]) and torch.equal(runs['1'][0][1], runs['1'][1][1])     # deterministic
    assert torch.equal(runs['1'][0][2], runs['1'][1][2])
    assert ((on[0] - off[0]).abs() <= 1e-4 * off[0].abs()).all(), (on[0], off[0])
    assert ((on[2] - off[2]).abs() <= 1e-5 * off[2].abs() + 1e-6).all()                                        # running statistics
    # gradients: the two plans' invstd differ by an ulp (fp32 vs rounded fp64 square root), and a train-mode step at this size
    # amplifies that like any other round-off (the reference's own fp32 / fp64 evaluations differ by percents in the first
    # layers: test_train_step_gradients_within_fp32_noise_of_fp64_oracle) -- plumbing check only: every parameter within 25 %,
    # the median within 6 % (measured 4 % on mobilenetv2-fpn at this size)
    flat = on[4]._flat
    gmax = max(off[1][o:o + p.numel()].double().norm().item() for p, o in zip(flat.params, flat.offsets))
    rel = []
    for p, o in zip(flat.params, flat.offsets):
        da, db = on[1][o:o + p.numel()].double(), off[1][o:o + p.numel()].double()
        assert (da - db).norm().item() <= 0.25 * db.norm().item() + 1e-6 * gmax, (o, (da - db).norm().item(), db.norm().item())
        rel.append((da - db).norm().item() / (db.norm().item() + 1e-6 * gmax))
    rel.sort()
    print('hand-over vs finalize: gradient difference median %.2e, max %.2e' % (rel[len(rel) // 2], rel[-1]))
    assert rel[len(rel) // 2] <= 6e-2


@pytest.mark.parametrize('B,H,W,scope', [(4, 256, 256, 'dw'), (3, 96, 160, 'dw'), (16, 512, 512, 'dw'),
                                         (4, 256, 256, 'all'), (3, 96, 160, 'all'), (16, 512, 512, 'all')])
def test_backward_handover_equals_backward_finalize_launches(B, H, W, scope, monkeypatch):
    """Backward twin (pq_bn_bstat): PQ_BN_BWD_ATOMIC=0 (a pq_bn_bwd_finalize launch behind every sink that finalises a BatchNorm
    output's gradient) against the hand-over -- scope 'dw' (the engine's default: the layers read by the walking depthwise
    backward only) and 'all' (round 4: every layer whose producer -- ring data gradient, fused narrow backward, walking depthwise
    backward, shortcut / route backward -- and readers -- ring data gradient, both weight-gradient kernels, fused narrow backward,
    walking depthwise -- support it; measured slower in the step, kept behind the switch).  The producer's sink adds its sums into
    accumulators, the readers derive p / q / r and store dgamma / dbeta.  The forward is the same plan in both runs: losses and running
    statistics are bit-identical; q and r come from the same partials summed exactly instead of in fp64 (an ulp at most), and
    the backward has no discontinuity to amplify that: every parameter gradient agrees to 1e-5 norm-wise.  The hand-over plan
    replays bit-identically (integer accumulation is order independent)."""
    from test_model_gpu import _targets
    from seedfill import seeded_images
    runs = {}
    # (the fused inverted-residual kernels take no part in the hand-over: this test is about the producer / consumer roles of the
    # unfused kernels; fused plans are checked against the CPU oracle by the full-size tests)
    monkeypatch.setenv('PQ_IRFUSE', '0')
    for mode in ('1', '0', '1'):
        monkeypatch.setenv('PQ_BN_BWD_ATOMIC', scope if mode == '1' else '0')
        net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn')))
        seeded_fill(net, 4)
        net = net.to(DEV)
        net.train()
        x = seeded_images(B, H, W, 3).to(DEV)
        tgt = tuple(t.to(DEV) for t in _targets(B, H, W, 20, seed=9))
        out = net(x, tgt)
        out['loss'].mean().backward()
        net.check_nan()
        plan = list(net._plans.values())[0]
        rec = (torch.stack([out[k] for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]).detach().clone(), net._flat.grad.clone(),
               torch.cat([b_.detach().flatten().float() for b_ in net.buffers()]), plan.bn_bwd_handover, net)
        runs.setdefault(mode, []).append(rec)
    on, off = runs['1'][0], runs['0'][0]
    # 26 depthwise layers in the net; not handed over: lists of more than 640 rows (the stride-8 level at the benchmark shape) and
    # projections whose data gradient runs on the register-staged GEMM family (192-channel layers): 19 at 512 x 512 batch 16
    assert on[3] >= (15 if scope == 'dw' else 45) and off[3] == 0, (on[3], off[3])
    kinds = list(on[4]._plans.values())[0].bn_bwd_handover_producers
    assert abi.OP_CONV1X1_DGRAD in kinds
    if B * H * W >= 16 * 512 * 512:                                      # BASELINE config 2: the fused narrow-layer backward adds too
        assert abi.OP_CONV1X1_BWD in kinds
    if scope == 'all':
        readers = list(on[4]._plans.values())[0].bn_bwd_handover_consumers
        assert {abi.OP_DW_BWD, abi.OP_ROUTE_BWD} <= kinds and {abi.OP_CONV1X1_DGRAD, abi.OP_CONV1X1_WGRAD, abi.OP_DW_BWD} <= readers
    assert torch.equal(runs['1'][0][1], runs['1'][1][1])                 # deterministic
    assert torch.equal(on[0], off[0]) and torch.equal(on[2], off[2])     # same forward
    flat = on[4]._flat
    gmax = max(off[1][o:o + p.numel()].double().norm().item() for p, o in zip(flat.params, flat.offsets))
    worst = 0.0
    for p, o in zip(flat.params, flat.offsets):
        da, db = on[1][o:o + p.numel()].double(), off[1][o:o + p.numel()].double()
        err = (da - db).norm().item()
        assert err <= 1e-5 * db.norm().item() + 1e-7 * gmax, (o, err, db.norm().item())
        worst = max(worst, err / (db.norm().item() + 1e-7 * gmax))
    print('backward hand-over vs finalize launches: worst gradient difference %.2e' % worst)


@pytest.mark.parametrize('B,H,W,Cc,O2,stride', [(2, 32, 32, 96, 64, 1), (2, 32, 32, 144, 64, 2), (1, 64, 64, 192, 96, 1)])
def test_dgrad_sink_to_walking_depthwise_backward_handover(B, H, W, Cc, O2, stride):
    """Kernel pair of the backward hand-over (pq_bn_bstat): the 1x1 projection's data gradient (ring kernel, gradient sink on
    the depthwise conv's output) adds its BatchNorm-backward sums into accumulators, the walking depthwise backward derives
    p / q / r in its prologue -- against the same two kernels with statistic rows + pq_bn_bwd_finalize in between: input
    gradient, its statistic rows and the weight gradient within 1e-6 (same arithmetic on sums that are exact instead of fp64),
    dgamma / dbeta / p / q / r as the finalize kernel writes them."""
    l = abi.lib()
    g = torch.Generator().manual_seed(B * H * W + Cc + stride)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    Mo, Mi = B * OH * OW, B * H * W
    ldp = (Cc + 15) // 16 * 16
    # tensors of the depthwise layer: raw input x (its own affine + relu6), raw output y (BatchNorm + relu6 in front of the 1x1)
    x, y = K.rnd(Mi, Cc, g=g).to(DEV), K.rnd(Mo, Cc, g=g).to(DEV)
    scx, shx = K.chan_vec(torch.rand(Cc, generator=g) + 0.5), K.chan_vec(K.rnd(Cc, g=g, scale=0.3))
    mean_x, inv_x = K.chan_vec(x.mean(0).cpu()), K.chan_vec((1 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)).cpu())
    gamma = K.chan_vec(torch.rand(Cc, generator=g) + 0.5)
    mu_y = y.mean(0).cpu()
    inv_y = 1 / torch.sqrt(y.var(0, unbiased=False).cpu() + 1e-5)
    mean_y, invstd_y = K.chan_vec(mu_y), K.chan_vec(inv_y)
    scy = K.chan_vec(gamma[:Cc].cpu() * inv_y)
    shy = K.chan_vec(K.rnd(Cc, g=g, scale=0.3))
    wdw = K.rnd(Cc, 9, g=g, scale=0.4).to(DEV)
    w2 = K.rnd(O2, Cc, g=g, scale=1 / math.sqrt(Cc))
    w2d, w2t = w2.to(DEV), w2.t().contiguous().to(DEV)
    dy2 = K.rnd(Mo, O2, g=g).to(DEV)
    nblk = l.pq_conv1x1_blocks_k(Mo, Cc, O2, 1)
    nbw = l.pq_dw3x3_bwd_walk_blocks(B, H, W, Cc, stride)

    def run(handover):
        gy = torch.full((Mo, Cc), float('nan'), device=DEV)
        rows = torch.full((nblk * 2 * ldp,), float('nan'), device=DEV)
        vec = {k: torch.full((ldp + 16,), float('nan'), device=DEV) for k in ('p', 'q', 'r', 'dgamma', 'dbeta')}
        slots = l.pq_bn_stat_slots(nblk)
        acc = torch.zeros(l.pq_bn_stat_bytes(ldp, slots) // 8, dtype=torch.int64, device=DEV)
        bad = torch.zeros(16, dtype=torch.int32, device=DEV)
        d = abi.Conv1x1Dgrad()
        d.dy = abi.GSrc(dy2.data_ptr(), None, None, None, None, None, O2, 0)
        d.w, d.wt = w2d.data_ptr(), w2t.data_ptr()
        d.dx = abi.GSink(gy.data_ptr(), y.data_ptr(), scy.data_ptr(), shy.data_ptr(), mean_y.data_ptr(), invstd_y.data_ptr(),
                         rows.data_ptr(), Cc, Cc, 2, 0, 1, ldp)
        d.M, d.I, d.O, d.sub = Mo, Cc, O2, 1
        plan = abi.GemmPlan()
        abi.check(l.pq_conv1x1_dgrad_plan(C.byref(d), C.byref(plan)))
        assert plan.family == 2 and l.pq_bn_bstat_ok(abi.OP_CONV1X1_DGRAD, C.byref(d), 0) == 1
        if handover:
            d.dx.stat_acc, d.dx.stat_bad, d.dx.stat_slots = acc.data_ptr(), bad.data_ptr(), slots
        abi.check(l.pq_conv1x1_dgrad(C.byref(d), K.stream()))
        if not handover:
            f = abi.BnBwdFin(rows.data_ptr(), gamma.data_ptr(), invstd_y.data_ptr(), vec['dgamma'].data_ptr(), vec['dbeta'].data_ptr(),
                             vec['p'].data_ptr(), vec['q'].data_ptr(), vec['r'].data_ptr(), nblk, ldp, Cc, 1, float(Mo), None, None)
            abi.check(l.pq_bn_bwd_finalize(C.byref(f), K.stream()))
        gx = torch.full((Mi, Cc), float('nan'), device=DEV)
        srows = torch.full((nbw * 2 * ldp,), float('nan'), device=DEV)
        scw = torch.full((nbw * Cc * 9 + 16,), float('nan'), device=DEV)
        dww = torch.full((Cc, 9), float('nan'), device=DEV)
        f5 = abi.DwBwd(abi.GSrc(gy.data_ptr(), y.data_ptr(), vec['p'].data_ptr(), vec['q'].data_ptr(), mean_y.data_ptr(), vec['r'].data_ptr(), Cc, Cc),
                       abi.Src(x.data_ptr(), scx.data_ptr(), shx.data_ptr(), Cc, 2), wdw.data_ptr(),
                       abi.GSink(gx.data_ptr(), x.data_ptr(), scx.data_ptr(), shx.data_ptr(), mean_x.data_ptr(), inv_x.data_ptr(),
                                 srows.data_ptr(), Cc, Cc, 2, 0, 1, ldp),
                       None, scw.data_ptr(), B, H, W, OH, OW, Cc, stride, 5)
        assert l.pq_bn_bstat_ok(abi.OP_DW_BWD, C.byref(f5), 1) == 1
        if handover:
            f5.dy_stat = abi.BnBstat(acc.data_ptr(), bad.data_ptr(), gamma.data_ptr(), invstd_y.data_ptr(), vec['dgamma'].data_ptr(),
                                     vec['dbeta'].data_ptr(), vec['p'].data_ptr(), vec['q'].data_ptr(), vec['r'].data_ptr(), slots, ldp,
                                     float(Mo), 1)
        abi.check(l.pq_dw3x3_bwd(C.byref(f5), K.stream()))
        f6 = abi.DwBwd(abi.GSrc(), abi.Src(), None, abi.GSink(), dww.data_ptr(), scw.data_ptr(), B, H, W, OH, OW, Cc, stride, 6)
        abi.check(l.pq_dw3x3_bwd(C.byref(f6), K.stream()))
        torch.cuda.synchronize()
        assert int(bad[0]) == 0
        return gy, gx, dww, srows, {k: v[:Cc].clone() for k, v in vec.items()}

    gy0, gx0, dw0, s0, v0 = run(False)
    gy1, gx1, dw1, s1, v1 = run(True)
    assert torch.equal(gy0, gy1) and torch.isfinite(gx0).all()            # the sink's gradient itself does not depend on the hand-over
    for k in ('dgamma', 'dbeta', 'p', 'q', 'r'):
        K.close(v1[k], v0[k], rt=1e-6, at=1e-7 * float(v0[k].abs().max()), what=k)
    scale = float(gx0.abs().max())
    K.close(gx1, gx0, rt=1e-5, at=1e-6 * scale, what='input gradient of the depthwise layer')
    K.close(dw1, dw0, rt=1e-5, at=1e-6 * float(dw0.abs().max()), what='depthwise weight gradient')
    K.close(s1.view(nbw, 2, ldp)[:, :, :Cc].sum(0), s0.view(nbw, 2, ldp)[:, :, :Cc].sum(0), rt=1e-5,
            at=1e-5 * float(s0.view(nbw, 2, ldp)[:, :, :Cc].sum(0).abs().max()), what='statistic rows of the input gradient')


# ---- the fixed-point image of the statistics accumulators at its limits (the low limb is read back signed)
@pytest.mark.gpu
@pytest.mark.parametrize('case', ['max_magnitude_4096', 'tiny_negative', 'mixed', 'poison'])
def test_stat_accumulator_image_at_the_limits(case):
    import math
    from fractions import Fraction
    l = abi.lib()
    g = torch.Generator().manual_seed(7)
    slots, ldp = (1, 16)
    if case == 'max_magnitude_4096':
        v = torch.full((4096,), 4.9e14)                       # |v| < 2^49, the most one slot may take
    elif case == 'tiny_negative':
        v = -torch.rand(1000, generator=g) * 1e-18            # floor(v) = -1, fraction rounds to 2^50: image = 0 (error < 2^-50 each)
    elif case == 'mixed':
        slots = 4
        v = (torch.rand(16384, generator=g) - 0.5) * torch.pow(10.0, torch.randint(-6, 12, (16384,), generator=g).float())
    else:
        v = torch.tensor([1.0, float('nan'), 2.0, float('inf'), 6.0e14])
    vd = v.to(DEV)
    acc = torch.zeros(slots * 2 * ldp * 2, dtype=torch.int64, device=DEV)
    bad = torch.zeros(16, dtype=torch.int32, device=DEV)
    abi.check(l.pq_bn_stat_selftest(vd.data_ptr(), v.numel(), slots, ldp, acc.data_ptr(), bad.data_ptr(), K.stream()))
    torch.cuda.synchronize()
    a = acc.view(slots, 2, ldp, 2)[:, 0, 0, :].cpu()          # plane 0, channel 0: (lo, hi) per slot
    lo, hi = int(a[:, 0].sum()), int(a[:, 1].sum())
    assert all(int(x) >= 0 for x in a[:, 0]), 'a low limb went negative (overflow)'
    got = Fraction(hi) + Fraction(lo, 2 ** 50)
    if case == 'poison':
        assert int(bad[0]) == 1
        assert got == Fraction(3)                              # the finite, in-range partials were still added
        return
    assert int(bad[0]) == 0
    exact = sum(Fraction(float(x)) for x in v)
    assert abs(got - exact) <= Fraction(v.numel(), 2 ** 50), (float(got), float(exact))
    if case == 'max_magnitude_4096':
        assert got == exact
    # one partial more than a slot may take is refused
    big = torch.zeros(4097, device=DEV)
    assert l.pq_bn_stat_selftest(big.data_ptr(), 4097, 1, ldp, acc.data_ptr(), bad.data_ptr(), K.stream()) != 0
    assert b'4096' in l.pq_last_error()
    if case == 'max_magnitude_4096':
        # ... and so is a producer whose launch would add more row blocks than that into one slot
        M, I, O = 128 * 4097, 16, 16
        x = torch.zeros(M, I, device=DEV); w = torch.zeros(O, I, device=DEV); y = torch.empty(M, O, device=DEV)
        assert l.pq_conv1x1_blocks_k(M, O, I, 0) > 4096
        a = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), None, None, I, 0), w.data_ptr(), None, y.data_ptr(), None, M, I, O, O, 16, 1, 0, 0, 0, 0, 0,
                           acc.data_ptr(), bad.data_ptr(), 1)
        assert l.pq_conv1x1_fwd(C.byref(a), K.stream()) != 0 and b'row blocks per accumulator slot' in l.pq_last_error()


def test_sink_accumulators_are_validated_or_refused():
    """pq_gsink.stat_acc: entries that honour it check the record (stat_bad, finalising sink with statistics, pitch, slots, adds per
    slot); entries that do not add into it refuse the call instead of silently writing `partial` rows."""
    l = abi.lib()
    buf = torch.zeros(4096, device=DEV)
    p = buf.data_ptr()

    def sink(stat_bad=True, finalize=1, slots=1, ldp=64):
        return abi.GSink(p, p, None, None, p, p, p, 32, 32, 0, 0, finalize, ldp, None, p, p if stat_bad else None, slots, 0)
    gs = abi.GSrc(p, p, p, p, p, p, 32, 32)
    # honouring entries: a malformed record is an argument error
    for sk in (sink(stat_bad=False), sink(finalize=0), sink(slots=5), sink(ldp=16)):
        r = abi.RouteBwd(p, sk, 1, 4, 4, 32, 32, 0, 1)
        assert l.pq_route_bwd(C.byref(r), K.stream()) == -1 and b"stat_acc" in l.pq_last_error()
        f5 = abi.DwBwd(gs, abi.Src(p, None, None, 32, 0), p, sk, None, p, 1, 8, 8, 8, 8, 32, 1, 5)
        assert l.pq_dw3x3_bwd(C.byref(f5), K.stream()) == -1 and b"stat_acc" in l.pq_last_error()
    # entries that cannot add into accumulators
    d = abi.DwDgrad(gs, p, sink(), 1, 8, 8, 8, 8, 32, 1)
    assert l.pq_dw3x3_dgrad(C.byref(d), K.stream()) == -3 and b"stat_acc" in l.pq_last_error()
    gd = abi.GconvDgrad(gs, p, sink(), 1, 8, 8, 8, 8, 32, 16, 1)
    assert l.pq_gconv3x3_dgrad(C.byref(gd), K.stream()) == -3 and b"stat_acc" in l.pq_last_error()
    qb = abi.QactBwd(p, sink(), p, 16, 32, 32, 1)
    assert l.pq_qact_bwd(C.byref(qb), K.stream()) == -3 and b"stat_acc" in l.pq_last_error()
