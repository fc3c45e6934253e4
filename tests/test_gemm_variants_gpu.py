"""Every 1x1-convolution GEMM instantiation the BASELINE nets select at their BENCHMARK shapes is value-tested.

The dispatch (``row_bm`` / ``g2_pick_bn`` / ``g2_wgrad_plan`` in ``csrc/gemm*.hip``) picks tile shapes from the size of the launch
grid, so small test shapes only ever reach the 64 x 64 tiles.  The cases below use the real layer shapes (the fp32 CPU matmul
of a 65536 x 256 x 256 layer is milliseconds), each case PINS the instantiation it was written for through
``pq_conv1x1_{fwd,dgrad,wgrad}_plan`` (so an eligibility rule cannot silently reroute it), and the last test walks the
compiled plans of the three BASELINE nets at their benchmark shapes and fails when they select an instantiation that has no
case here.

mirrors: nn.Conv2d(k=1) forward / backward inside the conv blocks of the reference, model/parser.py:389-410.
"""
import ctypes as C
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

import test_kernels_gpu as K  # noqa: E402
from pqdet_amd import abi  # noqa: E402

# plan key = (family, bm, bn, stages, amode, emode, variant); family 2 = LDS-DMA ring kernels (gemm2.hip), 1 = gemm.hip
# ---- forward: M, I, O, act, affine, bias, stats, expected instantiation
FWD = [
    (65536, 128, 256, 2, True, False, True, (2, 128, 128, 2, 1, 1, 0)),      # FPN 64 x 64 level
    (65536, 256, 128, 2, True, False, True, (2, 128, 128, 2, 1, 1, 0)),
    (65536, 256, 75, 2, True, True, False, (2, 128, 128, 2, 1, 0, 0)),       # detection head (bias, no BatchNorm), ragged N
    (16384, 512, 256, 2, True, False, True, (2, 128, 64, 2, 1, 1, 0)),
    (16384, 256, 512, 2, True, False, True, (2, 128, 128, 2, 1, 1, 0)),
    (16384, 352, 256, 0, False, False, True, (2, 128, 64, 2, 0, 1, 0)),      # concatenated input: operand as stored
    (65536, 160, 128, 0, False, False, True, (2, 128, 128, 2, 0, 1, 0)),
    (4096, 512, 1024, 2, True, False, True, (2, 128, 64, 2, 1, 1, 0)),
    (4096, 1024, 512, 2, True, False, True, (2, 64, 64, 2, 1, 1, 0)),
    (16384, 256, 160, 1, True, False, True, (2, 64, 128, 2, 1, 1, 0)),       # 64-row blocks x 128 columns
    (16390, 256, 150, 1, True, True, False, (2, 64, 128, 2, 1, 0, 0)),       # ... ragged in both directions, bias
    (65500, 240, 240, 1, True, False, True, (2, 128, 128, 2, 1, 1, 0)),      # contraction 16 modulo 32 on 128-wide tiles, ragged M, N
    (16384, 528, 528, 1, True, False, True, (2, 128, 128, 2, 1, 1, 0)),      # RegNetX-600M last stage
    (32768, 176, 176, 0, False, False, True, (2, 128, 128, 2, 0, 1, 0)),     # ... half k-tile with the raw operand
    (16384, 512, 128, 0, False, True, False, (2, 64, 64, 2, 0, 0, 0)),       # raw operand, plain store
    (65536, 128, 200, 0, False, True, False, (2, 128, 128, 2, 0, 0, 0)),
    (16384, 512, 200, 0, False, True, False, (2, 128, 64, 2, 0, 0, 0)),
    (16384, 256, 140, 0, False, True, False, (2, 64, 128, 2, 0, 0, 0)),
    (16384, 512, 224, 2, True, True, False, (2, 128, 64, 2, 1, 0, 0)),
    (4096, 512, 256, 2, True, True, False, (2, 64, 64, 2, 1, 0, 0)),
    (16384, 256, 140, 0, False, False, True, (2, 64, 128, 2, 0, 1, 0)),
    (4096, 512, 256, 0, False, False, True, (2, 64, 64, 2, 0, 1, 0)),
    # register-staged kernels (gemm.hip): short contractions on the big planes ("streaming" full-N tiles need >= 384 row blocks),
    # heads with 75 outputs
    (16384, 64, 384, 0, False, False, True, (1, 64, 64, 2, 0, 1, 0)),
    (4096, 352, 75, 1, False, True, False, (1, 64, 64, 2, 1, 0, 0)),
    (16384, 64, 384, 0, True, False, True, (1, 64, 64, 2, 1, 1, 0)),
    (65536, 32, 16, 2, True, False, True, (1, 128, 32, 1, 1, 1, 0)),
    (65536, 32, 48, 1, True, False, True, (1, 128, 64, 1, 1, 1, 0)),
    (65536, 80, 75, 1, False, True, False, (1, 128, 96, 1, 1, 0, 0)),
    (65536, 16, 96, 0, True, False, True, (1, 128, 96, 1, 1, 1, 0)),
    (65536, 96, 240, 1, False, False, True, (1, 128, 128, 2, 1, 1, 0)),
    (65536, 24, 144, 0, False, False, True, (1, 128, 160, 1, 0, 1, 0)),
    (65536, 24, 144, 0, True, False, True, (1, 128, 160, 1, 1, 1, 0)),
    (65536, 32, 192, 0, False, False, True, (1, 128, 192, 1, 0, 1, 0)),
    (65536, 32, 192, 0, True, False, True, (1, 128, 192, 1, 1, 1, 0)),
    # ... and their plain-store flavours (inference plans: no statistics): found by tests/test_zz_instantiations_gpu.py
    (65536, 24, 144, 0, False, False, False, (1, 128, 160, 1, 0, 0, 0)),
    (65536, 24, 144, 0, True, False, False, (1, 128, 160, 1, 1, 0, 0)),
    (65536, 32, 192, 0, False, False, False, (1, 128, 192, 1, 0, 0, 0)),
    (65536, 32, 192, 0, True, False, False, (1, 128, 192, 1, 1, 0, 0)),
    (65536, 32, 16, 2, True, False, False, (1, 128, 32, 1, 1, 0, 0)),
]
# ---- strided (RegNet projection shortcuts): B, IH, IW, I, O, expected forward instantiation
FWD_STRIDED = [
    (8, 60, 60, 240, 528, (1, 128, 32, 2, 1, 1, 0)),
    (4, 256, 256, 32, 48, (1, 128, 64, 1, 1, 1, 0)),
    (16, 64, 64, 96, 240, (1, 128, 64, 2, 1, 1, 0)),
]
# B, H, W, I, O -> (data gradient, weight gradient) instantiations
BWD_STRIDED = [
    (8, 60, 60, 240, 528, (1, 64, 64, 2, 2, 2, 0), (1, 64, 64, 1, 3, 0, 0)),
]

# ---- data gradient: M, I, O, two-tensor operand, sink activation, accumulate, statistics, expected (with transposed weights)
DGRAD = [
    (65536, 128, 256, True, 2, False, True, (2, 64, 128, 2, 2, 2, 0)),
    (65536, 256, 128, True, 0, True, True, (2, 64, 128, 2, 2, 2, 0)),
    (16384, 512, 256, True, 2, False, True, (2, 64, 128, 2, 2, 2, 0)),
    (16384, 256, 512, True, 2, True, False, (2, 64, 128, 2, 2, 2, 0)),
    (4096, 1280, 512, True, 2, False, True, (2, 64, 128, 2, 2, 2, 0)),
    (4096, 512, 1024, True, 2, False, True, (2, 64, 64, 2, 2, 2, 0)),
    (131072, 64, 128, True, 2, False, True, (2, 64, 64, 2, 2, 2, 0)),
    (65500, 252, 128, True, 1, True, True, (2, 64, 128, 2, 2, 2, 0)),        # ragged M, N
    (4096, 160, 960, True, 2, False, True, (2, 64, 64, 2, 2, 2, 0)),         # 160 = 2.5 column tiles: a wave's whole block lies past N (its
                                                                             # prefetched sink operands are clamped into the row, round 4)
    (65536, 240, 240, True, 1, False, True, (2, 64, 128, 2, 2, 2, 0)),       # contraction (O) 16 modulo 32
    (16384, 528, 528, True, 1, False, True, (2, 64, 128, 2, 2, 2, 0)),
    (65536, 128, 256, False, 2, False, True, (2, 64, 128, 2, 0, 2, 0)),      # one-tensor gradient operand (conv without BatchNorm)
    (65536, 256, 128, False, 0, True, True, (2, 64, 128, 2, 0, 2, 0)),
    (131072, 64, 128, False, 2, False, True, (2, 64, 64, 2, 0, 2, 0)),
    # 128-row data-gradient tiles: only from 2048 such tiles (round 3: 64-row tiles measured faster in the step below that); none of
    # the BASELINE nets reaches them any more, the instantiations stay for larger batches / planes
    (131072, 256, 128, True, 0, True, True, (2, 128, 64, 2, 2, 2, 0)),       # (a sink tile has at most two blocks per wave: 128 x 64)
    (131028, 252, 144, True, 1, False, True, (2, 128, 64, 2, 2, 2, 0)),      # ragged M, N; contraction 16 modulo 32
    (262144, 64, 128, True, 2, False, True, (2, 128, 64, 2, 2, 2, 0)),
    (131072, 256, 128, False, 0, True, True, (2, 128, 64, 2, 0, 2, 0)),
    (262144, 64, 128, False, 2, False, True, (2, 128, 64, 2, 0, 2, 0)),
    (4096, 512, 1024, False, 1, False, True, (2, 64, 64, 2, 0, 2, 0)),
    # register-staged kernels
    (16384, 176, 75, False, 1, False, False, (1, 64, 64, 2, 0, 2, 0)),
    (65536, 32, 192, True, 0, True, False, (1, 128, 32, 1, 2, 2, 0)),
    (65536, 48, 96, True, 1, False, False, (1, 128, 64, 1, 2, 2, 0)),
    (4096, 1024, 75, False, 2, False, True, (1, 128, 64, 2, 0, 2, 0)),
    (65536, 80, 75, False, 1, False, False, (1, 128, 96, 1, 0, 2, 0)),
    (65536, 256, 75, False, 2, False, True, (1, 128, 128, 2, 0, 2, 0)),
    (65536, 144, 32, True, 2, False, True, (1, 128, 160, 1, 2, 2, 0)),
    (65536, 192, 32, True, 2, False, True, (1, 128, 96, 1, 2, 2, 0)),      # a 192-wide SINK runs as two 96-wide column tiles
]

# ---- weight gradient: M, I, O, two-tensor operand, transformed input operand, expected (bm, bn, stages, amode, waves), splits
WGRAD = [
    (65536, 128, 256, True, True, (2, 128, 128, 2, 3, 0, 4), 128),
    (16384, 352, 256, True, False, (2, 128, 128, 2, 1, 0, 4), 40),           # concatenated input: operand as stored
    (4096, 512, 1024, True, True, (2, 128, 128, 2, 3, 0, 4), 8),
    (65536, 256, 128, False, True, (2, 128, 128, 2, 2, 0, 4), 128),           # one-tensor gradient operand
    (65536, 224, 112, False, False, (2, 128, 128, 2, 0, 0, 4), 128),          # ... both operands as stored, ragged tiles both ways
    (16416, 352, 224, True, True, (2, 128, 128, 2, 3, 0, 4), 40),            # ragged 128-tiles in O and I, uneven last split
    (256, 2560, 1280, True, True, (2, 128, 128, 2, 3, 0, 4), 1),             # a single row split
    (65536, 160, 128, True, False, (1, 64, 64, 1, 1, 0, 0), 86),             # 160 leaves a 128-tile 1/4 full: round-1 kernel
    (16384, 96, 576, True, True, (1, 64, 64, 1, 3, 0, 0), 29),
]
# ---- forward with the operand's BatchNorm DERIVED from the producer's accumulators (AMODE 3, tests/test_bn_handover_gpu.py):
# conv A (M, I -> O) feeds conv B (M, O -> C2): stats / bias of conv B, expected instantiation of conv B
HANDOVER = [
    (16384, 32, 192, 64, True, False, (2, 64, 64, 2, 3, 1, 0)),
    (4096, 256, 160, 960, True, False, (2, 128, 64, 2, 3, 1, 0)),
    (65536, 128, 256, 75, False, True, (2, 128, 128, 2, 3, 0, 0)),
    (16384, 128, 256, 512, True, False, (2, 128, 128, 2, 3, 1, 0)),
    (7200, 128, 352, 75, False, True, (2, 64, 64, 2, 3, 0, 0)),
    (16384, 128, 176, 176, True, False, (2, 64, 128, 2, 3, 1, 0)),            # RegNetX neck: behind a grouped 3x3 that hands its statistics over
]
WGRAD_BIAS = [(65536, 256, 75, False, (1, 64, 64, 1, 2, 0, 0))]                  # detection head: bias gradient in the same pass


def _id(case):
    return '%dx%dx%d' % case[:3]


@pytest.mark.parametrize('case', FWD, ids=_id)
def test_conv1x1_fwd_at_benchmark_shapes(case):
    K.check_conv1x1_fwd(*case[:7], expect=case[7])


@pytest.mark.parametrize('case', DGRAD, ids=_id)
def test_conv1x1_dgrad_at_benchmark_shapes(case):
    K.check_conv1x1_dgrad(*case[:7], expect_wt=case[7])


@pytest.mark.parametrize('case', WGRAD, ids=_id)
def test_conv1x1_wgrad_at_benchmark_shapes(case):
    M, I, O, two, xa, expect, splits = case
    K.check_conv1x1_wgrad(M, I, O, two, False, xa=xa, expect=expect)
    a = abi.Conv1x1Wgrad()            # the row-split count is part of what the case was written for
    a.dy.g = a.inp.x = a.dw = 256
    a.dy.y = 256 if two else None
    a.dy.ldg = a.dy.ldy = O
    a.inp.ld, a.inp.act, a.inp.scale = I, (2 if xa else 0), (256 if xa else None)
    a.M, a.I, a.O = M, I, O
    pl = abi.GemmPlan()
    abi.check(abi.lib().pq_conv1x1_wgrad_plan(C.byref(a), C.byref(pl)))
    assert pl.key() == tuple(expect) and pl.splits == splits, (pl.key(), pl.splits)


@pytest.mark.parametrize('case', WGRAD_BIAS, ids=_id)
def test_conv1x1_wgrad_with_bias_at_benchmark_shapes(case):
    M, I, O, two, expect = case
    K.check_conv1x1_wgrad(M, I, O, two, True, expect=expect)


@pytest.mark.parametrize('case', FWD_STRIDED, ids=lambda c: '%dx%dx%dx%d' % c[:4])
def test_conv1x1_fwd_strided_at_benchmark_shapes(case):
    K.check_conv1x1_fwd_strided_rows(*case[:5], act=1, stats=True, expect=case[5])


@pytest.mark.parametrize('case', BWD_STRIDED, ids=lambda c: '%dx%dx%dx%d' % c[:4])
def test_conv1x1_strided_backward_at_benchmark_shapes(case):
    K.check_conv1x1_strided_backward(*case[:5], True, expect=case[5], expect_w=case[6])


@pytest.mark.parametrize('case', HANDOVER, ids=lambda c: '%dx%dx%dx%d' % c[:4])
def test_conv1x1_fwd_with_handed_over_batchnorm_at_benchmark_shapes(case):
    import test_bn_handover_gpu as H
    H.test_conv1x1_to_conv1x1_handover(*case[:4], stats_b=case[4], bias_b=case[5], expect_b=case[6])


def _selected(model_name, size, batch):
    """{(kind, plan key)} over every 1x1 GEMM launch of one train step of the net at (batch, size)."""
    from pqdet_amd.engine import Plan
    from pqdet_amd.model import zoo
    from pqdet_amd.model.interpreter import DetectionModel
    net = DetectionModel(io.StringIO(zoo.cfg_text(model_name))).to('cuda')
    net.train()
    net._ensure_flat(torch.device('cuda', torch.cuda.current_device()))
    plan = Plan(net, (batch, 3, size, size), True, True)
    l = abi.lib()
    found = {}
    for ops in (plan.fwd, plan.bwd):
        for op in ops:
            kind = {abi.OP_CONV1X1_FWD: 'fwd', abi.OP_CONV1X1_DGRAD: 'dgrad', abi.OP_CONV1X1_WGRAD: 'wgrad'}.get(op.code)
            if kind is None:
                continue
            args = getattr(op.u, abi.OP_FIELD[op.code])
            if kind == 'wgrad' and not args.scratch:
                args.scratch = 256                     # scratch pointers are filled in by the engine after the loop
            pl = abi.GemmPlan()
            abi.check(getattr(l, 'pq_conv1x1_%s_plan' % kind)(C.byref(args), C.byref(pl)))
            found.setdefault((kind, pl.key()), (args.M, args.I, args.O))
    del plan, net
    torch.cuda.empty_cache()
    return found


def _covered():
    """The instantiations the parametrised kernel tests reach (their pinned plans + the small-shape lists of test_kernels_gpu)."""
    cov = set()
    for c in FWD:
        cov.add(('fwd', tuple(c[7])))
    for c in DGRAD:
        cov.add(('dgrad', tuple(c[7])))
    for c in WGRAD:
        cov.add(('wgrad', tuple(c[5])))
    for c in WGRAD_BIAS:
        cov.add(('wgrad', tuple(c[4])))
    for c in HANDOVER:
        cov.add(('fwd', tuple(c[6])))
    for c in FWD_STRIDED:
        cov.add(('fwd', tuple(c[5])))
    for c in BWD_STRIDED:
        cov.add(('dgrad', tuple(c[5])))
        cov.add(('wgrad', tuple(c[6])))
    return cov


@pytest.mark.parametrize('model_name,size,batch', [('mobilenetv2-fpn', 512, 16), ('regnetx-600m-fpn', 512, 16),
                                                   ('regnetx-600m-pan', 960, 8)])
def test_every_gemm_instantiation_of_the_baseline_nets_has_a_value_test(model_name, size, batch):
    found = _selected(model_name, size, batch)
    ring = {k: v for k, v in found.items() if k[1][0] == 2}        # LDS-DMA ring kernels (gemm2.hip)
    missing = {k: v for k, v in found.items() if k not in _covered()}
    assert not missing, 'no value test reaches these instantiations (kind, plan) -> first layer shape (M, I, O): %r' % missing
    print('%s: %d ring-kernel instantiations, %d register-staged ones' % (model_name, len(ring), len(found) - len(ring)))
    assert len(ring) >= 6
