"""BASELINE config-2 sizes (mobilenetv2-fpn, 512x512, batch 16; NMS over 16128 x 20 scores) checked through properties that
do not need the CPU oracle to finish: determinism, linearity of the backward in the upstream gradient, score order and
the suppression invariants of NMS at the largest candidate count the evaluator can produce."""
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pqdet_amd import tools  # noqa: E402
from pqdet_amd.dataset import RECOVER_BBOXES_REGISTER, labels  # noqa: E402
from pqdet_amd.model import zoo  # noqa: E402
from pqdet_amd.model.interpreter import DetectionModel  # noqa: E402

DEV = 'cuda'


def _iou_matrix(a, b):
    lt = torch.maximum(a[:, None, :2], b[None, :, :2])
    rb = torch.minimum(a[:, None, 2:4], b[None, :, 2:4])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter)


@pytest.mark.parametrize('model_name,batch,size', [('mobilenetv2-fpn', 16, 512),        # BASELINE config 2
                                                   ('regnetx-600m-fpn', 16, 512),       # BASELINE config 3 (per-GPU shape)
                                                   ('regnetx-600m-pan', 8, 960)])       # BASELINE config 4 (per-GPU shape)
def test_train_step_512_batch16_is_deterministic_and_linear_in_the_upstream_gradient(model_name, batch, size):
    torch.manual_seed(0)
    net = DetectionModel(io.StringIO(zoo.cfg_text(model_name))).to(DEV)
    net.train()
    x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(0)).to(DEV)
    tgt = tuple(t.to(DEV) for t in labels.synthetic_targets(batch, size, 20, seed=1))
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    runs = []
    for scale in (1.0, 1.0, 4.0):            # a power of two: scaling commutes with every fp32 rounding in a linear backward
        net.load_state_dict(sd0)                      # same running statistics going in
        out = net(x, tgt)
        (out['loss'].mean() * scale).backward()
        net.check_nan()
        runs.append((torch.stack([out[k] for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]).clone(),
                     net.flat_buffers()[1].clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])      # bit-identical replays
    assert torch.equal(runs[0][0], runs[2][0])
    g1, g4 = runs[0][1], runs[2][1]
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0
    err = (g4 - 4.0 * g1).abs().max().item()
    assert err <= 1e-6 * g4.abs().max().item()                                                 # backward is linear in dL
    # loss_per_branch sums to the total, the three parts sum to the total
    out = net(x, tgt)
    assert abs(sum(v.item() for v in out['loss_per_branch']) - out['loss'].item()) <= 1e-4 * out['loss'].item()
    parts = out['giou_loss'].item() + out['conf_loss'].item() + out['class_loss'].item()
    assert abs(parts - out['loss'].item()) <= 1e-4 * out['loss'].item()


def test_nms_invariants_at_the_largest_candidate_count():
    """16128 boxes x 20 classes, every score above the threshold (what a random-init model produces):
    322,560 candidates through pq_nms."""
    g = torch.Generator().manual_seed(3)
    n, ncls, thr = 16128, 20, 0.45
    xy = torch.rand(n, 2, generator=g) * 480
    wh = torch.rand(n, 2, generator=g) * 100 + 4
    sc = torch.rand(n, ncls, generator=g) * 0.89 + 0.11
    boxes = torch.cat([xy, xy + wh, sc], 1).to(DEV)
    dets = tools.torch_nms(boxes, 0.1, thr)
    k = dets.shape[0]
    assert dets.shape[1] == 6 and 0 < k < n * ncls
    score, cls = dets[:, 4], dets[:, 5].long()
    assert bool((score[:-1] >= score[1:]).all())                                   # score-descending
    assert bool((score > 0.1).all()) and int(cls.min()) >= 0 and int(cls.max()) < ncls
    # every kept row is one of the inputs with its own class score
    src = boxes[:, :4]
    for c in (0, 7, 19):
        mine = dets[cls == c]
        # no two kept boxes of a class overlap by more than the threshold
        iou = _iou_matrix(mine[:, :4], mine[:, :4])
        iou.fill_diagonal_(0)
        # (batched_nms compares IoUs of class-offset coordinates, x + cls*(max+1): float32 rounding of the shifted corners
        # moves an IoU by ~1e-5, so both invariants are checked with that slack)
        assert float(iou.max()) <= thr + 1e-4
        # every suppressed candidate of the class overlaps a kept box of higher (or equal) score by more than the threshold
        cand_scores = boxes[:, 4 + c]
        kept_mask = torch.zeros(n, dtype=torch.bool, device=DEV)
        match = (mine[:, None, :4] == src[None, :, :]).all(-1) & (mine[:, 4:5] == cand_scores[None, :])
        kept_idx = match.float().argmax(1)
        assert bool(match.any(1).all())
        kept_mask[kept_idx] = True
        sup = (~kept_mask).nonzero()[:, 0]
        sub = sup[torch.randperm(len(sup), generator=torch.Generator().manual_seed(c))[:2000].to(DEV)]
        iou2 = _iou_matrix(src[sub], mine[:, :4])
        ok = ((iou2 > thr - 1e-4) & (mine[None, :, 4] >= cand_scores[sub][:, None])).any(1)
        assert bool(ok.all())
    assert len(mine) > 0


def test_eval_pipeline_512_batch16_shapes_and_ranges():
    torch.manual_seed(1)
    net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn'))).to(DEV)
    net.eval()
    x = torch.randn(16, 3, 512, 512, generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        pred = net(x)
    assert pred.shape == (16, 16128, 25)
    assert bool(((pred[..., 4:] >= 0) & (pred[..., 4:] <= 1)).all())               # sigmoid outputs
    assert bool((pred[..., 2] > pred[..., 0]).all()) and bool((pred[..., 3] > pred[..., 1]).all())   # xmax > xmin (exp > 0)
    shapes = torch.tensor([[375., 500.]] * 16)
    rec = RECOVER_BBOXES_REGISTER['voc'](pred.clone(), torch.FloatTensor([512, 512]).to(DEV), shapes.to(DEV))
    assert rec.shape == (16, 16128, 24)
    # base_sample.py:98-139: only the top-left corner is clamped at 0 and only the bottom-right one at (orig - 1)
    assert float(rec[..., :2].min()) >= 0 and float(rec[..., 2].max()) <= 499 and float(rec[..., 3].max()) <= 374
    again = RECOVER_BBOXES_REGISTER['voc'](pred.clone(), torch.FloatTensor([512, 512]).to(DEV), shapes.to(DEV))
    assert torch.equal(rec, again)


def _relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _oracle_pair(name, seed):
    from oracle.net import OracleNet
    from seedfill import seeded_fill
    text = zoo.cfg_text(name)
    net = DetectionModel(io.StringIO(text))
    seeded_fill(net, seed)
    ref = OracleNet(io.StringIO(text))
    ref.load_state_dict(net.state_dict())
    return net.to(DEV), ref


def _losses_close(out, want, tol=1e-3):
    for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss'):
        assert abs(out[k].item() - want[k].item()) <= tol * abs(want[k].item()) + 1e-5, (k, out[k].item(), want[k].item())
    for i, v in enumerate(out['loss_per_branch']):
        assert abs(v.item() - want['loss_per_branch'][i].item()) <= tol * abs(want['loss_per_branch'][i].item()) + 1e-5


@pytest.mark.parametrize('name,B,S', [('mobilenetv2-fpn', 16, 512),          # BASELINE config 2 in full
                                      ('regnetx-600m-fpn', 16, 512),         # BASELINE config 3, per-GPU shape in full
                                      ('regnetx-600m-pan', 8, 960)])         # BASELINE config 4, per-GPU shape in full
def test_full_size_train_step_against_the_cpu_oracle(name, B, S):
    """BASELINE config 2 at its FULL size -- mobilenetv2-fpn, 512 x 512, batch 16, one TRAIN-mode step on the synthetic batch
    bench.py times -- against the CPU oracle in fp32 and fp64 (round 5: the same for configs 3 and 4, whose fused / folded plans
    differ: grouped 3x3 kernels, aliased ReLU shortcuts).  The only place where the 128-wide GEMM tiles, the streaming
    depthwise kernels and the 1M-row fused backward meet the reference's arithmetic end to end at the benchmark's shapes.
    Losses: 1e-3 (north_star's fp32 bar).  Gradients: train-mode BatchNorm + the discontinuous ignore mask + ReLU6 kinks make
    the step ill-conditioned even at this size -- the reference's own fp32 evaluation is 6e-2 (norm-wise, worst parameter)
    away from its fp64 evaluation (measured; printed below) -- so "5e-3 of the fp32 oracle" cannot hold for two fp32
    evaluations; what is asserted is that the HIP step is as close to fp64 as the reference's fp32 step is: per parameter
    e_hip <= max(5 e_ref, 2e-3) with at most 8 kink cases (<= max(10 e_ref, 2e-2)), median e_hip <= 2 x median e_ref, and
    >= 95 % of the parameters at e_hip <= 3 e_ref + 2e-3.  The well-conditioned 5e-3 check at this size is the next test."""
    import copy
    import numpy as np
    net, ref = _oracle_pair(name, 2)
    ref64 = copy.deepcopy(ref).double()
    net.train(), ref.train(), ref64.train()
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(0))
    tgt = labels.synthetic_targets(B, S, 20, seed=1)
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    out['loss'].mean().backward()
    net.check_nan()
    want = ref(x, tgt)
    want['loss'].mean().backward()
    w64 = ref64(x.double(), tuple(t.double() for t in tgt))
    w64['loss'].mean().backward()
    _losses_close(out, want)
    e_hip, e_ref, kinks = [], [], []
    for (n1, p1), (_, p2), (_, p3) in zip(net.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        if p3.grad.norm().item() < 1e-6:
            continue
        eh, er = _relerr(p1.grad, p3.grad), _relerr(p2.grad, p3.grad)
        e_hip.append(eh), e_ref.append(er)
        if eh > max(5 * er, 2e-3):
            kinks.append((n1, eh, er))
            assert eh <= max(10 * er, 2e-2), (n1, eh, er)
    e_hip, e_ref = np.array(e_hip), np.array(e_ref)
    print('full-size train step vs fp64: HIP median %.2e / max %.2e, reference fp32 median %.2e / max %.2e over %d parameters'
          % (np.median(e_hip), e_hip.max(), np.median(e_ref), e_ref.max(), len(e_hip)))
    assert len(kinks) <= 8, kinks
    assert np.median(e_hip) <= 2 * np.median(e_ref) + 1e-4
    assert (e_hip <= 3 * e_ref + 2e-3).mean() >= 0.95
    for (n1, b1), (_, b2) in zip(net.named_buffers(), ref.named_buffers()):
        if b1.is_floating_point():
            assert _relerr(b1, b2) < 1e-4, n1                  # running statistics after the step


@pytest.mark.parametrize('name,B,S,seed', [('mobilenetv2-fpn', 16, 512, 2), ('regnetx-600m-fpn', 16, 512, 3), ('regnetx-600m-pan', 8, 960, 5)])
def test_full_size_gradients_with_frozen_statistics_against_the_cpu_oracle(name, B, S, seed):
    """The WELL-CONDITIONED end-to-end check at the benchmark's layer shapes -- BASELINE configs 2, 3 and 4 at their full per-GPU
    sizes (the CPU oracle runs on the job's 16 cores: tests/conftest.py caps its thread pool at the cgroup quota):
    ``model.eval()`` + targets, i.e. BatchNorm on its running statistics (model/interpreter.py:72-85) like fixture
    G5b, so fp32 round-off is not amplified through batch statistics.  Losses 1e-3; >= 95 % of the parameters within 5e-3
    norm-wise of the fp32 oracle's gradient, none further than 5e-2 (a flipped ignore-mask cell / ReLU kink)."""
    net, ref = _oracle_pair(name, seed)
    net.eval(), ref.eval()
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(seed))
    tgt = labels.synthetic_targets(B, S, 20, seed=seed + 1)
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    out['loss'].mean().backward()
    net.check_nan()
    want = ref(x, tgt)
    want['loss'].mean().backward()
    _losses_close(out, want)
    errs = []
    for (n1, p1), (_, p2) in zip(net.named_parameters(), ref.named_parameters()):
        if p2.grad.norm().item() >= 1e-6:
            errs.append((_relerr(p1.grad, p2.grad), n1))
    errs.sort(reverse=True)
    print('%s %dx%d batch %d, frozen statistics: worst gradient errors vs the fp32 oracle %s' % (name, S, S, B, errs[:4]))
    assert errs[0][0] <= 5e-2, errs[:4]
    assert sum(e <= 5e-3 for e, _ in errs) >= 0.95 * len(errs), errs[:10]


def test_config4_train_step_960_losses_against_the_cpu_oracle():
    """regnetx-600m-pan, 960 x 960 (config 4's resolution), train mode: losses and running statistics against the fp32 oracle."""
    net, ref = _oracle_pair('regnetx-600m-pan', 5)
    net.train(), ref.train()
    x = torch.randn(8, 3, 960, 960, generator=torch.Generator().manual_seed(4))          # config 4's batch per GPU
    tgt = labels.synthetic_targets(8, 960, 20, seed=6)
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    out['loss'].mean().backward()
    net.check_nan()
    with torch.no_grad():
        want = ref(x, tgt)
    _losses_close(out, want)
    for (n1, b1), (_, b2) in zip(net.named_buffers(), ref.named_buffers()):
        if b1.is_floating_point():
            assert _relerr(b1, b2) < 1e-4, n1
