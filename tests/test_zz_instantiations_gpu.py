"""Every kernel instantiation that a train / inference step of a BASELINE net launches is also launched by a kernel-level VALUE
test of this suite.

The library notes "kernel expression | enclosing function" at every launch while its launch log is on (pq_launch_log; the pretty
name of a templated dispatcher carries the template arguments).  conftest.py switches the log on around every test of the
kernel-level modules (test_kernels_gpu, test_irblock_gpu, test_gemm_variants_gpu, test_bn_handover_gpu, test_qat_gpu) and
collects what they launched; this module -- named to run last -- replays one full-size step of each BASELINE configuration with
the log on and requires the step's set to be a subset.  A dispatch rule that sends a benchmark shape to an instantiation no value
test reaches (the round-4 fault lived in such a path) fails here, for EVERY kernel family: 1x1 GEMMs, grouped / depthwise /
stem convolutions, the fused narrow backward, the fused inverted-residual kernels, reductions, loss, optimiser.
(tests/test_gemm_variants_gpu.py additionally pins the run-time parameters of the GEMM plans: stages, splits.)
"""
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

import conftest  # noqa: E402
from pqdet_amd import abi  # noqa: E402


def _step_launches(model_name, size, batch, qat=False, train=True):
    from pqdet_amd import optim, tools
    from pqdet_amd.dataset import labels
    from pqdet_amd.model import zoo
    from pqdet_amd.model.interpreter import DetectionModel
    torch.manual_seed(0)
    net = DetectionModel(io.StringIO(zoo.cfg_text(model_name)), quant=qat).to('cuda')
    net.train()
    if qat:
        tools.prepare_qat(tools.fuse_model(net), backend='qnnpack')
    x = torch.randn(batch, 3, size, size, device='cuda')
    l = abi.lib()
    if train:
        opt = optim.Adam(net.parameters(), lr=4e-4, weight_decay=0.0)
        tgt = tuple(t.to('cuda') for t in labels.synthetic_targets(batch, size, 20, seed=1))
        l.pq_launch_log(1)
        opt.zero_grad()
        out = net(x, tgt)
        out['loss'].mean().backward()
        opt.step()
    else:
        net.eval()
        l.pq_launch_log(1)
        with torch.no_grad():
            net(x)
    torch.cuda.synchronize()
    got = conftest.dump_launch_log()
    l.pq_launch_log(0)
    del net, x
    torch.cuda.empty_cache()
    return got


CONFIGS = [('mobilenetv2-fpn', 512, 16, False, True), ('regnetx-600m-fpn', 512, 16, False, True), ('regnetx-600m-pan', 960, 8, False, True),
           ('mobilenetv2-fpn', 512, 12, True, True), ('mobilenetv2-fpn', 512, 16, False, False)]


@pytest.mark.parametrize('model_name,size,batch,qat,train', CONFIGS, ids=lambda v: str(v))
def test_every_kernel_a_baseline_step_launches_is_launched_by_a_value_test(model_name, size, batch, qat, train):
    if len(conftest.LAUNCHES_COVERED) < 50:
        pytest.skip('needs the kernel-level modules of the suite to have run in this session (run the whole GPU suite)')
    got = _step_launches(model_name, size, batch, qat, train)
    assert len(got) > 10
    missing = sorted(got - conftest.LAUNCHES_COVERED)
    assert not missing, '%d instantiations of this step are reached by no kernel-level value test:\n%s' % (len(missing), '\n'.join(missing))
    print('%s %dx%d batch %d%s%s: %d distinct launches, all covered (%d in the kernel tests)' % (
        model_name, size, size, batch, ' QAT' if qat else '', '' if train else ' inference', len(got), len(conftest.LAUNCHES_COVERED)))
