"""the on-disk checkpoint format of the reference is a drop-in here.

The layouts (top-level keys, the state dict's key / shape / dtype list with and without DataParallel's ``module.``
prefix, a backbone-prefix file) were recorded from the reference's own ``Trainer.save`` (``tests/golden/g12_checkpoints.json``,
generator ``make_golden_r2.py``); the files below are written in exactly that layout and loaded through ``tools.build_model``
(``tools.py:158-243``), ``load_weight`` / ``load_backbone`` (``:120-156``).  No GPU needed: weights are loaded on the CPU."""
import io
import json
import os
import subprocess
import sys

import pytest
import torch

from pqdet_amd import tools
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
from seedfill import seeded_fill

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _describe(sd):
    return [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()]


def _ckpt(sd, cfg_text, kind='normal', backend='none', step=1234):
    return {'step': step, 'AP': None, 'model': sd, 'cfg': cfg_text, 'type': kind, 'backend': backend}


def test_state_dict_layout_is_the_reference_checkpoint_layout(g12):
    net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn')))
    assert _describe(net.state_dict()) == g12['bare']['model']
    assert [['module.' + k, s, d] for k, s, d in _describe(net.state_dict())] == g12['dp']['model']
    assert set(g12['bare']['top']) == {'step', 'AP', 'model', 'cfg', 'type', 'backend'}
    assert g12['bare']['type'] == 'normal' and g12['bare']['backend'] == 'none'


@pytest.mark.parametrize('prefix', ['', 'module.'])
def test_build_model_resumes_reference_style_checkpoint(tmp_path, g12, prefix):
    cfg_text = zoo.cfg_text('mobilenetv2-fpn')
    src = DetectionModel(io.StringIO(cfg_text))
    seeded_fill(src, 3)
    sd = {prefix + k: v.clone() for k, v in src.state_dict().items()}
    assert [k for k, _, _ in g12['dp' if prefix else 'bare']['model']] == list(sd)
    path = str(tmp_path / 'model-7.pt')
    torch.save(_ckpt(sd, cfg_text), path)
    # cfg rebuilt from the checkpoint text (reference :205-209)
    model, info = tools.build_model(None, path, device='cpu', dataparallel=False)
    assert info == {'step': 1234, 'AP': None, 'cfg': cfg_text, 'type': 'normal', 'backend': 'none'}
    for (k, a), b in zip(src.state_dict().items(), model.state_dict().values()):
        assert torch.equal(a, b), k
    # explicit cfg path + clear_history (reference :200-202)
    cfg_file = tmp_path / 'net.cfg'
    cfg_file.write_text(cfg_text)
    model2, info2 = tools.build_model(str(cfg_file), path, clear_history=True, device='cpu', dataparallel=False)
    assert info2['step'] == 0
    assert torch.equal(model2.state_dict()['module_list.0.conv.weight'], src.state_dict()['module_list.0.conv.weight'])
    # a checkpoint whose keys do not fit the cfg is rejected like nn.Module.load_state_dict does
    bad = dict(sd)
    bad.pop(next(iter(bad)))
    torch.save(_ckpt(bad, cfg_text), path)
    with pytest.raises(RuntimeError):
        tools.build_model(None, path, device='cpu', dataparallel=False)


@pytest.mark.parametrize('prefix', ['', 'module.'])
def test_backbone_prefix_file(tmp_path, g12, prefix):
    cfg_text = zoo.cfg_text('mobilenetv2-fpn')
    donor = DetectionModel(io.StringIO(cfg_text))
    seeded_fill(donor, 9)
    want_keys = [k for k, _, _ in g12['backbone61']['model']]
    assert g12['backbone61']['n_keys'] == len(want_keys) and want_keys[0] == g12['backbone61']['first']
    full = donor.state_dict()
    bb = {prefix + k: full[k].clone() for k in want_keys}            # layers 0..61, cut by convert.partial's rule
    path = str(tmp_path / 'backbone.pt')
    torch.save(bb, path)
    cfg_file = tmp_path / 'net.cfg'
    cfg_file.write_text(cfg_text)
    torch.manual_seed(0)
    fresh = DetectionModel(io.StringIO(cfg_text)).state_dict()
    torch.manual_seed(0)
    model, info = tools.build_model(str(cfg_file), None, path, device='cpu', dataparallel=False)
    assert info == {}
    got = model.state_dict()
    for k in got:
        if k in want_keys:
            assert torch.equal(got[k], full[k]), k                   # taken from the backbone file
        else:
            assert torch.equal(got[k], fresh[k]), k                  # untouched (same seed, same init)


def test_qat_checkpoint_written_by_the_trainer_resumes(tmp_path):
    """trainer.py:194-201 stores ``self.model.state_dict()`` with 'type': 'qat'; tools.py:229-238 prepares a QAT model and
    loads it back.  A prepared model therefore presents (and accepts) the reference's QAT key layout."""
    names = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g9_names.json')))
    cfg_text = zoo.cfg_text('mobilenetv2-fpn')
    net = DetectionModel(io.StringIO(cfg_text), quant=True)
    seeded_fill(net, 0)
    net.train()
    tools.prepare_qat(tools.fuse_model(net), backend='qnnpack')
    with torch.no_grad():                                   # make every observer row distinct
        st = net._qat_state
        st[:, 0] = -torch.arange(1, st.shape[0] + 1, dtype=torch.float32) / 7
        st[:, 1] = torch.arange(1, st.shape[0] + 1, dtype=torch.float32) / 3
        st[:, 2] = 0.01 + torch.arange(st.shape[0], dtype=torch.float32) * 1e-4
        st[:, 3] = torch.arange(st.shape[0], dtype=torch.float32) % 200
    tools.disable_observer(net)
    sd = net.state_dict()
    assert _describe(sd) == [[k, s, d.replace('torch.', '')] for k, s, d in names['state_dict']]       # the reference's layout
    path = str(tmp_path / 'model-3.pt')
    torch.save(_ckpt(sd, cfg_text, 'qat', 'qnnpack', step=55), path)
    model, info = tools.build_model(None, path, device='cpu', dataparallel=False)                       # type read from the file
    assert info['type'] == 'qat' and info['step'] == 55 and model._qat and not model.qat_observer_enabled
    assert torch.equal(model._qat_state, net._qat_state)
    for (k, a), b in zip(net.plain_state_dict().items(), model.plain_state_dict().values()):
        assert torch.equal(a, b), k
    # 'module.'-prefixed QAT dict (a DataParallel-wrapped trainer would write that)
    torch.save(_ckpt({'module.' + k: v for k, v in sd.items()}, cfg_text, 'qat', 'qnnpack'), path)
    model3, _ = tools.build_model(None, path, device='cpu', dataparallel=False)
    assert torch.equal(model3._qat_state, net._qat_state)
    # the package's own layout is accepted too; an fbgemm checkpoint is refused up front
    model3.load_state_dict(net.plain_state_dict())
    torch.save(_ckpt(sd, cfg_text, 'qat', 'fbgemm'), path)
    with pytest.raises(NotImplementedError, match='fbgemm'):
        tools.build_model(None, path, device='cpu', dataparallel=False)


def test_launcher_aliases_resolve_reference_import_names():
    code = ('from pqdet_amd import launch; launch.install_aliases(); import tools, model.interpreter, model.parser, model.loss, '
            'dataset, eval.evaluator; import pqdet_amd.tools as t; assert tools is t; '
            'from model.interpreter import DetectionModel; from model.loss import loss_per_scale; print("ok")')
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert p.returncode == 0 and p.stdout.strip() == 'ok', p.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('nproc', [1, 2])
def test_reference_style_driver_runs_unmodified_under_the_launcher(tmp_path, nproc):
    """``python -m pqdet_amd.launch --nproc N tests/ref_style_driver.py``: the driver uses only the reference's names; with two
    ranks (both on the one test GPU, gloo) build_model wraps the model for gradient exchange and the replicas stay identical."""
    cfg_file = tmp_path / 'net.cfg'
    cfg_file.write_text(zoo.cfg_text('mobilenetv2-fpn'))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    cmd = [sys.executable, '-m', 'pqdet_amd.launch', '--nproc', str(nproc), '--backend', 'gloo',
           os.path.join(ROOT, 'tests', 'ref_style_driver.py'), str(cfg_file), str(tmp_path), '3']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert sorted(d['rank'] for d in lines) == list(range(nproc))
    for d in lines:
        assert d['step'] == 3 and d['resumed_equal'] and len(d['loss']) == 3 and all(v == v for v in d['loss'])
        assert d['wrapped'] == ('DistributedModel' if nproc > 1 else 'DetectionModel')
        assert d['keys_have_module_prefix'] == (nproc > 1)
    if nproc > 1:
        assert lines[0]['param_checksum'] == lines[1]['param_checksum']      # replicas in lock-step after 3 exchanged steps
