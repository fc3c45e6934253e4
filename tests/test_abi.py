"""The C-ABI shared library loads and exports everything include/pqdet_hip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from pqdet_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return abi.lib()


def test_every_declared_symbol_is_exported(lib):
    text = open(os.path.join(ROOT, 'include', 'pqdet_hip.h')).read()
    declared = set(re.findall(r'\b(pq_[a-z0-9_]+)\s*\(', text))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), name
        assert name in abi.SIGNATURES, 'binding missing for %s' % name
    assert set(abi.SIGNATURES) == declared


def test_struct_layouts_match_the_compiler(lib):
    pairs = {0: abi.Op, abi.OP_CONV1X1_FWD: abi.Conv1x1Fwd, abi.OP_CONV1X1_DGRAD: abi.Conv1x1Dgrad,
             abi.OP_CONV1X1_WGRAD: abi.Conv1x1Wgrad, abi.OP_DW_FWD: abi.DwFwd, abi.OP_DW_DGRAD: abi.DwDgrad,
             abi.OP_DW_WGRAD: abi.DwWgrad, abi.OP_STEM_FWD: abi.StemFwd, abi.OP_STEM_WGRAD: abi.StemWgrad,
             abi.OP_BN_FWD_FIN: abi.BnFwdFin, abi.OP_BN_BWD_FIN: abi.BnBwdFin, abi.OP_COPY_FWD: abi.Copy,
             abi.OP_ROUTE_BWD: abi.RouteBwd, abi.OP_DECODE: abi.Decode, abi.OP_LOSS_FWD: abi.YoloLoss,
             abi.OP_ADAM: abi.Adam, abi.OP_FILL: abi.Fill, abi.OP_DW_BWD: abi.DwBwd, abi.OP_GCONV_FWD: abi.GconvFwd,
             abi.OP_GCONV_DGRAD: abi.GconvDgrad, abi.OP_GCONV_WGRAD: abi.GconvWgrad, abi.OP_TRANSPOSE: abi.Transpose,
             103: abi.TransposeItem, 104: abi.CreateLabel, 105: abi.QatWeightItem, abi.OP_QAT_WPREP: abi.QatWeight,
             abi.OP_QACT_FWD: abi.Qact, abi.OP_QACT_BWD: abi.QactBwd, abi.OP_POOL_REDUCE: abi.PoolReduce,
             abi.OP_CONV1X1_BWD: abi.Conv1x1Bwd, abi.OP_BN_EVAL: abi.BnEval, abi.OP_LAYER_FWD: abi.LayerFwd, 106: abi.BnEvalItem, 107: abi.BnStat, 108: abi.Src, 109: abi.BnBstat, 100: abi.Recover, 101: abi.Nms, 102: abi.DecodeBwd,
             abi.OP_XDW_FWD: abi.XdwFwd, abi.OP_XSTAT_FWD: abi.XstatFwd, abi.OP_XDW_BWD: abi.XdwBwd,
             abi.OP_REDUCE_BATCH: abi.ReduceBatch, 110: abi.ReduceItem}
    for code, st in pairs.items():
        assert lib.pq_abi_sizeof(code) == C.sizeof(st), (code, st.__name__)


def test_argument_validation_needs_no_gpu(lib):
    a = abi.Conv1x1Fwd()
    assert lib.pq_conv1x1_fwd(C.byref(a), None) == -1
    assert b'null pointer' in lib.pq_last_error()
    assert lib.pq_version().startswith(b'pqdet_hip')
    assert lib.pq_yolo_loss_blocks(64, 64, 3) == 48
    assert lib.pq_conv1x1_wgrad_scratch_bytes(4096, 160, 960) > 0
    assert lib.pq_conv1x1_bwd_blocks(1 << 20, 16, 96) > 0 and lib.pq_conv1x1_bwd_blocks(4096, 160, 960) == 0
    assert lib.pq_conv1x1_bwd_pays(1 << 20, 16, 96) == 1 and lib.pq_conv1x1_bwd_pays(4096, 16, 96) == 0
    assert lib.pq_nms_scratch_bytes(16128, 20) > 16128 * 20 * 56


def test_sink_addend_is_validated_without_a_gpu(lib):
    """pq_gsink.addend (folded shortcuts): read through float4 loads by an ACCUMULATING sink only -- a misaligned address or a
    non-accumulating sink is refused by the entry point, before any launch."""
    import torch
    buf = torch.zeros(64)
    base = buf.data_ptr()
    for addend, acc in ((base + 4, 1), (base, 0)):
        sk = abi.GSink(base, base, None, None, None, None, None, 4, 4, 0, acc, 0, 0)
        sk.addend = addend
        r = abi.RouteBwd(base, sk, 1, 1, 4, 4, 4, 0, 1)
        assert lib.pq_route_bwd(C.byref(r), None) == -1
        assert b'addend' in lib.pq_last_error()
