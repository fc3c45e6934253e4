"""ctypes binding of ``libpqdet_hip.so`` (C-ABI declared in ``include/pqdet_hip.h``).

The library is built in-tree by :mod:`pqdet_amd.build` (hipcc, gfx950).  There is NO fallback: if the shared object is
missing or a symbol cannot be resolved, importing/using the compute path raises.  Structures below mirror the header
field for field; ``tests/test_abi.py`` checks every declared symbol is exported and that the struct sizes agree with
what the C compiler sees.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PQDET_HIP_LIB') or os.path.join(_HERE, 'lib', 'libpqdet_hip.so')   # env: perf experiments only

c_fp = C.c_void_p          # device float* (raw address)
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float

ACT_LINEAR, ACT_RELU, ACT_RELU6, ACT_LEAKY, ACT_LOGISTIC, ACT_TANH = range(6)
BOX_L1, BOX_GIOU, BOX_DIOU, BOX_CIOU, BOX_IOU = range(5)
BOX_KIND = {'l1': BOX_L1, 'giou': BOX_GIOU, 'diou': BOX_DIOU, 'ciou': BOX_CIOU, 'iou': BOX_IOU}


class BnStat(C.Structure):
    _fields_ = [('acc', c_fp), ('bad', c_fp), ('gamma', c_fp), ('beta', c_fp), ('running_mean', c_fp), ('running_var', c_fp),
                ('num_batches_tracked', c_fp), ('mean', c_fp), ('invstd', c_fp), ('scale', c_fp), ('shift', c_fp),
                ('slots', i32), ('ldp', i32), ('count', f32), ('eps', f32), ('momentum', f32), ('inv_count', f32)]


class BnBstat(C.Structure):
    _fields_ = [('acc', c_fp), ('bad', c_fp), ('gamma', c_fp), ('invstd', c_fp), ('dgamma', c_fp), ('dbeta', c_fp),
                ('p', c_fp), ('q', c_fp), ('r', c_fp), ('slots', i32), ('ldp', i32), ('count', f32), ('train', i32)]


class Src(C.Structure):
    _fields_ = [('x', c_fp), ('scale', c_fp), ('shift', c_fp), ('ld', i32), ('act', i32), ('qstate', c_fp), ('stat', BnStat)]


class GSrc(C.Structure):
    _fields_ = [('g', c_fp), ('y', c_fp), ('p', c_fp), ('q', c_fp), ('mu', c_fp), ('r', c_fp), ('ldg', i32), ('ldy', i32)]


class GSink(C.Structure):
    _fields_ = [('g', c_fp), ('y', c_fp), ('scale', c_fp), ('shift', c_fp), ('mean', c_fp), ('invstd', c_fp),
                ('partial', c_fp), ('ldg', i32), ('ldy', i32), ('act', i32), ('accumulate', i32), ('finalize', i32),
                ('ldp', i32), ('qstate', c_fp), ('stat_acc', c_fp), ('stat_bad', c_fp), ('stat_slots', i32), ('pad_', i32), ('addend', c_fp)]


class Conv1x1Fwd(C.Structure):
    _fields_ = [('inp', Src), ('w', c_fp), ('bias', c_fp), ('out', c_fp), ('partial', c_fp),
                ('M', i32), ('I', i32), ('O', i32), ('ldo', i32), ('ldp', i32),
                ('sub', i32), ('OH', i32), ('OW', i32), ('IH', i32), ('IW', i32), ('minmax', i32),
                ('stat_acc', c_fp), ('stat_bad', c_fp), ('stat_slots', i32)]


class Conv1x1Dgrad(C.Structure):
    _fields_ = [('dy', GSrc), ('w', c_fp), ('dx', GSink), ('M', i32), ('I', i32), ('O', i32),
                ('sub', i32), ('OH', i32), ('OW', i32), ('IH', i32), ('IW', i32), ('wt', c_fp), ('dy_stat', BnBstat)]


class Conv1x1Wgrad(C.Structure):
    _fields_ = [('dy', GSrc), ('inp', Src), ('dw', c_fp), ('dbias', c_fp), ('scratch', c_fp),
                ('M', i32), ('I', i32), ('O', i32),
                ('sub', i32), ('OH', i32), ('OW', i32), ('IH', i32), ('IW', i32), ('dy_stat', BnBstat)]


class Conv1x1Bwd(C.Structure):
    _fields_ = [('dy', GSrc), ('w', c_fp), ('inp', Src), ('dx', GSink), ('dw', c_fp), ('scratch', c_fp),
                ('M', i32), ('I', i32), ('O', i32), ('dy_stat', BnBstat)]


class GemmPlan(C.Structure):
    _fields_ = [('family', i32), ('bm', i32), ('bn', i32), ('stages', i32), ('amode', i32), ('emode', i32), ('splits', i32),
                ('variant', i32)]

    def key(self):
        return (self.family, self.bm, self.bn, self.stages, self.amode, self.emode, self.variant)


class GconvFwd(C.Structure):
    _fields_ = [('inp', Src), ('w', c_fp), ('out', c_fp), ('partial', c_fp),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('C', i32), ('gw', i32), ('stride', i32),
                ('ldo', i32), ('ldp', i32), ('stat_acc', c_fp), ('stat_bad', c_fp), ('stat_slots', i32)]


class GconvDgrad(C.Structure):
    _fields_ = [('dy', GSrc), ('w', c_fp), ('dx', GSink),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('C', i32), ('gw', i32), ('stride', i32)]


class GconvWgrad(C.Structure):
    _fields_ = [('dy', GSrc), ('inp', Src), ('dw', c_fp), ('scratch', c_fp),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('C', i32), ('gw', i32), ('stride', i32)]


class DwFwd(C.Structure):
    _fields_ = [('inp', Src), ('w', c_fp), ('out', c_fp), ('partial', c_fp),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('C', i32), ('stride', i32),
                ('ldo', i32), ('ldp', i32), ('minmax', i32), ('stat_acc', c_fp), ('stat_bad', c_fp), ('stat_slots', i32)]


class DwDgrad(C.Structure):
    _fields_ = [('dy', GSrc), ('w', c_fp), ('dx', GSink),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('C', i32), ('stride', i32)]


class DwWgrad(C.Structure):
    _fields_ = [('dy', GSrc), ('inp', Src), ('dw', c_fp), ('scratch', c_fp),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('C', i32), ('stride', i32)]


class DwBwd(C.Structure):
    _fields_ = [('dy', GSrc), ('inp', Src), ('w', c_fp), ('dx', GSink), ('dw', c_fp), ('scratch', c_fp),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('C', i32), ('stride', i32), ('part', i32),
                ('dy_stat', BnBstat)]


class StemFwd(C.Structure):
    _fields_ = [('image', c_fp), ('w', c_fp), ('out', c_fp), ('partial', c_fp),
                ('B', i32), ('CI', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('O', i32),
                ('stride', i32), ('ldo', i32), ('ldp', i32), ('minmax', i32)]


class StemWgrad(C.Structure):
    _fields_ = [('dy', GSrc), ('image', c_fp), ('dw', c_fp), ('scratch', c_fp),
                ('B', i32), ('CI', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('O', i32), ('stride', i32)]


class BnFwdFin(C.Structure):
    _fields_ = [('partial', c_fp), ('gamma', c_fp), ('beta', c_fp), ('running_mean', c_fp), ('running_var', c_fp),
                ('num_batches_tracked', c_fp), ('mean', c_fp), ('invstd', c_fp), ('scale', c_fp), ('shift', c_fp),
                ('nblk', i32), ('ldp', i32), ('C', i32), ('train', i32), ('count', f32), ('eps', f32), ('momentum', f32),
                ('scratch', c_fp), ('tickets', c_fp), ('sf', c_fp),
                ('planes', i32), ('act', i32), ('qstate', c_fp), ('mm', c_fp), ('avg_const', f32)]


class BnEvalItem(C.Structure):
    _fields_ = [('gamma', c_fp), ('beta', c_fp), ('running_mean', c_fp), ('running_var', c_fp), ('mean', c_fp), ('invstd', c_fp),
                ('scale', c_fp), ('shift', c_fp), ('sf', c_fp), ('eps', f32), ('C', i32)]


class BnEval(C.Structure):
    _fields_ = [('items', c_fp), ('n_items', i32)]


class BnBwdFin(C.Structure):
    _fields_ = [('partial', c_fp), ('gamma', c_fp), ('invstd', c_fp), ('dgamma', c_fp), ('dbeta', c_fp),
                ('p', c_fp), ('q', c_fp), ('r', c_fp), ('nblk', i32), ('ldp', i32), ('C', i32), ('train', i32),
                ('count', f32), ('scratch', c_fp), ('tickets', c_fp)]


class Copy(C.Structure):
    _fields_ = [('a', Src), ('b', Src), ('out', c_fp),
                ('B', i32), ('H', i32), ('W', i32), ('C', i32), ('ldo', i32), ('coff', i32), ('up', i32),
                ('gate', c_fp), ('ldgate', i32)]


class PoolReduce(C.Structure):
    _fields_ = [('a', Src), ('gout', c_fp), ('ldgo', i32), ('gate', c_fp), ('ldgate', i32), ('out', c_fp), ('ldo', i32),
                ('B', i32), ('HW', i32), ('C', i32), ('mul', f32)]


class RouteBwd(C.Structure):
    _fields_ = [('gout', c_fp), ('dx', GSink),
                ('B', i32), ('H', i32), ('W', i32), ('C', i32), ('ldgo', i32), ('coff', i32), ('up', i32),
                ('gate', c_fp), ('ldgate', i32), ('bcast', i32), ('mul', f32), ('dx2', GSink)]


class HeadView(C.Structure):
    _fields_ = [('base', c_fp), ('sb', i64), ('sh', i64), ('sw', i64), ('sc', i64)]


class Decode(C.Structure):
    _fields_ = [('raw', HeadView), ('out', c_fp),
                ('B', i32), ('H', i32), ('W', i32), ('A', i32), ('C', i32), ('stride', i32),
                ('rows_total', i32), ('row0', i32)]


class DecodeBwd(C.Structure):
    _fields_ = [('pred', c_fp), ('dpred', c_fp), ('draw', c_fp), ('view', HeadView),
                ('B', i32), ('H', i32), ('W', i32), ('A', i32), ('C', i32), ('stride', i32)]


class YoloLoss(C.Structure):
    _fields_ = [('inp', HeadView), ('label', c_fp), ('boxes', c_fp), ('partial', c_fp), ('losses', c_fp),
                ('nan_flag', c_fp), ('gscale', c_fp), ('grad', c_fp),
                ('gsb', i64), ('gsh', i64), ('gsw', i64), ('gsc', i64),
                ('B', i32), ('H', i32), ('W', i32), ('A', i32), ('C', i32), ('M', i32), ('stride', i32),
                ('box_loss', i32), ('from_raw', i32), ('ignore_thresh', f32), ('l1_gain', f32),
                ('totals', c_fp), ('head_slot', i32), ('n_heads', i32), ('gup', c_fp * 5), ('phase', i32),
                ('also', c_fp * 2)]


class Adam(C.Structure):
    _fields_ = [('param', c_fp), ('grad', c_fp), ('exp_avg', c_fp), ('exp_avg_sq', c_fp), ('n', i64),
                ('lr', f32), ('beta1', f32), ('beta2', f32), ('eps', f32), ('weight_decay', f32),
                ('bias_correction1', f32), ('bias_correction2', f32), ('grad_scale', f32)]


class Recover(C.Structure):
    _fields_ = [('pred', c_fp), ('delta', c_fp), ('ratio', c_fp), ('orig', c_fp), ('out', c_fp),
                ('B', i32), ('N', i32), ('C', i32), ('per_image', i32), ('mutate', i32)]


class Nms(C.Structure):
    _fields_ = [('bboxes', c_fp), ('dets', c_fp), ('count', c_fp), ('scratch', c_fp), ('scratch_bytes', i64),
                ('N', i32), ('C', i32), ('cap', i32), ('score_thr', f32), ('iou_thr', f32), ('cand_cap', i32),
                ('B', i32), ('img', c_fp)]


class CreateLabel(C.Structure):
    _fields_ = [('gt', c_fp), ('counts', c_fp), ('anchors', c_fp), ('label', c_fp * 3), ('boxes', c_fp * 3),
                ('box_counts', c_fp), ('H', i32 * 3), ('W', i32 * 3), ('stride', i32 * 3),
                ('B', i32), ('G', i32), ('C', i32), ('gt_per_grid', i32), ('cap', i32), ('iou_threshold', f32)]


class Fill(C.Structure):
    _fields_ = [('dst', c_fp), ('value', f32), ('n', i64)]


class TransposeItem(C.Structure):
    _fields_ = [('src', c_fp), ('dst', c_fp), ('rows', i32), ('cols', i32), ('tile0', i32), ('tiles_c', i32)]


class Transpose(C.Structure):
    _fields_ = [('items', c_fp), ('n_items', i32), ('total_tiles', i32)]


class QatWeightItem(C.Structure):
    _fields_ = [('w', c_fp), ('gamma', c_fp), ('running_var', c_fp), ('wq', c_fp), ('sf', c_fp), ('state', c_fp),
                ('dwq', c_fp), ('dw', c_fp), ('dgamma', c_fp), ('eps', f32), ('O', i32), ('K', i32)]


class QatWeight(C.Structure):
    _fields_ = [('items', c_fp), ('n_items', i32), ('observe', i32), ('quantise', i32), ('avg_const', f32)]


class Qact(C.Structure):
    _fields_ = [('inp', Src), ('out', c_fp), ('state', c_fp), ('partial', c_fp), ('rows', i64),
                ('C', i32), ('ldo', i32), ('flat', i32), ('observe', i32), ('quantise', i32), ('avg_const', f32)]


class QactBwd(C.Structure):
    _fields_ = [('gq', c_fp), ('dx', GSink), ('state', c_fp), ('rows', i32), ('C', i32), ('ldgq', i32), ('quantise', i32)]


class LayerFwd(C.Structure):
    _fields_ = [('inp', Src), ('w', c_fp), ('bias', c_fp), ('out', c_fp), ('kind', i32), ('B', i32), ('IH', i32), ('IW', i32),
                ('OH', i32), ('OW', i32), ('C', i32), ('k', i32), ('stride', i32), ('pad', i32), ('ldo', i32), ('O', i32),
                ('act', i32), ('nchw', i32)]


class XdwFwd(C.Structure):
    _fields_ = [('inp', Src), ('we', c_fp), ('mid', Src), ('yexp', c_fp), ('wd', c_fp), ('out', c_fp), ('partial', c_fp),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('I', i32), ('E', i32), ('stride', i32),
                ('ldy', i32), ('ldo', i32), ('ldp', i32), ('stat_acc', c_fp), ('stat_bad', c_fp), ('stat_slots', i32)]


class XstatFwd(C.Structure):
    _fields_ = [('inp', Src), ('we', c_fp), ('gamma', c_fp), ('beta', c_fp), ('running_mean', c_fp), ('running_var', c_fp),
                ('num_batches_tracked', c_fp), ('mean', c_fp), ('invstd', c_fp), ('scale', c_fp), ('shift', c_fp), ('scratch', c_fp),
                ('M', i32), ('I', i32), ('E', i32), ('eps', f32), ('momentum', f32), ('moments', c_fp)]


class XdwBwd(C.Structure):
    _fields_ = [('inp', Src), ('we', c_fp), ('mid_scale', c_fp), ('mid_shift', c_fp), ('mid_mean', c_fp), ('mid_invstd', c_fp),
                ('mid_gamma', c_fp), ('mid_act', i32), ('pad_', i32), ('dy', GSrc), ('wd', c_fp), ('dx', GSink), ('dwe', c_fp), ('dwd', c_fp),
                ('dgamma', c_fp), ('dbeta', c_fp), ('moments', c_fp), ('scratch', c_fp), ('gexp', c_fp), ('ldge', i32),
                ('B', i32), ('IH', i32), ('IW', i32), ('OH', i32), ('OW', i32), ('I', i32), ('E', i32), ('stride', i32)]


class ReduceItem(C.Structure):
    _fields_ = [('slab', c_fp), ('out0', c_fp), ('out1', c_fp), ('stride', i64), ('gstride', i64), ('splits', i32), ('n0', i32), ('n1', i32),
                ('ng', i32), ('perm_gw', i32), ('lanes', i32), ('block0', i32), ('pad_', i32)]


class ReduceBatch(C.Structure):
    _fields_ = [('items', c_fp), ('n_items', i32), ('n_blocks', i32)]


LAYER_MAXPOOL, LAYER_AVGPOOL, LAYER_IM2COL, LAYER_FC, LAYER_ACT, LAYER_PADROWS = range(6)


class _OpU(C.Union):
    _fields_ = [('c1f', Conv1x1Fwd), ('c1d', Conv1x1Dgrad), ('c1w', Conv1x1Wgrad),
                ('dwf', DwFwd), ('dwd', DwDgrad), ('dww', DwWgrad), ('dwb', DwBwd), ('stf', StemFwd), ('stw', StemWgrad),
                ('bnf', BnFwdFin), ('bnb', BnBwdFin), ('cp', Copy), ('rb', RouteBwd),
                ('dec', Decode), ('loss', YoloLoss), ('adam', Adam), ('fill', Fill),
                ('gcf', GconvFwd), ('gcd', GconvDgrad), ('gcw', GconvWgrad), ('tr', Transpose),
                ('qw', QatWeight), ('qa', Qact), ('qb', QactBwd), ('pr', PoolReduce), ('c1b', Conv1x1Bwd), ('bne', BnEval),
                ('lay', LayerFwd), ('xdf', XdwFwd), ('xsf', XstatFwd), ('xdb', XdwBwd), ('red', ReduceBatch)]


OPF_SIDE = 1
OPF_JOIN = 2
OPF_LAZY = 4
OPF_NOFORK = 8


class Op(C.Structure):
    _fields_ = [('code', i32), ('flags', i32), ('u', _OpU)]


(OP_CONV1X1_FWD, OP_CONV1X1_DGRAD, OP_CONV1X1_WGRAD, OP_DW_FWD, OP_DW_DGRAD, OP_DW_WGRAD, OP_STEM_FWD, OP_STEM_WGRAD,
 OP_BN_FWD_FIN, OP_BN_BWD_FIN, OP_COPY_FWD, OP_ROUTE_BWD, OP_DECODE, OP_LOSS_FWD, OP_LOSS_BWD, OP_ADAM, OP_FILL, OP_DW_BWD,
 OP_GCONV_FWD, OP_GCONV_DGRAD, OP_GCONV_WGRAD, OP_TRANSPOSE, OP_QAT_WPREP, OP_QAT_WGRAD, OP_QACT_FWD,
 OP_QACT_BWD, OP_POOL_REDUCE, OP_CONV1X1_BWD, OP_BN_EVAL, OP_LAYER_FWD, OP_XDW_FWD, OP_XSTAT_FWD, OP_XDW_BWD,
 OP_REDUCE_BATCH) = range(1, 35)

OP_FIELD = {
    OP_CONV1X1_FWD: 'c1f', OP_CONV1X1_DGRAD: 'c1d', OP_CONV1X1_WGRAD: 'c1w', OP_DW_FWD: 'dwf', OP_DW_DGRAD: 'dwd',
    OP_DW_WGRAD: 'dww', OP_STEM_FWD: 'stf', OP_STEM_WGRAD: 'stw', OP_BN_FWD_FIN: 'bnf', OP_BN_BWD_FIN: 'bnb',
    OP_COPY_FWD: 'cp', OP_ROUTE_BWD: 'rb', OP_DECODE: 'dec', OP_LOSS_FWD: 'loss', OP_LOSS_BWD: 'loss', OP_ADAM: 'adam',
    OP_FILL: 'fill', OP_DW_BWD: 'dwb', OP_GCONV_FWD: 'gcf', OP_GCONV_DGRAD: 'gcd', OP_GCONV_WGRAD: 'gcw',
    OP_TRANSPOSE: 'tr', OP_QAT_WPREP: 'qw', OP_QAT_WGRAD: 'qw', OP_QACT_FWD: 'qa', OP_QACT_BWD: 'qb',
    OP_POOL_REDUCE: 'pr', OP_CONV1X1_BWD: 'c1b', OP_BN_EVAL: 'bne', OP_LAYER_FWD: 'lay', OP_XDW_FWD: 'xdf', OP_XSTAT_FWD: 'xsf', OP_XDW_BWD: 'xdb',
    OP_REDUCE_BATCH: 'red',
}

# name -> (restype, argtypes); every symbol include/pqdet_hip.h declares
SIGNATURES = {
    'pq_conv1x1_fwd': (C.c_int, [C.POINTER(Conv1x1Fwd), C.c_void_p]),
    'pq_conv1x1_blocks': (i32, [i32, i32]),
    'pq_conv1x1_blocks_k': (i32, [i32, i32, i32, i32]),
    'pq_conv1x1_dgrad': (C.c_int, [C.POINTER(Conv1x1Dgrad), C.c_void_p]),
    'pq_conv1x1_wgrad': (C.c_int, [C.POINTER(Conv1x1Wgrad), C.c_void_p]),
    'pq_conv1x1_wgrad_scratch_bytes': (i64, [i32, i32, i32]),
    'pq_conv1x1_fwd_plan': (C.c_int, [C.POINTER(Conv1x1Fwd), C.POINTER(GemmPlan)]),
    'pq_conv1x1_dgrad_plan': (C.c_int, [C.POINTER(Conv1x1Dgrad), C.POINTER(GemmPlan)]),
    'pq_conv1x1_wgrad_plan': (C.c_int, [C.POINTER(Conv1x1Wgrad), C.POINTER(GemmPlan)]),
    'pq_dw3x3_fwd': (C.c_int, [C.POINTER(DwFwd), C.c_void_p]),
    'pq_dw3x3_blocks': (i32, [i32, i32, i32, i32]),
    'pq_dw3x3_fwd_blocks': (i32, [i32, i32, i32, i32, i32]),
    'pq_dw3x3_bwd_walk_blocks': (i32, [i32, i32, i32, i32, i32]),
    'pq_dw3x3_dgrad': (C.c_int, [C.POINTER(DwDgrad), C.c_void_p]),
    'pq_dw3x3_wgrad': (C.c_int, [C.POINTER(DwWgrad), C.c_void_p]),
    'pq_dw3x3_wgrad_scratch_bytes': (i64, [i32, i32, i32, i32]),
    'pq_dw3x3_bwd': (C.c_int, [C.POINTER(DwBwd), C.c_void_p]),
    'pq_gconv3x3_fwd': (C.c_int, [C.POINTER(GconvFwd), C.c_void_p]),
    'pq_gconv3x3_blocks': (i32, [i32, i32, i32]),
    'pq_gconv3x3_dgrad': (C.c_int, [C.POINTER(GconvDgrad), C.c_void_p]),
    'pq_gconv3x3_wgrad': (C.c_int, [C.POINTER(GconvWgrad), C.c_void_p]),
    'pq_gconv3x3_wgrad_scratch_bytes': (i64, [i32, i32, i32, i32, i32]),
    'pq_stem3x3_fwd': (C.c_int, [C.POINTER(StemFwd), C.c_void_p]),
    'pq_stem3x3_wgrad': (C.c_int, [C.POINTER(StemWgrad), C.c_void_p]),
    'pq_stem3x3_wgrad_scratch_bytes': (i64, [i32, i32, i32, i32, i32]),
    'pq_bn_fwd_finalize': (C.c_int, [C.POINTER(BnFwdFin), C.c_void_p]),
    'pq_bn_eval_batch': (C.c_int, [C.POINTER(BnEval), C.c_void_p]),
    'pq_layer_fwd': (C.c_int, [C.POINTER(LayerFwd), C.c_void_p]),
    'pq_xdw_fwd': (C.c_int, [C.POINTER(XdwFwd), C.c_void_p]),
    'pq_xdw_fwd_blocks': (i32, [i32, i32, i32, i32, i32, i32]),
    'pq_xstat_fwd': (C.c_int, [C.POINTER(XstatFwd), C.c_void_p]),
    'pq_xstat_fwd_scratch_bytes': (i64, [i32, i32]),
    'pq_xdw_bwd': (C.c_int, [C.POINTER(XdwBwd), C.c_void_p]),
    'pq_reduce_batch_plan': (i32, [C.POINTER(ReduceItem), i32]),
    'pq_reduce_batch': (C.c_int, [C.POINTER(ReduceBatch), C.c_void_p]),
    'pq_conv1x1_wgrad_reduce_item': (C.c_int, [C.POINTER(Conv1x1Wgrad), C.POINTER(ReduceItem)]),
    'pq_conv1x1_bwd_reduce_item': (C.c_int, [C.POINTER(Conv1x1Bwd), C.POINTER(ReduceItem)]),
    'pq_dw3x3_bwd_reduce_item': (C.c_int, [C.POINTER(DwBwd), C.POINTER(ReduceItem)]),
    'pq_gconv3x3_wgrad_reduce_item': (C.c_int, [C.POINTER(GconvWgrad), C.POINTER(ReduceItem)]),
    'pq_xdw_bwd_blocks': (i32, [i32, i32, i32, i32, i32, i32]),
    'pq_xdw_bwd_fix_blocks': (i32, [i32, i32]),
    'pq_xdw_bwd_scratch_bytes': (i64, [i32, i32, i32, i32, i32, i32]),
    'pq_bn_bwd_finalize': (C.c_int, [C.POINTER(BnBwdFin), C.c_void_p]),
    'pq_bn_finalize_scratch_bytes': (i64, [i32]),
    'pq_bn_finalize_tickets': (i32, [i32]),
    'pq_bn_stat_ok': (i32, [i32, C.c_void_p, i32]),
    'pq_bn_stat_selftest': (C.c_int, [C.c_void_p, i32, i32, i32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'pq_bn_bstat_ok': (i32, [i32, C.c_void_p, i32]),
    'pq_bn_stat_bytes': (i64, [i32, i32]),
    'pq_bn_stat_slots': (i32, [i32]),
    'pq_copy_fwd': (C.c_int, [C.POINTER(Copy), C.c_void_p]),
    'pq_route_bwd': (C.c_int, [C.POINTER(RouteBwd), C.c_void_p]),
    'pq_ew_blocks': (i32, [i32, i32]),
    'pq_pool_reduce': (C.c_int, [C.POINTER(PoolReduce), C.c_void_p]),
    'pq_conv1x1_bwd': (C.c_int, [C.POINTER(Conv1x1Bwd), C.c_void_p]),
    'pq_conv1x1_bwd_blocks': (i32, [i32, i32, i32]),
    'pq_conv1x1_bwd_pays': (i32, [i32, i32, i32]),
    'pq_conv1x1_bwd_scratch_bytes': (C.c_int64, [i32, i32, i32]),
    'pq_decode': (C.c_int, [C.POINTER(Decode), C.c_void_p]),
    'pq_decode_bwd': (C.c_int, [C.POINTER(DecodeBwd), C.c_void_p]),
    'pq_yolo_loss_fwd': (C.c_int, [C.POINTER(YoloLoss), C.c_void_p]),
    'pq_yolo_loss_bwd': (C.c_int, [C.POINTER(YoloLoss), C.c_void_p]),
    'pq_yolo_loss_blocks': (i32, [i32, i32, i32]),
    'pq_iou_pairs': (C.c_int, [c_fp, c_fp, c_fp, i32, i32, C.c_void_p]),
    'pq_qat_weight_prep': (C.c_int, [C.POINTER(QatWeight), C.c_void_p]),
    'pq_qat_weight_grad': (C.c_int, [C.POINTER(QatWeight), C.c_void_p]),
    'pq_qact_fwd': (C.c_int, [C.POINTER(Qact), C.c_void_p]),
    'pq_qact_blocks': (i32, [i64]),
    'pq_qact_bwd': (C.c_int, [C.POINTER(QactBwd), C.c_void_p]),
    'pq_transpose_batch': (C.c_int, [C.POINTER(Transpose), C.c_void_p]),
    'pq_adam_step': (C.c_int, [C.POINTER(Adam), C.c_void_p]),
    'pq_sign_l1': (C.c_int, [c_fp, c_fp, i64, f32, C.c_void_p]),
    'pq_fill': (C.c_int, [c_fp, f32, i64, C.c_void_p]),
    'pq_recover': (C.c_int, [C.POINTER(Recover), C.c_void_p]),
    'pq_create_label': (C.c_int, [C.POINTER(CreateLabel), C.c_void_p]),
    'pq_nms_last_path': (i32, []),
    'pq_nms': (C.c_int, [C.POINTER(Nms), C.c_void_p]),
    'pq_nms_scratch_bytes': (i64, [i32, i32]),
    'pq_nms_batch_scratch_bytes': (i64, [i32, i32, i32]),
    'pq_run_ops': (C.c_int, [C.POINTER(Op), i32, C.c_void_p]),
    'pq_host_device_pointer': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'pq_run_ops2': (C.c_int, [C.POINTER(Op), i32, C.c_void_p, C.c_void_p]),
    'pq_run_ops2_nojoin': (C.c_int, [C.POINTER(Op), i32, C.c_void_p, C.c_void_p]),
    'pq_run_ops_timed': (C.c_int, [C.POINTER(Op), i32, C.c_void_p, C.POINTER(C.c_float)]),
    'pq_run_ops2_timed': (C.c_int, [C.POINTER(Op), i32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    'pq_launch_log': (None, [i32]),
    'pq_launch_log_dump': (i64, [C.c_char_p, i64]),
    'pq_last_failed_op': (C.c_int, []),
    'pq_last_error': (C.c_char_p, []),
    'pq_version': (C.c_char_p, []),
    'pq_abi_sizeof': (i32, [i32]),
}

_lib = None


class PQError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library with typed entry points.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PQError('libpqdet_hip.so is not built: run `python -m pqdet_amd.build` '
                          '(hipcc --offload-arch=gfx950); there is no fallback path')
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        l = lib()
        raise PQError('%s failed (rc=%d, op #%d): %s' % (what or 'pqdet_hip call', rc, l.pq_last_failed_op(),
                                                           l.pq_last_error().decode()))


def ptr(t):
    """Device address of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()
