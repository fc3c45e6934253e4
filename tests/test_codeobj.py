"""No kernel of the shipped library may use scratch (private segment) memory or spill vector registers.

Read from the gfx950 code objects inside ``libpqdet_hip.so`` (no GPU needed).  Why this is a test: a kernel with a private
segment makes every queue it runs on depend on the runtime's scratch arena (set up by the command processor at dispatch, resized
on first use), a `.private_segment_fixed_size` appears silently when an accumulator array is indexed by a loop the compiler did
not unroll (round 3: 448 bytes per lane in the 128 x 192 gradient-sink tile) or when a kernel runs out of registers, and the one
GPU memory fault on record happened with such kernels in the step.  The vendor's radix-sort kernels
(rocPRIM, the > 8192-candidates-per-image NMS path) are reported but not ours to fix.
"""
import os

import pytest

from pqdet_amd import abi, codeobj


def _kernels():
    if not os.path.exists(os.path.join(codeobj.LLVM_BIN, 'llvm-readelf')):
        pytest.skip('llvm-readelf not found under %s' % codeobj.LLVM_BIN)
    ks = codeobj.kernels(abi.LIB_PATH)
    assert len(ks) > 100, 'code objects not found in %s' % abi.LIB_PATH
    return ks


def test_no_pq_kernel_uses_scratch_or_spills_vgprs():
    ks = _kernels()
    ours = [k for k in ks if k['name'].startswith('_ZN2pq')]
    assert len(ours) > 100
    bad = [(k['name'], k.get('private_segment_fixed_size', 0), k.get('vgpr_spill_count', 0)) for k in ours
           if k.get('private_segment_fixed_size', 0) or k.get('vgpr_spill_count', 0) or k.get('uses_dynamic_stack', False)]
    assert not bad, 'kernels with a private segment / spilled VGPRs:\n' + '\n'.join('%s: scratch %d B, %d VGPRs spilled' % b for b in bad)


def test_register_budgets_leave_the_occupancy_the_launch_bounds_promise():
    """VGPR + AGPR <= 512 everywhere (one wave per SIMD at least) and static LDS within the 160 KiB of a CU."""
    for k in _kernels():
        if not k['name'].startswith('_ZN2pq'):
            continue
        assert k.get('vgpr_count', 0) <= 512, k
        assert k.get("group_segment_fixed_size", 0) <= 160 * 1024, k


def test_sgpr_spill_budget():
    """Spilled SGPRs live in VGPR lanes (v_writelane / v_readlane: no memory, EXEC-independent), so they are legal -- but every spill
    is VALU work inside the kernel.  Budget: what the library has today (172 in the 128 x 128 gradient-sink tile of the register-staged
    GEMM family, two launches per step) as the ceiling for the old kernels, and a tight one for the kernels written since."""
    worst = {}
    for k in _kernels():
        name = k['name']
        if not name.startswith('_ZN2pq'):
            continue
        n = k.get('sgpr_spill_count', 0)
        assert n <= 176, '%s spills %d SGPRs (budget 176)' % (name, n)
        if any(t in name for t in ('xdw_fwd_kernel', 'xdwb_kernel', 'xgram_kernel', 'xstat_finalize', 'xfold_kernel', 'xfix_kernel', 'xslice')):
            assert n <= 40, '%s spills %d SGPRs (budget 40 for the fused inverted-residual kernels)' % (name, n)
        worst[name] = n
    assert worst
