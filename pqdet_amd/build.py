"""Build libpqdet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pqdet_amd.build          # incremental
    python -m pqdet_amd.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(HERE, 'lib')
LIB = os.path.join(OUT_DIR, 'libpqdet_hip.so')
ARCH = 'gfx950'
SOURCES = {            # file -> extra flags
    'api.hip': [],
    'gemm.hip': [],
    'gemm2.hip': ['-Wno-inline-asm'],
    'dwconv.hip': [],
    'dwtile.hip': [],
    'dwwalk.hip': [],
    'layers.hip': [],
    'gconv.hip': [],
    'pwbwd.hip': [],
    'irblock.hip': [],
    'irbwd.hip': [],
    'reduce.hip': [],
    'qat.hip': ['-ffp-contract=off'],
    'ew.hip': [],
    'yolo.hip': [],
    'post.hip': ['-ffp-contract=off'],   # NMS comparisons must not see FMA contraction (bit-exact selection)
}
COMMON = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-fno-gpu-rdc', '-Wno-unused-result',
          '-I' + os.path.join(HERE, '..', 'include')]


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, defines=(), tag: str = '') -> str:
    """defines/tag: build an ablation variant next to the real library (perf experiments only)."""
    global LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    lib = LIB if not tag else LIB.replace('.so', '_%s.so' % tag)
    hipcc = _hipcc()
    # every header under csrc/ (common.h, ldsdma.h, qatmath.h, ...) and the public one: editing any of them rebuilds all objects
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + \
        [os.path.join(HERE, '..', 'include', 'pqdet_hip.h')]
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace('.hip', (tag and '_' + tag) + '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, '-c', s, '-o', o] + COMMON + extra + ['-D' + d for d in defines])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([hipcc, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', lib] + objs)
    return lib


if __name__ == '__main__':
    defs = [a[2:] for a in sys.argv[1:] if a.startswith('-D')]
    tags = [a[6:] for a in sys.argv[1:] if a.startswith('--tag=')]
    print(build(force='--force' in sys.argv, defines=defs, tag=tags[0] if tags else ''))
