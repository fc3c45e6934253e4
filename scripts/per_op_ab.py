"""A/B of per-op times (each op replayed alone, hot caches) between two settings of an environment switch:
    python scripts/per_op_ab.py PQ_BN_ATOMIC 0 1 [class substring]"""
import os, subprocess, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, v0, v1 = sys.argv[1:4]
only = sys.argv[4] if len(sys.argv) > 4 else ''
res = []
for v in (v0, v1):
    env = dict(os.environ, PER_OP_ALL='1', MIN_US='0', **{var: v})
    if only: env['ONLY'] = only
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'phase_wall.py'), '16', '512'], env=env, capture_output=True, text=True).stdout
    lst = []
    for ln in out.splitlines():
        m = re.match(r'(fwd|bwd)#(\d+)\s+(.{28})\s(.*?)\s+([\d.]+) us', ln)
        if m: lst.append((m.group(1), int(m.group(2)), m.group(3).strip(), m.group(4).strip(), float(m.group(5))))
    lst.sort()
    rows, seen = {}, {}
    for tag, idx, name, hint, us in lst:          # key: n-th op of its class in program order (op indices shift between plans)
        n = seen.get((tag, name), 0); seen[(tag, name)] = n + 1
        rows[(tag, name, n)] = (name, hint, us)
    res.append(rows)
    for ln in out.splitlines()[:4]: print(v, ln)
a, b = res
tot = 0.0
for k in sorted(a):
    if k in b and (a[k][0] == b[k][0]):
        d = b[k][2] - a[k][2]
        tot += d
        if abs(d) > 1.0: print('%s %-28s #%-3d %-60s %7.1f -> %7.1f  (%+.1f)' % (k[0], k[1], k[2], a[k][1][:60], a[k][2], b[k][2], d))
print('sum of differences over matching ops: %+.1f us' % tot)
for k in sorted(set(a) ^ set(b)):
    src = a if k in a else b
    print('only in %s: %s %s %.1f us' % (v0 if k in a else v1, k, src[k][1][:50], src[k][2]))
