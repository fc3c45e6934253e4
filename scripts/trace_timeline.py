"""Timeline of ONE train step from a rocprofv3 --kernel-trace CSV: busy time per queue, union, idle gaps, overlap, and the
largest gaps between consecutive kernels of the main queue.  usage: trace_timeline.py <kernel_trace.csv> [step_index]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in rows]
ev.sort()
# a step starts at every adam_kernel end
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
lo, hi = adam[which - 1] + 1, adam[which] + 1
step = ev[lo:hi]
t0, t1 = step[0][0], step[-1][1]
print('step: %d kernels, %.3f ms wall (first start -> adam end)' % (len(step), (t1 - t0) / 1e6))
byq = collections.defaultdict(list)
for s, e, n, q in step: byq[q].append((s, e, n))
for q, lst in byq.items():
    print('queue %s: %d kernels, busy %.3f ms' % (q, len(lst), sum(e - s for s, e, _ in lst) / 1e6))
# union busy
iv = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('union busy %.3f ms, idle %.3f ms in %d gaps (>= 1 us: %d, sum %.3f ms)' % (busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps),
      sum(1 for g in gaps if g[0] >= 1000), sum(g[0] for g in gaps if g[0] >= 1000) / 1e6))
# time with >= 2 kernels running
pts = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
depth, last, t2 = 0, pts[0][0], 0
for t, d in pts:
    if depth >= 2: t2 += t - last
    depth += d; last = t
print('time with >= 2 kernels in flight: %.3f ms' % (t2 / 1e6))
mainq = max(byq.items(), key=lambda kv: len(kv[1]))[0]
lst = sorted(byq[mainq])
g = sorted(((lst[i + 1][0] - lst[i][1]) / 1e3, lst[i][2][:50], lst[i + 1][2][:50], (lst[i][1] - t0) / 1e6) for i in range(len(lst) - 1))
print('main queue %s: sum of inter-kernel gaps %.3f ms; histogram of gaps (us):' % (mainq, sum(x[0] for x in g) / 1e3))
import bisect
edges = [0, 1, 2, 3, 5, 8, 12, 20, 50, 1e9]
h = [0] * (len(edges) - 1); hs = [0.0] * (len(edges) - 1)
for x in g:
    k = bisect.bisect_right(edges, max(x[0], 0)) - 1; h[k] += 1; hs[k] += x[0]
for k in range(len(h)): print('  [%g, %g) us: %d gaps, %.3f ms' % (edges[k], edges[k + 1], h[k], hs[k] / 1e3))
print('largest gaps:')
for x in g[-12:]: print('  %.1f us after %s -> before %s (at %.2f ms)' % x)
# the tail of the step: what still runs after the main queue's last kernel in front of the optimiser (the join of the two streams)
adam_s = step[-1][0]
tail = [(s, e, n, q) for s, e, n, q in step[:-1] if e > adam_s - 400000]
print('last 400 us in front of adam_kernel (start us before adam, duration us, queue, kernel):')
for s, e, n, q in sorted(tail)[-28:]:
    print('  -%7.1f  %6.1f  q%s  %s' % ((adam_s - s) / 1e3, (e - s) / 1e3, 'M' if q == mainq else 'S', n[:90]))
