"""Kernels of ONE train step inside a time window, from a rocprofv3 --kernel-trace CSV (start, duration, queue, name).
usage: trace_window.py <kernel_trace.csv> <from_ms> <to_ms> [step_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lo_ms, hi_ms = float(sys.argv[2]), float(sys.argv[3])
which = int(sys.argv[4]) if len(sys.argv) > 4 else -2
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
step = ev[adam[which - 1] + 1: adam[which] + 1]
t0 = step[0][0]
for s, e, n, q in step:
    a = (s - t0) / 1e6
    if lo_ms <= a <= hi_ms:
        print('%8.3f ms  %7.1f us  q%s  %s' % (a, (e - s) / 1e3, q, n[:90]))
