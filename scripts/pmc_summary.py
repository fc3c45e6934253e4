"""Summarise rocprofv3 output for profiles/: per-kernel average duration (kernel_stats.csv) joined with the per-launch
FETCH_SIZE / WRITE_SIZE PMC counters of two separate --pmc passes.

usage: pmc_summary.py <stats_dir> <fetch_dir> <write_dir> [prefix]

Units / corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE shows
half of the bytes of wide (16 B/lane) coalesced reads, so hbm_read = 2 * FETCH_SIZE KiB; WRITE_SIZE is exact.  The
calibration row is adam_kernel, whose traffic is known exactly (reads 4 and writes 3 fp32 arrays of n elements).
"""
import collections
import csv
import glob
import os
import sys


def load_counter(d, prefix):
    acc = collections.defaultdict(lambda: [0.0, 0])
    f = glob.glob(os.path.join(d, prefix + '*counter_collection.csv'))[0]
    for row in csv.DictReader(open(f)):
        e = acc[row['Kernel_Name']]
        e[0] += float(row['Counter_Value'])
        e[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    stats_dir, fetch_dir, write_dir = sys.argv[1:4]
    prefix = sys.argv[4] if len(sys.argv) > 4 else ''
    stats = list(csv.DictReader(open(glob.glob(os.path.join(stats_dir, prefix + '*kernel_stats.csv'))[0])))
    fetch, write = load_counter(fetch_dir, prefix), load_counter(write_dir, prefix)
    print('kernel,calls,avg_us,pct_time,fetch_kib_raw_per_launch,hbm_read_mb_per_launch(2x),hbm_write_mb_per_launch,'
          'hbm_gb_per_s')
    for r in stats:
        n = r['Name']
        f, w = fetch.get(n), write.get(n)
        avg_us = float(r['AverageNs']) / 1e3
        rd = 2 * f * 1024 / 1e6 if f is not None else float('nan')
        wr = w * 1024 / 1e6 if w is not None else float('nan')
        bw = (rd + wr) / avg_us * 1e3 if f is not None and w is not None else float('nan')
        print('"%s",%s,%.2f,%s,%.1f,%.3f,%.3f,%.0f' % (n, r['Calls'], avg_us, r['Percentage'], f or 0, rd, wr, bw))


if __name__ == '__main__':
    main()
