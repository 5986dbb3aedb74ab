"""Per-kernel statistics of the STEADY-STATE steps of a rocprofv3 --kernel-trace CSV.

rocprofv3's own --stats summary covers the whole process, including MIOpen's find-time benchmark
launches during warm-up.  Every training step launches furthest-point sampling exactly once, so the
last `--steps` occurrences of the FPS kernel delimit the timed steps; everything that started after
the first of them is aggregated.

  python tools/trace_stats.py <kernel_trace.csv> --steps 2 [--top 60] > profiles/<name>.csv
"""
import argparse
import collections
import csv
import sys


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    return name if len(name) <= 110 else name[:107] + '...'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, required=True)
    ap.add_argument('--marker', default='fps_')      # fps_kernel (full update) or fps_pruned_kernel: one launch per step
    ap.add_argument('--top', type=int, default=80)
    args = ap.parse_args()
    rows = []
    with open(args.trace, newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [s for s, e, n in rows if args.marker in n]
    if len(marks) < args.steps:
        sys.exit('marker kernel seen %d times, need %d' % (len(marks), args.steps))
    t0 = marks[-args.steps]
    sel = [(s, e, n) for s, e, n in rows if s >= t0]
    span = max(e for s, e, n in sel) - t0
    agg = collections.defaultdict(lambda: [0, 0, 10 ** 18, 0])
    for s, e, n in sel:
        a = agg[n]
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    total = sum(a[1] for a in agg.values())
    # union of the kernel intervals = time with at least one kernel executing (both streams together)
    busy, cur_s, cur_e = 0, None, None
    for s_, e_, _ in sel:
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    busy += (cur_e - cur_s) if cur_e is not None else 0
    w = csv.writer(sys.stdout)
    w.writerow(['# steady-state window: %d steps, %.3f ms wall per step, %.3f ms summed kernel time per step, '
                '%.3f ms with at least one kernel running (GPU idle %.1f %%), %d launches per step'
                % (args.steps, span / args.steps / 1e6, total / args.steps / 1e6, busy / args.steps / 1e6,
                   100.0 * (1 - busy / span), len(sel) // args.steps)])
    w.writerow(['Name', 'CallsPerStep', 'TotalMsPerStep', 'AverageUs', 'Percentage', 'MinUs', 'MaxUs'])
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        w.writerow([short(n), '%.1f' % (a[0] / args.steps), '%.3f' % (a[1] / args.steps / 1e6), '%.2f' % (a[1] / a[0] / 1e3),
                    '%.2f' % (100.0 * a[1] / total), '%.2f' % (a[2] / 1e3), '%.2f' % (a[3] / 1e3)])


if __name__ == '__main__':
    main()
