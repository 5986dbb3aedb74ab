"""Average HBM traffic per launch of one kernel family from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes).

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <entry-point>

An entry point that launches SEVERAL kernels (camli_wino_conv3x3 = input transform + plane GEMMs + output transform) gives its
kernels as a '+'-separated list, the FIRST of which runs exactly once per launch of the entry point: the traffic per launch is
the sum of every listed kernel's counter over the pass divided by the number of launches of the first.

Kernels SHARED between entry points (the 1-D Winograd family: one input transform serves four entry points) are given as a
'~'-separated list: the traffic per launch is the sum of the listed kernels' MEAN counters (each runs once per launch of the
entry point; the mean of a shared kernel is taken over all its launches in the pass).

Units: the counters are KiB.  gfx950 correction (guide, section HBM): FETCH_SIZE counts a wide
coalesced 16-B/lane read stream at exactly half its bytes -> doubled; WRITE_SIZE is taken as is.
tools/pmc_probe.py re-checks both on a 256 MiB copy (131,083 KiB fetched / 262,144 KiB written).
"""
import csv
import json
import sys


def mean_counter(path, needle):
    vals = []
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            if needle in r['Kernel_Name']:
                vals.append(float(r['Counter_Value']))
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


def per_entry_launch(path, needles):
    total, first = 0.0, 0
    with open(path, newline='') as f:
        for r in csv.DictReader(f):
            hits = [n for n in needles if n in r['Kernel_Name']]
            if hits:
                total += float(r['Counter_Value'])
                first += needles[0] in r['Kernel_Name']
    return (total / first, first) if first else (0.0, 0)


def main():
    fetch_csv, write_csv, needle, entry = sys.argv[1:5]
    if '~' in needle:
        parts_f = [mean_counter(fetch_csv, n) for n in needle.split('~')]
        parts_w = [mean_counter(write_csv, n) for n in needle.split('~')]
        fetch, n1 = sum(p[0] for p in parts_f), parts_f[0][1]
        write, n2 = sum(p[0] for p in parts_w), parts_w[0][1]
    elif '+' in needle:
        fetch, n1 = per_entry_launch(fetch_csv, needle.split('+'))
        write, n2 = per_entry_launch(write_csv, needle.split('+'))
    else:
        fetch, n1 = mean_counter(fetch_csv, needle)
        write, n2 = mean_counter(write_csv, needle)
    out = {'entry_point': entry, 'kernel_substring': needle, 'launches_sampled': [n1, n2],
           'FETCH_SIZE_KiB_avg': round(fetch, 1), 'WRITE_SIZE_KiB_avg': round(write, 1),
           'correction': 'traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B (gfx950: wide coalesced reads counted at 1/2)',
           'traffic_bytes_per_launch': int((2 * fetch + write) * 1024),
           'traffic_bytes_per_launch_uncorrected': int((fetch + write) * 1024)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
