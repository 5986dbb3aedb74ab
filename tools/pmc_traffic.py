"""Average HBM traffic per launch of one kernel family from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes).

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <entry-point>

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


def main():
    fetch_csv, write_csv, needle, entry = sys.argv[1:5]
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
