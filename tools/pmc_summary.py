"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) into profiles/*.json.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <records_per_launch> <out.json>
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                name = row['Kernel_Name'].replace('besst::(anonymous namespace)::', '')
                name = re.sub(r'^void ', '', name)
                name = re.sub(r'\(.*$', '', name).strip()
                acc[name].append(float(row['Counter_Value']))
    return acc


def main():
    fetch_dir, write_dir, n_rec, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if not any(t in k for t in ('_kernel',)):
            continue
        kernels[k] = {'FETCH_SIZE_KB_mean': round(sum(f[k]) / max(1, len(f[k])), 1), 'FETCH_SIZE_launches': len(f[k]),
                      'WRITE_SIZE_KB_mean': round(sum(w[k]) / max(1, len(w[k])), 1), 'WRITE_SIZE_launches': len(w[k])}
    doc = {'_about': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py '
                     '--steps 5 --warmup 1 --no-stages --no-cpu-baseline --no-verify --breakdown-steps 0 --in-flight 0` on C2; '
                     'counter units are KB per dispatch. MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of '
                     'the bytes of a wide coalesced streaming read, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 for '
                     'stream_kernel; WRITE_SIZE and narrow/gather reads are uncalibrated. Summarised by tools/pmc_summary.py.',
           'kernels': kernels}
    sk = kernels.get('stream_kernel')
    if sk:
        doc['stream_kernel_traffic_bytes_per_launch'] = int((2 * sk['FETCH_SIZE_KB_mean'] + sk['WRITE_SIZE_KB_mean']) * 1024)
        doc['stream_kernel_algorithmic_bytes_per_launch'] = n_rec * 11
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc.get('kernels', {}).get('stream_kernel')), doc.get('stream_kernel_traffic_bytes_per_launch'))


if __name__ == '__main__':
    main()
