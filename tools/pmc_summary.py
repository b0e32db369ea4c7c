"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) into profiles/*.json.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <config> <records> <steps_in_run> <out.json>   (steps_in_run: informative)

Per kernel: mean KB per dispatch of each counter and dispatches per step; step_traffic_bytes = sum over the kernels of a
graph-build step of (2 * FETCH_SIZE + WRITE_SIZE) * 1024 * dispatches per step.  The factor 2 is the gfx950 correction
of MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports half of the bytes of a wide coalesced streaming read; narrow /
gathered reads and WRITE_SIZE are uncalibrated there, so the figure is an upper estimate for the gather-heavy kernels.
bench.py attaches it to roofline.traffic only when source_hash and records match the run.
"""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP_KERNELS = ('stream_kernel', 'ordered_kernel', 'stitch_kernel', 'compact_kernel', 'radix_hist_kernel',
                'radix_rowscan_kernel', 'radix_scatter_kernel', 'bucket_sort_kernel', 'bucket_reduce_kernel',
                'row_heads_kernel', 'row_scan_kernel', 'row_reduce_kernel', 'os_hist_kernel', 'os_offsets_kernel',
                'os_scatter_kernel', 'os_reduce_kernel', 'os_fixup_kernel', 'os_bucket_start_kernel', 'os_bucket_wave_kernel',
                'os_bucket_sort_kernel', 'os_bucket_rows_kernel', 'os_bucket_wave_lds_kernel', 'os_seg_tiles_kernel',
                'stitch_spans_kernel', 'presort_fixup_kernel', 'fused_wave_kernel', 'rg_group_kernel', 'rg_compact_kernel',
                'rg_tile_sums_kernel', 'rg_dst_kernel', 'rg_rows_kernel', 'rg_copy_kernel', 'msd_partition_kernel',
                'rl_list_kernel', 'rl_place_kernel', 'rl_rows_kernel')


def source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, 'besst_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(REPO, 'besst_amd', 'csrc', '*.h'))):
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def collect(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                name = row['Kernel_Name'].replace('besst::(anonymous namespace)::', '')
                name = re.sub(r'^void ', '', name)
                name = re.sub(r'[<(].*$', '', name).strip()
                acc[name].append(float(row['Counter_Value']))
    return acc


def main():
    fetch_dir, write_dir, config, n_rec, steps, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    f, w = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    kernels, total = {}, 0.0
    for k in sorted(set(f) | set(w)):
        if k not in STEP_KERNELS:
            continue
        fm = sum(f[k]) / max(1, len(f[k]))
        wm = sum(w[k]) / max(1, len(w[k]))
        # dispatches per step: relative to a kernel that runs exactly once per record loop / once per sort
        classify = k in ('stream_kernel', 'ordered_kernel', 'fused_kernel', 'fused_wave_kernel', 'stitch_kernel', 'compact_kernel',
                        'stitch_spans_kernel', 'presort_fixup_kernel')
        sort_anchor = next((a for a in ('rg_compact_kernel', 'os_offsets_kernel', 'msd_partition_kernel', 'radix_hist_kernel') if a in f), 'stitch_kernel')
        anchor = 'stitch_kernel' if classify else sort_anchor
        per_step = len(f[k]) / float(max(1, len(f.get(anchor, []))))
        if per_step < 0.5:          # not a kernel of the step (compact_kernel of the capacity probe, when the sort reads segments)
            continue
        kernels[k] = {'FETCH_SIZE_KB_mean': round(fm, 1), 'WRITE_SIZE_KB_mean': round(wm, 1),
                      'dispatches_per_step': round(per_step, 2),
                      'traffic_bytes_per_step': int((2 * fm + wm) * 1024 * per_step)}
        total += (2 * fm + wm) * 1024 * per_step
    doc = {'_about': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of bench.py on %s; '
                     'KB per dispatch; traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction on FETCH_SIZE, '
                     'MI355X_MICROARCH.md HBM section; gathered reads and WRITE_SIZE uncalibrated). tools/pmc_summary.py.' % config,
           'config': config, 'records': n_rec, 'steps_in_run': steps, 'source_hash': source_hash(),
           'step_traffic_bytes': int(total), 'kernels': kernels}
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({k: v['traffic_bytes_per_step'] for k, v in kernels.items()}), int(total))


if __name__ == '__main__':
    main()
