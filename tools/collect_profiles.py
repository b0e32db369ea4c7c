"""Copy the round's profile set from gpurun_out/final/ (tools/final_profiles.sh) into profiles/ under the round's names;
kernel-stats CSVs keep this library's kernels and the runtime's copy / fill kernels (the generator's torch kernels go).
usage: python tools/collect_profiles.py r05 [source dir]"""
import csv, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, 'gpurun_out', 'final')
dst = os.path.join(REPO, 'profiles')
for name in ('c3', 'c2', 'ingest', 'metrics'):
    rows = list(csv.DictReader(open(os.path.join(src, '%s_kernel_stats.csv' % name), newline='')))
    keep = [r for r in rows if 'besst' in r['Name'] or '__amd_rocclr' in r['Name'] or r['Name'] in ('copy_words_kernel', 'obs_sum_kernel')]
    with open(os.path.join(dst, '%s_%s_kernel_stats.csv' % (tag, name)), 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
        w.writeheader()
        w.writerows(keep)
for a, b in (('bench_default.json', 'bench_default.json'), ('c3_pmc_traffic.json', 'c3_pmc_traffic.json'),
             ('c2_pmc_traffic.json', 'c2_pmc_traffic.json'), ('metrics_pmc.json', 'metrics_pmc.json'),
             ('ingest_pmc.txt', 'ingest_pmc.txt'), ('c4_rank1.json', 'c4_rank1.json'),
             ('from_bam_2ranks_one_gpu.json', 'from_bam_2ranks_one_gpu.json'),
             ('bam_to_graph_c3_full.json', 'bam_to_graph_c3_full.json')):
    shutil.copy(os.path.join(src, a), os.path.join(dst, '%s_%s' % (tag, b)))
print(sorted(f for f in os.listdir(dst) if f.startswith(tag)))

# the C4-at-full-size-on-one-GPU object of the bench line, on its own
import json
line = json.loads(open(os.path.join(src, 'bench_default.json')).read().strip().splitlines()[-1])
for key in ('c4_single_gpu', 'c5_library_single_gpu', 'weak_scaling_n1'):
    if key in line:
        json.dump(line[key], open(os.path.join(dst, '%s_%s.json' % (tag, key)), 'w'), indent=1)
if os.path.isfile(os.path.join(src, 'record_loop_sq.json')):
    shutil.copy(os.path.join(src, 'record_loop_sq.json'), os.path.join(dst, '%s_record_loop_sq.json' % tag))
