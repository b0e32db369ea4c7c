"""Development helper: print one steady-state step of a rocprofv3 kernel trace as a timeline (name, start, duration, gap).
usage: python tools/timeline.py <dir with *_kernel_trace.csv> <first kernel of a step, e.g. stream_kernel> [step index]"""
import csv, glob, os, re, sys
d, first = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
rows = []
for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    with open(path, newline='') as fh:
        for r in csv.DictReader(fh):
            name = r['Kernel_Name'].replace('besst::(anonymous namespace)::', '')
            name = re.sub(r'^void ', '', name)
            name = re.sub(r'\(.*$', '', name)[:48]
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith(first)]
i0, i1 = starts[which], starts[which + 1]
t0 = rows[i0][0]
prev_end = None
for s, e, name, q in rows[i0:i1]:
    gap = '' if prev_end is None else '%+6.1f' % ((s - prev_end) / 1e3)
    print('%8.1f us  %6.1f us  gap %7s  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, name))
    prev_end = max(prev_end or e, e)
print('step: %.1f us' % ((rows[i1][0] - t0) / 1e3))
