"""Compact view of a rocprofv3 kernel_stats.csv: short kernel names, calls, average microseconds (development aid)."""
import csv
import re
import sys


def short(name):
    m = re.search(r'(?:besst::\(anonymous namespace\)::|besst::)(\w+)', name)
    if m:
        t = re.search(m.group(1) + r'<([^>]*)>', name)
        return m.group(1) + ('<%s>' % t.group(1) if t else '')
    return None


def main(path, only_besst=True, per=None):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            s = short(r['Name'])
            if s is None and only_besst:
                continue
            rows.append((s or r['Name'][:60], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
    rows.sort(key=lambda x: -x[3])
    calls = per or max(c for _, c, _, _ in rows)
    tot = 0.0
    for s, c, avg, total in rows:
        per_step = total / calls
        tot += per_step
        print('%-44s calls %5d  avg %9.1f us  per-step %9.1f us' % (s, c, avg, per_step))
    print('%-44s %38.1f us' % ('sum per step (calls of the busiest kernel)', tot))


if __name__ == '__main__':
    main(sys.argv[1], per=int(sys.argv[2]) if len(sys.argv) > 2 else None)
