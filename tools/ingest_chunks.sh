cd /root/repo
for cb in 2048 4096 7000 8192 16384; do
O=/tmp/ing_$cb; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && PROBE_MODES=device:$cb,device:$cb rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /root/repo/tools/ingest_probe.py C3 20000000 - 17 > $O/stats.log 2>&1)
grep "records/s" $O/stats.log | tail -1 | cut -c1-70
python - <<PY
import csv, glob
f = glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    n = r['Name']
    if 'inflate' in n: print('  cb=$cb %6s calls %10.1f us avg %10.1f ms total' % (r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
