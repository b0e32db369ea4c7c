import sys, json, os
sys.path.insert(0, '/root/repo')
import bench, torch
r = bench.bam_to_graph_timing(torch.device('cuda', 0), 'C3', pairs=50_000_000, realistic=True)
print(os.environ.get('BESST_STAGE_PREAD', '0'), json.dumps({k: r[k] for k in ('ingest_s', 'get_metrics_s', 'PE_s', 'total_s')}))
