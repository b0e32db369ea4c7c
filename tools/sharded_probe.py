"""Development probe (GPU box): cProfile of the sharded step with one rank forced through the RCCL orchestration."""
import cProfile, io, os, pstats, sys
os.environ.setdefault('BESST_FORCE_DISTRIBUTED', '1'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('LOCAL_RANK', '0'); os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['bench.py', '--gpus', '1', '--steps', '40', '--warmup', '3', '--no-verify', '--no-cpu-baseline', '--breakdown-steps', '0']
import bench
pr = cProfile.Profile(); pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats('distributed.py|pipeline.py|synchronize|item|cpu|_lib', 40)
    sys.stderr.write(s.getvalue()[:9000])
