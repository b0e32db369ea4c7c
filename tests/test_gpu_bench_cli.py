"""bench.py end to end on the GPU box: the default single-GPU line and the driver's N = 2 command line (torchrun,
both ranks on the box's one GPU over the gloo transport), at reduced sizes."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--pairs', '400000', '--contigs', '3000', '--steps', '5', '--warmup', '2', '--cpu-sample-records', '200000']


def _last_json_line(text):
    lines = [l for l in text.splitlines() if l.startswith('{')]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + SMALL, cwd=REPO, capture_output=True,
                         text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json_line(out.stdout)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 5 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['verified_vs_c_oracle'] is True
    assert d['config']['workload'].startswith('C3')
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(d['roofline'])
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(d['cpu_baseline'])
    assert d['stages']['linearize_calls_agree'] is True
    # kernels go by their own names: the record loop of a mate-pair library is the one-wave fused kernel
    loop = d['roofline']['record_loop_kernel']
    assert loop['name'] == 'fused_wave_kernel' and loop['name'] in d['kernel_ms'] and loop['avg_launch_ms'] > 0
    assert d['roofline']['dominant_kernel'] in d['kernel_ms']
    assert abs(d['value'] - 400000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']


def test_two_ranks_over_gloo_on_one_gpu():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BESST_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2'] + SMALL
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json_line(out.stdout)
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak'
    libs = d['config']['libraries']
    assert len(libs) == 2                                 # C4 shape: two libraries per step
    for lib in libs:
        assert lib['exchange_consistent'] is True
        assert all(lib['verified_vs_c_oracle'].values()), lib
    assert d['verified_vs_c_oracle'] is True
    assert abs(d['value'] - 2 * 2 * 400000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    # what a SCALE file is judged on: the CPU leg (rank 0, the port on the head of its slice + the committed calibration),
    # the traffic field with its reason, who took part, and the like-for-like one-GPU figure of the same shape
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(d['cpu_baseline']) and d['cpu_baseline']['value'] > 0
    assert d['roofline']['traffic'] is None and d['roofline']['traffic_note']
    assert d['rccl_ranks_seen'] == 2 and len(d['device_uuids']) == 2 and d['distinct_devices'] == 1   # (both on the box's one GPU)
    assert d['slices'] == 'contiguous'
    same = d['single_gpu_same_shape']
    assert same['value'] > 0 and abs(same['value'] - 2 * 400000 / (same['ms_per_step'] * 1e-3)) < 1e-3 * same['value']   # (ms rounded to 0.1 us)


def test_two_ranks_with_independent_slices():
    """--slices independent (rounds 2 and 3): every rank draws a whole-genome stream of its own."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BESST_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--slices',
           'independent'] + SMALL
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json_line(out.stdout)
    assert d['slices'] == 'independent' and d['verified_vs_c_oracle'] is True
    for lib in d['config']['libraries']:
        assert lib['exchange_consistent'] is True and all(lib['verified_vs_c_oracle'].values()), lib


def test_two_ranks_from_one_bam_file_per_library():
    """--from-bam: rank 0 writes each library as one sequencer-like BAM, both ranks ingest their slices on the GPU
    (distributed.ingest_slice), the timed steps and the oracle check run on the ingested records."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BESST_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--from-bam'] + SMALL
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json_line(out.stdout)
    assert d['n_gpus'] == 2 and d['verified_vs_c_oracle'] is True
    assert len(d['from_bam']) == 2
    for lib in d['from_bam']:
        assert lib['records'] == 2 * 2 * 400000 and sum(r['records'] for r in lib['per_rank']) == lib['records']
        assert [r['rank'] for r in lib['per_rank']] == [0, 1]
        assert all(r['ingest_s'] > 0 and r['staging_s'] >= 0 and r['chunks'] >= 1 for r in lib['per_rank'])
        assert lib['handshake_rounds'] >= 1 and lib['aggregate_records_per_s'] > 0
        assert lib['rank0_slice_alone']['records'] == lib['per_rank'][0]['records']
    assert [i['library'] for i in d['single_gpu_same_shape']['ingest']] == [1, 2] and d['single_gpu_same_shape']['ingest'][0]['records_per_s'] > 0
    for lib in d['config']['libraries']:
        assert lib['exchange_consistent'] is True and all(lib['verified_vs_c_oracle'].values()), lib
    assert abs(d['value'] - 2 * 2 * 400000 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']


def test_gpus_n_without_a_launcher_refuses_on_a_box_with_fewer_gpus():
    """`python bench.py --gpus 2` with WORLD_SIZE unset used to fall into the one-GPU path and print n_gpus 1.  Over RCCL it
    needs two devices: on this box it must say so, quickly, with a non-zero exit code and no JSON line."""
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('a multi-GPU box starts the ranks instead')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BESST_DIST_BACKEND')}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2'] + SMALL, cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and time.time() - t0 < 60
    assert 'one device per rank' in out.stderr and 'nothing was measured' in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]


def test_gpus_n_that_disagrees_with_the_launcher_is_refused():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, BESST_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '4'] + SMALL
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode != 0 and 'wrong n_gpus' in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """The self-launch (one rank per requested GPU under torch.distributed.run); gloo so that both fit this box's one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['BESST_DIST_BACKEND'] = 'gloo'
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2'] + SMALL, cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json_line(out.stdout)
    assert d['n_gpus'] == 2 and d['rccl_ranks_seen'] == 2 and d['verified_vs_c_oracle'] is True
    assert d['single_gpu_same_shape']['value'] > 0


def test_one_rank_of_the_c4_shape_is_the_first_point_of_the_curve():
    """--config C4 --gpus 1: the per-GPU shape of --gpus 2 / 4 / 8 through the same orchestration over RCCL with one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BESST_DIST_BACKEND')}
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--config', 'C4'] + SMALL, cwd=REPO,
                         env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json_line(out.stdout)
    assert d['n_gpus'] == 1 and d['backend'] == 'nccl' and d['rccl_ranks_seen'] == 1
    assert d['config']['workload'].startswith('C4-shaped') and len(d['config']['libraries']) == 2
    same = d['single_gpu_same_shape']
    assert same['value'] == d['value'] and same['ms_per_step'] > 0
    assert d['verified_vs_c_oracle'] is True
