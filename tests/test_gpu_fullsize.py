"""GPU parity at larger sizes against the C oracle, plus size-independent properties of the edge table."""
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from besst_amd import workload

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['two_pass', 'fused'])
def record_path(request, monkeypatch):
    """Every scenario runs through both forms of the record loop (besst_lib_params.record_path): stream_kernel +
    ordered_kernel, and fused_kernel."""
    monkeypatch.setenv('BESST_RECORD_PATH', '0' if request.param == 'two_pass' else '1')
    return request.param


def assert_table_equals_c_oracle(table, aligned, ctr, batch, wl):
    keys, payload, c_aligned, c_ctr = CO.record_loop(batch, wl['table'], wl['lib'], wl['node_bits'])
    rows = CO.edge_rows(keys, payload)
    assert aligned.tolist() == c_aligned.tolist()
    assert [ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates, ctr.reads_with_too_long_insert,
            ctr.fishy_reads, ctr.n_tuples, ctr.n_reach, ctr.prev_obs1, ctr.prev_obs2] == c_ctr.tolist()
    assert np.array_equal(table.key, rows['key'])
    assert np.array_equal(table.n.astype(np.int64), rows['n'])
    assert np.array_equal(table.first_idx.astype(np.int64), rows['first_idx'])
    assert np.array_equal(table.offset.astype(np.int64), rows['offset'])
    link = ~table.is_fishy
    assert np.array_equal(table.sum_obs[link], rows['sum_obs'][link])
    assert np.array_equal(table.sum_obs_sq[link], rows['sum_obs_sq'][link])
    assert np.array_equal(table.mask[link].astype(np.int64), rows['mask'][link])
    assert np.array_equal(table.obs_lo.astype(np.int64), rows['obs_lo'])
    assert np.array_equal(table.obs_hi.astype(np.int64), rows['obs_hi'])
    # properties that hold at any size
    assert np.all(np.diff(table.key.astype(np.uint64)) > 0)                      # strictly sorted, unique keys
    assert int(table.n.sum()) == ctr.n_tuples                                     # every tuple lands in one row
    assert np.array_equal(np.cumsum(np.r_[0, table.n[:-1]]), table.offset)        # slices tile the arrays
    o = table.obs_lo.astype(np.int64) + table.obs_hi
    assert int(table.sum_obs[link].sum()) == int(o.sum())
    assert int(table.sum_obs_sq[link].sum()) == int((o * o).sum())


@pytest.mark.parametrize('config,pairs,nc', [('C2', None, None),         # BASELINE.json configs[1] at FULL size: 10 k contigs / 10 M pairs
                                             ('C2', 1_000_000, 3000), ('C3', 600_000, 2000),
                                             ('C3', 5_000_000, 4000)])   # > 262144 tuples: 8-bit scanned sort path
def test_device_equals_c_oracle(config, pairs, nc):
    from besst_amd import device
    if pairs is None and os.environ.get('BESST_FULL_SIZE') == '0':
        pytest.skip('BESST_FULL_SIZE=0')
    wl = workload.make(config, 0, pairs=pairs, nc=nc)
    batch = wl['batch']
    with device.GraphContext(0) as ctx:
        ctx.set_contigs(**wl['table'])
        lib = wl['lib']
        ctx.set_library(lib['read_len'], lib['ins_size_threshold'], lib['min_mapq'], lib['orientation'],
                        lib['detect_duplicate'], lib['extend_paths'], lib['no_score'])
        ctx.push_records(batch)
        table, aligned, ctr = ctx.build_graph()
        assert_table_equals_c_oracle(table, aligned, ctr, batch, wl)


def _host_memory_gib():
    try:
        with open('/proc/meminfo') as fh:
            for line in fh:
                if line.startswith('MemAvailable:'):
                    return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def test_full_size_c3_resident_builder(record_path):
    """BASELINE.json configs[2] at FULL size - 100 k contigs / 200 M mate pairs with PE contamination, 400 M records =
    9.2 GB resident - through DeviceGraphBuilder.step(), the call bench.py times, against the C oracle on all 400 M
    records (slice-parallel over the host's cores; the one-thread run of bench.py's cpu_baseline takes minutes and is
    not repeated here).  Guarded only by what the box has: 60 GB of free HBM, 48 GiB of available host memory."""
    import torch
    from besst_amd import pipeline
    if os.environ.get('BESST_FULL_SIZE') == '0':
        pytest.skip('BESST_FULL_SIZE=0')
    if record_path != 'fused':
        pytest.skip('once, with the record loop the candidate density of a mate-pair library selects')
    free, _ = torch.cuda.mem_get_info(0)
    if free < 60e9 or _host_memory_gib() < 48:
        pytest.skip('needs 60 GB of free HBM and 48 GiB of host memory (%.0f GB / %.0f GiB here)' % (free / 1e9, _host_memory_gib()))
    dev = torch.device('cuda', 0)
    wl = workload.make_device(dev, 'C3', 0)
    rec = pipeline.DeviceRecords.from_columns(wl['cols'])
    assert rec.n == 400_000_000
    probe = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, 1)
    probe.set_contigs(**wl['table'])
    probe.reset()
    probe.classify(rec)
    n_tuples, _ = probe.read_sizes()
    del probe
    gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, int(n_tuples * 1.25) + 4096)
    gb.set_contigs(**wl['table'])
    for _ in range(2):
        gb.step(rec)
    table = gb.fetch_table()
    assert gb.sort_flags == 0, 'the run-grouped form serves the headline workload'
    ctr = gb.read_counters()
    batch = wl['batch']                                  # (the device columns copied to the host)
    cores = max(1, min(64, os.cpu_count() or 1))
    keys, payload, c_aligned, c_ctr = CO.record_loop(batch, wl['table'], wl['lib'], wl['node_bits'], threads=cores)
    rows = CO.edge_rows(keys, payload)
    assert [ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates, ctr.reads_with_too_long_insert,
            ctr.fishy_reads, ctr.n_tuples, ctr.n_reach, ctr.prev_obs1, ctr.prev_obs2] == c_ctr.tolist()
    assert gb.aligned.cpu().numpy().tolist() == c_aligned.tolist()
    assert np.array_equal(table.key, rows['key']) and np.array_equal(table.n.astype(np.int64), rows['n'])
    assert np.array_equal(table.first_idx.astype(np.int64), rows['first_idx'])
    assert np.array_equal(table.offset.astype(np.int64), rows['offset'])
    link = ~table.is_fishy
    assert np.array_equal(table.sum_obs[link], rows['sum_obs'][link])
    assert np.array_equal(table.sum_obs_sq[link], rows['sum_obs_sq'][link])
    assert np.array_equal(table.mask[link].astype(np.int64), rows['mask'][link])
    assert np.array_equal(table.obs_lo.astype(np.int64), rows['obs_lo'])
    assert np.array_equal(table.obs_hi.astype(np.int64), rows['obs_hi'])
    # size-independent properties
    assert np.all(np.diff(table.key.astype(np.uint64)) > 0)
    assert int(table.n.sum()) == ctr.n_tuples == len(keys)
    assert np.array_equal(np.cumsum(np.r_[0, table.n[:-1]]), table.offset)


def test_split_distribution_from_device_histogram():
    """Row a7 end to end on the device side: the count-per-value histogram of every reference-captured sample
    (tests/golden/unit_golden.json 'split', produced by BESST/find_bimodality.py:39-205 itself) is built by
    value_hist_kernel, fed to split_from_histogram, and must give the reference's clusters and moments exactly."""
    import json
    import os
    from besst_amd import device, find_bimodality
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'unit_golden.json')) as fh:
        gold = json.load(fh)['split']
    assert len(gold) >= 5
    with device.GraphContext(0) as ctx:
        for g in gold:
            vals = np.asarray(g['values'], dtype=np.int64)
            assert np.all(vals >= 0)
            n_bins = int(vals.max()) + 1
            hist, overflow = ctx.value_histogram(vals.astype(np.int32), n_bins)
            assert int(overflow) == 0 and int(hist.sum()) == len(vals)
            values = np.nonzero(hist)[0]
            counts = hist[values]
            split, m1, s1, m2, s2 = find_bimodality.split_from_histogram(values.tolist(), [int(c) for c in counts])
            if split < 0:
                got = ([], [], 0, 0, 0, 0)
            else:
                c1 = [int(v) for v, c in zip(values[:split], counts[:split]) for _ in range(int(c))]
                c2 = [int(v) for v, c in zip(values[split:], counts[split:]) for _ in range(int(c))]
                got = (c1, c2, m1, s1, m2, s2)
            assert got[0] == g['cluster1'] and got[1] == g['cluster2']
            assert (float(got[2]), float(got[3]), float(got[4]), float(got[5])) == \
                (g['mean1'], g['stddev1'], g['mean2'], g['stddev2'])


@pytest.mark.parametrize('seg_window,sort_form', [(None, 'runs'), (None, 'tuples'), ('2', 'tuples')])
def test_resident_builder_step_large_stream(seg_window, sort_form, monkeypatch):
    """DeviceGraphBuilder.step() - what bench.py times - on a mate-pair library large enough for the large-stream sort
    (6.4 M link tuples): the sort gets its digit histograms from stage 1 (besst_dev_classify_presort /
    besst_dev_reduce_presorted) - counted by compact_kernel after the two-pass record loop, by the fused record loop
    itself otherwise, and then the first chained-scan pass reads the tuples from the block segments (seg_window '2':
    with a two-block window, so that the tiles look their blocks up in memory) -, two chained-scan passes,
    wave-per-bucket sort + reduction ('tuples': BESST_REDUCE_NO_RUNS); 'runs' is the default form, chunks of the stream
    grouped into runs of equal keys and only the runs sorted (csrc/runs.hip), reading the same segments or the compacted
    stream.  Twice on the same builder (the second pass starts from the first one's leftovers
    in every workspace), against the C oracle."""
    import torch
    from besst_amd import pipeline
    if seg_window:
        monkeypatch.setenv('BESST_SEG_WINDOW', seg_window)
    dev = torch.device('cuda', 0)
    wl = workload.make_device(dev, 'C3', 0, pairs=30_000_000, nc=20_000)
    # every 37th contig is not in the table (a later pass: repeats and low-coverage contigs are gone): records on them
    # or with their mate on them add no coverage - the fused loop credits candidates early and has to take that back -
    # and the coverage of the contigs themselves is cleared by whichever kernel runs after the stitch
    table = {k: np.array(v, copy=True) for k, v in wl['table'].items()}
    table['cls'][::37] = 0
    wl['table'] = table
    rec = pipeline.DeviceRecords.from_columns(wl['cols'])
    probe = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, 1)
    probe.set_contigs(**wl['table'])
    probe.reset()
    probe.classify(rec)
    n_tuples, _ = probe.read_sizes()
    del probe
    assert n_tuples > 4_500_000
    gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, int(n_tuples * 1.25) + 4096)
    gb.set_contigs(**wl['table'])
    gb.sort_flags = pipeline.REDUCE_NO_RUNS if sort_form == 'tuples' else 0
    for _ in range(2):
        gb.step(rec)
    assert gb._args['presort'][1], 'the large-stream sort should take its histograms from stage 1'
    spec = gb._args['presort'][0]
    assert bool(spec.segmented) == bool(spec.in_record_loop) == (gb.params.record_path == 1)
    table = gb.fetch_table()
    assert gb.sort_flags == (pipeline.REDUCE_NO_RUNS if sort_form == 'tuples' else 0)
    ctr = gb.read_counters()
    assert_table_equals_c_oracle(table, gb.aligned.cpu().numpy(), ctr, wl['batch'], wl)


@pytest.mark.parametrize('config,pairs,nc,variant', [('C3', 6_000_000, 20_000, dict(chimeric_frac=0.03)),
                                                    ('C3', 6_000_000, 20_000, dict(chimeric_frac=0.5)),
                                                    ('C2', 3_000_000, 5_000, dict(order='name')),
                                                    ('C3', 3_000_000, 5_000, dict(order='name'))])
def test_variant_streams_resident_builder(config, pairs, nc, variant):
    """The workloads of bench.py's `robustness` object at reduced size, through DeviceGraphBuilder.step(): chimeric pairs
    (links on edges of their own: many runs per chunk, ten times the rows) and name-sorted streams (mates adjacent,
    pairs in random order: no clustering anywhere, stage 2 leaves the run-grouped form when the runs overflow)."""
    import torch
    from besst_amd import pipeline
    dev = torch.device('cuda', 0)
    wl = workload.make_device(dev, config, 0, pairs=pairs, nc=nc, **variant)
    rec = pipeline.DeviceRecords.from_columns(wl['cols'])
    gb = pipeline.DeviceGraphBuilder(dev, wl['asm'].nc, wl['node_bits'], wl['lib'], rec.n, rec.n)
    gb.set_contigs(**wl['table'])
    for _ in range(2):
        gb.step(rec)
    table = gb.fetch_table()
    ctr = gb.read_counters()
    assert_table_equals_c_oracle(table, gb.aligned.cpu().numpy(), ctr, wl['batch'], wl)
    if 'chimeric_frac' in variant:
        plain = workload.make_device(dev, config, 0, pairs=pairs, nc=nc)
        keys, _, _, _ = CO.record_loop(plain['batch'], plain['table'], plain['lib'], plain['node_bits'])
        assert len(table) > 1.5 * len(np.unique(keys))                           # the chimeric links did make edges


def oracle_in_slices(cols, asm, table, lib, node_bits, slice_records=100_000_000, threads=None, read_len=100):
    """The C oracle over device-resident columns without a host copy of the whole stream: slices of `slice_records`, the
    duplicate chain's state (CreateGraph.py:835-838,869-870) carried from slice to slice, counters and coverage summed,
    tuples concatenated in stream order."""
    from besst_amd import synth
    n = int(cols['tid'].shape[0])
    threads = threads or max(1, min(64, os.cpu_count() or 1))
    prev, keys, payload = (-1, -1), [], []
    aligned, ctr = None, None
    for lo in range(0, n, slice_records):
        part = synth.device_columns_to_batch(asm, {k: v[lo:lo + slice_records] for k, v in cols.items()}, read_len)
        k, p, a, c = CO.record_loop(part, table, lib, node_bits, prev=prev, threads=threads)
        prev = (int(c[8]), int(c[9]))
        keys.append(k)
        payload.append(p)
        aligned = a.copy() if aligned is None else aligned + a
        ctr = c.copy() if ctr is None else np.concatenate([ctr[:8] + c[:8], c[8:]])
        del part
    return np.concatenate(keys), np.concatenate(payload), aligned, ctr


def test_full_size_c4_one_gpu(record_path):
    """BASELINE.json configs[3] at FULL size on ONE GPU: 500 k contigs, two libraries of 5e8 read pairs = 1e9 records (25 GB)
    each - more than 2^30 records in one stream, 41-bit edge keys at full density - the PE library on the first-library
    table, then the mate-pair library with PE contamination on the contig table a previous pass leaves behind (scaffold ids
    counting on, MakeScaffolds.py:276), each through DeviceGraphBuilder.step() against the C oracle on every record."""
    import torch
    from besst_amd import pipeline, synth
    if os.environ.get('BESST_FULL_SIZE') == '0':
        pytest.skip('BESST_FULL_SIZE=0')
    if record_path != 'fused':
        pytest.skip('once (the record loop is selected per library by its candidate density)')
    os.environ.pop('BESST_RECORD_PATH', None)
    free, _ = torch.cuda.mem_get_info(0)
    if free < 150e9 or _host_memory_gib() < 64:
        pytest.skip('needs 150 GB of free HBM and 64 GiB of host memory (%.0f GB / %.0f GiB here)' % (free / 1e9, _host_memory_gib()))
    dev = torch.device('cuda', 0)
    cfg = synth.CONFIGS['C4']
    seed = synth.config_seed('C4')
    asm = synth.make_assembly(cfg['nc'], cfg['median'], seed)
    per_lib = cfg['pairs'] // len(cfg['libs'])
    assert per_lib == 500_000_000
    for li, spec in enumerate(cfg['libs']):
        lib = workload.library_constants(spec)
        thr = spec.mean + 4 * spec.sd
        table = workload.first_library_table(asm.lengths, thr) if li == 0 else \
            workload.later_library_table(asm, seed + 50 + li, thr, first_scaffold_id=asm.nc * li + 1)
        node_bits = workload.node_bits_for(table)
        assert node_bits >= 20
        cols = synth.simulate_library_device(asm, spec, per_lib, seed + 100 + li, dev)
        rec = pipeline.DeviceRecords.from_columns(cols)
        assert rec.n == 1_000_000_000
        probe = pipeline.DeviceGraphBuilder(dev, asm.nc, node_bits, lib, rec.n, 1)
        probe.set_contigs(**table)
        probe.reset()
        probe.classify(rec)
        n_tuples, _ = probe.read_sizes()
        del probe
        gb = pipeline.DeviceGraphBuilder(dev, asm.nc, node_bits, lib, rec.n, int(n_tuples * 1.1) + 4096)
        gb.set_contigs(**table)
        for _ in range(2):
            gb.step(rec)
        got = gb.fetch_table()
        ctr = gb.read_counters()
        keys, payload, c_aligned, c_ctr = oracle_in_slices(cols, asm, table, lib, node_bits)
        rows = CO.edge_rows(keys, payload)
        assert [ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates, ctr.reads_with_too_long_insert,
                ctr.fishy_reads, ctr.n_tuples, ctr.n_reach, ctr.prev_obs1, ctr.prev_obs2] == c_ctr.tolist()
        assert np.array_equal(gb.aligned.cpu().numpy(), c_aligned)
        link = ~got.is_fishy
        assert np.array_equal(got.key, rows['key']) and np.array_equal(got.n.astype(np.int64), rows['n'])
        assert np.array_equal(got.first_idx.astype(np.int64), rows['first_idx'])
        assert np.array_equal(got.offset.astype(np.int64), rows['offset'])
        assert np.array_equal(got.sum_obs[link], rows['sum_obs'][link])
        assert np.array_equal(got.sum_obs_sq[link], rows['sum_obs_sq'][link])
        assert np.array_equal(got.mask[link].astype(np.int64), rows['mask'][link])
        assert np.array_equal(got.obs_lo.astype(np.int64), rows['obs_lo'])
        assert np.array_equal(got.obs_hi.astype(np.int64), rows['obs_hi'])
        assert np.all(np.diff(got.key.astype(np.uint64)) > 0) and int(got.n.sum()) == ctr.n_tuples == len(keys)
        assert len(rows['key']) > 100_000 and ctr.nr_of_duplicates > 0
        print('C4 library %d (%s): %d tuples, %d edge rows, record path %s' % (li, spec.orientation, ctr.n_tuples, len(got),
                                                                               'fused' if gb.params.record_path else 'two-pass'))
        del gb, rec, cols, got, keys, payload, rows
        torch.cuda.empty_cache()


@pytest.mark.parametrize('lib_index,table_kind', [(1, 'later'), (0, 'first')])
def test_full_size_c5_library_one_gpu(record_path, lib_index, table_kind):
    """(lib_index 0: the 500 bp paired-end library on the first-library table - the sparse record loop, stream_kernel +
    ordered_kernel, past 2^31 records.)
    ONE library of BASELINE.json configs[4] (C5) at FULL size on one GPU: 2 M contigs, 1.33 G read pairs = 2.67 G records
    in one stream - more than 2^31 (every record index past that point needs its 32nd bit) - the 5 kb mate-pair library
    with PE contamination on the contig table a previous pass leaves behind (scaffold ids counting on from 2 M,
    MakeScaffolds.py:276: 47-bit edge keys).  DeviceGraphBuilder.step() against the C oracle on every record, in slices with
    the duplicate chain carried; the library-metrics pass over the whole stream against the oracle's counts."""
    import torch
    from besst_amd import pipeline, synth
    if os.environ.get('BESST_FULL_SIZE') == '0':
        pytest.skip('BESST_FULL_SIZE=0')
    if record_path != 'fused':
        pytest.skip('once (the record loop is selected per library by its candidate density)')
    os.environ.pop('BESST_RECORD_PATH', None)
    import gc
    gc.collect()
    torch.cuda.empty_cache()                                 # (what the test before this one left in the caching allocator)
    free, _ = torch.cuda.mem_get_info(0)
    if free < 200e9 or _host_memory_gib() < 64:
        pytest.skip('needs 200 GB of free HBM and 64 GiB of host memory (%.0f GB / %.0f GiB here)' % (free / 1e9, _host_memory_gib()))
    dev = torch.device('cuda', 0)
    wl = workload.make_device_windowed(dev, 'C5', lib_index, table=table_kind)
    asm, cols, table, lib, node_bits = wl['asm'], wl['cols'], wl['table'], wl['lib'], wl['node_bits']
    rec = pipeline.DeviceRecords.from_columns(cols)
    assert 0 <= 2_666_666_666 - rec.n <= 512 and rec.n > 1 << 31 and asm.nc == 2_000_000
    key = (cols['tid'][1:].to(torch.int64) << 32) | cols['pos'][1:].to(torch.int64)
    assert bool((key >= ((cols['tid'][:-1].to(torch.int64) << 32) | cols['pos'][:-1].to(torch.int64))).all())   # one sorted stream
    del key
    torch.cuda.empty_cache()
    probe = pipeline.DeviceGraphBuilder(dev, asm.nc, node_bits, lib, rec.n, 1)
    probe.set_contigs(**table)
    probe.reset()
    probe.classify(rec)
    n_tuples, _ = probe.read_sizes()
    del probe
    gb = pipeline.DeviceGraphBuilder(dev, asm.nc, node_bits, lib, rec.n, int(n_tuples * 1.1) + 4096)
    gb.set_contigs(**table)
    for _ in range(2):
        gb.step(rec)
    got = gb.fetch_table()
    ctr = gb.read_counters()
    keys, payload, c_aligned, c_ctr = oracle_in_slices(cols, asm, table, lib, node_bits)
    rows = CO.edge_rows(keys, payload)
    assert [ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates, ctr.reads_with_too_long_insert,
            ctr.fishy_reads, ctr.n_tuples, ctr.n_reach, ctr.prev_obs1, ctr.prev_obs2] == c_ctr.tolist()
    assert np.array_equal(gb.aligned.cpu().numpy(), c_aligned)
    link = ~got.is_fishy
    assert np.array_equal(got.key, rows['key']) and np.array_equal(got.n.astype(np.int64), rows['n'])
    assert np.array_equal(got.first_idx.astype(np.int64), rows['first_idx'])
    assert np.array_equal(got.offset.astype(np.int64), rows['offset'])
    assert np.array_equal(got.sum_obs[link], rows['sum_obs'][link])
    assert np.array_equal(got.sum_obs_sq[link], rows['sum_obs_sq'][link])
    assert np.array_equal(got.mask[link].astype(np.int64), rows['mask'][link])
    assert np.array_equal(got.obs_lo.astype(np.int64), rows['obs_lo'])
    assert np.array_equal(got.obs_hi.astype(np.int64), rows['obs_hi'])
    assert np.all(np.diff(got.key.astype(np.uint64)) > 0) and int(got.n.sum()) == ctr.n_tuples == len(keys)
    if table_kind == 'later':
        assert int(got.key.max()) >> 44, 'keys of a later pass on 2 M contigs need more than 44 bits'
    assert (gb.params.record_path == 1) == (lib_index == 1)      # dense mate pairs: the fused loop; sparse paired ends: two passes
    # records beyond 2^31 did make tuples: the last tuple's record lies in the last stretch of the stream
    assert len(rows['key']) > 100_000 and ctr.nr_of_duplicates > 0
    print('C5 library %d (%s): %d records, %d tuples, %d edge rows, key bits %d, record path %s'
          % (lib_index, wl['spec'].orientation, rec.n, ctr.n_tuples, len(got), gb.key_bits, 'fused' if gb.params.record_path else 'two-pass'))
    del gb, got, keys, payload, rows
    torch.cuda.empty_cache()
    # the library-metrics pass over all 2.67 G records in ONE call (besst_dev_metrics_sample refused n >= 2^31): the "1000
    # longest contigs" are put at the END of the header, so every sampled record lies beyond 2^31 and the scan has to walk
    # the whole stream to reach them (libmetrics.py:63-84, 293-303)
    top = np.zeros(asm.nc, np.uint8)
    top[asm.nc - 1000:] = 1
    lo = int(torch.searchsorted(cols['tid'], torch.tensor([asm.nc - 1000], dtype=torch.int32, device=dev))[0].item())
    assert lo > 1 << 31
    sampler = pipeline.DeviceMetricsSampler(dev, rec, asm.nc)
    sampler.set_top(top)
    local = sampler.count(lib['orientation'], lib['min_mapq'], lib['read_len']).cpu().numpy().copy()
    samples, state = sampler.emit(torch.zeros(3, dtype=torch.int64, device=dev), lib['orientation'], lib['min_mapq'],
                                  lib['read_len'], True)
    state = state.cpu().numpy()
    host = samples.cpu().numpy()
    lo -= lo % 4
    tail = synth.device_columns_to_batch(asm, {k: v[lo:] for k, v in cols.items()}, int(wl['spec'].read_len))
    want_isize, want_contam, c = CO.metrics_sample(tail, top, lib['orientation'], lib['min_mapq'], lib['read_len'], True)
    cap = pipeline.SAMPLE_CAP
    n_isize, n_contam = int(min(local[0], cap)), int(state[4])
    assert (n_isize, n_contam, int(state[3]), int(min(local[1], cap))) == tuple(int(x) for x in c[:4])
    assert n_isize > 10_000 and (n_contam > 1_000 or wl['spec'].contam_frac == 0)
    assert np.array_equal(host[:n_isize], want_isize) and np.array_equal(host[cap:cap + n_contam], want_contam)
