"""GPU parity of the multi-GPU kernel stages (tail / carry / emit / owner partition / pack / unpack / mapped reduce).

The GPU box has one MI355X, so W ranks are simulated in one process: each rank gets its own HipBackend on
cuda:0 with its contiguous slice of the stream, and the collectives are replaced by the equivalent tensor
shuffles.  The merged result must equal the single-process oracle (and hence the single-GPU build).
"""
import pytest

from tests import dist_util as DU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world,orientation,coverage,heads', [
    (2, 'fr', 'rider', 'gather'), (3, 'rf', 'allreduce', 'gather'), (8, 'fr', 'auto', 'gather'),
    (2, 'fr', 'allreduce', 'gather'), (2, 'fr', 'rider', 'exchange'), (3, 'rf', 'allreduce', 'exchange'),
    (8, 'fr', 'auto', 'exchange'), (5, 'rf', 'rider', 'exchange')])
def test_simulated_ranks_match_oracle(world, orientation, coverage, heads, monkeypatch):
    """heads: how the duplicate chain crosses slices - 'gather': the 16-byte tails are exchanged before the emit stage;
    'exchange': every slice emits with its first reaching record unresolved, the owners resolve the heads from the
    exchange headers (no tail exchange).
    coverage: how the coverage numerators and counters are summed - as a rider of the exchange regions (the
    receivers sum their sources' copies; 'auto' picks it for these small assemblies) or by the all-reduce."""
    import torch
    from besst_amd import distributed, workload
    monkeypatch.setenv('BESST_COVERAGE_EXCHANGE', coverage)
    wl = workload.make('C2', 0, pairs=150000, nc=700)
    if orientation == 'rf':
        wl = workload.make('C3', 0, pairs=150000, nc=300)
    dev = torch.device('cuda', 0)
    parts = DU.split_batch(wl['batch'], world)
    pair_cap = 8192
    backends = []
    for r in range(world):
        sub = dict(wl)
        sub['batch'] = parts[r]
        backends.append(distributed.HipBackend(dev, sub, r, world, pair_cap))
    for _ in range(2):
        tails = []
        for b in backends:
            b.reset()
            b.classify_scan()
            tails.append(b.classify_tail().clone())
        tails = torch.cat(tails)
        sends = []
        for b in backends:
            if heads == 'exchange':
                b.classify_emit_speculative()
                assert b.slice_info[:3].tolist() == tails[4 * b.rank:4 * b.rank + 3].tolist()
            else:
                b.classify_emit(tails)
            sends.append(b.partition().clone())
        if not backends[0].sums_ride_exchange:              # what the step's all-reduce does at this point
            assert coverage == 'allreduce'
            total = sum(b.pack_for_allreduce().clone() for b in backends)
            for b in backends:
                b.pack_for_allreduce().copy_(total)
        else:
            assert coverage != 'allreduce'
        region = backends[0].region
        for r, b in enumerate(backends):
            recv = torch.cat([sends[s][r * region:(r + 1) * region] for s in range(world)])
            b.unpack(recv)
            b.reduce()
        torch.cuda.synchronize()
    assert not any(b.overflowed() for b in backends)
    want_rows, want = DU.expected_rows(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
    assert want.nr_of_duplicates > 0 and want.fishy_reads > 0
    if heads == 'exchange':
        for b in backends:                                   # every owner saw every slice's description
            assert b.all_slice_info.view(world, 8)[:, :3].tolist() == tails.view(world, 4)[:, :3].tolist()
    # riders summed by every receiver, or all-reduced: each rank holds the global values
    for b in backends[1:]:
        assert torch.equal(b.aligned, backends[0].aligned)
        assert torch.equal(b.counter_words[:8], backends[0].counter_words[:8])
    aligned = backends[0].aligned.cpu()
    counters = backends[0].counter_words.cpu()
    assert aligned.tolist() == want.aligned
    assert counters.tolist() == [want.count, want.non_unique, want.non_unique_for_scaf, want.nr_of_duplicates,
                                 want.too_long, want.fishy_reads, len(want.tuples), want.n_reach]
    merged = {}
    lib = backends[0].lib
    for r, b in enumerate(backends):
        rows = DU.rows_from_table(b.local_table())
        for k in rows:
            assert lib.besst_owner_of_scaffold(k >> (2 + wl['node_bits']), world) == r
            if k & 1:                                  # fishy rows carry no observations
                rows[k]['lo'] = [0] * rows[k]['n']
                rows[k]['hi'] = [0] * rows[k]['n']
        assert not set(rows) & set(merged)
        merged.update(rows)
    for k, r in want_rows.items():
        if k & 1:
            r['s'] = r['s2'] = 0
    assert merged == want_rows


@pytest.mark.parametrize('world,config,pairs', [(2, 'C2', 3_000_000), (3, 'C3', 1_500_000), (8, 'C2', 200_000)])
def test_simulated_ranks_metrics_sample(world, config, pairs):
    """Sharded libmetrics scans: per-slice counts, prefix, samples written at their global positions; the sum of the
    ranks' buffers (what the all-reduce computes) equals the single-process scan.  The 3 M-pair case reaches the
    first-1,000,000 cut-off inside a later slice."""
    import numpy as np
    import torch
    from besst_amd import pipeline, workload
    from oracle import c_oracle as CO
    wl = workload.make(config, 0, pairs=pairs, nc=600)
    lib, asm, batch = wl['lib'], wl['asm'], wl['batch']
    top = np.zeros(asm.nc, np.uint8)
    top[np.lexsort((np.arange(asm.nc), -asm.lengths))[:500]] = 1
    dev = torch.device('cuda', 0)
    samplers = []
    for part in DU.split_batch(batch, world):
        s = pipeline.DeviceMetricsSampler(dev, pipeline.DeviceRecords(part, dev), asm.nc)
        s.set_top(top)
        samplers.append(s)
    args = (lib['orientation'], lib['min_mapq'], lib['read_len'])
    counts = [s.count(*args).clone() for s in samplers]
    total_samples = torch.zeros(2 * pipeline.SAMPLE_CAP, dtype=torch.int32, device=dev)
    total_state = torch.zeros(3, dtype=torch.int64, device=dev)
    before = torch.zeros(3, dtype=torch.int64, device=dev)
    for r, s in enumerate(samplers):
        samples, state = s.emit(before, *args)
        total_samples += samples
        total_state += state[3:6]
        before = before + counts[r]
    n_all = before.cpu().numpy()
    tot = total_state.cpu().numpy()
    w_isize, w_contam, w_counts = CO.metrics_sample(batch, top, *args)
    n_isize = int(min(n_all[0], pipeline.SAMPLE_CAP))
    host = total_samples.cpu().numpy()
    assert [n_isize, int(tot[1]), int(tot[0]), int(min(n_all[1], pipeline.SAMPLE_CAP))] == w_counts.tolist()
    assert np.array_equal(host[:n_isize], w_isize)
    assert np.array_equal(host[pipeline.SAMPLE_CAP:pipeline.SAMPLE_CAP + int(tot[1])], w_contam)
    assert not host[n_isize:pipeline.SAMPLE_CAP].any()
    if pairs >= 3_000_000:
        assert n_all[0] > pipeline.SAMPLE_CAP                  # the cut-off really was crossed


@pytest.mark.parametrize('n_records', [0, 1, 63, 64, 1000, 1024, 1025, 5000])
def test_tail_of_short_and_linkless_slices(n_records):
    """The slice's tail (the last record that reaches CreateEdge, from the block summaries): slices shorter than a
    block, exactly a block, empty ones, and slices without any reaching record (records of one contig only) against
    the C oracle's prev_obs after the same records."""
    import numpy as np
    import torch
    from besst_amd import distributed, workload
    from oracle import c_oracle as CO
    wl = workload.make('C2', 0, pairs=40000, nc=60)
    dev = torch.device('cuda', 0)
    batch = wl['batch']
    same = np.nonzero((batch.tid == batch.mtid))[0][:n_records]
    slices = [batch.slice(start, start + n_records) for start in (0, len(batch) // 3, len(batch) - n_records)]
    if len(same):
        slices.append(batch.take(same))                      # a single contig's interior has no link at all
    for k, part in enumerate(slices):
        sub = dict(wl)
        sub['batch'] = part
        b = distributed.HipBackend(dev, sub, 0, 1, 4096)
        b.reset()
        b.classify_scan()
        got = b.classify_tail().cpu().tolist()
        ctr = CO.record_loop(part, wl['table'], wl['lib'], wl['node_bits'])[3]
        want = [1, int(ctr[8]), int(ctr[9])] if int(ctr[7]) > 0 else [0]
        assert got[:len(want)] == want
        if k == 3:
            assert got[0] == 0


def test_simulated_ranks_score_their_own_edges():
    """Edge scoring per owner rank (besst_dev_score_edges on each rank's rows) equals scoring the single-GPU table."""
    import numpy as np
    import torch
    from besst_amd import device, distributed, workload
    wl = workload.make('C2', 0, pairs=400000, nc=500)
    asm, lib = wl['asm'], wl['lib']
    dev = torch.device('cuda', 0)
    world = 3
    parts = DU.split_batch(wl['batch'], world)
    backends = []
    for r in range(world):
        sub = dict(wl)
        sub['batch'] = parts[r]
        backends.append(distributed.HipBackend(dev, sub, r, world, 16384))
    tails = []
    for b in backends:
        b.reset()
        b.classify_scan()
        tails.append(b.classify_tail().clone())
    tails = torch.cat(tails)
    sends = []
    for b in backends:
        b.classify_emit(tails)
        sends.append(b.partition().clone())
    region = backends[0].region
    for r, b in enumerate(backends):
        b.unpack(torch.cat([sends[s][r * region:(r + 1) * region] for s in range(world)]))
        b.reduce()
    torch.cuda.synchronize()

    def pick(table):
        rows = np.nonzero((~table.is_fishy) & ((table.mask & 1) != 0) & (table.n >= 5))[0].astype(np.uint32)
        len1 = asm.lengths[(table.u[rows] >> 1) - 1].astype(np.int32)
        len2 = asm.lengths[(table.v[rows] >> 1) - 1].astype(np.int32)
        return rows, np.zeros(len(rows), np.uint8), len1, len2

    got = {}
    for b in backends:
        table = b.local_table()
        rows, swap, len1, len2 = pick(table)
        gap, sd0, ks, flags = b.gb.score_edges(rows, swap, len1, len2, lib['mean'], lib['sd'], lib['read_len'])
        for i, row in enumerate(rows):
            got[int(table.key[row])] = (float(gap[i]), float(sd0[i]), int(ks[i]), int(flags[i]))
    with device.GraphContext(0) as ctx:
        ctx.set_contigs(**wl['table'])
        ctx.set_library(lib['read_len'], lib['ins_size_threshold'], lib['min_mapq'], lib['orientation'],
                        lib['detect_duplicate'], lib['extend_paths'], lib['no_score'])
        ctx.push_records(wl['batch'])
        table, _, _ = ctx.build_graph()
        rows, swap, len1, len2 = pick(table)
        res = ctx.score_edges(rows, swap, len1, len2, lib['mean'], lib['sd'], lib['read_len'])
        gap, sd0, ks, flags = res[0], res[1], res[2], res[3]
        want = {int(table.key[row]): (float(gap[i]), float(sd0[i]), int(ks[i]), int(flags[i]))
                for i, row in enumerate(rows)}
    assert len(want) > 50 and got == want


@pytest.mark.parametrize('flags', [dict(detect_duplicate=False), dict(extend_paths=False), dict(no_score=True),
                                   dict(detect_duplicate=False, extend_paths=False)])
@pytest.mark.parametrize('heads', ['gather', 'exchange'])
def test_simulated_ranks_library_flags(flags, heads):
    """The slice-boundary carry with the other CreateEdge regimes: duplicates kept (-d off), no G' (second CreateEdge
    call absent), no_score (G' only)."""
    import torch
    from besst_amd import distributed, workload
    wl = workload.make('C2', 0, pairs=120000, nc=500)
    wl['lib'] = dict(wl['lib'], **flags)
    dev = torch.device('cuda', 0)
    world = 3
    parts = DU.split_batch(wl['batch'], world)
    backends = []
    for r in range(world):
        sub = dict(wl)
        sub['batch'] = parts[r]
        backends.append(distributed.HipBackend(dev, sub, r, world, 8192))
    tails = []
    for b in backends:
        b.reset()
        b.classify_scan()
        tails.append(b.classify_tail().clone())
    tails = torch.cat(tails)
    sends = []
    for b in backends:
        if heads == 'exchange':
            b.classify_emit_speculative()
        else:
            b.classify_emit(tails)
        sends.append(b.partition().clone())
    region = backends[0].region
    for r, b in enumerate(backends):
        b.unpack(torch.cat([sends[s][r * region:(r + 1) * region] for s in range(world)]))
        b.reduce()
    torch.cuda.synchronize()
    want_rows, want = DU.expected_rows(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
    assert backends[0].sums_ride_exchange            # 500 contigs: the counters were summed by the unpack stage
    for b in backends:
        assert b.counter_words.cpu().tolist() == [want.count, want.non_unique, want.non_unique_for_scaf,
                                                  want.nr_of_duplicates, want.too_long, want.fishy_reads,
                                                  len(want.tuples), want.n_reach]
    merged = {}
    for b in backends:
        rows = DU.rows_from_table(b.local_table())
        for k in rows:
            if k & 1:
                rows[k]['lo'] = [0] * rows[k]['n']
                rows[k]['hi'] = [0] * rows[k]['n']
        merged.update(rows)
    for k, r in want_rows.items():
        if k & 1:
            r['s'] = r['s2'] = 0
    assert merged == want_rows


@pytest.mark.parametrize('heads', ['gather', 'exchange'])
def test_large_slices_take_the_sort_tile_partition(heads):
    """Slices of more than 4 M records: the owner partition keeps the sort's 4096-tuple tiles and the row-scanned
    table (smaller slices use 1024-tuple tiles).  Two simulated ranks, union of the owned rows against the C oracle
    on the whole stream."""
    import numpy as np
    import torch
    from besst_amd import distributed, workload
    from oracle import c_oracle as CO
    world = 2
    wl = workload.make('C3', 0, pairs=4_400_000, nc=3000)
    dev = torch.device('cuda', 0)
    parts = DU.split_batch(wl['batch'], world)
    assert min(len(p) for p in parts) > 4096 * 1024
    backends = []
    for r in range(world):
        sub = dict(wl)
        sub['batch'] = parts[r]
        backends.append(distributed.HipBackend(dev, sub, r, world, 1 << 20))
    tails = []
    for b in backends:
        b.reset()
        b.classify_scan()
        tails.append(b.classify_tail().clone())
    tails = torch.cat(tails)
    sends = []
    for b in backends:
        if heads == 'exchange':
            b.classify_emit_speculative()
        else:
            b.classify_emit(tails)
        sends.append(b.partition().clone())
    region = backends[0].region
    for r, b in enumerate(backends):
        b.unpack(torch.cat([sends[s][r * region:(r + 1) * region] for s in range(world)]))
        b.reduce()
    torch.cuda.synchronize()
    assert not any(b.overflowed() for b in backends)
    keys, payload, aligned, c_ctr = CO.record_loop(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
    rows = CO.edge_rows(keys, payload)
    assert len(keys) > 500_000
    tables = [b.local_table() for b in backends]
    key = np.concatenate([t.key for t in tables])
    order = np.argsort(key, kind='stable')
    assert np.array_equal(key[order], rows['key'])                       # disjoint owners, complete union
    for name, want in (('n', rows['n']), ('first_idx', rows['first_idx'])):
        got = np.concatenate([getattr(t, name) for t in tables]).astype(np.int64)[order]
        assert np.array_equal(got, want), name
    link = (rows['key'] & np.uint64(1)) == 0
    got = np.concatenate([t.sum_obs for t in tables])[order]
    assert np.array_equal(got[link], rows['sum_obs'][link])
    # coverage numerators and counters: all-reduce path here (3000 contigs fit the rider, so check whichever ran)
    total = backends[0].aligned.cpu().numpy() if backends[0].sums_ride_exchange else \
        sum(b.aligned.cpu().numpy() for b in backends)
    assert total.tolist() == aligned.tolist()
    if backends[0].sums_ride_exchange:
        assert backends[1].counter_words.cpu().numpy().tolist() == c_ctr[:8].tolist()


@pytest.mark.parametrize('which,world,flags', [(0, 2, {}), (1, 2, {}), (2, 3, {}), (5, 2, {}),
                                               (3, 2, dict(detect_duplicate=False)), (4, 3, dict(extend_paths=False)),
                                               (6, 2, dict(no_score=True))])
def test_slice_boundary_right_before_a_duplicate(which, world, flags):
    """The `which`-th duplicate record of the stream becomes the FIRST record of a slice: its slice emits the record
    provisionally, the owner of its key has to drop the tuple again and every rank has to correct the counters and
    the global emit indexes behind it.  Both the gathered-tails path and the header path must agree with the
    single-process oracle."""
    import numpy as np
    import torch
    from besst_amd import distributed, workload
    from oracle import c_oracle as CO
    wl = workload.make('C2', 0, pairs=100000, nc=500)
    wl['lib'] = dict(wl['lib'], **flags)
    batch = wl['batch']
    n = len(batch)

    def dups_in_prefix(k):
        return int(CO.record_loop(batch.take(slice(0, k)), wl['table'], wl['lib'], wl['node_bits'])[3][3])
    total = dups_in_prefix(n)
    assert total > which
    lo, hi = 0, n                                            # smallest prefix holding which + 1 duplicates
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if dups_in_prefix(mid) >= which + 1:
            hi = mid
        else:
            lo = mid
    cut = hi - 1                                             # record `cut` is the duplicate
    bounds = [0, cut, n] if world == 2 else [0, cut // 2, cut, n]
    parts = [batch.take(slice(bounds[r], bounds[r + 1])) for r in range(world)]
    want_rows, want = DU.expected_rows(batch, wl['table'], wl['lib'], wl['node_bits'])
    dev = torch.device('cuda', 0)
    for heads in ('gather', 'exchange'):
        backends = []
        for r in range(world):
            sub = dict(wl)
            sub['batch'] = parts[r]
            backends.append(distributed.HipBackend(dev, sub, r, world, 16384))
        tails = []
        for b in backends:
            b.reset()
            b.classify_scan()
            tails.append(b.classify_tail().clone())
        tails = torch.cat(tails)
        sends = []
        for b in backends:
            if heads == 'exchange':
                b.classify_emit_speculative()
            else:
                b.classify_emit(tails)
            sends.append(b.partition().clone())
        if heads == 'exchange':
            info = backends[-1].slice_info.tolist()
            assert info[3] == 1 and info[7] >= 0            # the last slice's head was emitted provisionally
        region = backends[0].region
        for r, b in enumerate(backends):
            b.unpack(torch.cat([sends[s][r * region:(r + 1) * region] for s in range(world)]))
            b.reduce()
        torch.cuda.synchronize()
        assert backends[0].sums_ride_exchange
        for b in backends:
            assert b.counter_words.cpu().tolist() == [want.count, want.non_unique, want.non_unique_for_scaf,
                                                      want.nr_of_duplicates, want.too_long, want.fishy_reads,
                                                      len(want.tuples), want.n_reach], heads
        merged = {}
        for b in backends:
            rows = DU.rows_from_table(b.local_table())
            for k in rows:
                if k & 1:
                    rows[k]['lo'] = [0] * rows[k]['n']
                    rows[k]['hi'] = [0] * rows[k]['n']
            merged.update(rows)
        for k, r in want_rows.items():
            if k & 1:
                r['s'] = r['s2'] = 0
        assert merged == want_rows, heads
        assert sum(int(b.flags.cpu()[0]) for b in backends) == len(want.tuples)


@pytest.mark.parametrize('heads', ['gather', 'exchange'])
def test_empty_and_linkless_slices_in_the_middle(heads):
    """Rank 1 holds no records at all, rank 2 only records of one contig (nothing reaches CreateEdge): the chain has
    to pass through both to rank 3."""
    import numpy as np
    import torch
    from besst_amd import distributed, workload
    wl = workload.make('C2', 0, pairs=80000, nc=400)
    batch = wl['batch']
    n = len(batch)
    same = np.nonzero((batch.tid == batch.mtid) & (batch.tid == batch.tid[n // 2]))[0]
    same = same[(same > n // 3)][:300]
    assert same.shape[0] > 10
    a = int(same[0])
    # slices: [0, a) | empty | the linkless records | everything behind them
    rest = np.setdiff1d(np.arange(a, n), same)
    parts = [batch.take(slice(0, a)), batch.take(slice(0, 0)), batch.take(same), batch.take(rest)]
    world = 4
    whole = batch.take(np.concatenate([np.arange(0, a), same, rest]))
    want_rows, want = DU.expected_rows(whole, wl['table'], wl['lib'], wl['node_bits'])
    dev = torch.device('cuda', 0)
    backends = []
    for r in range(world):
        sub = dict(wl)
        sub['batch'] = parts[r]
        backends.append(distributed.HipBackend(dev, sub, r, world, 16384))
    tails = []
    for b in backends:
        b.reset()
        b.classify_scan()
        tails.append(b.classify_tail().clone())
    tails = torch.cat(tails)
    assert tails.view(world, 4)[1:3, 0].tolist() == [0, 0]
    sends = []
    for b in backends:
        if heads == 'exchange':
            b.classify_emit_speculative()
        else:
            b.classify_emit(tails)
        sends.append(b.partition().clone())
    region = backends[0].region
    for r, b in enumerate(backends):
        b.unpack(torch.cat([sends[s][r * region:(r + 1) * region] for s in range(world)]))
        b.reduce()
    torch.cuda.synchronize()
    for b in backends:
        assert b.counter_words.cpu().tolist() == [want.count, want.non_unique, want.non_unique_for_scaf,
                                                  want.nr_of_duplicates, want.too_long, want.fishy_reads,
                                                  len(want.tuples), want.n_reach]
        assert b.aligned.cpu().tolist() == want.aligned
    merged = {}
    for b in backends:
        rows = DU.rows_from_table(b.local_table())
        for k in rows:
            if k & 1:
                rows[k]['lo'] = [0] * rows[k]['n']
                rows[k]['hi'] = [0] * rows[k]['n']
        merged.update(rows)
    for k, r in want_rows.items():
        if k & 1:
            r['s'] = r['s2'] = 0
    assert merged == want_rows


@pytest.mark.parametrize('world', [3, 8])
def test_simulated_ranks_ingest_their_part_of_a_bam_file(world, tmp_path):
    """The whole multi-GPU path from the file: rank r ingests slice (r, W) of the BAM on the GPU (besst_ctx_push_bam_device_slice:
    its slice of the stream, cut at BGZF block boundaries), the sharded build runs on the context's columns where they lie
    (besst_ctx_record_pointers, no copy), and the merged edge tables equal the single-process oracle's on the whole file."""
    import torch
    from besst_amd import bamio, distributed, workload
    wl = workload.make('C3', 0, pairs=150000, nc=300)
    path = str(tmp_path / 'lib.bam')
    bamio.write_bam(path, wl['batch'], threads=3, level=1, realistic=True)
    dev = torch.device('cuda', 0)
    pair_cap = 16384
    backends, bams, total = [], [], 0
    slices, rereads = distributed.ingest_all_slices(path, world, device_index=0, threads=2)
    assert rereads == 0                                      # (htslib's layout: every slice begins with its first block's first byte)
    for r, (bam, cols) in enumerate(slices):
        assert bam.ingest.on_device == 1 and bam.boundary == (0, 0)
        bams.append(bam)
        total += len(bam)
        sub = {k: v for k, v in wl.items() if k not in ('batch', 'cols', '_rec')}
        sub['cols'] = cols
        assert int(sub['cols']['tid'].shape[0]) == len(bam)
        backends.append(distributed.HipBackend(dev, sub, r, world, pair_cap))
    assert total == len(wl['batch'])
    tails = []
    for b in backends:
        b.reset()
        b.classify_scan()
        tails.append(b.classify_tail().clone())
    sends = []
    for b in backends:
        b.classify_emit_speculative()
        sends.append(b.partition().clone())
    if not backends[0].sums_ride_exchange:
        tot = sum(b.pack_for_allreduce().clone() for b in backends)
        for b in backends:
            b.pack_for_allreduce().copy_(tot)
    region = backends[0].region
    for r, b in enumerate(backends):
        b.unpack(torch.cat([sends[s][r * region:(r + 1) * region] for s in range(world)]))
        b.reduce()
    torch.cuda.synchronize()
    assert not any(b.overflowed() for b in backends)
    want_rows, want = DU.expected_rows(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
    assert backends[0].aligned.cpu().tolist() == want.aligned
    assert backends[0].counter_words.cpu().tolist()[:8] == [want.count, want.non_unique, want.non_unique_for_scaf, want.nr_of_duplicates,
                                                            want.too_long, want.fishy_reads, len(want.tuples), want.n_reach]
    merged = {}
    for r, b in enumerate(backends):
        rows = DU.rows_from_table(b.local_table())
        for k in rows:
            if k & 1:
                rows[k]['lo'] = [0] * rows[k]['n']
                rows[k]['hi'] = [0] * rows[k]['n']
        assert not set(rows) & set(merged)
        merged.update(rows)
    for k, r in want_rows.items():
        if k & 1:
            r['s'] = r['s2'] = 0
    assert merged == want_rows
    del backends
    for bam in bams:
        bam.close()
