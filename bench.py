#!/usr/bin/env python
"""bench.py - read-pairs/s into the scaffold graph on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one resident batch: record loop (classify), duplicate
chain, ordered tuple emission, radix sort and edge-table reduction, all enqueued on one HIP stream
through the besst_dev_* C ABI with the record columns already in HBM.  No host round trip happens
inside a step.  At N > 1 every rank owns a contiguous slice of the (tid,pos)-sorted stream
(weak scaling on the C4 shape - 500 k contigs, two libraries, one eighth of each library's pairs per GPU) and every
rank checks its share against the C oracle (verify_sharded_vs_oracle); see besst_amd/distributed.py.

N = 1 (the default): the workload is BASELINE.json configs[2] (C3: 100 k contigs / 200 M mate pairs with PE
contamination) at FULL size - the largest single-GPU config - generated on the GPU; configs[1] (C2) is measured
afterwards and rides along as the "c2" object of the same line.

Prints ONE JSON line on rank 0 (see the repo prompt for the contract) with two extra objects:
  roofline     - the WHOLE step priced at SURVEY 8(d)'s (38 + 32 f) bytes per read pair over the step time and
                 the 8 TB/s peak; the record loop's kernel (the dominant one) under its own name, with its algorithmic
                 bytes and its PMC bytes over its HIP-event duration, as an extra;
                 traffic = HBM bytes per step from the committed rocprofv3 PMC passes of this very build
  cpu_baseline - the pure-Python oracle (port of the reference's record loop) timed on this box's
                 host cores over a bounded sample of the same stream, with the port's measured speed
                 relative to the real reference (oracle/cpu_port_calibration.json)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
RECORD_LOOP_SLOTS = 0b111   # stream_kernel, fused_kernel, fused_wave_kernel: besst_prof_slot_name(0..2)
RECORD_LOOP_KERNELS = ('stream_kernel', 'fused_kernel', 'fused_wave_kernel')


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--settle-seconds', type=float, default=0.3,
                    help='untimed passes before the warm-up steps (setup: the clocks of an idle GPU need work to settle)')
    ap.add_argument('--config', default=None,
                    help='BASELINE.json config of the workload; default C3 (configs[2], the largest single-GPU config) '
                         'at N = 1, a C2-sized slice per rank at N > 1')
    ap.add_argument('--also', default='C2', help='second config measured after the headline one at N = 1 ("" = none)')
    ap.add_argument('--pairs', type=int, default=None, help='override pairs per GPU (smoke runs)')
    ap.add_argument('--contigs', type=int, default=None)
    ap.add_argument('--cpu-sample-records', type=int, default=20_000_000,
                    help='records of the stream the Python port is timed on; 0 skips it (C port + check only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown-steps', type=int, default=3)
    ap.add_argument('--copies', type=int, default=None,
                    help='resident copies of the record columns cycled through by consecutive steps, so that a step '
                         'cannot be served from the 256 MiB Infinity Cache left warm by the previous one '
                         '(default: 3 for C2, 1 for C3 whose 9.2 GB of records cannot stay cached)')
    ap.add_argument('--no-verify', action='store_true', help='skip the full-size check against the C oracle')
    ap.add_argument('--no-stages', action='store_true', help='skip the separate metrics / scoring stage timings')
    ap.add_argument('--no-robustness', action='store_true',
                    help='skip the "robustness" object (C3 with chimeric pairs, name-sorted C2)')
    ap.add_argument('--slices', choices=('independent', 'contiguous'), default='contiguous',
                    help='N > 1: what a rank holds of a library - "contiguous": the rank-th (tid, pos)-contiguous cut of ONE '
                         'stream, what distributed.ingest_slice yields on a real file (a slice\'s tuples concentrate on few '
                         'owners); "independent": a whole-genome stream of its own per rank (rounds 2 and 3)')
    ap.add_argument('--from-bam', action='store_true',
                    help='--gpus N: rank 0 writes every library as ONE sequencer-like BAM (untimed), every rank ingests its '
                         'slice of the file on its GPU (distributed.ingest_slice, timed and reported per rank), and the timed '
                         'steps run on the ingested records; default 12.5 M pairs per library and GPU')
    ap.add_argument('--in-flight', type=int, default=3,
                    help='library passes kept in flight (one HIP stream each) for the extra "overlapped" figure; '
                         '0 skips it.  The headline value is always measured with ONE pass at a time.')
    return ap.parse_args()


def cpu_baseline(batch, table, lib, n_sample):
    """Time the oracle's record loop (single thread) on the first n_sample records."""
    from oracle import py_oracle as O
    n = min(len(batch), n_sample)
    rec = {k: getattr(batch, k)[:n].tolist() for k in ('tid', 'mtid', 'pos', 'mpos', 'flag', 'mapq', 'qlen')}
    tab = dict(cls=table['cls'].tolist(), scaf=table['scaf_id'].tolist(), slen=table['scaf_len'].tolist(),
               cpos=table['ctg_pos'].tolist(), clen=table['ctg_len'].tolist(),
               cdir=[bool(x) for x in table['direction'].tolist()])
    p = O.LibParams(read_len=lib['read_len'], ins_size_threshold=lib['ins_size_threshold'], min_mapq=lib['min_mapq'],
                    orientation=lib['orientation'])
    t0 = time.perf_counter()
    res = O.record_loop(rec, tab, p)
    dt = time.perf_counter() - t0
    return dict(value=(n / 2.0) / dt, unit='read-pairs/s', cores=1, kind='port',
                sample='first %d records (%d pairs) of the same stream, oracle/py_oracle.record_loop, %.1f s'
                       % (n, n // 2, dt)), res


def measured_copy_bandwidth(device):
    """Device-to-device copy of 1 GiB (report the fraction of a MEASURED ceiling too): GB/s moved,
    read + write."""
    import torch
    a = torch.empty(1 << 30, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * (1 << 30) / dt / 1e9


C_PORT_TIMING = {}


_H2D = {}


def measured_h2d_bandwidth(device):
    """GB/s of a pinned host buffer -> HBM copy on this box (1 GiB, best of three): what the COMPRESSED file crosses PCIe at
    best - the peak the ingest's `ingest_roofline` is priced against, measured in the same run."""
    import torch
    key = str(device)
    if key not in _H2D:
        n = 1 << 30
        host = torch.empty(n, dtype=torch.uint8).pin_memory()
        dev = torch.empty(n, dtype=torch.uint8, device=device)
        best = 0.0
        for _ in range(4):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            dev.copy_(host, non_blocking=True)
            torch.cuda.synchronize(device)
            best = max(best, n / (time.perf_counter() - t0) / 1e9)
        del host, dev
        _H2D[key] = best
    return _H2D[key]


def ingest_roofline(device, compressed_bytes, seconds):
    """The device ingest moves the compressed file across PCIe once: compressed GB/s over the box's measured H2D rate."""
    peak = measured_h2d_bandwidth(device)
    gbps = compressed_bytes / seconds / 1e9
    return {'bound': 'pcie-h2d', 'achieved': round(gbps, 2), 'peak': round(peak, 1), 'unit': 'GB/s', 'frac': round(gbps / peak, 4),
            'what': 'compressed bytes of the file / ingest_s over a pinned host -> HBM copy of 1 GiB measured in this run'}


def verify_full(runner, wl):
    """Second half of the cpu_baseline leg (rank 0, N=1): the C restatement is timed over the WHOLE workload, on one
    thread and on all host cores, and its edge table doubles as the full-size parity check of the device's."""
    import numpy as np
    from oracle import c_oracle as CO
    table = runner.gb.fetch_table()
    ctr = runner.gb.read_counters()
    t0 = time.perf_counter()
    keys, payload, aligned, c_ctr = CO.record_loop(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
    C_PORT_TIMING['seconds'] = time.perf_counter() - t0
    C_PORT_TIMING['records'] = len(wl['batch'])
    from besst_amd._lib import effective_cpus
    cores = effective_cpus()                             # (affinity and cgroup quota: not the machine's CPU count)
    best = None
    for _ in range(2):                                   # the first call pays thread start-up and page faults
        t0 = time.perf_counter()
        mt = CO.record_loop(wl['batch'], wl['table'], wl['lib'], wl['node_bits'], threads=cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    C_PORT_TIMING['mt_seconds'] = best
    C_PORT_TIMING['mt_cores'] = cores
    C_PORT_TIMING['mt_equal'] = bool(np.array_equal(mt[0], keys) and np.array_equal(mt[3], c_ctr))
    rows = CO.edge_rows(keys, payload)
    link = ~table.is_fishy
    ok = (np.array_equal(table.key, rows['key']) and np.array_equal(table.n.astype(np.int64), rows['n'])
          and np.array_equal(table.sum_obs[link], rows['sum_obs'][link])
          and np.array_equal(table.sum_obs_sq[link], rows['sum_obs_sq'][link])
          and np.array_equal(table.first_idx.astype(np.int64), rows['first_idx'])
          and np.array_equal(table.obs_lo.astype(np.int64), rows['obs_lo'])
          and np.array_equal(table.obs_hi.astype(np.int64), rows['obs_hi'])
          and runner.gb.aligned.cpu().numpy().tolist() == aligned.tolist()
          and [ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates,
               ctr.reads_with_too_long_insert, ctr.fishy_reads, ctr.n_tuples, ctr.n_reach, ctr.prev_obs1,
               ctr.prev_obs2] == c_ctr.tolist())
    return bool(ok)


def stage_timings(wl):
    """Wall time of the other stages through the host-buffer C ABI (reported separately)."""
    import numpy as np
    from besst_amd import _lib, device, pipeline
    batch, lib, asm = wl['batch'], wl['lib'], wl['asm']
    out = {}
    lib_h = _lib.load()
    with device.GraphContext(0) as ctx:
        ctx.set_contigs(**wl['table'])
        ctx.set_library(lib['read_len'], lib['ins_size_threshold'], lib['min_mapq'], lib['orientation'],
                        lib['detect_duplicate'], lib['extend_paths'], lib['no_score'])
        t0 = time.perf_counter()
        ctx.push_records(batch)
        out['h2d_push_ms'] = (time.perf_counter() - t0) * 1e3
        top = np.zeros(asm.nc, np.uint8)
        top[np.lexsort((np.arange(asm.nc), -asm.lengths))[:1000]] = 1
        ctx.metrics_sample(top, lib['orientation'], lib['min_mapq'], lib['read_len'], True)
        lib_h.besst_prof_enable(0xffffffff)
        t0 = time.perf_counter()
        _, _, counts = ctx.metrics_sample(top, lib['orientation'], lib['min_mapq'], lib['read_len'], True)
        out['metrics_scan_ms'] = (time.perf_counter() - t0) * 1e3
        out['metrics_kernels_ms'] = pipeline.prof_collect().get('metrics_kernels', (0.0, 0))[0]
        out['metrics_records_scanned'] = int(counts.records_scanned)

        def gated_share(mask, records):
            """Share of the pass's waves that read all five columns: a wave reads the reference ids of its 4 x 256 records
            (sub-tile u of a 4096-record tile: records u * 1024 + 256 w + [0, 256) for wave w) and the other four columns only
            if one of them lies on a top-1000 contig (csrc/metrics.hip: tile_flags / eval_tile)."""
            import torch
            n = int(records) // 4096 * 4096
            if n == 0:
                return 1.0
            tid = wl['cols']['tid'][:n] if 'cols' in wl else torch.from_numpy(np.ascontiguousarray(batch.tid[:n])).cuda()
            top = torch.from_numpy(mask.astype(np.bool_)).to(tid.device)
            hit = top[tid.long().clamp_(0, asm.nc - 1)] & (tid >= 0) & (tid < asm.nc)
            per_wave = hit.view(-1, 4, 4, 256).any(dim=3).any(dim=1)           # (tile, sub, wave, record) -> (tile, wave)
            return float(per_wave.float().mean().item())

        def roofline(records, ms, mask):
            # SURVEY 8(d) prices the library-metrics pass at 22 B/pair (15 B/record: tid mtid tlen flag mapq, read once).  The
            # kernel needs less: every record's reference id (4 B) and the other 11 B only for the waves that hold a record
            # on a top-1000 contig.  `frac` is on those NEEDED bytes (never above 1); the priced figure rides along, labelled.
            share = gated_share(mask, records)
            needed = records * (4.0 + 11.0 * share)
            gbps = needed / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            priced = records / 2 * 22 / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            moved = None
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r06_metrics_pmc.json')) as fh:
                    moved = json.load(fh).get('moved_bytes_per_record')
            except (OSError, ValueError):
                pass
            return {'records_scanned': int(records), 'kernel_ms': round(ms, 4), 'bound': 'hbm',
                    'model': 'needed bytes = 4 B/record (reference id) + 11 B/record for the share of waves that hold a record '
                             'on one of the 1000 longest contigs (mtid tlen flag mapq)',
                    'gated_wave_share': round(share, 4), 'needed_bytes_per_record': round(4.0 + 11.0 * share, 3),
                    'achieved_GBps': round(gbps, 1), 'peak_GBps': 8000.0, 'frac': round(gbps / 8000.0, 4),
                    'moved_bytes_per_record_pmc': moved,
                    'frac_on_moved_bytes': None if moved is None or ms <= 0 else round(records * moved / (ms * 1e-3) / 1e9 / 8000.0, 4),
                    'priced_22B_per_pair': {'achieved_GBps': round(priced, 1), 'of_peak': round(priced / 8000.0, 4),
                                            'note': 'SURVEY 8(d)\'s price; above what the kernel reads, so this ratio can pass 1 '
                                                    '- it is not a roofline fraction'}}
        out['metrics_roofline'] = roofline(counts.records_scanned, out['metrics_kernels_ms'], top)
        # the same pass forced over the whole library: a top-1000 mask of three short contigs never fills the samples
        # (libmetrics.py:293-303 then scans to the end of the file)
        few = np.zeros(asm.nc, np.uint8)
        few[np.argsort(asm.lengths, kind='stable')[:3]] = 1
        ctx.metrics_sample(few, lib['orientation'], lib['min_mapq'], lib['read_len'], True)
        pipeline.prof_collect()
        _, _, counts_all = ctx.metrics_sample(few, lib['orientation'], lib['min_mapq'], lib['read_len'], True)
        out['metrics_roofline_full_scan'] = roofline(counts_all.records_scanned,
                                                     pipeline.prof_collect().get('metrics_kernels', (0.0, 0))[0], few)
        ctx.build_graph()
        t0 = time.perf_counter()
        table, _, _ = ctx.build_graph()
        out['ctx_build_graph_ms'] = (time.perf_counter() - t0) * 1e3
        rows = np.nonzero((~table.is_fishy) & ((table.mask & 1) != 0) & (table.n >= 5))[0].astype(np.uint32)
        if rows.shape[0]:
            len1 = (asm.lengths[(table.u[rows] >> 1) - 1]).astype(np.int32)
            len2 = (asm.lengths[(table.v[rows] >> 1) - 1]).astype(np.int32)
            swap = np.zeros(rows.shape[0], np.uint8)
            ctx.score_edges(rows, swap, len1, len2, lib['mean'], lib['sd'], lib['read_len'])
            pipeline.prof_collect()
            t0 = time.perf_counter()
            ctx.score_edges(rows, swap, len1, len2, lib['mean'], lib['sd'], lib['read_len'])
            dt = time.perf_counter() - t0
            out['score_kernel_ms'] = pipeline.prof_collect().get('score_kernels', (0.0, 0))[0]
            out['score_ms'] = dt * 1e3
            out['scored_edges'] = int(rows.shape[0])
            out['score_edges_per_s'] = rows.shape[0] / dt
            # the same edges through the log-normal branch (param.lognormal: a skewed library, CreateGraph.py:522-531):
            # the library read as LogNormal(ln mean, sd / mean ... 0.25) - the gap is an argmax over ~400 gaps x n links
            import math
            from besst_amd import mathstats_compat as MC
            ln_mu, ln_sigma = math.log(lib['mean']), 0.25
            ln = (ln_mu, ln_sigma, MC.lognormal_support(ln_mu, ln_sigma), int(0.8 * lib['ins_size_threshold']))
            ctx.score_edges(rows, swap, len1, len2, lib['mean'], lib['sd'], lib['read_len'], lognormal=ln)
            pipeline.prof_collect()
            t0 = time.perf_counter()
            gap_ln = ctx.score_edges(rows, swap, len1, len2, lib['mean'], lib['sd'], lib['read_len'], lognormal=ln)[0]
            dt = time.perf_counter() - t0
            out['score_lognormal_kernel_ms'] = pipeline.prof_collect().get('score_kernels', (0.0, 0))[0]
            out['score_lognormal_ms'] = dt * 1e3
            out['score_lognormal_links'] = int(table.n[rows].sum())
            # a sample of the edges against the host restatement (not timed; ~1 ms per edge)
            take = rows[:: max(1, rows.shape[0] // 300)][:300]
            lo, hi = table.obs_lo, table.obs_hi
            bad = 0
            for k, rr in zip(range(0, rows.shape[0], max(1, rows.shape[0] // 300)), take.tolist()):
                if not (2 * lib['sd'] < len1[k] and 2 * lib['sd'] < len2[k]):
                    continue
                a, b = int(table.offset[rr]), int(table.offset[rr]) + int(table.n[rr])
                want = min(MC.lognormal_GapEstimator(ln_mu, ln_sigma, lib['read_len'], (lo[a:b] + hi[a:b]).tolist(),
                                                     int(len1[k]), c2_len=int(len2[k])), ln[3])
                bad += int(gap_ln[k]) != want
            out['score_lognormal_sample_mismatches'] = bad
        lib_h.besst_prof_enable(0)
        out.update(linearize_timing())
        out.update(chain_timing())
        out.update(scorepaths_timing())
        pairs = len(batch) // 2
        out['pcie_inclusive_pairs_per_s'] = pairs / ((out['h2d_push_ms'] + out['ctx_build_graph_ms']) * 1e-3)
    return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}


def dropin_timing(device, config, pairs=None, contigs=None):
    """Wall time of the drop-in as BESST calls it (runBESST:168,182): libmetrics.get_metrics + CreateGraph.PE on a host
    RecordBatch (what besst_amd.bamio.read_bam returns), everything included - upload, device passes, download of the
    edge table, and the Python side that fills Contig/Scaffold objects and the networkx-compatible graphs.  `library_s`
    is the share spent inside the C ABI (transfers + kernels), the rest is interpreter time."""
    import io
    import tempfile
    from besst_amd import CreateGraph, Parameter, device as dev_mod, libmetrics, session, workload
    wl = workload.make_device(device, config, 0, pairs=pairs, nc=contigs)
    batch = wl['batch']
    del wl['cols']
    p = Parameter.parameter()
    p.scaffold_indexer = 1; p.min_mapq = 11; p.lower_cov_cutoff = 0.001; p.cov_cutoff = None; p.first_lib = True
    p.orientation = wl['lib']['orientation']; p.detect_duplicate = True; p.extend_paths = True; p.no_score = False
    p.detect_haplotype = False; p.print_scores = False; p.max_contig_overlap = 200; p.pass_number = 1
    p.information_file = io.StringIO(); p.output_directory = tempfile.mkdtemp(prefix='besst_amd_')
    p.contig_index = dict(enumerate(batch.references))

    C_dict = {name: _SeqLen(int(n)) for name, n in zip(batch.references, batch.lengths)}
    dev_mod.CALL_SECONDS = {}
    CreateGraph.STAGE_SECONDS = {}
    t0 = time.perf_counter()
    libmetrics.get_metrics(batch, p, p.information_file)
    t1 = time.perf_counter()
    G, Gp = CreateGraph.PE({}, {}, p.information_file, C_dict, p, {}, {}, batch)
    t2 = time.perf_counter()
    lib_s = dict(dev_mod.CALL_SECONDS)
    dev_mod.CALL_SECONDS = None
    pe_stages = {k: round(v, 3) for k, v in CreateGraph.STAGE_SECONDS.items()}
    CreateGraph.STAGE_SECONDS = None
    session.close_session(batch)
    total = t2 - t0
    # the graphs are backed by columns: a node's containers are made when it is first read.  What PE no longer pays it has
    # only deferred for a consumer that walks EVERYTHING - said here: the first walk over every edge of both graphs (all
    # neighbour dictionaries and edge attribute dictionaries made, in bulk), outside total_s.
    tw = time.perf_counter()
    edges_g, edges_gp = len(G.edges()), len(Gp.edges())
    walk_s = time.perf_counter() - tw
    return {'records': len(batch), 'get_metrics_s': round(t1 - t0, 3), 'PE_s': round(t2 - t1, 3), 'total_s': round(total, 3),
            'library_s': round(sum(lib_s.values()), 3), 'library_calls_s': {k: round(v, 3) for k, v in lib_s.items()},
            'PE_stages_s': pe_stages, 'host_share': round(1.0 - sum(lib_s.values()) / total, 3), 'edges_G': G.number_of_edges(),
            'edges_G_prime': Gp.number_of_edges(), 'pairs_per_s': (len(batch) // 2) / total,
            'first_full_walk_of_both_graphs_s': round(walk_s, 3), 'edges_walked': edges_g + edges_gp}


def bam_to_graph_timing(device, config, pairs=None, realistic=False):
    """BAM bytes -> scored graphs, everything included (SURVEY 8(f) rank 1 + the path): the config's stream is written as
    a BAM file (native writer, untimed scaffolding; /dev/shm when there is one), then timed: bamio.ResidentBam - the
    compressed file uploaded chunk by chunk, BGZF inflate + record decode on the GPU (csrc/bgzf_gpu.hip) - followed by
    libmetrics.get_metrics and CreateGraph.PE on the resident records.  The host form of the ingest (reader threads +
    pinned staging) is timed on the same file beside it, and the records the two leave in HBM are compared.
    realistic: bases and qualities that compress like a sequencer's (~70 B/record on disk, 14 000 DEFLATE symbols per
    block) instead of constant bytes (~16 B/record, long matches) - the file a user brings, and the one the headline
    config's stage is measured on since round 4."""
    import io
    import shutil
    import tempfile
    from besst_amd import CreateGraph, Parameter, _lib, bamio, libmetrics, session, workload
    wl = workload.make_device(device, config, 0, pairs=pairs)
    batch = wl['batch']
    del wl['cols']
    base = '/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > 64 * len(batch) else None
    tmp = tempfile.mkdtemp(prefix='besst_amd_bam_', dir=base)
    path = os.path.join(tmp, 'lib.bam')
    cores = _lib.effective_cpus()
    try:
        t0 = time.perf_counter()
        bamio.write_bam(path, batch, level=1, realistic=realistic)
        write_s = time.perf_counter() - t0
        size = os.path.getsize(path)
        p = Parameter.parameter()
        p.scaffold_indexer = 1; p.min_mapq = 11; p.lower_cov_cutoff = 0.001; p.cov_cutoff = None; p.first_lib = True
        p.orientation = wl['lib']['orientation']; p.detect_duplicate = True; p.extend_paths = True; p.no_score = False
        p.detect_haplotype = False; p.print_scores = False; p.max_contig_overlap = 200; p.pass_number = 1
        p.information_file = io.StringIO(); p.output_directory = tmp
        p.contig_index = dict(enumerate(batch.references))
        C_dict = {name: _SeqLen(int(n)) for name, n in zip(batch.references, batch.lengths)}
        n_rec = len(batch)
        del batch, wl
        threads = bamio.reader_threads()
        t0 = time.perf_counter()
        bam = bamio.ResidentBam(path, threads=threads, chunk_records=4 << 20)
        t1 = time.perf_counter()
        libmetrics.get_metrics(bam, p, p.information_file)
        t2 = time.perf_counter()
        G, Gp = CreateGraph.PE({}, {}, p.information_file, C_dict, p, {}, {}, bam)
        t3 = time.perf_counter()
        st = bam.ingest
        out = {'records': n_rec, 'pairs_of_the_config': 'all' if pairs is None else '%d (a slice)' % pairs, 'bam_bytes': size,
               'file': ('sequencer-like: pseudo-random bases, slowly changing qualities' if realistic else
                        'constant bases and qualities') + ', %.1f B/record compressed' % (size / n_rec),
               'reader_threads': threads, 'usable_cpus': cores, 'machine_cpus': os.cpu_count(),
               'write_bam_s_untimed': round(write_s, 2),
               'ingest_form': ('device: BGZF inflate + record decode on the GPU (besst_ctx_push_bam_device; inflate kernel: %s form)'
                               % ('first' if os.environ.get('BESST_INFLATE') == '1' else 'second')) if st.on_device else
                              'host: reader threads + pinned staging (besst_ctx_push_bam)',
               'ingest_s': round(t1 - t0, 3), 'ingest_records_per_s': n_rec / (t1 - t0),
               'ingest_compressed_GBps': round(size / (t1 - t0) / 1e9, 2),
               'ingest_staging_s' if st.on_device else 'ingest_decode_s': round(st.decode_seconds, 3),
               'ingest_wait_s': round(st.copy_wait_seconds, 3), 'ingest_chunks': int(st.chunks), 'h2d_bytes': int(st.bytes_h2d),
               'inflated_bytes': int(st.inflated_bytes), 'bgzf_blocks': int(st.blocks),
               'get_metrics_s': round(t2 - t1, 3), 'PE_s': round(t3 - t2, 3), 'total_s': round(t3 - t0, 3),
               'pairs_per_s': (n_rec // 2) / (t3 - t0), 'edges_G': G.number_of_edges(), 'edges_G_prime': Gp.number_of_edges()}
        if st.on_device:
            out['ingest_roofline'] = ingest_roofline(device, size, t1 - t0)
            # the same ingest once more: the first read of a file that has just been written is the slower one on the host
            # side (staging reads 20-25 GB/s, 40-50 from the second read on), and then the inflate kernel is the bound
            t0 = time.perf_counter()
            again = bamio.ResidentBam(path, threads=threads, chunk_records=4 << 20)
            dt = time.perf_counter() - t0
            out['ingest_repeat'] = {'ingest_s': round(dt, 3), 'ingest_records_per_s': n_rec / dt,
                                    'ingest_staging_s': round(again.ingest.decode_seconds, 3),
                                    'ingest_wait_s': round(again.ingest.copy_wait_seconds, 3),
                                    'ingest_roofline': ingest_roofline(device, size, dt)}
            again.close()
        # the other ingest form on the same file (ingest only), and whether the two leave the same records in HBM
        t0 = time.perf_counter()
        other = bamio.ResidentBam(path, threads=threads, chunk_records=4 << 20, mode='host' if st.on_device else 'device')
        t1 = time.perf_counter()
        agree = len(other) == len(bam)
        step = 16 << 20
        for lo in range(0, len(bam) if agree else 0, step):
            a, b = bam.ctx.fetch_records(lo, min(step, len(bam) - lo)), other.ctx.fetch_records(lo, min(step, len(bam) - lo))
            agree = agree and all(np.array_equal(a[k], b[k]) for k in a)
        out['ingest_other_form'] = {'form': 'device' if other.ingest.on_device else 'host', 'ingest_s': round(t1 - t0, 3),
                                    'ingest_records_per_s': n_rec / (t1 - t0), 'h2d_bytes': int(other.ingest.bytes_h2d)}
        out['ingest_forms_agree'] = bool(agree)
        other.close()
        session.close_session(bam)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bam_ingest_timing(device, config, pairs):
    """Ingest only, both forms, on a file whose bases and qualities compress like a sequencer's (~3.3 x: pseudo-random
    bases, slowly changing qualities - five DEFLATE symbols of six are literals) instead of the constant bytes the other
    BAM stages write (~13 x, long matches): the inflate kernel's other regime."""
    import shutil
    import tempfile
    from besst_amd import _lib, bamio, workload
    wl = workload.make_device(device, config, 0, pairs=pairs)
    batch = wl['batch']
    del wl['cols']
    base = '/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > 128 * len(batch) else None
    tmp = tempfile.mkdtemp(prefix='besst_amd_bam_', dir=base)
    path = os.path.join(tmp, 'lib.bam')
    try:
        t0 = time.perf_counter()
        bamio.write_bam(path, batch, level=1, realistic=True)
        write_s = time.perf_counter() - t0
        n_rec, size = len(batch), os.path.getsize(path)
        del batch, wl
        out = {'records': n_rec, 'bam_bytes': size, 'bytes_per_record_compressed': round(size / n_rec, 1),
               'write_bam_s_untimed': round(write_s, 2), 'usable_cpus': _lib.effective_cpus()}
        held = []
        for mode in ('device', 'host'):
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                bam = bamio.ResidentBam(path, mode=mode)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    if best is not None:
                        best[1].close()
                    best = (dt, bam)
                else:
                    bam.close()
            dt, bam = best
            out[mode] = {'ingest_s': round(dt, 3), 'records_per_s': n_rec / dt, 'compressed_GBps': round(size / dt / 1e9, 2),
                         'on_device': int(bam.ingest.on_device), 'h2d_bytes': int(bam.ingest.bytes_h2d),
                         'inflated_bytes': int(bam.ingest.inflated_bytes)}
            if bam.ingest.on_device:
                out[mode]['ingest_roofline'] = ingest_roofline(device, size, dt)
            held.append(bam)
        agree = len(held[0]) == len(held[1])
        step = 16 << 20
        for lo in range(0, n_rec if agree else 0, step):
            a, b = held[0].ctx.fetch_records(lo, min(step, n_rec - lo)), held[1].ctx.fetch_records(lo, min(step, n_rec - lo))
            agree = agree and all(np.array_equal(a[k], b[k]) for k in a)
        out['forms_agree'] = bool(agree)
        for bam in held:
            bam.close()
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class _SeqLen(object):                                   # PE only takes len() of a contig's sequence and stores it
    __slots__ = ('n',)

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, k):                            # a contig filtered for low coverage is written out as FASTA
        return 'N' * len(range(*k.indices(self.n))) if isinstance(k, slice) else 'N'


def linearize_workload(n_scaf, n_edges, seed=20240929):
    """Seeded scored scaffold graph for the linearisation stage: random link edges, half of
    the scores from a small pool (ties, zeros, 0.8 ratios), half uniform in [0, 2]."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2 * n_scaf, n_edges).astype(np.int32)
    b = rng.integers(0, 2 * n_scaf, n_edges).astype(np.int32)
    keep = (a >> 1) != (b >> 1)
    a, b = a[keep], b[keep]
    lo, hi = np.minimum(a, b).astype(np.int64), np.maximum(a, b).astype(np.int64)
    _, first = np.unique(lo * (2 * n_scaf) + hi, return_index=True)
    first.sort()
    a, b = a[first], b[first]
    pool = np.array([0.0, 0.5, 0.8, 1.0, 1.0, 1.25, 2.0])
    score = np.where(rng.random(a.shape[0]) < 0.5, rng.choice(pool, a.shape[0]), np.round(rng.random(a.shape[0]) * 2, 2))
    return a, b, score.astype(np.float64)


def _host_memory_gib():
    try:
        with open('/proc/meminfo') as fh:
            for line in fh:
                if line.startswith('MemAvailable:'):
                    return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def linearize_timing(n_scaf=2_000_000, n_edges=3_000_000):
    """MakeScaffolds steps 1-4 on a C5-sized scored edge table: wall time through the host-pointer call (copies
    included) and with the table resident in HBM."""
    import torch
    from besst_amd import MakeScaffolds as MS, _lib
    a, b, score = linearize_workload(n_scaf, n_edges)
    m = int(a.shape[0])
    MS.linearize_arrays(n_scaf, a, b, score)
    t0 = time.perf_counter()
    res = MS.linearize_arrays(n_scaf, a, b, score)
    host_ms = (time.perf_counter() - t0) * 1e3
    lib_h = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    da, db, ds = (torch.from_numpy(x).to(dev) for x in (a, b, score))
    ws = torch.empty(lib_h.besst_dev_linearize_workspace_bytes(n_scaf, m), dtype=torch.uint8, device=dev)
    alive = torch.empty(m, dtype=torch.uint8, device=dev)
    removed = torch.empty(n_scaf, dtype=torch.uint8, device=dev)
    amb = torch.empty(2 * n_scaf, dtype=torch.uint8, device=dev)
    top = torch.empty(2 * n_scaf, dtype=torch.float64, device=dev)
    sec = torch.empty(2 * n_scaf, dtype=torch.float64, device=dev)
    best = torch.empty(2 * n_scaf, dtype=torch.int32, device=dev)
    counters = np.zeros(8, np.int64)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run():
        _lib.check(lib_h.besst_dev_linearize(stream, 15, n_scaf, m, da.data_ptr(), db.data_ptr(), ds.data_ptr(),
                                             ws.data_ptr(), ws.numel(), alive.data_ptr(), removed.data_ptr(),
                                             amb.data_ptr(), top.data_ptr(), sec.data_ptr(), best.data_ptr(),
                                             _lib.ptr(counters)), 'besst_dev_linearize')
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    dev_ms = (time.perf_counter() - t0) * 1e3
    same = bool(np.array_equal(alive.cpu().numpy().astype(bool), res['alive2'])
                and np.array_equal(removed.cpu().numpy(), res['removed_by']))
    return {'linearize_scaffolds': n_scaf, 'linearize_edges': m, 'linearize_rounds': int(counters[4]),
            'linearize_host_call_ms': host_ms, 'linearize_resident_ms': dev_ms,
            'linearize_edges_per_s': m / (dev_ms * 1e-3), 'linearize_calls_agree': same}


def chain_timing(n_scaf=2_000_000, seed=7):
    """Chain extraction (NewContigsScaffolds' list ranking) on a C5-sized linearised graph: paths of 1..40 scaffolds,
    wall time through the host-pointer call (copies and the per-pass synchronisation included)."""
    from besst_amd import MakeScaffolds as MS
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n_scaf)
    runs = rng.integers(1, 41, n_scaf)
    ends = np.cumsum(runs)
    k = int(np.searchsorted(ends, n_scaf)) + 1
    first = np.concatenate(([0], ends[:k - 1]))
    is_first = np.zeros(n_scaf, bool)
    is_first[first[first < n_scaf]] = True
    sides = rng.integers(0, 2, n_scaf)
    a = np.arange(n_scaf - 1)
    joined = ~is_first[a + 1]                               # perm[a] -- perm[a + 1] are neighbours on a path
    u = 2 * perm[a[joined]] + (1 - sides[a[joined]])
    v = 2 * perm[a[joined] + 1] + sides[a[joined] + 1]
    link = np.full(2 * n_scaf, -1, np.int32)
    link[u], link[v] = v, u
    gap = np.zeros(2 * n_scaf, np.int32)
    gap[u] = gap[v] = rng.integers(1, 500, u.shape[0])
    slen = rng.integers(200, 20000, n_scaf)
    order = rng.permutation(2 * n_scaf).astype(np.int32)
    MS.chain_arrays(n_scaf, link, gap, slen, order)
    t0 = time.perf_counter()
    _, _, _, passes = MS.chain_arrays(n_scaf, link, gap, slen, order)
    ms = (time.perf_counter() - t0) * 1e3
    return {'chain_scaffolds': n_scaf, 'chain_paths': int(is_first.sum()), 'chain_passes': passes,
            'chain_host_call_ms': ms, 'chain_scaffolds_per_s': n_scaf / (ms * 1e-3)}


def scorepaths_workload(n_scaf=50_000, n_links=150_000, n_paths=200_000, seed=11):
    """Seeded link graph (CSR) and a batch of alternating random walks over it for the ScorePaths stage."""
    rng = np.random.default_rng(seed)
    n_nodes = 2 * n_scaf
    a = rng.integers(0, n_nodes, n_links)
    b = rng.integers(0, n_nodes, n_links)
    keep = (a >> 1) != (b >> 1)
    a, b = a[keep], b[keep]
    w = rng.integers(1, 50, a.shape[0])
    src = np.concatenate([a, b])
    order = np.argsort(src, kind='stable')
    col = np.ascontiguousarray(np.concatenate([b, a]).astype(np.int32)[order])
    weight = np.ascontiguousarray(np.concatenate([w, w]).astype(np.int32)[order])
    row_ptr = np.zeros(n_nodes + 1, np.int64)
    np.cumsum(np.bincount(src, minlength=n_nodes), out=row_ptr[1:])
    lens = rng.choice([2, 3, 4, 6, 8, 12, 20, 40], n_paths)
    path_ptr = np.zeros(n_paths + 1, np.int64)
    np.cumsum(lens, out=path_ptr[1:])
    nodes = np.empty(int(path_ptr[-1]), np.int32)
    cur = rng.integers(0, n_nodes, n_paths)
    for step in range(int(lens.max())):
        live = np.nonzero(lens > step)[0]
        nodes[path_ptr[live] + step] = cur[live]
        if step % 2 == 0:
            deg = row_ptr[cur[live] + 1] - row_ptr[cur[live]]
            pick = row_ptr[cur[live]] + (rng.integers(0, 1 << 30, live.shape[0]) % np.maximum(deg, 1))
            nxt = np.where(deg > 0, col[np.minimum(pick, col.shape[0] - 1)], rng.integers(0, n_nodes, live.shape[0]))
        else:
            nxt = cur[live] ^ 1
        cur[live] = nxt
    return n_nodes, row_ptr, col, weight, path_ptr, nodes


def scorepaths_timing():
    """ScorePaths weights of 200 k candidate paths (mean 12 ends) on a 50 k-scaffold link graph, everything resident."""
    import torch
    from besst_amd import _lib
    n_nodes, row_ptr, col, weight, path_ptr, nodes = scorepaths_workload()
    n_paths = path_ptr.shape[0] - 1
    lib_h = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    d = [torch.from_numpy(x).to(dev) for x in (row_ptr, col, weight, path_ptr, nodes)]
    good = torch.zeros(n_paths, dtype=torch.int64, device=dev)
    bad = torch.zeros(n_paths, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run():
        _lib.check(lib_h.besst_dev_score_paths(stream, n_nodes, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                               n_paths, d[3].data_ptr(), d[4].data_ptr(), 0, good.data_ptr(),
                                               bad.data_ptr()), 'besst_dev_score_paths')
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    return {'scorepaths_paths': int(n_paths), 'scorepaths_path_ends': int(path_ptr[-1]), 'scorepaths_ms': ms,
            'scorepaths_paths_per_s': n_paths / (ms * 1e-3)}


def scorepaths_cpu_baseline(n_sample=20_000):
    """cpu_baseline leg: the restatement of ScorePaths' weights (oracle/scorepaths_oracle.py, one thread) on the
    first paths of the same batch."""
    from oracle import scorepaths_oracle as PO
    _, row_ptr, col, weight, path_ptr, nodes = scorepaths_workload()
    rp, cl, wl, nl, pp = row_ptr.tolist(), col.tolist(), weight.tolist(), nodes.tolist(), path_ptr.tolist()
    t0 = time.perf_counter()
    for p in range(n_sample):
        PO.link_weights(rp, cl, wl, nl[pp[p]:pp[p + 1]], False)
    dt = time.perf_counter() - t0
    return {'value': n_sample / dt, 'unit': 'paths/s', 'cores': 1, 'kind': 'port',
            'sample': 'first %d paths of the batch, oracle/scorepaths_oracle.link_weights, %.1f s' % (n_sample, dt)}


def linearize_cpu_baseline(n_scaf=200_000, n_edges=300_000):
    """cpu_baseline leg: the sequential restatement of steps 1-4 (oracle/scaffold_oracle.py, one thread) on a
    bounded graph of the same shape."""
    from oracle import scaffold_oracle as SO
    a, b, score = linearize_workload(n_scaf, n_edges)
    la, lb, ls = a.tolist(), b.tolist(), score.tolist()
    t0 = time.perf_counter()
    SO.linearize(n_scaf, la, lb, ls)
    dt = time.perf_counter() - t0
    return {'value': len(la) / dt, 'unit': 'edges/s', 'cores': 1, 'kind': 'port',
            'sample': '%d scaffolds / %d scored link edges, oracle/scaffold_oracle.linearize, %.1f s'
                      % (n_scaf, len(la), dt)}


def source_hash():
    """sha256 over the kernel sources: ties a committed PMC summary to the build it was collected on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, 'besst_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(REPO, 'besst_amd', 'csrc', '*.h'))):
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_step_traffic(config, n_rec, kernel=None):
    """HBM bytes per graph-build step (all kernels of one step; of one kernel when it is named) from the committed rocprofv3 PMC passes of THIS build
    on THIS workload (profiles/r06_<config>_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc
    passes, gfx950 x2 correction on FETCH_SIZE, tools/pmc_summary.py), else None - a summary collected on other
    kernel sources or another record count says nothing about this run."""
    path = os.path.join(REPO, 'profiles', 'r06_%s_pmc_traffic.json' % config.lower())
    try:
        with open(path) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None
    if doc.get('source_hash') != source_hash() or doc.get('records') != n_rec:
        return None
    if kernel is not None:
        return (doc.get('kernels', {}).get(kernel) or {}).get('traffic_bytes_per_step')
    return doc.get('step_traffic_bytes')


def reference_calibration():
    try:
        with open(os.path.join(REPO, 'oracle', 'cpu_port_calibration.json')) as fh:
            doc = json.load(fh)
        return {'port_over_reference': round(doc['port_over_reference'], 3),
                'reference_pairs_per_s_there': round(doc['reference_pairs_per_s'], 1),
                'port_pairs_per_s_there': round(doc['port_pairs_per_s'], 1),
                'measured': 'build container, one core, %s; tools/calibrate_cpu_port.py (real BESST/CreateGraph.py '
                            'record loop vs oracle/py_oracle.record_loop on one stream)' % doc['workload']}
    except (OSError, ValueError, KeyError):
        return None


def measure_single(args, device, config, steps, warmup, copies, pairs=None, contigs=None, verify=True, **variant):
    """One GPU, one library pass at a time, records resident: timed steps, per-kernel breakdown, full-size check.
    Returns (result dict, workload, runner)."""
    import torch
    from besst_amd import _lib, pipeline, workload
    lib_h = _lib.load()
    t0 = time.perf_counter()
    wl = workload.make_device(device, config, 0, pairs=pairs, nc=contigs, **variant)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    lib = wl['lib']
    runner = SingleGpu(device, wl, copies)
    n_rec = runner.rec.n
    n_pairs = n_rec // 2
    # setup, not measurement: the clocks of an idle GPU take a few hundred milliseconds of work to settle (one box of the
    # pool measured 5.3 ms per step in a 30 ms timed region that its own per-kernel breakdown, taken right after, put
    # at 2.9), so the passes run untimed for 0.3 s before the W warm-up steps
    t_setup = time.perf_counter()
    while time.perf_counter() - t_setup < args.settle_seconds:
        for _ in range(4):
            runner.step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        runner.step()
    torch.cuda.synchronize()
    runner.check_capacity()
    # the streaming kernel is timed with HIP events inside the timed region, on every 4th launch: an event pair costs
    # the stream ~3 us, i.e. ~5 % of a C2 step when every launch carries one
    lib_h.besst_prof_sample_every(4 if steps >= 8 else 1)
    lib_h.besst_prof_enable(RECORD_LOOP_SLOTS)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = pipeline.prof_collect()
    lib_h.besst_prof_enable(0)
    lib_h.besst_prof_sample_every(1)
    lib_h.besst_prof_enable(0xffffffff)
    for _ in range(args.breakdown_steps):
        runner.step()
    torch.cuda.synchronize()
    breakdown = {k: round(v[0] / max(1, args.breakdown_steps), 4) for k, v in pipeline.prof_collect().items()}
    lib_h.besst_prof_enable(0)
    n_tuples, n_rows = runner.sizes()
    f = n_tuples / float(n_pairs)
    step_s = elapsed / steps
    alg_step = n_pairs * (38.0 + 32.0 * f)               # 2 x 19 B of records + 16 B per tuple written and read
    # the record loop's kernel - the dominant one of every config - under its own name, priced at ITS algorithmic bytes:
    # stream_kernel reads tid, mtid, mapq, qlen of every record (11 B; the candidates' other columns are ordered_kernel's);
    # the fused forms read all seven columns once (19 B) and write every tuple (16 B)
    loop_kernel = next((k for k in RECORD_LOOP_KERNELS if k in prof), None)
    cls_ms, cls_launches = prof.get(loop_kernel, (0.0, 0))
    cls_avg_s = (cls_ms / max(1, cls_launches)) * 1e-3
    own = n_rec * 11.0 if loop_kernel == 'stream_kernel' else n_rec * 19.0 + n_tuples * 16.0
    own_pmc = pmc_step_traffic(config, n_rec, loop_kernel)
    dominant = max(breakdown.items(), key=lambda kv: kv[1])[0] if breakdown else None
    verified = None
    if verify and not args.no_verify and not args.no_cpu_baseline:
        verified = verify_full(runner, wl)
    res = {
        'value': n_pairs / step_s,
        'ms_per_step': step_s * 1e3,
        'workload': '%s: %d contigs / %d read-pairs, one %s library%s, %d records resident in HBM (%d cop%s cycled)'
                    % (config, wl['asm'].nc, n_pairs, lib['orientation'],
                       ' with %.0f %% PE contamination' % (100 * wl['spec'].contam_frac) if wl['spec'].contam_frac else '',
                       n_rec, copies, 'y' if copies == 1 else 'ies')
                    + (' [variant: %s]' % ', '.join('%s=%s' % kv for kv in sorted(variant.items())) if variant else ''),
        'records': n_rec, 'link_tuples_per_pair': round(f, 5), 'link_tuples': n_tuples, 'edge_rows': n_rows,
        'roofline': {
            'bound': 'hbm', 'scope': 'whole graph-build step, SURVEY 8(d): (38 + 32 f) bytes per read pair',
            'achieved': round(alg_step / step_s / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(alg_step / step_s / 1e9 / HBM_PEAK_GBS, 4),
            'traffic': pmc_step_traffic(config, n_rec),
            'algorithmic_bytes_per_step': alg_step, 'bytes_per_pair': round(38.0 + 32.0 * f, 3),
            'dominant_kernel': dominant,
            'record_loop_kernel': {'name': loop_kernel, 'algorithmic_bytes_per_launch': own,
                                   'bytes': '11 B/record' if loop_kernel == 'stream_kernel' else '19 B/record + 16 B/tuple',
                                   'avg_launch_ms': round(cls_avg_s * 1e3, 4), 'launches_timed': int(cls_launches),
                                   'GBps': round(own / cls_avg_s / 1e9, 1) if cls_avg_s > 0 else None,
                                   'frac': round(own / cls_avg_s / 1e9 / HBM_PEAK_GBS, 4) if cls_avg_s > 0 else None,
                                   'pmc_bytes_per_launch': own_pmc,
                                   'pmc_GBps': round(own_pmc / cls_avg_s / 1e9, 1) if own_pmc and cls_avg_s > 0 else None}},
        'kernel_ms': breakdown,
        'verified_vs_c_oracle': verified,
        'generate_s': round(gen_s, 2),
        'stage2_form': 'tuple by tuple (the runs overflowed)' if runner.gb.sort_flags else 'default',
    }
    return res, wl, runner


def robustness(args, device):
    """The regimes DESIGN.md section 6 names as leaving a fast path, measured the same way as the headline (results stay
    exact: each is checked against the C oracle on its own stream):
      * C3 with 3 % of the contig-spanning pairs chimeric - their links sit on edges of their own, so the run-grouped
        stage 2 sees more runs per chunk and the edge table gets an order of magnitude more rows;
      * C2 name-sorted - mates adjacent, pairs in random order: no wave shares a contig, candidates do not cluster and
        consecutive tuples share no key (stage 2 falls back from runs to the tuple-by-tuple sort)."""
    import torch
    out = {}
    for label, config, steps, variant in (('c3_chimeric_3pct', 'C3', 10, dict(chimeric_frac=0.03)),
                                          ('c2_name_sorted', 'C2', 20, dict(order='name'))):
        C_PORT_TIMING.clear()
        try:
            res, wl, runner = measure_single(args, device, config, steps, 3, 1 if config == 'C3' else 3, **variant)
        except Exception as e:                               # noqa: BLE001 - the bench line must still be printed
            out[label] = {'error': str(e).splitlines()[0][:200] if str(e) else type(e).__name__}
            continue
        out[label] = {k: res[k] for k in ('workload', 'ms_per_step', 'value', 'link_tuples', 'edge_rows', 'kernel_ms',
                                          'verified_vs_c_oracle', 'stage2_form')}
        out[label]['variant'] = variant
        out[label]['roofline_frac'] = res['roofline']['frac']
        del res, wl, runner
        torch.cuda.empty_cache()
    C_PORT_TIMING.clear()
    return out


def cpu_legs(args, wl, with_stage_ports):
    """cpu_baseline object: the Python port on a bounded sample, the C port timings verify_full left behind."""
    if args.cpu_sample_records > 0:
        base, _ = cpu_baseline(wl['batch'], wl['table'], wl['lib'], args.cpu_sample_records)
    else:                    # --cpu-sample-records 0: skip the Python port, keep the C port + full-size check
        base = dict(value=None, unit='read-pairs/s', cores=1, kind='port', sample='python port skipped')
    cal = reference_calibration()
    if cal:
        base['reference_calibration'] = cal
        if base['value']:
            base['reference_equivalent_value'] = base['value'] / cal['port_over_reference']
    if C_PORT_TIMING:      # the C restatement (oracle/besst_oracle.c), whole stream: record loop only
        base['c_port'] = {'value': C_PORT_TIMING['records'] / 2.0 / C_PORT_TIMING['seconds'],
                          'unit': 'read-pairs/s', 'cores': 1,
                          'sample': 'whole stream (%d records), oracle/besst_oracle.c record loop, %.2f s'
                                    % (C_PORT_TIMING['records'], C_PORT_TIMING['seconds'])}
        base['c_port_all_cores'] = {'value': C_PORT_TIMING['records'] / 2.0 / C_PORT_TIMING['mt_seconds'],
                                    'unit': 'read-pairs/s', 'cores': C_PORT_TIMING['mt_cores'],
                                    'equals_sequential': C_PORT_TIMING['mt_equal'],
                                    'sample': 'whole stream, contiguous slices, best of two calls, %.3f s'
                                              % C_PORT_TIMING['mt_seconds']}
    if with_stage_ports:
        base['linearize_port'] = linearize_cpu_baseline()
        base['scorepaths_port'] = scorepaths_cpu_baseline()
    return base


METRIC = 'read-pairs/sec into scaffold graph; edge-set match + gap MAE vs CPU ref'
DTYPE = 'int32/int64 (fp64 for read_len truncation)'


def main_single(args, device, result_fd):
    """N = 1: the headline is BASELINE.json configs[2] (C3, the largest single-GPU config) at full size; configs[1]
    (C2) rides along as a second object of the same line."""
    import torch
    copies = args.copies if args.copies is not None else (1 if args.config == 'C3' else 3)
    res, wl, runner = measure_single(args, device, args.config, args.steps, args.warmup, copies, args.pairs, args.contigs)
    out = {
        'metric': METRIC, 'value': res['value'], 'unit': 'read-pairs/s', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': res['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
        'config': {'workload': res['workload'], 'records_per_gpu': res['records'],
                   'link_tuples_per_pair': res['link_tuples_per_pair'], 'edge_rows': res['edge_rows'],
                   'parallelism': 'single GPU', 'generator': 'besst_amd.synth.simulate_library_device (torch ops on the '
                   'GPU, seeded), %.1f s' % res['generate_s']},
        'roofline': res['roofline'], 'kernel_ms': res['kernel_ms'], 'verified_vs_c_oracle': res['verified_vs_c_oracle'],
    }
    out['roofline']['measured_d2d_copy_GBps'] = round(measured_copy_bandwidth(device), 1)
    if args.in_flight > 1:
        out['overlapped'] = overlapped_throughput(runner, wl, device, args.in_flight, max(args.steps, 30))
    if not args.no_stages:
        del runner
        torch.cuda.empty_cache()
        out['stages'] = stage_timings(wl)
    else:
        del runner
    out['cpu_baseline'] = None if args.no_cpu_baseline else cpu_legs(args, wl, not args.no_stages)
    del wl
    torch.cuda.empty_cache()
    if not args.no_stages:
        # end to end through the reference's two entry points, from host records to the scored graphs
        out['stages']['dropin_' + args.config.lower()] = dropin_timing(device, args.config, args.pairs, args.contigs)
        if args.also and args.also != args.config and args.pairs is None:
            out['stages']['dropin_' + args.also.lower()] = dropin_timing(device, args.also)
        if args.pairs is None:
            # BAM bytes -> graphs: C2 always, the headline config when the host has the memory for its 400 M-record file
            for cfg_name in ([args.also] if args.also else []) + [args.config]:
                # (the headline config as a quarter of its pairs: writing and reading back a 400 M-record BAM file would
                # double the run time of the default bench on a host limited to 16 CPUs)
                big = cfg_name not in ('C1', 'C2')
                if big and _host_memory_gib() < 64:
                    continue
                try:
                    out['stages']['bam_to_graph_' + cfg_name.lower()] = bam_to_graph_timing(
                        device, cfg_name, pairs=50_000_000 if big else None, realistic=big)
                    if big:
                        out['stages']['bam_ingest_sequencer_like'] = bam_ingest_timing(device, cfg_name, pairs=20_000_000)
                except Exception as e:                       # noqa: BLE001 - the bench line must still be printed
                    out['stages']['bam_to_graph_' + cfg_name.lower()] = {'error': str(e).splitlines()[0][:200] if str(e) else type(e).__name__}
    if args.also and args.also != args.config and args.pairs is None:
        C_PORT_TIMING.clear()
        res2, wl2, runner2 = measure_single(args, device, args.also, 20, 3, 3)
        over2 = overlapped_throughput(runner2, wl2, device, args.in_flight, 60) if args.in_flight > 1 else None
        del runner2
        second = {k: res2[k] for k in ('value', 'ms_per_step', 'workload', 'records', 'link_tuples_per_pair',
                                        'edge_rows', 'roofline', 'kernel_ms', 'verified_vs_c_oracle')}
        second['unit'] = 'read-pairs/s'
        second['steps'], second['warmup'] = 20, 3
        if over2 is not None:
            second['overlapped'] = over2
        if not args.no_cpu_baseline and C_PORT_TIMING:
            second['c_port'] = {'value': C_PORT_TIMING['records'] / 2.0 / C_PORT_TIMING['seconds'], 'cores': 1}
        out[args.also.lower()] = second
    if not args.no_robustness and not args.no_stages and args.pairs is None and args.config == 'C3':
        out['robustness'] = robustness(args, device)
        try:
            out['c4_single_gpu'] = c4_single_gpu(args, device)
        except Exception as e:                               # noqa: BLE001 - the bench line must still be printed
            out['c4_single_gpu'] = {'error': str(e).splitlines()[0][:200] if str(e) else type(e).__name__}
        try:
            out['c5_library_single_gpu'] = c5_library_single_gpu(args, device)
        except Exception as e:                               # noqa: BLE001 - the bench line must still be printed
            out['c5_library_single_gpu'] = {'error': str(e).splitlines()[0][:200] if str(e) else type(e).__name__}
        # the N = 1 point of the weak-scaling curve the driver's --gpus 2 / 4 / 8 lines belong to: the per-GPU shape of C4
        # (one eighth of each library) with one rank through the sharded orchestration over RCCL - this line's own value is
        # C3, a different workload; efficiency at N is value(N) / (N x this object's value)
        out['weak_scaling_n1'] = weak_scaling_first_point(args)
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + '\n').encode())


def weak_scaling_first_point(args):
    """`bench.py --gpus 1 --config C4` in a process of its own (the process group, its buffers and the C4 assembly stay
    out of this one), after this process has given its HBM back."""
    import subprocess
    import torch
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BESST_DIST_BACKEND')}
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--config', 'C4', '--steps', str(args.steps), '--warmup',
           str(args.warmup), '--no-cpu-baseline']
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
        if res.returncode != 0 or len(lines) != 1:
            return {'error': 'rc %d: %s' % (res.returncode, res.stderr.strip().splitlines()[-1][:200] if res.stderr.strip() else '')}
        d = json.loads(lines[0])
        return {'command': 'python bench.py --gpus 1 --config C4', 'workload': d['config']['workload'], 'value': d['value'],
                'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'n_gpus': d['n_gpus'], 'backend': d['backend'],
                'rccl_ranks_seen': d['rccl_ranks_seen'], 'roofline_frac': d['roofline']['frac'],
                'link_tuples_per_pair': d['config']['link_tuples_per_pair'],
                'exchange_consistent': all(l['exchange_consistent'] for l in d['config']['libraries'])}
    except Exception as e:                                   # noqa: BLE001 - the bench line must still be printed
        return {'error': str(e).splitlines()[0][:200] if str(e) else type(e).__name__}


def main():
    args = parse_args()
    # BESST_BENCH_WATCHDOG=<seconds>: every thread's Python stack to stderr at that interval (where is a run that does not
    # come back?)
    if os.environ.get('BESST_BENCH_WATCHDOG'):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['BESST_BENCH_WATCHDOG']), repeat=True, file=sys.stderr)
    # Only the final JSON line may reach stdout: RCCL prints a version banner to fd 1 when the process group is
    # created, so everything before the result is routed to stderr at the file-descriptor level.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    # A container limited to a few CPUs of a large host (the GPU boxes: 16 of 256): torch sizes its thread pools by the
    # machine, and a few hundred threads spinning behind every small CPU op get the whole cgroup throttled for most of each
    # scheduling period - the host side of a step (a dozen launches and collectives) then takes milliseconds.  Sized by what
    # the process may use, unless the caller has said otherwise (BESST_BENCH_THREADS=0 leaves the defaults).
    if os.environ.get('BESST_BENCH_THREADS', '1') != '0':
        from besst_amd import _lib as _l
        usable = str(max(1, min(_l.effective_cpus(), 16)))
        for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
            os.environ.setdefault(var, usable)
        os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
    import torch

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no GPU visible; besst_amd has no CPU path)')
    # BESST_DIST_BACKEND=gloo: several ranks on ONE GPU - a check of the sharded orchestration (RCCL refuses two ranks on
    # one device); collectives are host-staged there, so its timings say nothing about a multi-GPU node.
    backend = os.environ.get('BESST_DIST_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU), or say why not.  Never fall
        # through to the one-GPU line for a request of N.
        if backend == 'nccl' and n_dev < args.gpus:
            raise SystemExit('bench.py --gpus %d: %d GPU(s) visible here and RCCL wants one device per rank - nothing was '
                             'measured (BESST_DIST_BACKEND=gloo runs the ranks on the GPUs there are: an orchestration '
                             'check, not a scaling figure)' % (args.gpus, n_dev))
        import socket
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
        sock.close()
        os.dup2(result_fd, 1)                                # the ranks route their own output
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
                                  str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py --gpus %d under a launcher with WORLD_SIZE=%d: the line would carry the wrong n_gpus - '
                         'start it with --nproc-per-node %d (or without a launcher: it starts its own ranks)'
                         % (args.gpus, world, args.gpus))
    if backend == 'gloo':
        local_rank %= n_dev
    elif local_rank >= n_dev:
        raise SystemExit('bench.py: rank %d wants GPU %d of %d visible - RCCL needs one device per rank' % (rank, local_rank, n_dev))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    # BESST_FORCE_DISTRIBUTED=1, or --config C4 / C5 with one GPU: ONE rank through the sharded orchestration over RCCL - the
    # N = 1 point of the scaling curve (the same shape per GPU as N = 2, 4, 8) instead of the C3 line
    force_dist = os.environ.get('BESST_FORCE_DISTRIBUTED') == '1' or (world == 1 and args.config in ('C4', 'C5'))
    if world == 1 and not force_dist:
        if args.config is None:
            args.config = 'C3'
        return main_single(args, device, result_fd)
    return main_sharded(args, device, rank, world, backend, force_dist, result_fd)


def c5_library_single_gpu(args, device):
    """ONE library of BASELINE.json configs[4] (C5) at FULL size on one GPU: 2 M contigs, 1.33 G read pairs = 2.67 G records
    (66.7 GB) in one stream - the first one past 2^31 records - the 5 kb mate-pair library on the table a previous pass leaves
    behind (scaffold ids counting on from 2 M: 45-bit keys).  Correctness at this size is
    tests/test_gpu_fullsize.py::test_full_size_c5_library_one_gpu (every record against the C oracle); here it is timed."""
    import torch
    from besst_amd import pipeline, workload
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(device)
    if free < 200e9:
        return {'skipped': 'needs 200 GB of free HBM (%.0f GB here)' % (free / 1e9)}
    t0 = time.perf_counter()
    wl = workload.make_device_windowed(device, 'C5', 1)
    rec = pipeline.DeviceRecords.from_columns(wl['cols'])
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    asm, nb, lib, table = wl['asm'], wl['node_bits'], wl['lib'], wl['table']
    probe = pipeline.DeviceGraphBuilder(device, asm.nc, nb, lib, rec.n, 1)
    probe.set_contigs(**table)
    probe.reset()
    probe.classify(rec)
    n_tuples, _ = probe.read_sizes()
    del probe
    gb = pipeline.DeviceGraphBuilder(device, asm.nc, nb, lib, rec.n, int(n_tuples * 1.1) + 4096)
    gb.set_contigs(**table)
    for _ in range(3):
        gb.step(rec)
    torch.cuda.synchronize()
    steps = max(5, min(args.steps, 20))
    t0 = time.perf_counter()
    for _ in range(steps):
        gb.step(rec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_t, n_r = gb.read_sizes()
    pairs = rec.n / 2.0
    f = n_t / pairs
    alg = pairs * (38.0 + 32.0 * f)
    return {'workload': 'one C5 library at full size on one GPU: %d contigs, %d records (> 2^31) resident, %s N(%g, %g), later-pass '
                        'contig table' % (asm.nc, rec.n, wl['spec'].orientation, wl['spec'].mean, wl['spec'].sd),
            'ms_per_step': round(dt * 1e3, 4), 'value': pairs / dt, 'unit': 'read-pairs/s', 'steps': steps,
            'records': rec.n, 'link_tuples': n_t, 'edge_rows': n_r, 'node_bits': nb, 'key_bits': gb.key_bits,
            'link_tuples_per_pair': round(f, 5), 'record_path': 'fused' if gb.params.record_path else 'two-pass',
            'roofline_frac': round(alg / dt / 1e9 / HBM_PEAK_GBS, 4), 'algorithmic_bytes_per_step': alg,
            'generate_s': round(gen_s, 1), 'hbm_allocated_bytes': int(torch.cuda.memory_allocated(device)),
            'verified': 'tests/test_gpu_fullsize.py::test_full_size_c5_library_one_gpu (same seeds, every record against the C '
                        'oracle; the metrics pass on records beyond 2^31)'}


def c4_single_gpu(args, device):
    """BASELINE.json configs[3] (C4) at FULL size on ONE GPU: 500 k contigs, the PE library on the first-library table, then
    the mate-pair library on the table a previous pass leaves behind - 5e8 read pairs = 1e9 records (25 GB) each, both
    resident at once, a step = both libraries' passes one after the other (and, labelled, with three passes in flight).
    Correctness of this shape at this size is tests/test_gpu_fullsize.py::test_full_size_c4_one_gpu (every record against
    the C oracle); here it is timed."""
    import torch
    from besst_amd import pipeline, synth, workload
    free, _ = torch.cuda.mem_get_info(device)
    if free < 150e9:
        return {'skipped': 'needs 150 GB of free HBM (%.0f GB here)' % (free / 1e9)}
    cfg = synth.CONFIGS['C4']
    seed = synth.config_seed('C4')
    asm = synth.make_assembly(cfg['nc'], cfg['median'], seed)
    per_lib = cfg['pairs'] // len(cfg['libs'])
    libs, t_gen = [], time.perf_counter()
    for li, spec in enumerate(cfg['libs']):
        lib = workload.library_constants(spec)
        thr = spec.mean + 4 * spec.sd
        table = workload.first_library_table(asm.lengths, thr) if li == 0 else \
            workload.later_library_table(asm, seed + 50 + li, thr, first_scaffold_id=asm.nc * li + 1)
        nb = workload.node_bits_for(table)
        rec = pipeline.DeviceRecords.from_columns(synth.simulate_library_device(asm, spec, per_lib, seed + 100 + li, device))
        probe = pipeline.DeviceGraphBuilder(device, asm.nc, nb, lib, rec.n, 1)
        probe.set_contigs(**table)
        probe.reset()
        probe.classify(rec)
        n_tuples, _ = probe.read_sizes()
        del probe
        gb = pipeline.DeviceGraphBuilder(device, asm.nc, nb, lib, rec.n, int(n_tuples * 1.1) + 4096)
        gb.set_contigs(**table)
        libs.append((spec, rec, gb, nb, lib, table))
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t_gen

    def step():
        for _, rec, gb, _, _, _ in libs:
            gb.step(rec)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    steps = max(5, min(args.steps, 20))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out_libs, tuples = [], 0
    for spec, rec, gb, nb, lib, table in libs:
        n_t, n_r = gb.read_sizes()
        tuples += n_t
        out_libs.append({'library': '%s N(%g, %g)' % (spec.orientation, spec.mean, spec.sd), 'records': rec.n, 'node_bits': nb,
                         'key_bits': gb.key_bits, 'link_tuples': n_t, 'edge_rows': n_r,
                         'record_path': 'fused' if gb.params.record_path else 'two-pass'})
    pairs = per_lib * len(libs)
    f = tuples / float(pairs)
    alg = pairs * (38.0 + 32.0 * f)
    res = {'workload': 'C4 at full size on one GPU: %d contigs, %d libraries x %d read-pairs (%d records resident)'
                       % (asm.nc, len(libs), per_lib, 2 * pairs),
           'ms_per_step': round(dt * 1e3, 4), 'value': pairs / dt, 'unit': 'read-pairs/s', 'steps': steps,
           'link_tuples_per_pair': round(f, 5), 'libraries': out_libs, 'generate_s': round(gen_s, 1),
           'roofline_frac': round(alg / dt / 1e9 / HBM_PEAK_GBS, 4), 'algorithmic_bytes_per_step': alg,
           'hbm_allocated_bytes': int(torch.cuda.max_memory_allocated(device)),
           'verified': 'tests/test_gpu_fullsize.py::test_full_size_c4_one_gpu (same seeds, every record against the C oracle)'}
    # three passes in flight (labelled: never the figure above): the second library's pass under the first one's tail
    if args.in_flight > 1 and torch.cuda.mem_get_info(device)[0] > 110e9:
        pools = []
        for spec, rec, gb, nb, lib, table in libs:
            pool = pipeline.PassPool(device, asm.nc, nb, lib, rec.n, gb.tup_cap, 2)
            pool.set_contigs(**table)
            pools.append((pool, rec))
        torch.cuda.synchronize()
        for _ in range(2):
            for pool, rec in pools:
                pool.submit(rec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for pool, rec in pools:
                pool.submit(rec)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / steps
        res['overlapped'] = {'in_flight': 2 * len(pools), 'ms_per_step': round(dt2 * 1e3, 4), 'value': pairs / dt2}
        del pools
    del libs
    torch.cuda.empty_cache()
    return res


def owner_checksums(keys, payload, owners, world):
    """[world, 4] order-independent summaries (count, sum key, sum payload, sum of a mix) of a tuple multiset per owner,
    in wrapping 64-bit arithmetic - numpy or torch tensors alike."""
    import torch
    k = torch.as_tensor(keys.view(np.int64) if isinstance(keys, np.ndarray) else keys)
    p = torch.as_tensor(payload.view(np.int64) if isinstance(payload, np.ndarray) else payload)
    o = torch.as_tensor(owners).to(torch.int64)
    mix = k ^ (p * -7046029254386353131)                     # 0x9E3779B97F4A7C15 as a signed 64-bit constant
    out = torch.zeros(world, 4, dtype=torch.int64, device=k.device)
    out[:, 0].index_add_(0, o, torch.ones_like(k))
    out[:, 1].index_add_(0, o, k)
    out[:, 2].index_add_(0, o, p)
    out[:, 3].index_add_(0, o, mix)
    return out


def verify_sharded_vs_oracle(job, wl, rank, world, device, backend_name):
    """N >= 1, any size: every rank checks ITS share against the C oracle, nothing is rebuilt on one rank.
      * the oracle's record loop runs on the rank's own slice (host), with the duplicate chain's carry-in taken from
        the tails of the slices before it (all-gather of three words);
      * its tuples are split by key owner and summarised per owner (count + three wrapping sums); the summaries are
        summed over the ranks, and each owner compares its line with the same summary of the tuples it RECEIVED;
      * the owner's edge table must equal the oracle's aggregation of those received tuples (keys, link counts, sums,
        per-link observations in arrival order, first-occurrence index);
      * coverage and counters: sums of the oracles' per-slice values against the all-reduced device values."""
    import torch
    import torch.distributed as dist
    from besst_amd import distributed
    from oracle import c_oracle as CO
    b = job.backend
    cpu = torch.device('cpu')
    coll_dev = cpu if backend_name == 'gloo' else device
    from besst_amd._lib import effective_cpus
    threads = max(1, effective_cpus() // max(1, world))
    batch, table, lib, nb = wl['batch'], wl['table'], wl['lib'], wl['node_bits']
    keys, payload, aligned, ctr = CO.record_loop(batch, table, lib, nb, threads=threads)
    tails = torch.zeros(world, 3, dtype=torch.int64, device=coll_dev)
    tails[rank] = torch.tensor([1 if ctr[7] > 0 else 0, int(ctr[8]), int(ctr[9])], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(tails)
    prev = (-1, -1)
    for j in range(rank):
        if int(tails[j, 0]):
            prev = (int(tails[j, 1]), int(tails[j, 2]))
    if prev != (-1, -1):
        keys, payload, aligned, ctr = CO.record_loop(batch, table, lib, nb, prev=prev, threads=threads)
    scaf = (keys >> np.uint64(2 + nb)).astype(np.uint64)
    owners = ((((scaf * np.uint64(2654435761)) & np.uint64(0xffffffff)) >> np.uint64(15)) % np.uint64(world)).astype(np.int64)
    want = owner_checksums(keys, payload, owners, world).to(coll_dev)
    sums = torch.cat([torch.from_numpy(aligned.astype(np.int64)), torch.from_numpy(ctr[:8].astype(np.int64))]).to(coll_dev)
    if world > 1:
        dist.all_reduce(want)
        dist.all_reduce(sums)
    n_recv, n_rows = b.sizes()
    rk, rp = b.rkeys[:n_recv], b.rpayload[:n_recv]
    got = owner_checksums(rk, rp, torch.full((n_recv,), rank, dtype=torch.int64, device=device), world)[rank].cpu()
    ok_exchange = bool(torch.equal(got, want[rank].cpu()))
    table_d = b.local_table()
    hk, hp = rk.cpu().numpy().view(np.uint64), rp.cpu().numpy().view(np.uint64)
    rows = CO.edge_rows(hk, hp)
    gidx = b.gidx[:n_recv].cpu().numpy().view(np.uint32).astype(np.int64)
    link = ~table_d.is_fishy
    ok_rows = bool(np.array_equal(table_d.key, rows['key']) and np.array_equal(table_d.n.astype(np.int64), rows['n'])
                   and np.array_equal(table_d.sum_obs[link], rows['sum_obs'][link])
                   and np.array_equal(table_d.sum_obs_sq[link], rows['sum_obs_sq'][link])
                   and np.array_equal(table_d.obs_lo.astype(np.int64), rows['obs_lo'])
                   and np.array_equal(table_d.obs_hi.astype(np.int64), rows['obs_hi'])
                   and np.array_equal(table_d.first_idx.astype(np.int64), gidx[rows['first_idx']] if n_recv else rows['first_idx']))
    dev_sums = torch.cat([b.aligned.cpu(), b.counter_words.cpu()[:8]])
    ok_sums = bool(torch.equal(dev_sums, sums.cpu()))
    last = (-1, -1)
    for j in range(world):
        if int(tails[j, 0]):
            last = (int(tails[j, 1]), int(tails[j, 2]))
    ok_prev = job.final_prev_obs() == last
    flags = torch.tensor([int(ok_exchange), int(ok_rows), int(ok_sums), int(ok_prev)], dtype=torch.int64, device=coll_dev)
    if world > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    return {'tuples_received_match_oracle_per_owner': bool(flags[0]), 'owner_edge_tables_match_oracle': bool(flags[1]),
            'coverage_and_counters_match_oracle': bool(flags[2]), 'final_prev_obs_matches': bool(flags[3])}


def main_sharded(args, device, rank, world, backend_name, force_dist, result_fd):
    """N > 1 (or one rank forced through the sharded orchestration): BASELINE.json configs[3] (C4) shaped weak scaling -
    ONE assembly of 500 k contigs, TWO libraries (PE 500 bp on the first-library contig table, then MP 5 kb with PE
    contamination on the table a previous pass leaves behind); every rank holds one eighth of each library's pairs
    (62.5 M pairs per library and GPU: the full config on eight GPUs), as a contiguous slice of the rank-ordered stream.
    A step = both library passes, each: per-record pass on the slice -> one all-to-all by key owner -> per-owner sort and
    reduction."""
    import torch
    import torch.distributed as dist
    from besst_amd import _lib, distributed, pipeline, synth, workload
    if 'MASTER_ADDR' not in os.environ:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ.setdefault('MASTER_PORT', '29531')
    # a rank that never arrives must end the job, not hold it: every collective gives up after this long (gloo raises in
    # the waiting ranks; under RCCL the watchdog tears the process down) - BESST_COLLECTIVE_TIMEOUT seconds, default 900 (rank 0 may spend a minute writing a library's file while the others wait)
    import datetime
    limit = float(os.environ.get('BESST_COLLECTIVE_TIMEOUT', '900'))
    kw = {'timeout': datetime.timedelta(seconds=limit)} if limit > 0 else {}     # (0: the backend's own default)
    if backend_name == 'gloo':
        dist.init_process_group('gloo', rank=rank, world_size=world, **kw)
    else:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device, **kw)
    # ---- who is here: the ranks the collective library sees and the devices they run on (a SCALE file whose ranks share a
    # device, or whose group is smaller than --gpus, is not a scaling measurement) - BEFORE anything is allocated
    uuid = str(getattr(torch.cuda.get_device_properties(device), 'uuid', '')) or 'device-%d' % device.index
    uuids = [None] * world
    if world > 1:
        dist.all_gather_object(uuids, uuid)
    else:
        uuids = [uuid]
    sharing = uuids.count(uuid)                              # ranks on this rank's GPU
    if backend_name != 'gloo' and len(set(uuids)) < world:
        raise SystemExit('bench.py --gpus %d: the %d ranks sit on %d distinct GPU(s) (%s) - RCCL needs one device per rank; '
                         'nothing was measured' % (world, world, len(set(uuids)), ', '.join(sorted(set(uuids)))))
    config = args.config or 'C4'
    cfg = synth.CONFIGS[config]
    seed = synth.config_seed(config)
    n_ctg = int(args.contigs if args.contigs is not None else cfg['nc'])
    asm = synth.make_assembly(n_ctg, cfg['median'], seed)
    per_lib = int(args.pairs if args.pairs is not None else
                  (12_500_000 if args.from_bam else cfg['pairs'] // len(cfg['libs']) // 8))
    cap_env = os.environ.get('BESST_PAIR_CAPACITY')     # start from a (too small) region capacity: grow-and-retry path
    # the step's HBM budget, item by item (distributed.memory_budget), against what the GPU has free - before anything
    # is allocated.  Tuples per record as in tools/memory_budget.py: twice what the libraries' streams measure.
    budget = {}
    for li, spec in enumerate(cfg['libs']):
        n_tup = int(2 * per_lib * (0.214 if spec.orientation == 'rf' else 0.014))
        budget['library %d' % (li + 1)] = distributed.memory_budget(
            2 * per_lib, n_ctg, world, int(cap_env) if cap_env else int(n_tup * 1.5 / world) + 4096,
            int(n_tup * 1.25) + 4096)['total']
    free_hbm, total_hbm = torch.cuda.mem_get_info(device)
    need = sum(budget.values())
    # what drawing a library leaves in flight beside the step's buffers: its columns twice (unsorted parts + sorted copy),
    # sort key and permutation, and one chunk of the generator's temporaries
    draw = 2 * per_lib * (2 * 25 + 16) + int(min(32_000_000, per_lib * 1.3) * 260)
    if (need + draw) * sharing > 0.9 * free_hbm:
        # (ranks that share a GPU - gloo - share its HBM: eight C4-sized ranks on one device once sat in the allocator
        # for the rest of the box's time limit instead of failing)
        raise SystemExit('bench.py --gpus %d: a rank needs %.1f GB of HBM for the step (%s) + %.1f GB while a library is drawn, '
                         '%d rank(s) share this GPU, %.1f GB are free - nothing was measured; use --pairs to shrink the slices'
                         % (world, need / 1e9, ', '.join('%s %.1f GB' % (k, v / 1e9) for k, v in budget.items()),
                            draw / 1e9, sharing, free_hbm / 1e9))
    jobs, wls = [], []
    from_bam = [] if args.from_bam else None
    keep_bams = []
    for li, spec in enumerate(cfg['libs']):
        if args.from_bam:
            cols, report, bam = ingest_library_from_bam(asm, spec, per_lib, seed + 100 + li, device, rank, world, li)
            from_bam.append(report)
            keep_bams.append(bam)                            # (owns the columns' memory)
        else:
            cols = synth.simulate_library_device(asm, spec, per_lib, seed + 100 + li + 7919 * rank, device,
                                                 window=(rank, world) if (args.slices == 'contiguous' and world > 1) else None)
        thr = spec.mean + 4 * spec.sd
        table = (workload.first_library_table(asm.lengths, thr) if li == 0 else
                 workload.later_library_table(asm, seed + 50 + li, thr, first_scaffold_id=asm.nc * li + 1))
        wl = workload.DeviceWorkload(config=config, asm=asm, cols=cols, table=table, lib=workload.library_constants(spec),
                                     node_bits=workload.node_bits_for(table), pairs=per_lib, spec=spec)
        wls.append(wl)
        jobs.append(distributed.ShardedGraphBuild(device, wl, rank, world, pair_capacity=int(cap_env) if cap_env else None))
    lib_h = _lib.load()

    def step():
        for job in jobs:
            job.step()

    # ---- the like-for-like one-GPU figure: the SAME shape (this rank's slice of both libraries) through the same
    # orchestration with a group of one, timed on rank 0 before the timed region - weak-scaling efficiency is value(N) /
    # (N x this), not against the C3 line that `--gpus 1` prints
    same_shape = None
    if world > 1:
        solo_groups = [dist.new_group([r]) for r in range(world)]      # (collective: every rank makes every group)
        if rank == 0:
            solo = [distributed.ShardedGraphBuild(device, wl, 0, 1, group=solo_groups[0],
                                                  pair_capacity=int(cap_env) if cap_env else None) for wl in wls]
            for k in range(40):
                for job in solo:
                    job.step()
            torch.cuda.synchronize()
            for job in solo:
                job.check_capacity()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for job in solo:
                    job.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            same_shape = {'ms_per_step': round(dt * 1e3, 4), 'value': per_lib * len(solo) / dt, 'unit': 'read-pairs/s',
                          'what': 'rank 0\'s slice of every library through the sharded orchestration with a group of one '
                                  '(all-to-all to itself), %d steps, before the timed region' % args.steps}
            del solo
            torch.cuda.empty_cache()
        dist.barrier()

    # setup, not measurement: RCCL opens its channels and the builders size their exchange regions on the first passes
    # (a run whose only untimed passes were two warm-up steps once measured 2.8 ms per step instead of 1.36), and an idle
    # GPU's clocks settle: 150 passes (0.2-0.5 s)
    for k in range(150):                                 # (a FIXED count: every pass holds collectives)
        step()
        if k % 8 == 7:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    for job in jobs:
        job.check_capacity()
    lib_h.besst_prof_sample_every(1)
    lib_h.besst_prof_enable(RECORD_LOOP_SLOTS)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    pipeline.prof_collect()
    lib_h.besst_prof_enable(0)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if backend_name == 'gloo' else device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    lib_h.besst_prof_enable(0xffffffff)
    for _ in range(args.breakdown_steps):
        step()
    torch.cuda.synchronize()
    breakdown = {k: round(v[0] / max(1, args.breakdown_steps), 4) for k, v in pipeline.prof_collect().items()}
    lib_h.besst_prof_enable(0)
    libs_out, n_tuples_total, checks = [], 0, []
    for li, (job, wl) in enumerate(zip(jobs, wls)):
        n_tuples, n_rows = job.sizes()
        n_tuples_total += n_tuples
        b = job.backend
        emitted = int(b.counter_words.cpu()[6].item())
        entry = {'library': '%s N(%g, %g)%s' % (wl['spec'].orientation, wl['spec'].mean, wl['spec'].sd,
                                                 ' + %.0f %% PE contamination' % (100 * wl['spec'].contam_frac)
                                                 if wl['spec'].contam_frac else ''),
                 'record_path': 'fused' if b.gb.params.record_path else 'two-pass', 'node_bits': wl['node_bits'],
                 'link_tuples': n_tuples, 'edge_rows': n_rows,
                 'exchange_consistent': bool(emitted == n_tuples and not b.overflowed())}
        if not args.no_verify and not args.no_cpu_baseline:
            try:
                entry['verified_vs_c_oracle'] = verify_sharded_vs_oracle(job, wl, rank, world, device, backend_name)
            except Exception as e:                       # noqa: BLE001 - the bench line must still be printed
                entry['verified_vs_c_oracle'] = 'error: %s' % (str(e).splitlines()[0][:200] if str(e) else type(e).__name__)
        libs_out.append(entry)
    if rank == 0:
        total_pairs = per_lib * len(jobs) * world
        step_s = elapsed / args.steps
        f = n_tuples_total / float(total_pairs)
        alg_step = total_pairs * (38.0 + 32.0 * f)
        ok = all(isinstance(e.get('verified_vs_c_oracle'), dict) and all(e['verified_vs_c_oracle'].values()) for e in libs_out)
        out = {
            'metric': METRIC, 'value': total_pairs / step_s, 'unit': 'read-pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': step_s * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
            'config': {'workload': '%s-shaped: %d contigs, %d libraries, %d read-pairs per library and GPU (%d in all), '
                                   'records resident in HBM; a step = every library\'s pass'
                                   % (config, asm.nc, len(jobs), per_lib, total_pairs),
                       'records_per_gpu': 2 * per_lib * len(jobs), 'link_tuples_per_pair': round(f, 5),
                       'parallelism': 'stream-slice x%d + key-owner all-to-all (RCCL)' % world, 'libraries': libs_out,
                       'hbm_budget_bytes_per_gpu': need, 'hbm_allocated_bytes_rank0': int(torch.cuda.max_memory_allocated(device))},
            'roofline': {'bound': 'hbm', 'scope': 'whole step, SURVEY 8(d): (38 + 32 f) bytes per read pair, over the '
                                                  'aggregate peak of all GPUs',
                         'achieved': round(alg_step / step_s / 1e9, 1), 'peak': HBM_PEAK_GBS * world, 'unit': 'GB/s',
                         'frac': round(alg_step / step_s / 1e9 / (HBM_PEAK_GBS * world), 4), 'traffic': None,
                         'traffic_note': 'PMC passes (rocprofv3 --pmc) are taken per process on one GPU: profiles/ holds '
                                         'them for the single-GPU step; none was collected under torch.distributed.run',
                         'algorithmic_bytes_per_step': alg_step},
            'kernel_ms': breakdown,
            'verified_vs_c_oracle': ok if not (args.no_verify or args.no_cpu_baseline) else None,
            'rccl_ranks_seen': int(dist.get_world_size()), 'backend': backend_name, 'device_uuids': uuids,
            'distinct_devices': len(set(uuids)), 'slices': args.slices if world > 1 else 'whole stream',
            'single_gpu_same_shape': same_shape if world > 1 else {
                'ms_per_step': round(step_s * 1e3, 4), 'value': total_pairs / step_s, 'unit': 'read-pairs/s',
                'what': 'this line: one rank through the sharded orchestration (all-to-all to itself) - the N = 1 point of '
                        'the weak-scaling curve, the per-GPU shape of --gpus 2 / 4 / 8'},
            'cpu_baseline': sharded_cpu_baseline(args, wls[0]),
        }
        if from_bam is not None:
            out['from_bam'] = from_bam
            if isinstance(out.get('single_gpu_same_shape'), dict):
                # the one-rank figure of the same shape: rank 0's slice of every file ingested alone (the others waiting)
                out['single_gpu_same_shape']['ingest'] = [dict(lib['rank0_slice_alone'] or {}, library=lib['library']) for lib in from_bam]
            out['slices'] = 'slices of one BAM file per library (distributed.ingest_slice)'
            out['data'] = 'synthetic (written as BAM files by rank 0, untimed; ingested by every rank on its GPU)'
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + '\n').encode())
    dist.barrier()
    dist.destroy_process_group()


def ingest_library_from_bam(asm, spec, per_lib, seed, device, rank, world, li):
    """--from-bam: ONE file for the library (rank 0 draws world x per_lib pairs, writes them as a sequencer-like BAM into
    /dev/shm - untimed scaffolding), then every rank ingests its slice on its GPU.  Timed: rank 0's slice on its own first
    (the other ranks wait: the like-for-like one-rank figure), then all slices together - what tells whether the host's page
    cache and cores or the GPUs bound an N-GPU ingest.  -> (column dict, report for the bench line on rank 0, ResidentBam)."""
    import shutil
    import tempfile
    import torch
    import torch.distributed as dist
    from besst_amd import bamio, distributed, synth
    multi = world > 1 and dist.is_initialized()
    box = [None]
    if rank == 0:
        cols = synth.simulate_library_device(asm, spec, per_lib * world, seed, device)
        batch = synth.device_columns_to_batch(asm, cols, int(spec.read_len))
        del cols
        torch.cuda.empty_cache()
        base = '/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > 128 * len(batch) else None
        tmp = tempfile.mkdtemp(prefix='besst_amd_bam_', dir=base)
        path = os.path.join(tmp, 'library%d.bam' % (li + 1))
        t0 = time.perf_counter()
        bamio.write_bam(path, batch, realistic=True)
        box[0] = (path, len(batch), os.path.getsize(path), round(time.perf_counter() - t0, 2))
        del batch
    if multi:
        dist.broadcast_object_list(box, src=0)
    path, n_records, n_bytes, write_s = box[0]
    solo = None
    try:
        if multi:
            if rank == 0:                                    # slice 0 alone: it begins behind the header, nothing to settle
                t0 = time.perf_counter()
                alone = bamio.ResidentBam(path, device_index=device.index, part=(0, world), first_skip=-1)
                dt = time.perf_counter() - t0
                solo = {'records': len(alone), 'ingest_s': round(dt, 4), 'records_per_s': len(alone) / dt,
                        'staging_s': round(alone.ingest.decode_seconds, 4), 'wait_s': round(alone.ingest.copy_wait_seconds, 4)}
                alone.close()
            dist.barrier()
        info = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bam, cols = distributed.ingest_slice(path, rank, world, device_index=device.index, info=info)
        dt = time.perf_counter() - t0
        mine = {'rank': rank, 'records': len(bam), 'ingest_s': round(dt, 4), 'staging_s': round(bam.ingest.decode_seconds, 4),
                'wait_s': round(bam.ingest.copy_wait_seconds, 4), 'chunks': int(bam.ingest.chunks), 'rounds': info.get('rounds', 1),
                'reads': info.get('reads', 1), 'compressed_bytes': int(bam.ingest.bytes_h2d)}
        per_rank = [mine]
        if multi:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
    finally:
        if multi:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(os.path.dirname(path), ignore_errors=True)
    slowest = max(r['ingest_s'] for r in per_rank)
    assert sum(r['records'] for r in per_rank) == n_records
    report = {'library': li + 1, 'file': 'sequencer-like: pseudo-random bases, slowly changing qualities, %.1f B/record compressed'
                                         % (n_bytes / float(n_records)),
              'bam_bytes': n_bytes, 'records': n_records, 'write_bam_s_untimed': write_s, 'per_rank': per_rank,
              'ingest_s_slowest_rank': slowest, 'aggregate_records_per_s': n_records / slowest,
              'handshake_rounds': max(r['rounds'] for r in per_rank), 'slices_read_twice': sum(r['reads'] - 1 for r in per_rank),
              'rank0_slice_alone': solo}
    return cols, report, bam


def sharded_cpu_baseline(args, wl):
    """Rank 0's cpu_baseline of an N > 1 line: the Python port on the head of ITS slice of the first library (a bounded
    sample, one core) + the committed calibration against the real reference loop."""
    if args.no_cpu_baseline or args.cpu_sample_records <= 0:
        return {'value': None, 'unit': 'read-pairs/s', 'cores': 1, 'kind': 'port', 'sample': 'skipped (--no-cpu-baseline)'}
    from besst_amd import synth
    n = int(min(args.cpu_sample_records, wl['cols']['tid'].shape[0]))
    head = synth.device_columns_to_batch(wl['asm'], {k: v[:n] for k, v in wl['cols'].items()}, int(wl['spec'].read_len))
    base, _ = cpu_baseline(head, wl['table'], wl['lib'], n)
    base['sample'] = 'rank 0, library 1: ' + base['sample']
    cal = reference_calibration()
    if cal:
        base['reference_calibration'] = cal
        base['reference_equivalent_value'] = base['value'] / cal['port_over_reference']
    return base


def overlapped_throughput(runner, wl, device, in_flight, steps):
    """Independent library passes on separate HIP streams (each with its own builder and workspace): the small
    latency-bound kernels of one pass run under the streaming kernel of another.  Reported next to the headline
    figure, never as it: with passes sharing the chip the per-launch duration of stream_kernel no longer measures
    the kernel, so the roofline object always comes from the one-pass-at-a-time region."""
    import torch
    from besst_amd import pipeline
    pool = pipeline.PassPool(device, wl['asm'].nc, wl['node_bits'], wl['lib'], runner.rec.n, runner.cap, in_flight)
    pool.set_contigs(**wl['table'])
    torch.cuda.synchronize()
    recs = runner.recs
    builders = pool.builders

    def run(k):
        for i in range(k):
            pool.submit(recs[i % len(recs)])
    run(2 * in_flight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sizes = [gb.read_sizes() for gb in builders]
    ok = all(sz == runner.sizes() for sz in sizes)
    pairs = runner.rec.n // 2
    return {'in_flight': in_flight, 'steps': steps, 'ms_per_step': round(dt / steps * 1e3, 5),
            'value': pairs / (dt / steps), 'unit': 'read-pairs/s', 'edge_tables_match_single_pass': bool(ok)}


class SingleGpu(object):
    def __init__(self, device, wl, copies=1):
        from besst_amd import pipeline
        if 'cols' in wl:
            self.recs = [pipeline.DeviceRecords.from_columns(wl['cols'], copy=(k > 0)) for k in range(max(1, copies))]
        else:
            self.recs = [pipeline.DeviceRecords(wl['batch'], device) for _ in range(max(1, copies))]
        self.rec = self.recs[0]
        self.i = 0
        # tuple capacity: every record may emit one tuple; sized down after the first measured pass
        probe = pipeline.DeviceGraphBuilder(device, wl['asm'].nc, wl['node_bits'], wl['lib'], self.rec.n, 1)
        probe.set_contigs(**wl['table'])
        probe.reset()
        probe.classify(self.rec)
        n_tuples, _ = probe.read_sizes()
        del probe
        self.cap = int(n_tuples * 1.25) + 4096
        self.gb = pipeline.DeviceGraphBuilder(device, wl['asm'].nc, wl['node_bits'], wl['lib'], self.rec.n, self.cap)
        self.gb.set_contigs(**wl['table'])

    def step(self):
        self.gb.step(self.recs[self.i % len(self.recs)])
        self.i += 1

    def sizes(self):
        return self.gb.read_sizes()

    def check_capacity(self):
        n, _ = self.gb.read_sizes()
        if n > self.cap:
            raise SystemExit('tuple capacity exceeded')


if __name__ == '__main__':
    main()
