"""Synthetic workloads for bench.py and the full-size property tests.

A workload = one library of one BASELINE.json config: the record stream, the contig table as
InitializeObjects would build it for a first library (CreateGraph.py:729-786: scaffold ids are a
running index in header order, contigs >= contig_threshold are 'large', the rest 'small') and the
per-library constants the record loop needs.
"""
import numpy as np

from . import synth


def first_library_table(lengths, contig_threshold, first_scaffold_id=1):
    lengths = np.asarray(lengths, dtype=np.int64)
    nc = lengths.shape[0]
    present = lengths > 0
    scaf = np.zeros(nc, dtype=np.int64)
    scaf[present] = first_scaffold_id + np.arange(int(present.sum()))
    cls = np.where(lengths >= contig_threshold, 1, np.where(present, 2, 0))
    return dict(scaf_id=scaf.astype(np.int32), scaf_len=lengths.astype(np.int32), ctg_pos=np.zeros(nc, dtype=np.int32),
                ctg_len=lengths.astype(np.int32), direction=np.ones(nc, dtype=np.uint8), cls=cls.astype(np.uint8))


def node_bits_for(table):
    max_id = int(table['scaf_id'].max()) if len(table['scaf_id']) else 1
    return max(1, int(max_id * 2 + 1).bit_length())


def make(config='C2', lib_index=0, pairs=None, nc=None, seed_offset=0, tid_offset=0, reads_seed_offset=0):
    """seed_offset: a different assembly AND different reads; reads_seed_offset: the same assembly (contig table),
    an independent set of read pairs - what rank r of a sharded run holds: its slice of one library's stream."""
    cfg = synth.CONFIGS[config]
    spec = cfg['libs'][lib_index]
    n_pairs = int(pairs if pairs is not None else cfg['pairs'] // len(cfg['libs']))
    n_ctg = int(nc if nc is not None else cfg['nc'])
    seed = synth.config_seed(config) + 1000 * seed_offset
    asm = synth.make_assembly(n_ctg, cfg['median'], seed)
    batch = synth.simulate_library(asm, spec, n_pairs, seed + 100 + lib_index + 7919 * reads_seed_offset)
    lib = dict(read_len=float(spec.read_len), ins_size_threshold=spec.mean + 6 * spec.sd, min_mapq=11,
               orientation=spec.orientation, detect_duplicate=True, extend_paths=True, no_score=False,
               mean=spec.mean, sd=spec.sd)
    table = first_library_table(asm.lengths, spec.mean + 4 * spec.sd)
    return dict(config=config, asm=asm, batch=batch, table=table, lib=lib, node_bits=node_bits_for(table),
                pairs=n_pairs, spec=spec)


class DeviceWorkload(dict):
    """A workload whose record columns were drawn on the GPU (synth.simulate_library_device); the host RecordBatch
    (wl['batch'], what the oracles read) is only materialised when somebody asks for it."""

    def __missing__(self, key):
        if key == 'batch':
            self['batch'] = synth.device_columns_to_batch(self['asm'], self['cols'], int(self['spec'].read_len))
            return self['batch']
        raise KeyError(key)


def make_device(device, config='C2', lib_index=0, pairs=None, nc=None, seed_offset=0, reads_seed_offset=0,
                chimeric_frac=None, order='coordinate'):
    """``make`` with the records generated on ``device`` (a torch device): the full-size mate-pair configs take
    seconds instead of minutes.  Same assembly, table and library constants as ``make``; the read pairs come from
    torch's generator instead of numpy's, so the two streams are different samples of the same model.  chimeric_frac /
    order: the variants of a config that bench.py's `robustness` object measures (chimeric pairs that put links on
    edges of their own; a name-sorted stream)."""
    import copy
    cfg = synth.CONFIGS[config]
    spec = cfg['libs'][lib_index]
    if chimeric_frac is not None:
        spec = copy.copy(spec)
        spec.chimeric_frac = float(chimeric_frac)
    n_pairs = int(pairs if pairs is not None else cfg['pairs'] // len(cfg['libs']))
    n_ctg = int(nc if nc is not None else cfg['nc'])
    seed = synth.config_seed(config) + 1000 * seed_offset
    asm = synth.make_assembly(n_ctg, cfg['median'], seed)
    cols = synth.simulate_library_device(asm, spec, n_pairs, seed + 100 + lib_index + 7919 * reads_seed_offset, device,
                                         order=order)
    lib = dict(read_len=float(spec.read_len), ins_size_threshold=spec.mean + 6 * spec.sd, min_mapq=11,
               orientation=spec.orientation, detect_duplicate=True, extend_paths=True, no_score=False,
               mean=spec.mean, sd=spec.sd)
    table = first_library_table(asm.lengths, spec.mean + 4 * spec.sd)
    return DeviceWorkload(config=config, asm=asm, cols=cols, table=table, lib=lib, node_bits=node_bits_for(table),
                          pairs=n_pairs, spec=spec)


def make_device_windowed(device, config, lib_index, windows=8, pairs=None, table='later'):
    """One library of a config too large to draw in one piece (a C5 library: 1.33 G pairs = 2.67 G records - more than
    2^31 elements in any of the generator's tensors): the genome is cut into `windows` stretches, each stretch's records are
    drawn and sorted on the GPU (synth.simulate_library_device(window=...)) and written into their place in the
    preallocated columns; the concatenation is ONE coordinate-sorted stream.  table: 'first' (InitializeObjects' table) or
    'later' (the state a previous pass leaves: scaffold ids counting on from nc * lib_index + 1, MakeScaffolds.py:276)."""
    import torch
    cfg = synth.CONFIGS[config]
    spec = cfg['libs'][lib_index]
    n_pairs = int(pairs if pairs is not None else cfg['pairs'] // len(cfg['libs']))
    seed = synth.config_seed(config)
    asm = synth.make_assembly(cfg['nc'], cfg['median'], seed)
    per = [n_pairs // windows + (1 if k < n_pairs % windows else 0) for k in range(windows)]
    n_rec = 2 * n_pairs
    dt = dict(tid=torch.int32, mtid=torch.int32, pos=torch.int32, mpos=torch.int32, tlen=torch.int32, flag=torch.int16,
              mapq=torch.uint8, qlen=torch.int16)
    cols = {k: torch.empty(n_rec + 64 * windows, dtype=v, device=device) for k, v in dt.items()}
    at = 0
    for k in range(windows):
        part = synth.simulate_library_device(asm, spec, per[k], seed + 100 + lib_index + 7919 * k, device, window=(k, windows))
        m = int(part['tid'].shape[0])
        for name in dt:
            cols[name][at:at + m] = part[name]
        at += m
        del part
        torch.cuda.empty_cache()
    # (a stretch holds 2 x its pairs records less one per chunk of the generator at most: a chunk's odd record count is
    # rounded up to whole pairs)
    assert n_rec - 64 * windows <= at <= n_rec
    cols = {k: v[:at] for k, v in cols.items()}
    thr = spec.mean + 4 * spec.sd
    tab = first_library_table(asm.lengths, thr) if table == 'first' else \
        later_library_table(asm, seed + 50 + lib_index, thr, first_scaffold_id=asm.nc * lib_index + 1)
    return DeviceWorkload(config=config, asm=asm, cols=cols, table=tab, lib=library_constants(spec),
                          node_bits=node_bits_for(tab), pairs=n_pairs, spec=spec)


def later_library_table(asm, seed, contig_threshold, max_run=5, first_scaffold_id=1):
    """Contig table of a library >= 2: the state a previous pass leaves behind (MakeScaffolds.py:362-414) - random runs
    of 1..max_run adjacent contigs chained into scaffolds with random per-contig direction, cumulative position (+ true
    gap) and scaffold length - classified by CleanObjects' rule for the new library (CreateGraph.py:788-810:
    scaffolds shorter than the new contig_threshold become small).  Vectorised form of synth.chain_scaffolds (same
    model, its own random stream), for the 500 k - 2 M contig configs."""
    rng = np.random.default_rng(seed)
    nc = asm.nc
    runs = rng.integers(1, max_run + 1, nc)                       # more than enough runs
    ends = np.cumsum(runs)
    k = int(np.searchsorted(ends, nc, side='left')) + 1
    runs = runs[:k].copy()
    runs[-1] -= int(ends[k - 1] - nc)
    scaf_index = np.repeat(np.arange(k), runs)
    first = np.concatenate(([0], np.cumsum(runs)[:-1]))           # first contig of every scaffold
    step = asm.lengths + asm.gaps                                 # contig + the gap behind it
    excl = np.concatenate(([0], np.cumsum(step)[:-1]))
    position = excl - excl[first][scaf_index]
    last = first + runs - 1
    scaf_len = (position[last] + asm.lengths[last])[scaf_index]
    direction = rng.random(nc) < 0.5
    cls = np.where(scaf_len >= contig_threshold, 1, 2)
    table = dict(scaf_id=(first_scaffold_id + scaf_index).astype(np.int32), scaf_len=scaf_len.astype(np.int32),
                 ctg_pos=position.astype(np.int32), ctg_len=asm.lengths.astype(np.int32),
                 direction=direction.astype(np.uint8), cls=cls.astype(np.uint8))
    return table


def library_constants(spec):
    return dict(read_len=float(spec.read_len), ins_size_threshold=spec.mean + 6 * spec.sd, min_mapq=11,
                orientation=spec.orientation, detect_duplicate=True, extend_paths=True, no_score=False,
                mean=spec.mean, sd=spec.sd)
