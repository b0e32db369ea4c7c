"""Multi-GPU scaffold-graph build: one process per GPU, RCCL over xGMI.

Partitioning
  phase 1 (stream order)  rank r owns the r-th contiguous slice of the (tid,pos)-sorted record stream and
                          runs the per-record kernel on it.  CreateEdge's duplicate rule compares a record
                          with the previous record that reached CreateEdge *anywhere earlier in the stream*
                          (CreateGraph.py:835-838,869-870).  Default ('exchange'): a slice leaves its first
                          reaching record unresolved and describes it in the headers of the phase-2 regions;
                          the owners resolve those heads, so nothing is exchanged before the emit stage.
                          Other modes: every rank publishes the last such observation of its slice (16
                          bytes) and picks its incoming prev_obs from the all-gathered tails.
  phase 2 (key owners)    every link/fishy tuple is routed to owner = hash(min scaffold of the key) mod W with
                          ONE equal-split all-to-all of fixed-capacity regions (count in the region header, so
                          no size exchange and no host round trip).  A stable partition on the sender plus
                          source-rank-ordered regions on the receiver keep the global BAM order, hence per-edge
                          observation order and first-occurrence order survive the exchange.
  phase 3                 each rank sorts and reduces the keys it owns; coverage numerators and counters are
                          all-reduced (sum).  Edge scoring is embarrassingly parallel per owner.

xGMI is point-to-point, so the all-to-all puts each (src,dst) region on its own link; the only ring-style
collectives are the tiny tail all-gather and the coverage/counter all-reduces.

The kernel stages sit behind a small backend interface: ``HipBackend`` (the product: hand-written HIP through
the besst_dev_* C ABI on torch-owned HBM).  CPU tests inject an oracle-backed stand-in over gloo to exercise
exactly this file's orchestration with world_size 2.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def _host_staged(t, group):
    """True when a device tensor has to travel over a gloo group: the two-ranks-on-one-GPU check (RCCL refuses two
    ranks on one device; tests/test_gpu_two_ranks.py, ``BESST_DIST_BACKEND=gloo python bench.py``).  The collectives
    then go through host copies - a debugging transport, never what a multi-GPU node runs."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


class _Done(object):
    def wait(self):
        return True


def _all_gather_into(out, x, group):
    if _host_staged(x, group):
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, x.cpu(), group=group)
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, x, group=group)


def _all_to_all(out, x, group):
    if _host_staged(x, group):
        host = torch.empty(x.shape, dtype=x.dtype)
        dist.all_to_all_single(host, x.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, x, group=group)


def _all_reduce(x, group, op=dist.ReduceOp.SUM, async_op=False):
    if _host_staged(x, group):
        host = x.cpu()
        dist.all_reduce(host, op=op, group=group)
        x.copy_(host)
        return _Done()
    return dist.all_reduce(x, op=op, group=group, async_op=async_op)


RIDER_MAX_BYTES = 256 * 1024
RECORD_BYTES = 25                # tid mtid pos mpos tlen i32, flag qlen u16, mapq u8


def memory_budget(n_records, n_contigs, world, pair_capacity, tuple_capacity=None, coverage_mode='auto'):
    """HBM bytes one rank needs for one library of a sharded build: what DeviceRecords, HipBackend and the exchange of
    ShardedGraphBuild allocate, item by item, from the library's own workspace-size functions (no GPU needed).
    bench.py --gpus N checks the total of its libraries against the free HBM before it allocates anything, and DESIGN.md
    section 5 carries the table for BASELINE.json configs[3] / configs[4] on eight GPUs (tools/memory_budget.py)."""
    from . import pipeline
    lib = _lib.load()
    n_records, n_contigs, world = int(n_records), int(n_contigs), int(world)
    pair_cap = int(pair_capacity + (pair_capacity & 1))
    recv_cap = pair_cap * world
    part_cap = min(int(tuple_capacity), n_records) if tuple_capacity else n_records
    sum_bytes = (n_contigs + 8) * 8
    rider = sum_bytes if coverage_mode == 'rider' or (coverage_mode == 'auto' and sum_bytes <= RIDER_MAX_BYTES) else 0
    region = int(lib.besst_dev_exchange_stride_bytes(pair_cap, rider))
    rec_cap = max(1, n_records)
    items = {
        'records (8 columns, resident between get_metrics and PE)': n_records * RECORD_BYTES,
        'contig table + coverage / counter state': int(lib.besst_dev_contig_table_bytes(n_contigs))
                                                   + 2 * (n_contigs + (pipeline.COUNTER_BYTES + 32) // 8 + 2) * 8,
        'tuple stream of the slice (key + payload per record slot)': rec_cap * 16,
        'record-loop workspace (block segments, summaries)': int(lib.besst_dev_classify_workspace_bytes(rec_cap)),
        'partition workspace': int(lib.besst_dev_reduce_workspace_bytes(part_cap)),
        'all-to-all regions, send + receive (world x region each)': 2 * world * region,
        'received tuples (key, payload, emit index)': recv_cap * 20,
        'sort / reduce workspace of the owned tuples': int(lib.besst_dev_reduce_workspace_bytes(recv_cap)),
        'edge rows + observations of the owned tuples': recv_cap * 48,
    }
    return {'items': items, 'total': int(sum(items.values())), 'pair_capacity': pair_cap, 'region_bytes': region,
            'received_capacity': recv_cap}


class BuildInputs(object):
    """What one rank's graph build works on: the record columns of its slice in HBM (pipeline.DeviceRecords), the contig
    table columns (CreateGraph.contig_table: the same on every rank), the library constants of the record loop and the
    node width of the edge keys.  ``of()`` also takes a bench / test workload dict (workload.make)."""
    __slots__ = ('rec', 'n_contigs', 'node_bits', 'lib', 'table')

    def __init__(self, rec, n_contigs, node_bits, lib, table):
        self.rec, self.n_contigs, self.node_bits, self.lib, self.table = rec, int(n_contigs), int(node_bits), lib, table

    @classmethod
    def of(cls, src, device):
        if isinstance(src, cls):
            return src
        return cls(device_records(src, device), src['asm'].nc, src['node_bits'], src['lib'], src['table'])


def device_records(wl, device):
    """The workload's record columns in HBM, uploaded (or adopted, when they were generated on the GPU) once."""
    from . import pipeline
    rec = wl.get('_rec') if isinstance(wl, dict) else None
    if rec is None:
        rec = pipeline.DeviceRecords.from_columns(wl['cols']) if 'cols' in wl else pipeline.DeviceRecords(wl['batch'], device)
        try:
            wl['_rec'] = rec
        except TypeError:
            pass
    return rec


class SliceIngestError(_lib.BesstDeviceError):
    """A slice of the file could not be read on some rank; raised on EVERY rank in the same round of ingest_slice's check
    (``rank``: the first rank that failed, ``status``: its library status or None)."""

    def __init__(self, rank, status, text):
        _lib.BesstDeviceError.__init__(self, 'ingest_slice: rank %d: %s' % (rank, text), status=status)
        self.rank = rank


def _read_slice(bamio, path, dev, threads, rank, world, skip, chunk_blocks):
    """One attempt at a slice.  A GUESSED first record start (skip < 0, rank > 0) that leads nowhere - the walk from it runs
    into bytes that are no record - answers BESST_ERR_UNSUPPORTED: None, the caller reads again once the slice before it
    has said where the slice begins."""
    try:
        return bamio.ResidentBam(path, device_index=dev, threads=threads, part=(rank, world), first_skip=skip,
                                 chunk_blocks=chunk_blocks)
    except _lib.BesstDeviceError as e:
        if skip < 0 and rank > 0 and e.status == _lib.ERR_UNSUPPORTED:
            return None
        raise


def _first_wrong(pairs):
    """The first slice whose offset is not what the slice before it reports (None: a guess that led nowhere); slice 0 begins
    behind the header and is always right."""
    for r in range(1, len(pairs)):
        if pairs[r] is None or pairs[r - 1] is None or pairs[r][0] != pairs[r - 1][1]:
            return r
    return None


def _first_error(pairs):
    for r, p in enumerate(pairs):
        if isinstance(p, tuple) and len(p) == 3 and p[0] == 'error':
            return r, p[1], p[2]
    return None


def ingest_slice(path, rank, world, device_index=None, threads=None, gather=None, chunk_blocks=0, group=None, info=None):
    """Rank ``rank``'s slice of a BAM file for the sharded build: slice (rank, world) of the file - cut at BGZF block
    boundaries every rank finds on its own - is inflated and decoded on the rank's GPU (besst_ctx_push_bam_device_slice), and
    the resident columns are handed on where they lie (besst_ctx_record_pointers).  -> (bamio.ResidentBam, column dict for
    ``wl['cols']`` / pipeline.DeviceRecords.from_columns).  The ResidentBam owns the memory: keep it while the columns are
    in use, close() it afterwards.

    Any block layout: a record belongs to the slice it begins in, and where a slice's first record begins is the one thing a
    rank cannot know alone - it guesses, the ranks gather every slice's (offset used, bytes of the last record that lie in
    the next slice), and a rank whose offset is not what the slice before it reports reads its slice again from there
    (htslib's layout: all zeros, one round).  ``gather``: callable (rank's pair) -> list of all ranks' pairs; default
    torch.distributed.all_gather_object over ``group`` when a process group of that size is up (a single rank needs none).

    A rank whose read fails for any other reason (rank 0, out of memory, a CRC / inflate failure, a re-read that still leads
    nowhere) gathers ('error', status, text) in place of its pair: every rank sees it in the same round and raises
    SliceIngestError together - nobody is left waiting in a collective.  ``info`` (a dict): receives 'rounds' (gathers
    until the slices had settled) and 'reads' (how often this rank read its slice)."""
    from . import bamio
    dev = rank if device_index is None else device_index
    if world == 1:
        bam = bamio.ResidentBam(path, device_index=dev, threads=threads, chunk_blocks=chunk_blocks)
        return bam, bam.ctx.record_tensors()
    if gather is None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) == world):
            raise RuntimeError('ingest_slice: pass gather= or initialise a process group of %d ranks' % world)

        def gather(pair):
            out = [None] * world
            dist.all_gather_object(out, pair, group=group)
            return out
    skip, bam, failed, error = -1, None, False, None
    info = info if info is not None else {}
    info.update(rounds=0, reads=0)
    for _ in range(world + 1):
        info['rounds'] += 1
        if bam is None and not failed and error is None:
            info['reads'] += 1
            try:
                bam = _read_slice(bamio, path, dev, threads, int(rank), int(world), skip, chunk_blocks)
                failed = bam is None
            except Exception as e:                           # whatever it is, the other ranks have to hear of it
                error = ('error', getattr(e, 'status', None), '%s: %s' % (type(e).__name__, e))
        pairs = gather(error if error is not None else (tuple(bam.boundary) if bam is not None else None))
        bad = _first_error(pairs)
        if bad is not None:
            if bam is not None:
                bam.close()
            raise SliceIngestError(*bad)
        # slices 0 .. k - 1 are right when each one's offset is what the slice before it reports; the first that is not -
        # or whose guess led nowhere - reads its slice again from there (what IT reports may change with that, so the slices
        # behind it are checked again in the next round: at most world - 1 rounds, one in htslib's layout); every rank sees
        # the same pairs
        wrong = _first_wrong(pairs)
        if wrong is None:
            return bam, bam.ctx.record_tensors()
        if wrong == rank:
            if bam is not None:
                bam.close()
            bam, skip, failed = None, pairs[rank - 1][1], False
    if bam is not None:
        bam.close()
    raise SliceIngestError(-1, None, 'the slices did not settle')


def ingest_all_slices(path, world, device_index=0, threads=None, chunk_blocks=0):
    """ingest_slice for every rank of ``world`` in ONE process (simulated ranks on one GPU; tests): the same protocol - guess,
    compare every slice's offset with what the slice before reports, read the first wrong slice again - without a process
    group.  -> list of (bamio.ResidentBam, column dict), rereads (how many slices were read twice)."""
    from . import bamio

    def read(r, skip):
        return _read_slice(bamio, path, device_index, threads, r, world, skip, chunk_blocks)
    bams = [read(r, -1) for r in range(world)]
    rereads = 0
    for _ in range(world + 1):
        wrong = _first_wrong([b.boundary if b is not None else None for b in bams])
        if wrong is None:
            return [(b, b.ctx.record_tensors()) for b in bams], rereads
        skip = bams[wrong - 1].boundary[1]
        if bams[wrong] is not None:
            bams[wrong].close()
        bams[wrong] = read(wrong, skip)
        rereads += 1
    raise RuntimeError('ingest_all_slices: the slices did not settle')


class HipBackend(object):
    """Kernel stages of one rank on its GPU."""

    def __init__(self, device, inputs, rank, world, pair_capacity, tuple_capacity=None):
        """inputs: BuildInputs (or a workload dict, BuildInputs.of)."""
        from . import pipeline
        self.pipeline = pipeline
        self.lib = _lib.load()
        self.device = device
        self.rank, self.world = rank, world
        self.inputs = inp = BuildInputs.of(inputs, device)
        self.rec = inp.rec
        self.pair_cap = int(pair_capacity + (pair_capacity & 1))
        self.recv_cap = self.pair_cap * world
        self.gb = pipeline.DeviceGraphBuilder(device, inp.n_contigs, inp.node_bits, inp.lib, self.rec.n, self.recv_cap)
        self.gb.set_contigs(**inp.table)
        # coverage numerators and the 8 summable counter words are adjacent in the builder's state block, so ONE
        # in-place all-reduce sums both
        self._sum_buf = self.gb.state[:self.gb.n_contigs + 8]
        # Small assemblies send that block as a RIDER behind the tuples of every exchange region and the receivers
        # sum the riders of their sources: the all-reduce disappears from the step.  World copies of the block travel,
        # so beyond RIDER_MAX_BYTES (32 k contigs) the all-reduce moves fewer bytes and stays.
        mode = os.environ.get('BESST_COVERAGE_EXCHANGE', 'auto')          # 'auto' | 'rider' | 'allreduce'
        sum_bytes = int(self._sum_buf.numel()) * 8
        self.rider_bytes = sum_bytes if mode == 'rider' or (mode == 'auto' and sum_bytes <= RIDER_MAX_BYTES) else 0
        self.sums_ride_exchange = self.rider_bytes > 0
        self.region = self.lib.besst_dev_exchange_stride_bytes(self.pair_cap, self.rider_bytes)
        u8 = dict(dtype=torch.uint8, device=device)
        self.send = torch.zeros(world * self.region, **u8)
        # the partition reads its input speculatively up to this capacity: never beyond the emit buffers (one tuple per
        # record at most)
        self.part_cap = min(int(tuple_capacity), self.rec.n) if tuple_capacity else self.rec.n
        self.ws_part = torch.empty(self.lib.besst_dev_reduce_workspace_bytes(self.part_cap), **u8)
        self.tail = torch.zeros(4, dtype=torch.int32, device=device)
        # tail mode 'exchange': the slice head travels in the exchange headers (see ShardedGraphBuild.step)
        self.slice_info = torch.zeros(8, dtype=torch.int32, device=device)
        self.all_slice_info = torch.zeros(world * 8, dtype=torch.int32, device=device)
        self.heads_ride_exchange = False
        self._args = {}
        self.rkeys = torch.empty(self.recv_cap, dtype=torch.int64, device=device)
        self.rpayload = torch.empty(self.recv_cap, dtype=torch.int64, device=device)
        self.gidx = torch.empty(self.recv_cap, dtype=torch.int32, device=device)
        self.flags = self.gb.spare[:2]                                    # n_recv, overflow (zeroed by gb.reset)

    # -- tensors the orchestration reduces across ranks --
    @property
    def aligned(self):
        return self.gb.aligned

    @property
    def counter_words(self):
        return self.gb.small[:64].view(torch.int64)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        self.gb.reset()
        self.heads_ride_exchange = False

    # The argument lists of the stage calls never change (all buffers are allocated once), so they are marshalled
    # once: at ~230 us per step the Python side of a ctypes call with 20 pointer arguments is not free.
    def _call(self, name, fn, build, stream=None):
        args = self._args.get(name)
        if args is None:
            args = self._args[name] = build()
        _lib.check(fn(self._stream() if stream is None else C.c_void_p(stream.cuda_stream), *args), name)

    def classify_scan(self):
        g, r, p = self.gb, self.rec, self.pipeline._p
        if 'classify_scan' not in self._args:
            g.params.record_path = g.record_path(r)          # sampled once (synchronises), before the first step
        g.params.mate_bits = getattr(r, 'mate_bits_ptr', None)
        self._call('classify_scan', self.lib.besst_dev_classify_scan, lambda: (
            r.n, p(r.tid), p(r.mtid), p(r.pos), p(r.mpos), p(r.flag), p(r.mapq), p(r.qlen),
            g.n_contigs, p(g.table), C.byref(g.params), g.node_bits, p(g.aligned), g._small(0), p(g.ws1),
            g.ws1.numel()))

    def classify_tail(self):
        g = self.gb
        self._call('classify_tail', self.lib.besst_dev_classify_tail, lambda: (
            self.rec.n, self.pipeline._p(self.tail), self.pipeline._p(g.ws1), g.ws1.numel()))
        return self.tail

    def classify_emit(self, tails):
        """tails: the gathered world x 4 int32 (contiguous); the stitch kernel picks this rank's incoming prev_obs."""
        g, p = self.gb, self.pipeline._p
        if self._args.get('tails_ptr') != tails.data_ptr():
            self._args.pop('classify_emit', None)
            self._args['tails_ptr'] = tails.data_ptr()
        self._call('classify_emit', self.lib.besst_dev_classify_emit, lambda: (
            self.rec.n, g.params.detect_duplicate, g._carry, p(g.keys), p(g.payload), g._n_out,
            g._small(0), p(g.ws1), g.ws1.numel(), g.n_contigs, p(g.table), p(g.aligned), p(tails), self.rank, None))

    def classify_emit_speculative(self):
        """Emit without knowing the slices before this one: the slice's first reaching record stays unresolved and
        is described in ``slice_info``; partition() puts that into the exchange headers and unpack() - on every
        owner - resolves the heads of all slices in stream order."""
        g, p = self.gb, self.pipeline._p
        self.heads_ride_exchange = True
        self._call('classify_emit_spec', self.lib.besst_dev_classify_emit, lambda: (
            self.rec.n, g.params.detect_duplicate, g._carry, p(g.keys), p(g.payload), g._n_out,
            g._small(0), p(g.ws1), g.ws1.numel(), g.n_contigs, p(g.table), p(g.aligned), None, self.rank,
            p(self.slice_info)))

    def partition(self):
        g, p = self.gb, self.pipeline._p
        if self.heads_ride_exchange:
            self._call('partition_spec', self.lib.besst_dev_partition, lambda: (
                self.part_cap, g._n_out, g.node_bits, self.world, p(g.keys), p(g.payload), self.pair_cap,
                p(self.send), p(self.ws_part), self.ws_part.numel(),
                p(self._sum_buf) if self.rider_bytes else None, self.rider_bytes, p(self.slice_info)))
            return self.send
        self._call('partition', self.lib.besst_dev_partition, lambda: (
            self.part_cap, g._n_out, g.node_bits, self.world, p(g.keys), p(g.payload), self.pair_cap,
            p(self.send), p(self.ws_part), self.ws_part.numel(),
            p(self._sum_buf) if self.rider_bytes else None, self.rider_bytes, None))
        return self.send

    def unpack(self, recv):
        p = self.pipeline._p
        if self._args.get('recv_ptr') != recv.data_ptr():
            self._args.pop('unpack', None)
            self._args.pop('unpack_spec', None)
            self._args['recv_ptr'] = recv.data_ptr()
        if self.heads_ride_exchange:
            g = self.gb
            self._call('unpack_spec', self.lib.besst_dev_unpack, lambda: (
                self.world, self.pair_cap, p(recv), p(self.rkeys), p(self.rpayload), p(self.gidx),
                C.c_void_p(self.flags.data_ptr()), C.c_void_p(self.flags.data_ptr() + 4),
                p(self._sum_buf) if self.rider_bytes else None, self.rider_bytes, 1, self.rank,
                g.params.detect_duplicate, p(self.all_slice_info), g._small(0)))
            return
        self._call('unpack', self.lib.besst_dev_unpack, lambda: (
            self.world, self.pair_cap, p(recv), p(self.rkeys), p(self.rpayload), p(self.gidx),
            C.c_void_p(self.flags.data_ptr()), C.c_void_p(self.flags.data_ptr() + 4),
            p(self._sum_buf) if self.rider_bytes else None, self.rider_bytes, 0, self.rank, 0, None, None))

    def reduce(self):
        g, p = self.gb, self.pipeline._p
        args = self._args.get('dev_reduce')
        if args is None:
            args = self._args['dev_reduce'] = (
                self.recv_cap, C.c_void_p(self.flags.data_ptr()), g.key_bits, p(self.rkeys), p(self.rpayload),
                p(g.row_key), p(g.row_mask), p(g.row_n), p(g.row_sum), p(g.row_sum_sq), p(g.row_first), p(g.row_offset),
                p(g.obs_lo), p(g.obs_hi), g._n_rows, p(g.ws2), g.ws2.numel(), p(self.gidx), g.key_base)

        def again():        # (g.read_sizes() repeats the call with BESST_REDUCE_NO_RUNS when the run-grouped form overflows)
            _lib.check(self.lib.besst_dev_reduce_flags(self._stream(), *args, g.sort_flags), 'dev_reduce')
        g._redo = again
        again()

    def pack_for_allreduce(self):
        return self._sum_buf

    def unpack_after_allreduce(self):
        pass

    def overflowed(self):
        return bool(self.flags.cpu()[1].item())

    def sizes(self):
        """(tuples received, edge rows owned) - synchronises."""
        _, rows = self.gb.read_sizes()
        return int(self.flags.cpu()[0].item()), rows

    def local_table(self):
        """This rank's owned edge rows as a host EdgeTable (first_idx = global emit index)."""
        from .device import EdgeTable
        L, r = self.sizes()
        g = self.gb

        def h(t, n, dt):
            return t[:n].cpu().numpy().view(dt)
        return EdgeTable(h(g.row_key, r, np.uint64), h(g.row_mask, r, np.uint32), h(g.row_n, r, np.uint32),
                         h(g.row_sum, r, np.int64), h(g.row_sum_sq, r, np.int64), h(g.row_first, r, np.uint32),
                         h(g.row_offset, r, np.uint32), g.node_bits, h(g.obs_lo, L, np.int32),
                         h(g.obs_hi, L, np.int32))


class ShardedGraphBuild(object):
    """Orchestrates one sharded graph-build step; ``backend`` supplies the per-rank kernel stages."""

    def __init__(self, device, inputs, rank, world, backend=None, group=None, pair_capacity=None):
        """inputs: BuildInputs or a workload dict (unused when a backend is handed in)."""
        self.rank, self.world, self.group = rank, world, group
        if backend is None:
            inputs = BuildInputs.of(inputs, device)
            tuple_capacity = None
            if pair_capacity is None:
                pair_capacity, tuple_capacity = self._probe_pair_capacity(device, inputs, world, group)
            backend = HipBackend(device, inputs, rank, world, pair_capacity, tuple_capacity)
        self.backend = backend
        self._tails = None
        # How a slice learns the duplicate chain's state at its first record (see step()): 'exchange' (default) - it does
        # not, the heads are resolved by the owners after the all-to-all; 'late' - an all-gather of the slices' 16-byte
        # tails between the per-record pass and the emit stage (the fallback: one more collective on the critical path,
        # nothing speculative).  Two more variants of round 2 - a backward search for the tail run before or beside the
        # per-record pass - were never measurable across devices and have been removed.
        self.tail_mode = os.environ.get('BESST_TAIL_MODE', 'exchange')   # 'exchange' | 'late'
        if self.tail_mode not in ('exchange', 'late'):
            raise ValueError("BESST_TAIL_MODE must be 'exchange' or 'late'")
        self._recv = None
        # BESST_ALLREDUCE_ASYNC=1: the coverage/counter all-reduce overlaps the tuple exchange and the sort on its own
        # communicator (so that it is not serialised behind the all-to-all); BESST_SIDE_GROUP=0 keeps even that on
        # the default communicator.  See step() for why the default is the plain in-order all-reduce.
        self.allreduce_async = os.environ.get('BESST_ALLREDUCE_ASYNC', '0') == '1'
        want_side = (dist.is_initialized() and group is None and os.environ.get('BESST_SIDE_GROUP', '1') != '0'
                     and self.allreduce_async and not getattr(backend, 'sums_ride_exchange', False))
        self.side_group = dist.new_group() if want_side else group

    @staticmethod
    def probe_tuples(device, inputs):
        """Tuples the slice emits: one untimed local pass of the record loop, no collective."""
        from . import pipeline
        inp = BuildInputs.of(inputs, device)
        probe = pipeline.DeviceGraphBuilder(device, inp.n_contigs, inp.node_bits, inp.lib, inp.rec.n, 1)
        probe.set_contigs(**inp.table)
        probe.reset()
        probe.classify(inp.rec)
        n_out, _ = probe.read_sizes()
        return int(n_out)

    @staticmethod
    def _probe_pair_capacity(device, inputs, world, group=None):
        """Size the exchange regions (tuples per (src,dst) pair, 1.5x slack) from the probe pass, the largest over ranks."""
        n_out = ShardedGraphBuild.probe_tuples(device, inputs)
        cap = torch.tensor([int(n_out * 1.5 / world) + 4096], dtype=torch.int64, device=device)
        if dist.is_initialized():
            _all_reduce(cap, group, op=dist.ReduceOp.MAX)        # (the build's own group: a group of one must not wait for the world)
        return int(cap.item()), int(n_out * 1.25) + 4096

    def step(self):
        b = self.backend
        b.reset()
        device_backend = hasattr(b, 'classify_emit_speculative')
        mode = self.tail_mode if device_backend else 'late'
        if mode == 'exchange':
            # No tail exchange at all: every slice emits with its first reaching record unresolved and describes it
            # in the headers of the all-to-all regions; the owners replay the chain over the slices, resolve the heads,
            # drop a head's tuple where it was a duplicate and correct the summed counters (unpack_kernel).  The step
            # is then one collective (plus the all-reduce of large assemblies).
            b.classify_scan()
            b.classify_emit_speculative()
            tails = None
        elif device_backend:
            # 'late': the tail comes from the per-record pass's block summaries, the gather sits on the critical path
            if self._tails is None:
                self._tails = torch.zeros(self.world * 4, dtype=torch.int32, device=b.device)
            b.classify_scan()
            _all_gather_into(self._tails, b.classify_tail(), self.group)
            tails = self._tails
        else:                                            # CPU stand-in backends (gloo has no all_gather_into_tensor)
            b.classify_scan()
            tail = b.classify_tail()
            gathered = [torch.empty_like(tail) for _ in range(self.world)]
            dist.all_gather(gathered, tail, group=self.group)
            self._tails = tails = torch.cat(gathered)
        if mode != 'exchange':
            b.classify_emit(tails)
        # Coverage numerators and counters are final here.  Default: a plain all-reduce in stream order on the main
        # communicator.  BESST_ALLREDUCE_ASYNC=1 issues it asynchronously on a second communicator instead, so that
        # it overlaps the tuple exchange and the sort; with one rank over RCCL that was 17 us SLOWER per step (192
        # vs 175 us: the event hand-overs between the streams cost more than the 80 KB all-reduce), and issuing it
        # after the sort from a side stream another 28 us slower, so overlap stays opt-in until it can be measured
        # across xGMI.
        if getattr(b, 'sums_ride_exchange', False):
            summed = _Done()                       # the block rides the all-to-all (HipBackend.rider_bytes)
        elif self.allreduce_async:
            summed = _all_reduce(b.pack_for_allreduce(), self.side_group, async_op=True)
        else:
            _all_reduce(b.pack_for_allreduce(), self.group)
            summed = _Done()
        send = b.partition()
        if self._recv is None:
            self._recv = torch.empty_like(send)
        _all_to_all(self._recv, send, self.group)
        b.unpack(self._recv)
        b.reduce()
        summed.wait()
        b.unpack_after_allreduce()

    def check_capacity(self, grow=True):
        """The exchange regions have a fixed capacity (sized from a probe pass with 1.5x slack).  If any rank
        truncated a region in the last step, every rank learns it (all-reduce of the flag) and - with ``grow`` -
        rebuilds its buffers with twice the capacity and repeats the step, so that a skewed owner distribution
        costs a re-run of one step instead of the job.  A rank that cannot read its flag, or cannot allocate the doubled
        regions, says so in the same all-reduce: every rank raises RankFailure together and nobody enters the repeated
        step's all-to-all alone."""
        dev = self.backend.aligned.device

        def agree(over, failed):
            flag = torch.tensor([1 if over else 0, 1 if failed else 0], dtype=torch.int32, device=dev)
            if dist.is_initialized():
                _all_reduce(flag, self.group, op=dist.ReduceOp.MAX)
            over, failed = (int(x) for x in flag.cpu().tolist())
            return bool(over), bool(failed)
        for _ in range(6):
            mine, why = False, None
            try:
                mine = bool(self.backend.overflowed())
            except Exception as e:                           # noqa: BLE001 - reported to every rank below
                why = '%s: %s' % (type(e).__name__, e)
            over, failed = agree(mine, why is not None)
            if failed:
                raise _lib.RankFailure('exchange capacity check failed on some rank' + (' (here: %s)' % why if why else ''))
            if not over:
                return
            if not grow or not isinstance(self.backend, HipBackend):
                raise _lib.BesstDeviceError('exchange region overflow: raise pair_capacity')
            old = self.backend
            args = (old.device, old.inputs, self.rank, self.world, old.pair_cap * 2, old.part_cap)
            self.backend = old = None                        # (free the old regions before the doubled ones are made)
            try:
                self.backend = HipBackend(*args)
            except Exception as e:                           # noqa: BLE001
                why = '%s: %s' % (type(e).__name__, e)
            _, failed = agree(False, why is not None)
            if failed:
                raise _lib.RankFailure('growing the exchange regions to %d pairs failed on some rank%s'
                                       % (args[4], ' (here: %s)' % why if why else ''))
            self._tails = None
            self._recv = None
            self.step()
        raise _lib.BesstDeviceError('exchange region overflow persists after growing the regions 64x')

    def sizes(self):
        """Global (tuples, edge rows) summed over ranks - synchronises."""
        n, r = self.backend.sizes()
        t = torch.tensor([n, r], dtype=torch.int64, device=self.backend.aligned.device)
        _all_reduce(t, self.group)
        return int(t[0].item()), int(t[1].item())

    def gather_edges(self, dst=0):
        """Final gather of the owned edge rows to rank `dst`: returns the list of every rank's
        ``backend.local_table()`` there (keys are disjoint: each key has one owner), None elsewhere.  Host side,
        once per library; the rows' ``first_idx`` is the global emit index, so sorting the union by it restores the
        reference's first-occurrence order."""
        mine = self.backend.local_table()
        out = [None] * self.world if self.rank == dst else None
        dist.gather_object(mine, out, dst=dst, group=self.group)
        return out

    def final_prev_obs(self):
        """counter.prev_obs1/2 after the last record of the global stream."""
        if getattr(self.backend, 'heads_ride_exchange', False):
            tails = self.backend.all_slice_info.cpu().numpy().reshape(self.world, 8)[:, :3]
        else:
            tails = self._tails.cpu().numpy().reshape(self.world, 4)
        prev = (-1, -1)
        for j in range(self.world):
            if tails[j, 0]:
                prev = (int(tails[j, 1]), int(tails[j, 2]))
        return prev


class ShardedMetricsSample(object):
    """libmetrics' three BAM scans over a stream sharded in contiguous slices.

    The reference takes the FIRST 1,000,000 qualifying observations in stream order (libmetrics.py:83,302), so a
    slice has to know how many came before it: every rank counts its slice, the counts are all-gathered (3 int64
    per rank), and each rank then writes its samples at their global positions in zeroed 1,000,000-entry buffers.
    Positions are disjoint across ranks, hence ONE all-reduce(sum) of the buffers is the ordered sample of the
    whole stream on every rank, and a second tiny one sums counter_total / n_contam.  ``backend`` supplies
    ``count() -> int64[3]`` and ``emit(before) -> (samples int32[2e6], state int64[8])``
    (pipeline.DeviceMetricsSampler on the GPU; an oracle-backed stand-in in the gloo tests)."""

    def __init__(self, backend, rank, world, group=None):
        self.backend, self.rank, self.world, self.group = backend, rank, world, group

    def sample(self, orientation, min_mapq, read_len, want_isize=True):
        from .pipeline import SAMPLE_CAP
        b = self.backend
        local = b.count(orientation, min_mapq, read_len).clone()
        if _host_staged(local, self.group):
            counts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(self.world)]
            dist.all_gather(counts, local.cpu(), group=self.group)
            counts = [c.to(local.device) for c in counts]
        else:
            counts = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(counts, local, group=self.group)
        before = torch.zeros_like(local)
        for r in range(self.rank):
            before += counts[r]
        samples, state = b.emit(before, orientation, min_mapq, read_len, want_isize)
        _all_reduce(samples, self.group)
        tot = state[3:6].clone()
        _all_reduce(tot, self.group)
        total = torch.stack(counts).sum(0).cpu().numpy()
        tot = tot.cpu().numpy()
        n_isize = int(min(total[0], SAMPLE_CAP)) if want_isize else 0
        n_contam = int(tot[1])
        host = samples.cpu().numpy()
        return (host[:n_isize].copy(), host[SAMPLE_CAP:SAMPLE_CAP + n_contam].copy(),
                dict(n_isize=n_isize, n_contam=n_contam, counter_total=int(tot[0]),
                     sample_counter=int(min(total[1], SAMPLE_CAP)), records_scanned=int(tot[2])))
