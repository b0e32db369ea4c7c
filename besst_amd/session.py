"""Per-library device session: one alignment stream resident in HBM.

The reference hands the same open ``pysam.Samfile`` first to ``libmetrics.get_metrics`` and then to
``CreateGraph.PE`` (runBESST:162-182) and re-reads the file up to four times.  Here the stream is
uploaded once (flat SoA columns) and both entry points work on the resident copy; the session is
cached on the ``bam_file`` object so the second call finds it.
"""
import weakref

from . import device
from .records import RecordBatch

_sessions = weakref.WeakKeyDictionary()
# bam_file objects that cannot be weakly referenced or hashed (some pysam handles): keyed by id(); such a session lives
# until close_session(bam_file) is called (the CLI does so after CreateGraph.PE)
_sessions_by_id = {}


class Session(object):
    is_follower = False                     # (sharded.ShardedSession: every rank but 0 follows rank 0 through CreateGraph.PE)

    def __init__(self, batch, device_index=0):
        self.batch = batch
        self.ctx = device.GraphContext(device_index)
        # reference count defines the tid range; classes are filled in by CreateGraph.PE
        n = len(batch.references)
        zeros = [0] * n
        self.ctx.set_contigs(scaf_id=zeros, scaf_len=zeros, ctg_pos=zeros, ctg_len=zeros, direction=zeros, cls=zeros)
        self.ctx.push_records(batch)

    @classmethod
    def resident(cls, bam):
        """Session over a bamio.ResidentBam: its records are in HBM already, in its own context."""
        self = cls.__new__(cls)
        self.batch = bam
        self.ctx = bam.ctx
        return self

    def metrics_sample(self, top_mask, orientation, min_mapq, read_len, want_isize):
        return self.ctx.metrics_sample(top_mask, orientation, min_mapq, read_len, want_isize)

    def stream_order(self):
        """Index of the first record that breaks the coordinate order, or None (libmetrics.get_metrics' input guard)."""
        return self.ctx.stream_order()[0]

    # CreateGraph.PE's epilogue: nothing to tell anybody on one GPU
    def done(self, param):
        pass

    def abort(self, exc):
        pass

    def close(self):
        self.ctx.close()


def open_session(bam_file, device_index=0):
    """Session for a RecordBatch or any pysam-like object (materialised once, slow host loop)."""
    try:
        sess = _sessions.get(bam_file)
    except TypeError:
        sess = None
    if sess is None:
        entry = _sessions_by_id.get(id(bam_file))
        if entry is not None and entry[0] is bam_file:
            sess = entry[1]
    if sess is not None:
        return sess
    from . import sharded
    if hasattr(bam_file, 'engine') and hasattr(bam_file, 'head'):         # bamio.ShardedBam: this rank's slice is in HBM
        sess = sharded.session_for_bam(bam_file)
    elif hasattr(bam_file, 'ingest') and hasattr(bam_file, 'ctx'):        # bamio.ResidentBam
        sess = Session.resident(bam_file)
    elif sharded.active_group() is not None:
        # a stream every rank holds, under a process group: rank r works on the r-th contiguous slice of it
        rank, world = sharded.active_group()
        sess = sharded.session_for_batch(RecordBatch.from_pysam_like(bam_file), rank, world, sharded.PROCESS_GROUP)
    else:
        sess = Session(RecordBatch.from_pysam_like(bam_file), device_index)
    try:
        _sessions[bam_file] = sess
    except TypeError:
        _sessions_by_id[id(bam_file)] = (bam_file, sess)      # the strong reference keeps id() from being reused
    return sess


def close_session(bam_file):
    try:
        sess = _sessions.pop(bam_file, None)
    except TypeError:
        sess = None
    if sess is None:
        entry = _sessions_by_id.pop(id(bam_file), None)
        sess = entry[1] if entry is not None and entry[0] is bam_file else None
    if sess is not None:
        sess.close()
