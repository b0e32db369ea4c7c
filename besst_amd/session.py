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


class Session(object):
    def __init__(self, batch, device_index=0):
        self.batch = batch
        self.ctx = device.GraphContext(device_index)
        # reference count defines the tid range; classes are filled in by CreateGraph.PE
        n = len(batch.references)
        zeros = [0] * n
        self.ctx.set_contigs(scaf_id=zeros, scaf_len=zeros, ctg_pos=zeros, ctg_len=zeros, direction=zeros, cls=zeros)
        self.ctx.push_records(batch)

    def metrics_sample(self, top_mask, orientation, min_mapq, read_len, want_isize):
        return self.ctx.metrics_sample(top_mask, orientation, min_mapq, read_len, want_isize)

    def close(self):
        self.ctx.close()


def open_session(bam_file, device_index=0):
    """Session for a RecordBatch or any pysam-like object (materialised once, slow host loop)."""
    try:
        sess = _sessions.get(bam_file)
    except TypeError:
        sess = None
    if sess is not None:
        return sess
    batch = RecordBatch.from_pysam_like(bam_file)
    sess = Session(batch, device_index)
    try:
        _sessions[bam_file] = sess
    except TypeError:
        pass
    return sess


def close_session(bam_file):
    sess = _sessions.pop(bam_file, None)
    if sess is not None:
        sess.close()
