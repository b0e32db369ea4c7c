"""The input guard's pass (besst_ctx_stream_order / besst_dev_stream_order, csrc/metrics.hip): is the resident stream sorted by
(reference id, position), reference -1 last?  Against the numpy statement of the same rule (tests/fake_device.stream_order_of)."""
import numpy as np
import pytest

from besst_amd import device
from besst_amd.records import RecordBatch
from tests import fake_device

pytestmark = pytest.mark.gpu


def _batch(tid, pos):
    n = len(tid)
    z = np.zeros(n, np.int32)
    return RecordBatch(['c%d' % i for i in range(8)], [1000] * 8, tid=tid, mtid=tid, pos=pos, mpos=z, tlen=z, flag=z.astype(np.uint16),
                       mapq=z.astype(np.uint8), qlen=z.astype(np.uint16))


def _sorted_stream(rng, n, unplaced=0):
    tid = np.sort(rng.integers(0, 8, n - unplaced)).astype(np.int32)
    pos = np.zeros(n - unplaced, np.int32)
    for t in range(8):
        m = tid == t
        pos[m] = np.sort(rng.integers(0, 1000, int(m.sum())))
    return np.concatenate([tid, np.full(unplaced, -1, np.int32)]), np.concatenate([pos, np.full(unplaced, -1, np.int32)])


@pytest.mark.parametrize('n', [0, 1, 2, 3, 4, 5, 7, 8, 1023, 1024, 1025, 4099, 300000])
def test_stream_order_matches_numpy(n):
    rng = np.random.default_rng(n)
    with device.GraphContext(0) as ctx:
        ctx.set_contigs(*[[0] * 8] * 6)
        cases = []
        tid, pos = _sorted_stream(rng, n, unplaced=min(n // 3, 5))
        cases.append((tid, pos))
        for k in ([1, n - 1, n // 2, 4, 5, 1024] if n > 1 else []):
            if 0 < k < n:
                t2, p2 = tid.copy(), pos.copy()
                t2[k], p2[k] = 0, -1                         # smaller than anything but an equal key in front of it
                cases.append((t2, p2))
        if n > 10:
            t2, p2 = tid.copy(), pos.copy()
            t2[3] = -1                                       # an unplaced read in the middle: what follows lies in front of it
            cases.append((t2, p2))
        for t, p in cases:
            b = _batch(t, p)
            ctx.clear_records()
            ctx.push_records(b)
            assert ctx.stream_order() == fake_device.stream_order_of(b)
        assert ctx.stream_order()[0] is not None or n <= 10


def test_unsorted_stream_warns_on_the_device_path(capsys):
    from besst_amd import libmetrics, session
    from tests import golden_util as GU
    from tests.test_gpu_dropin import make_param
    doc, batch = GU.load('fr_given')
    order = np.arange(len(batch))
    order[1000], order[5000] = order[5000], order[1000]
    shuffled = batch.take(order)
    param = make_param(doc['overrides'])
    libmetrics.get_metrics(shuffled, param, param.information_file)
    session.close_session(shuffled)
    assert param.stream_unsorted_at == 1001 and 'Need indexed bamfiles' in capsys.readouterr().err
    param = make_param(doc['overrides'])
    libmetrics.get_metrics(batch, param, param.information_file)
    session.close_session(batch)
    assert not hasattr(param, 'stream_unsorted_at')
