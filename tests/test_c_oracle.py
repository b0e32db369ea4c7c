"""The C oracle must agree with the Python oracle (itself pinned to the reference's golden vectors)."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import py_oracle as O
from tests import golden_util as GU
from tests.test_gpu_graph_build import _oracle_inputs


@pytest.mark.parametrize('name', GU.scenario_names())
def test_c_record_loop_matches_python_oracle(name):
    doc, batch = GU.load(name)
    p, rec, tab = _oracle_inputs(doc, batch)
    res = O.LoopResult(len(tab['cls']))
    res.tuples = []
    O.record_loop(rec, tab, p, res=res)
    table = dict(cls=np.array(tab['cls']), scaf_id=np.array(tab['scaf']), scaf_len=np.array(tab['slen']),
                 ctg_pos=np.array(tab['cpos']), ctg_len=np.array(tab['clen']), direction=np.array(tab['cdir']))
    nb = max(1, int(max(tab['scaf']) * 2 + 1).bit_length())
    lib = dict(read_len=p.read_len, ins_size_threshold=p.ins_size_threshold, min_mapq=p.min_mapq,
               orientation=p.orientation, detect_duplicate=p.detect_duplicate, extend_paths=p.extend_paths,
               no_score=p.no_score)
    keys, payload, aligned, ctr = CO.record_loop(batch, table, lib, nb)
    assert aligned.tolist() == res.aligned
    assert ctr.tolist() == [res.count, res.non_unique, res.non_unique_for_scaf, res.nr_of_duplicates, res.too_long,
                            res.fishy_reads, len(res.tuples), res.n_reach, res.prev[0], res.prev[1]]
    want_keys = [(((u << nb) | v) << 1) | f for (u, v, f, ou, ov, m) in res.tuples]
    want_pl = [ou | ((ov | (m << 30)) << 32) for (u, v, f, ou, ov, m) in res.tuples]
    assert keys.tolist() == want_keys and payload.tolist() == want_pl


@pytest.mark.parametrize('name', ['fr_infer', 'rf_contam', 'fr_edgecases'])
def test_c_metrics_sample_matches_reference_metrics(name):
    """Feeding the C sampler's lists through the Python finishing must reproduce the golden metrics."""
    doc, batch = GU.load(name)
    p = O.LibParams(**doc['overrides'])
    rec = GU.rec_lists(batch)
    q = O.LibParams(**doc['overrides'])
    O.get_metrics(rec, batch.lengths, q)
    top = np.zeros(len(batch.lengths), np.uint8)
    top[list(O.top_contig_indexes(list(batch.lengths)))] = 1
    isize, contam, counts = CO.metrics_sample(batch, top, q.orientation, q.min_mapq, q.read_len)
    # rebuild the python-side samples the same way the oracle does and compare list-for-list
    want_isize, want_contam = [], []
    for i in range(len(batch)):
        if top[rec['tid'][i]] if 0 <= rec['tid'][i] < len(top) else False:
            f, tl, t, m, mq = rec['flag'][i], rec['tlen'][i], rec['tid'][i], rec['mtid'][i], rec['mapq'][i]
            inn, out = O.is_innie(f, tl, t, m, mq, q.min_mapq), O.is_outie(f, tl, t, m, mq, q.min_mapq)
            if (q.orientation == 'fr' and inn) or (q.orientation == 'rf' and out):
                want_isize.append(abs(tl))
            if q.orientation == 'fr' and out and q.read_len < abs(tl) + 2 * q.read_len:
                want_contam.append(abs(tl))
            if q.orientation == 'rf' and inn and q.read_len < abs(tl):
                want_contam.append(abs(tl))
    assert isize.tolist() == want_isize[:1000000]
    assert contam.tolist() == want_contam


@pytest.mark.parametrize('config,threads', [('C2', 2), ('C2', 7), ('C3', 16)])
def test_threaded_record_loop_equals_sequential(config, threads):
    """The slice-parallel variant (bench.py's all-cores CPU baseline) walks back to the previous reaching record
    for its incoming prev_obs; tuples, coverage and counters must equal the sequential loop."""
    import numpy as np
    from besst_amd import workload
    from oracle import c_oracle as CO
    wl = workload.make(config, 0, pairs=120000, nc=500)
    want = CO.record_loop(wl['batch'], wl['table'], wl['lib'], wl['node_bits'])
    got = CO.record_loop(wl['batch'], wl['table'], wl['lib'], wl['node_bits'], threads=threads)
    assert want[3][3] > 0          # duplicates exist, so the chain matters
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
