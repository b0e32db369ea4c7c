"""Helpers shared by the GPU parity tests: run the device graph build and the oracle side by side."""
import numpy as np

from besst_amd import device


def table_columns(tab):
    return dict(scaf_id=np.asarray(tab['scaf'], dtype=np.int32), scaf_len=np.asarray(tab['slen'], dtype=np.int32),
                ctg_pos=np.asarray(tab['cpos'], dtype=np.int32), ctg_len=np.asarray(tab['clen'], dtype=np.int32),
                direction=np.asarray(tab['cdir'], dtype=np.uint8), cls=np.asarray(tab['cls'], dtype=np.uint8))


def device_build(batch, tab, p, ctx=None, chunks=1):
    own = ctx is None
    ctx = ctx or device.GraphContext(0)
    try:
        ctx.set_contigs(**table_columns(tab))
        ctx.set_library(p.read_len, p.ins_size_threshold, p.min_mapq, p.orientation, p.detect_duplicate,
                        p.extend_paths, p.no_score)
        ctx.clear_records()
        n = len(batch)
        step = (n + chunks - 1) // chunks if n else 1
        for s in range(0, n, step):
            ctx.push_records(batch.slice(s, min(n, s + step)))
        return ctx.build_graph()
    finally:
        if own:
            ctx.close()


def assert_matches_oracle(table, aligned, ctr, loop, n_contigs):
    """Bit-exact comparison of the device edge table with the oracle's LoopResult."""
    assert (ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates,
            ctr.reads_with_too_long_insert, ctr.fishy_reads) == \
        (loop.count, loop.non_unique, loop.non_unique_for_scaf, loop.nr_of_duplicates, loop.too_long,
         loop.fishy_reads)
    assert (ctr.prev_obs1, ctr.prev_obs2) == loop.prev
    assert aligned.tolist() == list(loop.aligned[:n_contigs])
    want_links = {(r.u, r.v): r for r in loop.edges.values() if r.n}
    want_fishy = {(r.u, r.v): r.fishy for r in loop.edges.values() if r.fishy}
    got_links, got_fishy = {}, {}
    keys = table.key.tolist()
    assert keys == sorted(keys) and len(set(keys)) == len(keys), 'rows must be strictly sorted by key'
    for i in range(len(table)):
        pair = (int(table.u[i]), int(table.v[i]))
        assert pair[0] < pair[1]
        if table.is_fishy[i]:
            got_fishy[pair] = int(table.n[i])
        else:
            got_links[pair] = i
    assert got_fishy == want_fishy
    assert set(got_links) == set(want_links)
    assert ctr.n_tuples == sum(r.n + r.fishy for r in loop.edges.values())
    for pair, i in got_links.items():
        r = want_links[pair]
        lo, hi = int(table.offset[i]), int(table.offset[i]) + int(table.n[i])
        assert (int(table.n[i]), int(table.sum_obs[i]), int(table.sum_obs_sq[i]), int(table.mask[i])) == \
            (r.n, r.obs, r.obs_sq, r.mask), pair
        assert table.obs_lo[lo:hi].tolist() == r.obs_u and table.obs_hi[lo:hi].tolist() == r.obs_v, pair
    # first-occurrence order of link rows (dict insertion order in the reference)
    dev_order = sorted(got_links, key=lambda pr: int(table.first_idx[got_links[pr]]))
    ora_order = [k for k, r in sorted(want_links.items(), key=lambda kv: kv[1].first_idx)]
    assert dev_order == ora_order
