"""The stitch stage cuts the stream's block summaries into spans of 4096 blocks and hands prev_obs / tuple offsets from
span to span through per-span aggregates (csrc/classify.hip: stitch_spans_kernel, stitch_kernel).  A stream needs 67 M
records for a second span, so BESST_STITCH_SPAN shrinks the span: with 1 and 3 blocks per span the small scenarios below
cross span borders all the time - in the single-GPU build (both forms of the record loop) and in the sharded build,
whose slices resolve their first reaching record either from gathered tails or, unresolved, at the owners."""
import numpy as np
import pytest

from besst_amd import workload
from tests import test_gpu_distributed_sim as SIM
from tests.test_gpu_fullsize import assert_table_equals_c_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('span', ['1', '3'])
@pytest.mark.parametrize('path', ['0', '1'])
@pytest.mark.parametrize('config,pairs,nc', [('C2', 150_000, 400), ('C3', 120_000, 150)])
def test_single_gpu_build_across_span_borders(span, path, config, pairs, nc, monkeypatch):
    from besst_amd import device
    monkeypatch.setenv('BESST_STITCH_SPAN', span)
    monkeypatch.setenv('BESST_RECORD_PATH', path)
    wl = workload.make(config, 0, pairs=pairs, nc=nc)
    batch = wl['batch']
    assert len(batch) > 10 * 16384                        # a dozen and more block summaries
    with device.GraphContext(0) as ctx:
        ctx.set_contigs(**wl['table'])
        lib = wl['lib']
        ctx.set_library(lib['read_len'], lib['ins_size_threshold'], lib['min_mapq'], lib['orientation'],
                        lib['detect_duplicate'], lib['extend_paths'], lib['no_score'])
        ctx.push_records(batch)
        for _ in range(2):
            table, aligned, ctr = ctx.build_graph()
            assert_table_equals_c_oracle(table, aligned, ctr, batch, wl)


@pytest.mark.parametrize('span', ['1', '2'])
@pytest.mark.parametrize('world,orientation,coverage,heads', [(3, 'rf', 'allreduce', 'gather'), (2, 'fr', 'rider', 'exchange'),
                                                              (5, 'rf', 'rider', 'exchange')])
def test_sharded_build_across_span_borders(span, world, orientation, coverage, heads, monkeypatch):
    monkeypatch.setenv('BESST_STITCH_SPAN', span)
    SIM.test_simulated_ranks_match_oracle(world, orientation, coverage, heads, monkeypatch)


@pytest.mark.parametrize('span', ['1'])
@pytest.mark.parametrize('which,world,flags', [(0, 2, {}), (2, 3, {}), (3, 2, dict(detect_duplicate=False))])
def test_duplicate_right_behind_a_slice_border(span, which, world, flags, monkeypatch):
    monkeypatch.setenv('BESST_STITCH_SPAN', span)
    SIM.test_slice_boundary_right_before_a_duplicate(which, world, flags)
