"""GPU parity on the BASELINE.json configs that no other test touches, at reduced read-pair counts but with the configs'
own assembly sizes, libraries and flags, each against the C oracle (oracle/besst_oracle.c):

  C1  the Travis run (.travis.yml:15): 1836 contigs / 1 M pairs shaped like testdata/testset1 (whose BAM is not in the
      checkout), flags -m 4000 -s 500 -k 3000 -T 6000, with scoring and with --no_score
  C4  two libraries on 500 k contigs (41-bit edge keys): PE 500 bp on the first-library table, then MP 5 kb with PE
      contamination on the contig table a previous pass leaves behind (chained scaffolds, mixed directions, CleanObjects'
      classification by the new library's threshold)
  C5  three libraries on 2 M contigs (45-bit keys: link words no longer fit 64 bits next to the stream index on the
      large-stream path): PE 500, MP 5 kb, MP N(10000, 1000)
"""
import numpy as np
import pytest

from oracle import c_oracle as CO

pytestmark = pytest.mark.gpu


def _check(wl_cols, asm, table, lib, node_bits):
    import torch
    from besst_amd import pipeline, synth
    dev = torch.device('cuda', 0)
    rec = pipeline.DeviceRecords.from_columns(wl_cols)
    gb = pipeline.DeviceGraphBuilder(dev, asm.nc, node_bits, lib, rec.n, rec.n)
    gb.set_contigs(**table)
    gb.step(rec)
    torch.cuda.synchronize()
    got = gb.fetch_table()
    ctr = gb.read_counters()
    batch = synth.device_columns_to_batch(asm, wl_cols)
    keys, payload, aligned, c_ctr = CO.record_loop(batch, table, lib, node_bits)
    rows = CO.edge_rows(keys, payload)
    link = ~got.is_fishy
    assert np.array_equal(got.key, rows['key'])
    assert np.array_equal(got.n.astype(np.int64), rows['n'])
    assert np.array_equal(got.sum_obs[link], rows['sum_obs'][link])
    assert np.array_equal(got.sum_obs_sq[link], rows['sum_obs_sq'][link])
    assert np.array_equal(got.first_idx.astype(np.int64), rows['first_idx'])
    assert np.array_equal(got.mask.astype(np.int64), rows['mask'])
    assert np.array_equal(got.obs_lo.astype(np.int64), rows['obs_lo'])
    assert np.array_equal(got.obs_hi.astype(np.int64), rows['obs_hi'])
    assert gb.aligned.cpu().numpy().tolist() == aligned.tolist()
    assert [ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates, ctr.reads_with_too_long_insert,
            ctr.fishy_reads, ctr.n_tuples, ctr.n_reach, ctr.prev_obs1, ctr.prev_obs2] == c_ctr.tolist()
    return len(rows['key']), int(c_ctr[0])


@pytest.mark.parametrize('no_score', [False, True])
def test_c1_travis_flags(no_score):
    import torch
    from besst_amd import synth, workload
    cfg = synth.CONFIGS['C1']
    asm = synth.make_assembly(cfg['nc'], cfg['median'], synth.config_seed('C1'))
    spec = cfg['libs'][0]
    cols = synth.simulate_library_device(asm, spec, cfg['pairs'], synth.config_seed('C1') + 100, torch.device('cuda', 0))
    lib = workload.library_constants(spec)
    lib.update(ins_size_threshold=6000, no_score=no_score)         # -T 6000 (an int, as argparse hands it over)
    table = workload.first_library_table(asm.lengths, 3000)        # -k 3000
    rows, useful = _check(cols, asm, table, lib, workload.node_bits_for(table))
    assert rows > 100 and useful > 1000


@pytest.mark.parametrize('config,pairs', [('C4', 2_000_000), ('C5', 1_500_000)])
def test_multi_library_configs(config, pairs):
    import torch
    from besst_amd import synth, workload
    cfg = synth.CONFIGS[config]
    seed = synth.config_seed(config)
    asm = synth.make_assembly(cfg['nc'], cfg['median'], seed)
    dev = torch.device('cuda', 0)
    for li, spec in enumerate(cfg['libs']):
        lib = workload.library_constants(spec)
        thr = spec.mean + 4 * spec.sd
        if li == 0:
            table = workload.first_library_table(asm.lengths, thr)
        else:
            # scaffold ids keep growing across passes (param.scaffold_indexer, MakeScaffolds.py:276)
            table = workload.later_library_table(asm, seed + 50 + li, thr, first_scaffold_id=asm.nc * li + 1)
            assert len(set(table['direction'].tolist())) == 2 and int(table['ctg_pos'].max()) > 0
        node_bits = workload.node_bits_for(table)
        assert node_bits >= {'C4': 20, 'C5': 22}[config]             # >= 41- / 45-bit edge keys
        cols = synth.simulate_library_device(asm, spec, pairs, seed + 100 + li, dev)
        rows, useful = _check(cols, asm, table, lib, node_bits)
        assert rows > 1000 and useful > 1000, (config, li, rows, useful)
        del cols
        torch.cuda.empty_cache()
