#!/usr/bin/env python
"""Calibrate bench.py's cpu_baseline ("port": oracle/py_oracle.record_loop) against the REAL reference.

Build container only (needs /root/reference).  One seeded C2-shaped stream, one core:
  reference : BESST/CreateGraph.py PE() up to its own 'ELAPSED reading file' line (InitializeObjects +
              InitializeGraph + the record loop, CreateGraph.py:45-213), driven through tests/refharness
  port      : oracle/py_oracle.record_loop on the same records and contig table
Writes oracle/cpu_port_calibration.json; bench.py copies `port_over_reference` into cpu_baseline so that the
port's pairs/s on the GPU box can be read as reference pairs/s.
"""
import io
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from besst_amd import synth, workload          # noqa: E402
from oracle import py_oracle as O                # noqa: E402
from tests.refharness import driver, loader      # noqa: E402


class TimedInfo(io.StringIO):
    def __init__(self):
        io.StringIO.__init__(self)
        self.t_loop_end = None

    def write(self, s):
        if self.t_loop_end is None and 'ELAPSED reading file' in s:
            self.t_loop_end = time.perf_counter()
        return io.StringIO.write(self, s)


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    mods = loader.load()
    wl = workload.make('C2', 0, pairs=pairs, nc=nc)
    batch, table, lib = wl['batch'], wl['table'], wl['lib']
    thr = float(lib['mean'] + 4 * lib['sd'])
    out = {'workload': 'C2-shaped: %d contigs / %d read-pairs (%d records), fr N(500,50)' % (nc, pairs, len(batch)),
           'host': os.uname().nodename, 'python': sys.version.split()[0], 'runs': []}
    for rep in range(2):
        param = driver.make_param(mods, mean_ins_size=lib['mean'], std_dev_ins_size=lib['sd'],
                                  ins_size_threshold=lib['ins_size_threshold'], contig_threshold=thr,
                                  read_len=100, orientation='fr', lognormal=False, contamination_ratio=False, empirical_distribution=None)
        info = TimedInfo()
        param.information_file = info
        param.contig_index = dict(enumerate(batch.references))
        length_of = dict(zip(batch.references, batch.lengths))
        C_dict = {name: 'A' * int(length_of[name]) for name in batch.references}
        t0 = time.perf_counter()
        mods['CreateGraph'].PE({}, {}, info, C_dict, param, {}, {}, batch)
        ref_s = info.t_loop_end - t0
        rec = {k: getattr(batch, k).tolist() for k in ('tid', 'mtid', 'pos', 'mpos', 'flag', 'mapq', 'qlen')}
        tab = dict(cls=table['cls'].tolist(), scaf=table['scaf_id'].tolist(), slen=table['scaf_len'].tolist(),
                   cpos=table['ctg_pos'].tolist(), clen=table['ctg_len'].tolist(),
                   cdir=[bool(x) for x in table['direction'].tolist()])
        p = O.LibParams(read_len=lib['read_len'], ins_size_threshold=lib['ins_size_threshold'], min_mapq=lib['min_mapq'],
                        orientation=lib['orientation'])
        t0 = time.perf_counter()
        res = O.record_loop(rec, tab, p)
        port_s = time.perf_counter() - t0
        out['runs'].append({'reference_s': ref_s, 'port_s': port_s, 'port_useful_reads': res.count})
    ref_s = min(r['reference_s'] for r in out['runs'])
    port_s = min(r['port_s'] for r in out['runs'])
    out['reference_pairs_per_s'] = pairs / ref_s
    out['port_pairs_per_s'] = pairs / port_s
    out['port_over_reference'] = ref_s / port_s
    out['note'] = ('reference = CreateGraph.PE through its record loop over in-memory record views (no BGZF / pysam '
                   'decode, which the real tool pays on top); port = oracle/py_oracle.record_loop; one core each')
    with open(os.path.join(REPO, 'oracle', 'cpu_port_calibration.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
