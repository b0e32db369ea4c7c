"""Scaffold-graph construction: drop-in for ``BESST/CreateGraph.py`` with the record loop on the MI355X.

``PE(Contigs, Scaffolds, Information, C_dict, param, small_contigs, small_scaffolds, bam_file)``
keeps the reference signature (CreateGraph.py:45) and contract: it returns
``(G, G_prime)`` as networkx-1.x compatible graphs with nodes ``(scaffold_id, 'L'|'R')`` and link-edge
attributes ``nr_links, obs, obs_sq, observations, gap, score``, mutates the four object dicts and
``param`` in place, and writes repeats.fa / repeats_log.tsv / low_coverage_contigs.fa.

Split of work:
  * device (libbesst_amd.so): the per-record loop (CreateGraph.py:111-211) including coverage sums,
    fishy-read counts, PosDir calculators, the CreateEdge duplicate chain and acceptance rule, then the
    radix sort + segmented reduction that turns link tuples into edge rows, and the per-edge ML gap /
    expected sigma / KS statistic of GiveScoreOnEdges (:498-614);
  * host (this file): object initialisation, assembling the graphs from the edge rows in first-occurrence
    order (= dict insertion order of the reference), the O(#contigs) coverage statistics and the
    O(#edges) filters whose semantics depend on graph iteration order (:355-404).
"""
from __future__ import print_function

import gc
import math
from collections import deque
import operator
import os
import sys
from itertools import compress, repeat
from time import time

import numpy as np

from . import Contig, Scaffold, e_nr_links, mathstats_compat, session
from . import GenerateOutput as GO
from .Parameter import counters
from .device import CLS_LARGE, CLS_SMALL, MASK_G, MASK_GPRIME
from .mathstats_compat import MaxObsDistr
from .nxcompat import Graph


# Wall time of PE's stages on the leading rank (seconds, summed over calls), filled only while this is a dict: bench.py and
# tests/pe_host_profile.py set it to {} to report where PE's host time goes.
STAGE_SECONDS = None


class _ObjectColumns(object):
    """Columns of the contig dictionaries that PE has at hand anyway (lengths from the header, the coverage it has just
    computed), kept for the duration of one PE call so that the stages that follow do not read them back attribute by
    attribute from 100 k objects.  Valid while a dictionary's membership is unchanged (PE's filters retire scaffolds and
    graph nodes, never contig entries); a dictionary whose size differs from the noted one is read the slow way."""

    def __init__(self):
        self.by_id = {}

    def note(self, contigs, objs=None, **cols):
        entry = self.by_id.get(id(contigs))
        if entry is None or entry['size'] != len(contigs):
            entry = self.by_id[id(contigs)] = dict(size=len(contigs), objs=objs if objs is not None else list(contigs.values()))
        entry.update(cols)

    def get(self, contigs, name):
        entry = self.by_id.get(id(contigs))
        if entry is None or entry['size'] != len(contigs):
            return None
        return entry.get(name)


_COLUMNS = None             # the running PE call's _ObjectColumns (None outside PE: every helper reads the objects)


def _objects_of(contigs):
    objs = _COLUMNS.get(contigs, 'objs') if _COLUMNS is not None else None
    return objs if objs is not None else list(contigs.values())


def _cached_column(contigs, objs, attribute, dtype):
    col = _COLUMNS.get(contigs, attribute) if _COLUMNS is not None else None
    return col if col is not None else _column(objs, attribute, dtype)


class _Stages(object):
    __slots__ = ('t',)

    def __init__(self):
        from time import perf_counter
        self.t = perf_counter() if STAGE_SECONDS is not None else None

    def mark(self, name):
        if self.t is not None:
            from time import perf_counter
            now = perf_counter()
            STAGE_SECONDS[name] = STAGE_SECONDS.get(name, 0.0) + now - self.t
            self.t = now


def PE(Contigs, Scaffolds, Information, C_dict, param, small_contigs, small_scaffolds, bam_file):
    """CreateGraph.PE (CreateGraph.py:45).  The cyclic garbage collector is paused for the duration of the call: the
    objects and graphs built here are hundreds of thousands of small containers that reference nothing but numbers and
    each other, and every allocation burst makes the collector re-walk the growing heap (building the two graphs of a
    100 k-contig assembly: 1.8 s with it, 0.4 s without)."""
    global _COLUMNS
    was_enabled = gc.isenabled()
    gc.disable()
    _COLUMNS = _ObjectColumns()
    try:
        return _PE(Contigs, Scaffolds, Information, C_dict, param, small_contigs, small_scaffolds, bam_file)
    finally:
        _COLUMNS = None
        if was_enabled:
            gc.enable()


def _PE(Contigs, Scaffolds, Information, C_dict, param, small_contigs, small_scaffolds, bam_file):
    """One GPU: the whole of PE.  Under a process group (besst_amd.sharded): rank 0 leads - it runs PE's host side and
    returns the graphs -, every other rank follows rank 0 through the collective stages and returns empty graphs."""
    sess = session.open_session(bam_file)
    if sess.is_follower:
        sess.follow(param)
        return (Graph(), Graph())
    try:
        out = _PE_leader(sess, Contigs, Scaffolds, Information, C_dict, param, small_contigs, small_scaffolds)
    except BaseException as exc:
        if not getattr(exc, 'on_every_rank', False):          # (a failed collective stage has been raised on all ranks already)
            sess.abort(exc)
        raise
    sess.done(param)
    return out


def _PE_leader(sess, Contigs, Scaffolds, Information, C_dict, param, small_contigs, small_scaffolds):
    G = Graph()
    G_prime = Graph()
    print('Parsing BAM file...', file=Information)
    batch = sess.batch
    stages = _Stages()

    if param.first_lib:
        start_init = time()
        InitializeObjects(batch, Contigs, Scaffolds, param, Information, G_prime, small_contigs, small_scaffolds, C_dict)
        print('Time initializing BESST objects: ', time() - start_init, file=Information)
    else:
        start_clean = time()
        CleanObjects(Contigs, Scaffolds, param, Information, small_contigs, small_scaffolds)
        print('Time cleaning BESST objects for next library: ', time() - start_clean, file=Information)
    stages.mark('objects (InitializeObjects / CleanObjects)')

    if len(Scaffolds) == 0:
        if param.output_directory and not os.path.isfile(param.output_directory + '/repeats.fa'):
            open(param.output_directory + '/repeats.fa', 'w').close()
        return (G, G_prime)

    # The scaffold ends every graph starts with (InitializeGraph, CreateGraph.py:710-722) are noted as columns; the
    # graphs themselves are built ONCE, at the end, from what the filters below leave (GraphPlan.build).
    tot_start = time()
    plan_G, plan_Gp = GraphPlan(), GraphPlan()
    if param.no_score:
        InitializeGraph(small_scaffolds, plan_Gp, Information)
        InitializeGraph(Scaffolds, plan_Gp, Information)
    elif param.extend_paths:
        InitializeGraph(Scaffolds, plan_G, Information)
        InitializeGraph(small_scaffolds, plan_Gp, Information)
        InitializeGraph(Scaffolds, plan_Gp, Information)
    else:
        InitializeGraph(Scaffolds, plan_G, Information)
    print('Total time elapsed for initializing Graph: ', time() - tot_start, file=Information)

    # ---- record loop on the device ------------------------------------------------------------------------
    print('Reading bam file and creating scaffold graph...', file=Information)
    staart = time()
    table_cols, tids_of_group = contig_table(batch.references, Contigs, small_contigs, Scaffolds, small_scaffolds)
    stages.mark('graph plans + contig table')
    ctx = sess.ctx
    ctx.set_contigs(**table_cols)
    ctx.set_library(param.read_len, param.ins_size_threshold, param.min_mapq, param.orientation,
                    param.detect_duplicate, param.extend_paths, param.no_score)
    table, aligned, ctr = ctx.build_graph(lazy_observations=True)
    counter = counters(ctr.count, ctr.non_unique, ctr.non_unique_for_scaf, ctr.nr_of_duplicates,
                       ctr.prev_obs1, ctr.prev_obs2, ctr.reads_with_too_long_insert)
    stages.mark('device: record loop, edge table, fetch')
    links = LinkTable(table)
    plan_G.take(links, MASK_G)
    plan_Gp.take(links, MASK_GPRIME)

    print('ELAPSED reading file:', time() - staart, file=Information)
    print('NR OF FISHY READ LINKS: ', ctr.fishy_reads, file=Information)
    print('Number of USEFUL READS (reads mapping to different contigs uniquly): ', counter.count, file=Information)
    print('Number of non unique reads (at least one read non-unique in read pair) that maps to different contigs '
          '(filtered out from scaffolding): ', counter.non_unique, file=Information)
    print('Reads with too large insert size from "USEFUL READS" (filtered out): ',
          counter.reads_with_too_long_insert, file=Information)
    print('Initial number of edges in G (the graph with large contigs): ', plan_G.number_of_edges(), file=Information)
    print('Initial number of edges in G_prime (the full graph of all contigs before removal of repats): ',
          plan_Gp.number_of_edges(), file=Information)
    if param.detect_duplicate:
        print('Number of duplicated reads indicated and removed: ', counter.nr_of_duplicates, file=Information)

    # ---- coverage (CreateGraph.py:237-246) -----------------------------------------------------------------
    # a contig that is not in THIS library's BAM header has no aligned bases: the reference starts every contig's
    # counter at 0 (CreateGraph.py:89-95), so its coverage is 0 there too
    aligned = np.asarray(aligned, dtype=np.int64)
    for contigs, tids in zip((Contigs, small_contigs), tids_of_group):
        if len(contigs):
            objs = _objects_of(contigs)
            numer = np.where(tids >= 0, aligned[np.maximum(tids, 0)], 0)
            length = _cached_column(contigs, objs, 'length', np.int64)
            cov = numer / length.astype(np.float64)                       # (int / float(length), as the reference divides)
            for cont, value in zip(objs, cov.tolist()):
                cont.coverage = value
            _COLUMNS.note(contigs, objs, length=length, coverage=cov)

    if param.first_lib and param.lower_cov_cutoff:
        filter_low_coverage_contigs(Contigs, Scaffolds, plan_G, param, plan_Gp, small_contigs, small_scaffolds, Information)

    mean_cov, std_dev_cov = CalculateMeanCoverage(Contigs, Information, param)
    param.mean_coverage = mean_cov
    param.std_dev_coverage = std_dev_cov

    if param.first_lib:
        RepeatDetector(Contigs, Scaffolds, plan_G, param, plan_Gp, small_contigs, small_scaffolds, Information)
    print('Number of edges in G (after repeat removal): ', plan_G.number_of_edges(), file=Information)
    print('Number of edges in G_prime (after repeat removal): ', plan_Gp.number_of_edges(), file=Information)

    RemoveBugEdges(plan_G, plan_Gp, links, param, Information)
    print('Number of edges in G (after filtering for buggy flag stats reporting): ', plan_G.number_of_edges(), file=Information)
    print('Number of edges in G_prime  (after filtering for buggy flag stats reporting): ', plan_Gp.number_of_edges(),
          file=Information)

    infer_spurious_link_count_threshold(plan_Gp, param)
    if not param.edgesupport:
        param.edgesupport = 5
        print('Letting -e be {0} for this library.'.format(param.edgesupport), file=Information)
    else:
        print('User has set -e to be {0} for this library.'.format(param.edgesupport), file=Information)

    counter_low_support = plan_G.drop(links.n < param.edgesupport)
    print('Removed {0} edges from graph G of border contigs.'.format(counter_low_support), file=Information)
    remove_edges_below_threshold(plan_Gp, param)

    stages.mark('coverage, repeats, filters (columns)')
    scores = None
    if not param.no_score:
        scores = GiveScoreOnEdges(plan_G, Scaffolds, small_scaffolds, Contigs, param, Information, 'G', ctx)
    stages.mark('GiveScoreOnEdges (device scoring + host scalars)')
    plan_G.build(G, scores)
    stages.mark('assemble G')
    plan_Gp.build(G_prime, None)
    stages.mark('assemble G_prime')
    print('Number of edges in G_prime  (after removing edges under -e threshold (if not specified, default is '
          '-e 3): ', plan_Gp.number_of_edges(), file=Information)
    print('\n -------------------------------------------------------------\n', file=Information)
    print('Nr of contigs/scaffolds included in this pass: ' + str(len(Scaffolds) + len(small_scaffolds)),
          file=Information)
    print('Out of which {0} acts as border contigs.'.format(len(Scaffolds)), file=Information)
    return (G, G_prime)


# -----------------------------------------------------------------------------------------------------------
# device boundary helpers
# -----------------------------------------------------------------------------------------------------------
def contig_table(references, Contigs, small_contigs, Scaffolds, small_scaffolds):
    """Flatten the object dicts into the per-tid table the kernels gather from: one column per attribute and dictionary
    (attribute getters and dictionary look-ups mapped in C), scattered to the contigs' places in the BAM header.  Also
    returns, per dictionary, the tid of every contig in dictionary order (-1: not in this library's header)."""
    n = len(references)
    cols = dict(scaf_id=np.zeros(n, np.int32), scaf_len=np.zeros(n, np.int32), ctg_pos=np.zeros(n, np.int32),
                ctg_len=np.zeros(n, np.int32), direction=np.zeros(n, np.uint8), cls=np.zeros(n, np.uint8))
    # a name that occurs twice in the header keeps its first place (the later entries stay absent)
    tid_of = None
    tids_of_group = []
    for contigs, scaffolds, k in ((Contigs, Scaffolds, CLS_LARGE), (small_contigs, small_scaffolds, CLS_SMALL)):
        count = len(contigs)
        tids = _COLUMNS.get(contigs, 'tid') if _COLUMNS is not None else None
        if tids is None:
            if tid_of is None:
                tid_of = dict(zip(reversed(references), range(n - 1, -1, -1)))
            tids = np.fromiter(map(tid_of.get, contigs, repeat(-1)), dtype=np.int64, count=count)
        tids_of_group.append(tids)
        if count == 0:
            continue
        objs = _objects_of(contigs)
        here = tids >= 0
        # (a contig of the small dictionary that is also a key of the large one cannot exist: the dictionaries are disjoint)
        scaf = _cached_column(contigs, objs, 'scaffold', np.int64)
        slen = _COLUMNS.get(contigs, 's_length') if _COLUMNS is not None else None
        if slen is None:
            s_len = dict(zip(scaffolds, map(operator.attrgetter('s_length'), scaffolds.values())))
            slen = np.fromiter(map(s_len.__getitem__, scaf.tolist()), dtype=np.int64, count=count)
        at = tids[here]
        cols['cls'][at] = k
        cols['scaf_id'][at] = scaf[here]
        cols['scaf_len'][at] = slen[here]
        cols['ctg_pos'][at] = _cached_column(contigs, objs, 'position', np.int64)[here]
        cols['ctg_len'][at] = _cached_column(contigs, objs, 'length', np.int64)[here]
        cols['direction'][at] = _cached_column(contigs, objs, 'direction', np.bool_)[here]
    return cols, tids_of_group


class LinkData(dict):
    """Attribute dict of a link edge.  'observations' - one int per link, in BAM order (CreateGraph.py:849,862) - is cut
    out of the device's observation column the first time anything asks for it (its readers are
    MakeScaffolds.py:322,331,425,1139: a few edges per extended path); every other key is an ordinary item."""
    __slots__ = ('_col', '_lo', '_hi')                       # unset on a LinkData built anywhere: a plain dict until GraphPlan lends it a column

    def _cut(self):
        col = getattr(self, '_col', None)
        if col is not None:
            self._col = None
            dict.__setitem__(self, 'observations', col[self._lo:self._hi].tolist())

    def __missing__(self, key):
        if key == 'observations' and getattr(self, '_col', None) is not None:
            self._cut()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    # whatever looks at the dict as a whole sees the complete dict
    def __iter__(self):
        self._cut()
        return dict.__iter__(self)

    def __len__(self):
        self._cut()
        return dict.__len__(self)

    def __contains__(self, key):
        if key == 'observations':
            self._cut()
        return dict.__contains__(self, key)

    def __eq__(self, other):
        self._cut()
        if isinstance(other, LinkData):
            other._cut()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def get(self, key, default=None):
        if key == 'observations':
            self._cut()
        return dict.get(self, key, default)

    # writers: the lazily cut list must not overwrite, or hide from, what a caller puts there
    def __setitem__(self, key, value):
        if key == 'observations':
            self._col = None                                 # the caller's value replaces the column for good
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        if key == 'observations':
            self._cut()
        dict.__delitem__(self, key)

    def pop(self, key, *default):
        if key == 'observations':
            self._cut()
        return dict.pop(self, key, *default)

    def popitem(self):
        self._cut()
        return dict.popitem(self)

    def setdefault(self, key, default=None):
        if key == 'observations':
            self._cut()
        return dict.setdefault(self, key, default)

    def update(self, *args, **kw):
        self._cut()
        dict.update(self, *args, **kw)

    def clear(self):
        self._col = None
        dict.clear(self)

    def keys(self):
        self._cut()
        return dict.keys(self)

    def items(self):
        self._cut()
        return dict.items(self)

    def values(self):
        self._cut()
        return dict.values(self)

    def copy(self):
        self._cut()
        return dict(self)

    def __repr__(self):
        self._cut()
        return dict.__repr__(self)


class ObservationColumn(object):
    """`observations` of every link, one int per link in BAM order, grouped by edge row: a slice of it is an edge's list.
    Taken from the table when a slice is first asked for (EdgeTable.observation_sums: by then the background fetch is
    normally done)."""
    __slots__ = ('_table', '_col')

    def __init__(self, table):
        self._table, self._col = table, None

    def __getitem__(self, index):
        col = self._col
        if col is None:
            col = self._col = self._table.observation_sums()
            self._table = None
        return col[index]


class LinkTable(object):
    """The device's edge rows as columns, link rows in order of first occurrence in the BAM.

    The reference creates an edge the first time CreateEdge accepts a link for it (CreateGraph.py:842-849), so the
    adjacency order of its graphs = order of first occurrence; ``first_idx`` is monotone in that order.  ``u < v`` are
    node codes (scaffold id * 2 + (side == 'R')); fishy rows (BWA-quirk counts, :141-163) are kept by node pair."""

    def __init__(self, table):
        fishy = np.flatnonzero(table.is_fishy)
        rows = np.flatnonzero(~table.is_fishy)
        rows = rows[np.argsort(table.first_idx[rows], kind='stable')]
        self.rows = rows                                     # device row of every link (for the scoring stage)
        self.u, self.v = table.u[rows], table.v[rows]
        self.n = table.n[rows].astype(np.int64)
        self.obs, self.obs_sq = table.sum_obs[rows], table.sum_obs_sq[rows]
        self.mask = table.mask[rows]
        self.lo = table.offset[rows].astype(np.int64)
        self.pair = table.key[rows] >> np.uint64(1)
        self.fishy_pair = table.key[fishy] >> np.uint64(1)
        self.fishy_n = table.n[fishy].astype(np.int64)
        # one observation per link: obs1 + obs2 (the device keeps the two ends apart for the scoring stage); the column is
        # summed on the device and crosses PCIe beside this call's host work (device.ObservationSource)
        self.observations = ObservationColumn(table)

    def __len__(self):
        return int(self.rows.shape[0])


class GraphPlan(object):
    """One of the two graphs while it is still columns: the scaffolds whose two ends are its nodes (in InitializeGraph
    order) and a liveness flag per link row.  Every filter of PE is a mask over the link columns; node and adjacency
    order of the graph built at the end equal those of a graph that had the edges inserted and removed one by one,
    because removing dictionary entries keeps the order of the rest."""

    def __init__(self):
        self.sid, self.length = [], []
        self.links = None
        self.alive = None

    def add_scaffolds(self, scaffolds):
        self.sid.extend(scaffolds)
        self.length.extend(map(operator.attrgetter('s_length'), scaffolds.values()))

    def take(self, links, mask_bit):
        self.links = links
        self.alive = (links.mask & mask_bit) != 0
        self.sid_arr = np.asarray(self.sid, dtype=np.int64)
        self.len_arr = np.asarray(self.length, dtype=np.int64)
        size = int(max(self.sid_arr.max() if self.sid_arr.size else 0,
                       (links.v.max() >> 1) if len(links) else 0)) + 1
        self.node_alive = np.zeros(size, dtype=bool)          # by scaffold id
        self.node_alive[self.sid_arr] = True

    def remove_scaffolds(self, sids):
        """remove_nodes_from of both ends of every scaffold: the nodes and every edge at them go."""
        sids = np.asarray([s for s in sids if s < self.node_alive.shape[0]], dtype=np.int64)
        if sids.size == 0:
            return
        self.node_alive[sids] = False
        lk = self.links
        self.alive &= self.node_alive[lk.u >> 1] & self.node_alive[lk.v >> 1]

    def drop(self, which):
        """remove_edge for the live links selected by the mask; returns how many went."""
        hit = self.alive & which
        self.alive &= ~which
        return int(hit.sum())

    def number_of_nodes(self):
        return 2 * int(self.node_alive[self.sid_arr].sum())

    def number_of_edges(self):
        return self.number_of_nodes() // 2 + int(self.alive.sum())

    def node_rank(self):
        """Position of every node code in the graph's node order (scaffolds in InitializeGraph order, 'L' before 'R')."""
        keep = self.sid_arr[self.node_alive[self.sid_arr]]
        rank = np.full(2 * self.node_alive.shape[0], np.iinfo(np.int64).max, dtype=np.int64)
        rank[2 * keep] = 2 * np.arange(keep.shape[0])
        rank[2 * keep + 1] = 2 * np.arange(keep.shape[0]) + 1
        return rank

    def edges_order(self, select=None):
        """Live link rows (optionally only the selected ones) in the order G.edges() lists them, and for each whether the
        edge is reported from its larger node code: networkx walks the nodes in node order and lists every edge at its
        first endpoint, in adjacency (= first occurrence) order."""
        idx = np.flatnonzero(self.alive if select is None else (self.alive & select))
        rank = self.node_rank()
        ru, rv = rank[self.links.u[idx]], rank[self.links.v[idx]]
        order = np.lexsort((idx, np.minimum(ru, rv)))
        return idx[order], (rv < ru)[order]

    def build(self, graph, scores):
        """Fill the (empty) graph: nodes with their 'length', the intra-scaffold edges (CreateGraph.py:710-722), the
        surviving links.  The graph gets its node keys at once - in node order - and a column source behind them
        (nxcompat.LazyDict / GraphColumns): a node's neighbour dictionary and the attribute dictionaries of its edges are
        made when the node is first read, all that is still unmade in bulk the first time something walks the whole graph.
        A graph that already holds nodes is filled the general way."""
        keep = self.node_alive[self.sid_arr]
        sids, lengths = self.sid_arr[keep].tolist(), self.len_arr[keep].tolist()
        count = len(sids)
        nodes = [None] * (2 * count)
        nodes[0::2], nodes[1::2] = zip(sids, repeat('L')), zip(sids, repeat('R'))
        lk = self.links
        idx = np.flatnonzero(self.alive)
        source = GraphColumns(nodes, lengths, lk, idx, scores)
        # where every node code sits in `nodes`
        slot = np.full(2 * self.node_alive.shape[0], -1, dtype=np.int64)
        kept = self.sid_arr[keep]
        slot[2 * kept] = 2 * np.arange(count)
        slot[2 * kept + 1] = 2 * np.arange(count) + 1
        su, sv = slot[lk.u[idx]], slot[lk.v[idx]]
        general = bool(graph._node) or not hasattr(graph, 'adopt') or bool((su < 0).any() or (sv < 0).any())
        if general:
            # a graph that is not empty, or a link end no InitializeGraph call has seen (cannot happen with the record
            # loop's rules): as add_node / add_edge would
            for scaffold_, length in zip(sids, lengths):
                graph.add_scaffold(scaffold_, length)
            side = ('L', 'R')
            for p, (a, b) in enumerate(zip(lk.u[idx].tolist(), lk.v[idx].tolist())):
                na, nb = (a >> 1, side[a & 1]), (b >> 1, side[b & 1])
                graph.add_edge(na, nb)
                data = source.data(p)
                graph._adj[na][nb] = data
                graph._adj[nb][na] = data
            return
        source.index(su, sv)
        tokens = dict(zip(nodes, range(2 * count)))          # (hashed once; the two dictionaries copy it)
        graph.adopt(tokens, source.node_side(), tokens, source)


class GraphColumns(object):
    """What stands behind a graph that GraphPlan.build has filled: the nodes (in node order; slot 2 k / 2 k + 1 = the two
    ends of the k-th scaffold), the live link rows in adjacency (= first occurrence) order, and per node the slice of a
    CSR list that names its links.  `one(slot)` makes the neighbour dictionary of one node - the other end of its scaffold
    first (InitializeGraph's edge), then its links in order -, `rest(adj)` every dictionary still unmade, in bulk.  The
    attribute dictionary of an edge is made once and shared by its two ends, whichever is read first."""

    def __init__(self, nodes, lengths, links, idx, scores):
        self.nodes, self.lengths = nodes, lengths
        self.links, self.idx = links, idx
        self.scores = scores
        self.datas = [None] * int(idx.shape[0])
        self.inner = [None] * (len(nodes) // 2)
        self.gaps = self.vals = None
        self.n = self.s1 = self.s2 = self.lo = None
        self.su = self.sv = self.ptr = None

    def index(self, su, sv):
        """The slots of every link's two ends; the CSR over the node slots - for every slot the positions of its links (in
        link order) and the slot at the other end - is made when a single node is first asked for."""
        self.su, self.sv = su, sv

    def _csr(self):
        su, sv = self.su, self.sv
        m = int(su.shape[0])
        owner = np.concatenate((su, sv))
        pos = np.concatenate((np.arange(m), np.arange(m)))
        order = np.lexsort((pos, owner))
        self.link_of = pos[order]
        self.other = np.concatenate((sv, su))[order]
        self.ptr = np.concatenate(([0], np.cumsum(np.bincount(owner, minlength=len(self.nodes))))).tolist()

    def _columns(self):
        if self.n is None:
            lk, idx = self.links, self.idx
            self.n, self.s1, self.s2 = lk.n[idx].tolist(), lk.obs[idx].tolist(), lk.obs_sq[idx].tolist()
            lo = lk.lo[idx]
            self.lo, self.hi = lo.tolist(), (lo + lk.n[idx]).tolist()
            scores, m = self.scores, int(idx.shape[0])
            self.scores = None
            if scores is not None and m:
                # GiveScoreOnEdges scores every live link of G, in G.edges() order: gap / score by link position
                order = np.argsort(scores[0], kind='stable')
                if order.shape[0] == m and np.array_equal(np.asarray(scores[0])[order], idx):
                    order = order.tolist()
                    self.gaps = [scores[1][k] for k in order]
                    self.vals = [scores[2][k] for k in order]
                else:
                    at = dict(zip(idx.tolist(), range(m)))
                    self.gaps, self.vals = [_UNSCORED] * m, [_UNSCORED] * m
                    for k, gap, score in zip(np.asarray(scores[0]).tolist(), scores[1], scores[2]):
                        self.gaps[at[k]] = gap
                        self.vals[at[k]] = score

    def data(self, p):
        d = self.datas[p]
        if d is None:
            self._columns()
            if self.gaps is not None and self.gaps[p] is not _UNSCORED:
                d = LinkData(nr_links=self.n[p], obs=self.s1[p], obs_sq=self.s2[p], gap=self.gaps[p], score=self.vals[p])
            else:
                d = LinkData(nr_links=self.n[p], obs=self.s1[p], obs_sq=self.s2[p])
            d._col, d._lo, d._hi = self.links.observations, self.lo[p], self.hi[p]
            self.datas[p] = d
        return d

    def _inner(self, k):
        d = self.inner[k]
        if d is None:
            d = self.inner[k] = {'nr_links': None}
        return d

    def one(self, slot):
        nodes = self.nodes
        nb = {nodes[slot ^ 1]: self._inner(slot >> 1)}
        if self.ptr is None:
            self._csr()
        a, b = self.ptr[slot], self.ptr[slot + 1]
        if b > a:
            data = self.data
            for j, p in zip(self.other[a:b].tolist(), self.link_of[a:b].tolist()):
                nb[nodes[j]] = data(p)
        return nb

    def rest(self, adj):
        """Every neighbour dictionary still unmade, the way the graph used to be built in one go: the edge dictionaries in a
        comprehension, the neighbour dictionaries from the scaffold pairs, then two insertions per link in link order."""
        nodes = self.nodes
        self._columns()
        datas = self.datas
        m = len(datas)
        if m and all(d is None for d in datas):
            col = self.links.observations
            if self.gaps is not None and _UNSCORED not in self.gaps:
                datas = [LinkData(nr_links=n, obs=s1, obs_sq=s2, gap=g, score=v)
                         for n, s1, s2, g, v in zip(self.n, self.s1, self.s2, self.gaps, self.vals)]
            elif self.gaps is None:
                datas = [LinkData(nr_links=n, obs=s1, obs_sq=s2) for n, s1, s2 in zip(self.n, self.s1, self.s2)]
            else:
                datas = [self.data(p) for p in range(m)]
            for d, a, b in zip(datas, self.lo, self.hi):
                d._col = col
                d._lo = a
                d._hi = b
            self.datas = datas
        else:
            datas = self.datas = [self.data(p) for p in range(m)]
        raw = dict.__getitem__
        unmade = [k for k, node in enumerate(nodes) if dict.__contains__(adj, node) and raw(adj, node).__class__ is int]
        if len(unmade) == len(nodes):
            inner = [{'nr_links': None} if d is None else d for d in self.inner]
            nbrs = [None] * len(nodes)
            left, right = nodes[0::2], nodes[1::2]
            nbrs[0::2], nbrs[1::2] = [{r: d} for r, d in zip(right, inner)], [{l: d} for l, d in zip(left, inner)]
            for d, i, j in zip(datas, self.su.tolist(), self.sv.tolist()):
                nbrs[i][nodes[j]] = d
                nbrs[j][nodes[i]] = d
            dict.update(adj, zip(nodes, nbrs))
        else:
            put = dict.__setitem__
            for k in unmade:
                put(adj, nodes[k], self.one(k))

    def node_side(self):
        return _NodeAttributes(self.nodes, self.lengths)

    def link_arrays(self):
        """The link edges as G.edges() would list them - every edge from its first endpoint in node order, a node's edges in
        adjacency (= link) order - without making a single container: (slot of the first endpoint, slot of the second, link
        position).  A node's slot is its place in the node order: 2 k / 2 k + 1 for the k-th scaffold."""
        su, sv = self.su, self.sv
        lo, hi = np.minimum(su, sv), np.maximum(su, sv)
        order = np.lexsort((np.arange(su.shape[0]), lo))
        return lo[order], hi[order], order

    def link_scores(self):
        """score per link position, or None when the graph's links carry none."""
        self._columns()
        return self.vals


class _NodeAttributes(object):
    """The node attribute dictionaries ({'length': scaffold length}, CreateGraph.py:713-716) behind a filled graph."""

    def __init__(self, nodes, lengths):
        self.nodes, self.lengths = nodes, lengths

    def one(self, slot):
        return {'length': self.lengths[slot >> 1]}

    def rest(self, node):
        raw, put, lengths = dict.__getitem__, dict.__setitem__, self.lengths
        for k, n in enumerate(self.nodes):
            if dict.__contains__(node, n) and raw(node, n).__class__ is int:
                put(node, n, {'length': lengths[k >> 1]})


_UNSCORED = object()


# -----------------------------------------------------------------------------------------------------------
# host-side stages, same names as the reference
# -----------------------------------------------------------------------------------------------------------
def InitializeGraph(dict_with_scaffolds, graph, Information):
    graph.add_scaffolds(dict_with_scaffolds)
    return ()


def CalculateStats(sorted_contig_lengths, sorted_contig_lengths_small, param, Information):
    """N50 / L50 of the length-sorted large group followed by the small one (CreateGraph.py:727-757): the first position
    where the running length reaches half the assembly - a cumulative sum and one search (integers: exact)."""
    lengths = np.concatenate([np.asarray(sorted_contig_lengths, dtype=np.int64).reshape(-1),
                              np.asarray(sorted_contig_lengths_small, dtype=np.int64).reshape(-1)])
    N50, L50 = 0, 0
    if lengths.size:
        reached = np.flatnonzero(np.cumsum(lengths) >= param.tot_assembly_length / 2.0)
        # the reference stops at the first group that yields a non-zero N50; zero-length entries cannot occur here
        if reached.size:
            L50 = int(reached[0]) + 1
            N50 = int(lengths[reached[0]])
    print('L50: ', L50, 'N50: ', N50, 'Initial contig assembly length: ', param.tot_assembly_length, file=Information)
    return (N50, L50)


def InitializeObjects(bam_file, Contigs, Scaffolds, param, Information, G_prime, small_contigs, small_scaffolds, C_dict):
    """One contig and one scaffold object per sequence of the FASTA that the BAM header names, in header order, scaffold
    ids counting on from param.scaffold_indexer; contigs of at least param.contig_threshold bases are the large ones
    (CreateGraph.py:729-786).  The selection is a mask over the header's length column; the objects are made in
    comprehensions and handed to the dictionaries in bulk."""
    contig_threshold = param.contig_threshold
    cont_lengths = [int(nr) for nr in bam_file.lengths]
    cont_names = bam_file.references
    contig_lengths = [len(c_seq) for c_seq in C_dict.values()]
    param.tot_assembly_length = sum(contig_lengths)
    N50, L50 = CalculateStats(np.sort(np.asarray(contig_lengths, dtype=np.int64))[::-1], [], param, Information)
    param.current_L50 = L50
    param.current_N50 = N50
    n = len(cont_names)
    lens = np.asarray(cont_lengths, dtype=np.int64)
    known = np.fromiter(map(C_dict.__contains__, cont_names), dtype=np.bool_, count=n)
    if len(set(cont_names)) != n:
        # a name that occurs twice in the header: the reference takes the first entry (the sequence is gone from C_dict by
        # the time the second comes)
        seen = set()
        for i, name in enumerate(cont_names):
            if name in seen:
                known[i] = False
            seen.add(name)
    large = known & (lens >= contig_threshold)
    chosen = np.flatnonzero(known & (large | (lens > 0)))
    is_large = large[chosen].tolist()
    names = [cont_names[i] for i in chosen.tolist()]
    lengths = lens[chosen].tolist()
    first_id = param.scaffold_indexer
    ids = range(first_id, first_id + len(names))
    seqs = list(map(C_dict.pop, names))
    # (instances made and their slots filled a column at a time, all inside C: a third of the constructor calls' time)
    count = len(names)
    contigs = _bulk_objects(Contig.contig, count, name=names, scaffold=ids, direction=repeat(True), position=repeat(0),
                            length=lengths, coverage=repeat(None), repeat=repeat(False), is_haplotype=repeat(False),
                            sequence=seqs)
    scaffolds = _bulk_objects(Scaffold.scaffold, count, name=ids, contigs=[[c] for c in contigs], s_length=lengths)
    fresh = not Contigs and not small_contigs
    if all(is_large):
        Contigs.update(zip(names, contigs))
        Scaffolds.update(zip(ids, scaffolds))
    else:
        is_small = (~large[chosen]).tolist()
        Contigs.update(zip(compress(names, is_large), compress(contigs, is_large)))
        Scaffolds.update(zip(compress(ids, is_large), compress(scaffolds, is_large)))
        small_contigs.update(zip(compress(names, is_small), compress(contigs, is_small)))
        small_scaffolds.update(zip(compress(ids, is_small), compress(scaffolds, is_small)))
    param.scaffold_indexer = first_id + len(names)
    if _COLUMNS is not None and fresh and len(Contigs) + len(small_contigs) == len(names):
        # what the record loop's contig table and the coverage statistics read of these objects is known here as columns:
        # header place, scaffold id, lengths; position 0 and forward direction for a contig that is its own scaffold
        big = large[chosen]
        sid_col = np.arange(first_id, first_id + len(names), dtype=np.int64)
        len_col = lens[chosen]
        for group, sel in ((Contigs, big), (small_contigs, ~big)):
            count = int(sel.sum())
            if count != len(group):
                continue
            objs = list(compress(contigs, is_large if group is Contigs else (~big).tolist()))
            _COLUMNS.note(group, objs, tid=chosen[sel].astype(np.int64), scaffold=sid_col[sel], s_length=len_col[sel],
                          length=len_col[sel], position=np.zeros(count, np.int64), direction=np.ones(count, np.bool_))
    return ()


def _bulk_objects(cls, count, **columns):
    """`count` instances of a __slots__ class with their attributes set from columns - what calling the class `count`
    times with those arguments leaves, without a Python frame per object."""
    objs = list(map(object.__new__, repeat(cls, count)))
    for name, values in columns.items():
        deque(map(getattr(cls, name).__set__, objs, values), maxlen=0)
    return objs


def _column(objects, attribute, dtype):
    return np.fromiter(map(operator.attrgetter(attribute), objects), dtype=dtype, count=len(objects))


def CleanObjects(Contigs, Scaffolds, param, Information, small_contigs, small_scaffolds):
    """Scaffolds below the library's contig threshold move to the small dictionaries (CreateGraph.py:759-784): one
    length column per dictionary, a mask, and a loop over the scaffolds that actually move."""
    large_ids = list(Scaffolds)
    large_len = _column([Scaffolds[i] for i in large_ids], 's_length', np.int64)
    small_len = _column(list(small_scaffolds.values()), 's_length', np.int64)
    N50, L50 = CalculateStats(np.sort(large_len)[::-1], np.sort(small_len)[::-1], param, Information)
    param.current_L50 = L50
    param.current_N50 = N50
    moving = np.flatnonzero(large_len < param.contig_threshold)
    for k in moving.tolist():
        scaffold_ = large_ids[k]
        S_obj = Scaffolds.pop(scaffold_)
        GO.ChangeToSmallContigs(Contigs, S_obj.contigs, small_contigs)
        small_scaffolds[scaffold_] = S_obj
    print('Nr of contigs/scaffolds that was singeled out due to length constraints ' + str(int(moving.size)),
          file=Information)
    return ()


def _retire_scaffolds(selected, scaffold_dict, graphs):
    """Contigs taken out of the scaffolding (low coverage, repeats): their scaffold objects and both scaffold ends go."""
    gone = [c.scaffold for c in selected]
    for scaf_ in gone:
        del scaffold_dict[scaf_]
    for g in graphs:
        g.remove_scaffolds(gone)


def _coverage_groups(Contigs, Scaffolds, G, G_prime, small_contigs, small_scaffolds, param):
    """The two contig dictionaries as (objects in dictionary order, coverage column, scaffold dictionary, graphs)."""
    out = []
    for contigs, scaffolds, graphs in ((Contigs, Scaffolds, (G, G_prime) if param.extend_paths else (G,)),
                                       (small_contigs, small_scaffolds, (G_prime,))):
        objs = _objects_of(contigs)
        out.append((objs, _cached_column(contigs, objs, 'coverage', np.float64), scaffolds, graphs))
    return out


def filter_low_coverage_contigs(Contigs, Scaffolds, G, param, G_prime, small_contigs, small_scaffolds, Information):
    """-z_min (CreateGraph.py:407-433): a mask over each dictionary's coverage column."""
    print('Removing low coverage contigs if -z_min specified..', file=Information)
    low_coverage_contigs = []
    for objs, cov, scaffolds, graphs in _coverage_groups(Contigs, Scaffolds, G, G_prime, small_contigs, small_scaffolds, param):
        chosen = [objs[k] for k in np.flatnonzero(cov < param.lower_cov_cutoff).tolist()]
        _retire_scaffolds(chosen, scaffolds, graphs)
        low_coverage_contigs.extend(chosen)
    GO.PrintOut_low_cowerage_contigs(low_coverage_contigs, Contigs, param.output_directory, small_contigs)
    print('Removed a total of: ', len(low_coverage_contigs), ' low coverage contigs. With coverage lower than ',
          param.lower_cov_cutoff, file=Information)


def _acc(values):
    """sum() of the reference: left to right (np.sum adds pairwise and rounds differently)."""
    values = np.asarray(values, dtype=np.float64)
    return float(np.cumsum(values)[-1]) if values.size else 0


def _mean_and_std(xs):
    xs = np.asarray(xs, dtype=np.float64)
    n = float(xs.size)
    mean = _acc(xs) / n
    return mean, (_acc(xs ** 2 - 2 * xs * mean + mean ** 2) / (n - 1)) ** 0.5


def RemoveOutliers(mean_cov, std_dev, cov_list):
    k = MaxObsDistr(len(cov_list), 0.95)
    cov_list = np.asarray(cov_list, dtype=np.float64)
    filtered_list = cov_list[(cov_list < mean_cov + k * std_dev) & (cov_list < 2 * mean_cov)]
    return cov_list.size > filtered_list.size, filtered_list


def CalculateMeanCoverage(Contigs, Information, param):
    """Mean / sd of the coverage of the 50 000 longest contigs with the extreme ones trimmed away
    (CreateGraph.py:875-927), on a length and a coverage column."""
    objs = _objects_of(Contigs)
    lengths = _cached_column(Contigs, objs, 'length', np.int64)
    longest = np.argsort(-lengths, kind='stable')[:50000]                 # ties keep dictionary order, as sorted() does
    cov = _cached_column(Contigs, objs, 'coverage', np.float64)[longest]
    cov_of_longest_contigs = cov[cov > 0]
    if cov_of_longest_contigs.size <= 1:
        sys.exit('Too few contigs to calculate coverage on. Got: {0} contigs. If you have specified  -z_min or '
                 '--min_mapq, consider lower them. If not, check the BAM file for proper alignments. Exiting here '
                 'before scaffolding...'.format(int(cov_of_longest_contigs.size)))
    mean_cov, std_dev = _mean_and_std(cov_of_longest_contigs)
    n = float(cov_of_longest_contigs.size)
    print('Mean coverage before filtering out extreme observations = ', mean_cov, file=Information)
    print('Std dev of coverage before filtering out extreme observations= ', std_dev, file=Information)
    print('Number of contigs used in calc of coverage before filtering: ', n, file=Information)
    extreme_obs_occur = True
    while extreme_obs_occur:
        extreme_obs_occur, filtered_list = RemoveOutliers(mean_cov, std_dev, cov_of_longest_contigs)
        n = float(filtered_list.size)
        if n == 0 or _acc(filtered_list) == 0:
            break
        mean_cov, std_dev = _mean_and_std(filtered_list)
        cov_of_longest_contigs = filtered_list
    print('Mean coverage after filtering = ', mean_cov, file=Information)
    print('Std coverage after filtering = ', std_dev, file=Information)
    print('Number of contigs used in calc of coverage after filtering: ', n, file=Information)
    print('Length of longest contig in calc of coverage: ', int(lengths[longest[0]]), file=Information)
    print('Length of shortest contig in calc of coverage: ', int(lengths[longest[-1]]), file=Information)
    return (mean_cov, std_dev)


def RepeatDetector(Contigs, Scaffolds, G, param, G_prime, small_contigs, small_scaffolds, Information):
    """Contigs above the repeat coverage threshold leave the graphs, contigs at half coverage are marked as potential
    haplotypes (CreateGraph.py:959-1018): two masks over each dictionary's coverage column."""
    mean_cov, std_dev = param.mean_coverage, param.std_dev_coverage
    k = MaxObsDistr(len(Contigs), 0.95)
    repeat_thresh = param.cov_cutoff if param.cov_cutoff else max(mean_cov + k * std_dev, 2 * mean_cov - 3 * std_dev)
    print('Detecting repeats..', file=Information)
    Repeats = []
    count_hapl = 0
    for objs, cov, scaffolds, graphs in _coverage_groups(Contigs, Scaffolds, G, G_prime, small_contigs, small_scaffolds, param):
        chosen = [objs[j] for j in np.flatnonzero(cov > repeat_thresh).tolist()]
        _retire_scaffolds(chosen, scaffolds, graphs)
        Repeats.extend(chosen)
        if param.detect_haplotype:
            for j in np.flatnonzero(cov < mean_cov / 2.0 + param.hapl_threshold * std_dev).tolist():
                objs[j].is_haplotype = True
                count_hapl += 1
    GO.repeat_contigs_logger(Repeats, Contigs, param.output_directory, small_contigs, param)
    GO.PrintOutRepeats(Repeats, Contigs, param.output_directory, small_contigs)
    print('Removed a total of: ', len(Repeats), ' repeats. With coverage larger than ', repeat_thresh, file=Information)
    if param.detect_haplotype:
        print('Marked a total of: ', count_hapl, ' potential haplotypes.', file=Information)
    return (Contigs, Scaffolds, G)


def RemoveBugEdges(G, G_prime, links, param, Information):
    """Drop an edge when the BWA-quirk read count of its node pair reaches its link count (CreateGraph.py:690-708)."""
    edges_removed = 0
    if links.fishy_pair.size and len(links):
        by_pair = np.argsort(links.pair, kind='stable')
        at = np.searchsorted(links.pair[by_pair], links.fishy_pair)
        at[at >= by_pair.shape[0]] = 0
        k = by_pair[at]
        hit = (links.pair[k] == links.fishy_pair) & (links.fishy_n >= links.n[k])
        bug = np.zeros(len(links), dtype=bool)
        bug[k[hit]] = True
        if param.extend_paths:
            edges_removed = G_prime.drop(bug)
            G.drop(bug)
        else:
            edges_removed = G.drop(bug)
    print('Number of BWA buggy edges removed: ', edges_removed, file=Information)
    return ()


def infer_spurious_link_count_threshold(G_prime, param):
    nr_nodes = G_prime.number_of_nodes() / 2
    contamination_ratio = param.contamination_ratio if param.contamination_ratio else 0
    cov = param.mean_coverage * (1 - contamination_ratio)
    link_params = e_nr_links.Param(param.mean_ins_size, param.std_dev_ins_size, cov, param.read_len, 0)
    gap = param.mean_ins_size + param.std_dev_ins_size - 2 * param.read_len
    expected = e_nr_links.ExpectedLinks(100000, 100000, gap, link_params)
    link_number, count = np.unique(G_prime.links.n[G_prime.alive], return_counts=True)
    total_included_edges = 0
    for link_number, count in zip(link_number[::-1].tolist(), count[::-1].tolist()):
        total_included_edges += count
        print('Nodes: {0}.\t Total edges with over {1} links:{2}. \tAverage density: {3}'.format(
            nr_nodes, link_number, total_included_edges, total_included_edges / float(nr_nodes)),
            file=param.information_file)
    param.expected_links_over_mean_plus_stddev = 5 if expected < 5 else int(expected)
    print('Letting filtering threshold in high complexity regions be {0} for this library.'.format(
        param.expected_links_over_mean_plus_stddev), file=param.information_file)


def remove_edges_below_threshold(graph, param):
    """Dense-region pruning (CreateGraph.py:355-404).  Order dependent: the thin edges are visited in G.edges() order and
    an edge goes only while both its ends still have more than four neighbours - a sequential sweep over the thin edges
    with the degrees in a column (every node starts with its intra-scaffold edge plus its live links)."""
    print('Remove edges in high complexity areas.', file=param.information_file)
    lk = graph.links
    limit = param.expected_links_over_mean_plus_stddev
    thin, _ = graph.edges_order(lk.n < limit)
    removed = 0
    if thin.size:
        live = np.flatnonzero(graph.alive)
        degree = (np.bincount(np.concatenate([lk.u[live], lk.v[live]]), minlength=2 * graph.node_alive.shape[0]) + 1).tolist()
        gone = []
        for k, a, b in zip(thin.tolist(), lk.u[thin].tolist(), lk.v[thin].tolist()):
            if degree[a] > 4 and degree[b] > 4:
                degree[a] -= 1
                degree[b] -= 1
                gone.append(k)
        removed = len(gone)
        graph.alive[gone] = False
    print('Removed total of {0} edges in high density areas.'.format(removed), file=param.information_file)
    counter_low_support = graph.drop(lk.n < param.edgesupport)
    print('Removed an additional of {0} edges with low support from full graph G_prime of all contigs.'.format(
        counter_low_support), file=param.information_file)


def get_conditional_stddevs(steps, empirical_isize_distr, max_isize):
    """Expected std-dev of the spanning insert size given the gap, from the empirical distribution
    (CreateGraph.py:436-469): for every gap of `steps` the density f(x) * max(0, x - gap + 1), its mean and sigma (sums
    in index order, as the reference's sum() over the list), repeated for the gaps up to the next step."""
    expected = []
    previous_gap = 0
    items = list(empirical_isize_distr.items())
    for gap in steps:
        density = [0] * (max_isize + 1)
        for x, f_x in items:
            w_x = max(0, x - gap + 1)
            if w_x > 0:
                density[x] = f_x * w_x
        tot = 0
        for v in density:
            tot = tot + v
        tot = float(tot)
        acc = 0
        for i, v in enumerate(density):
            acc = acc + i * v
        mu = acc / tot
        acc = 0
        for i, v in enumerate(density):
            acc = acc + (i - mu) ** 2 * v
        sigma = math.sqrt(acc / tot)
        expected.extend([sigma] if gap == 0 else [sigma] * (gap - previous_gap))
        previous_gap = gap
    return expected


def dense_distribution(empirical_isize_distr, max_isize):
    """param.empirical_distribution as a float64 column indexed by insert size (0 where it has no entry)."""
    f = np.zeros(max_isize + 1, dtype=np.float64)
    keys = np.fromiter(empirical_isize_distr.keys(), dtype=np.int64, count=len(empirical_isize_distr))
    vals = np.fromiter(empirical_isize_distr.values(), dtype=np.float64, count=len(empirical_isize_distr))
    ok = (keys >= 0) & (keys <= max_isize)                   # (a negative insert size never gets a positive weight, :446)
    f[keys[ok]] = vals[ok]
    return f


def expand_conditional_stddevs(steps, sigmas):
    """The flattened list of CreateGraph.py:460-467: the sigma of a step stands for the gaps since the step before."""
    expected = []
    previous_gap = 0
    for gap, sigma in zip(steps, np.asarray(sigmas).tolist()):
        expected.extend([sigma] if gap == 0 else [sigma] * (gap - previous_gap))
        previous_gap = gap
    return expected


def GiveScoreOnEdges(G, Scaffolds, small_scaffolds, Contigs, param, Information, plot, ctx):
    """Score every link edge of G (CreateGraph.py:473-614, normal and log-normal branch).

    The device returns per edge the ML gap, the expected std-dev and the integer KS numerator h; the
    remaining scalar arithmetic below is evaluated with the reference's expressions so the floats agree.
    """
    lognormal = None
    if param.lognormal:
        # skewed library (libmetrics: skew_adj > 0.5): gaps from the log-normal estimator over the raw observations,
        # expected sigma from the empirical distribution conditioned on the gap (CreateGraph.py:485-494).  The
        # reference's `range(0, int(max_isize*0.8), max_isize/50)` is Python 2 integer division.  Both run on the
        # device: one workgroup per step for the sigmas, one per edge for the gaps.
        emp_distr = param.empirical_distribution
        max_isize = sorted(emp_distr.keys())[-1]
        steps = list(range(0, int(max_isize * 0.8), max_isize // 50))
        cond_sd = expand_conditional_stddevs(steps, ctx.conditional_stddevs(dense_distribution(emp_distr, max_isize), steps))
        log_norm_max_gap = len(cond_sd) - 1
        lognormal = (param.lognormal_mean, param.lognormal_sigma,
                     mathstats_compat.lognormal_support(param.lognormal_mean, param.lognormal_sigma), log_norm_max_gap)
    score_file = None
    if param.print_scores:
        score_file = open(os.path.join(param.output_directory, 'score_file_pass_{0}.tsv'.format(param.pass_number)), 'w')
        print('scf1/ctg1\to1\tscf2/ctg2\to2\tgap\tlink_variation_score\tlink_dispersity_score\tnumber_of_links',
              file=score_file)

    lk = G.links
    idx, from_v = G.edges_order()                            # G.edges(): every link, reported from its first endpoint
    first = np.where(from_v, lk.v[idx], lk.u[idx])
    second = np.where(from_v, lk.u[idx], lk.v[idx])
    s_length = np.zeros(G.node_alive.shape[0], dtype=np.int64)
    for group in (small_scaffolds, Scaffolds):
        if group:
            ids = np.fromiter(group, dtype=np.int64, count=len(group))
            ids_in = ids < s_length.shape[0]
            s_length[ids[ids_in]] = _column(list(group.values()), 's_length', np.int64)[ids_in]
    len1_a, len2_a = s_length[first >> 1], s_length[second >> 1]
    # l1 belongs to the first endpoint (:568-579); the device keeps the observations of the smaller node code first
    gap_d, sd0_d, ks_h, flags = ctx.score_edges(lk.rows[idx], from_v.astype(np.uint8), len1_a, len2_a, param.mean_ins_size,
                                                param.std_dev_ins_size, param.read_len, lognormal=lognormal)
    if lognormal is not None:
        # :549-553 - conditional_stddevs[int(gap)] for a positive gap, [0] otherwise (only read where flags & 1)
        at = np.where((gap_d > 0) & ((flags & 1) != 0), gap_d, 0.0).astype(np.int64)
        sd0_d = np.asarray(cond_sd, dtype=np.float64)[at]
    gap_d, sd0_d, ks_h, flags = gap_d.tolist(), sd0_d.tolist(), ks_h.tolist(), flags.tolist()
    len1, len2 = len1_a.tolist(), len2_a.tolist()
    n_l, obs_l, obs_sq_l = lk.n[idx].tolist(), lk.obs[idx].tolist(), lk.obs_sq[idx].tolist()
    gaps, scores = [0] * len(n_l), [None] * len(n_l)
    side = ('L', 'R')
    for j, n in enumerate(n_l):
        mean_ = obs_l[j] / float(n)
        # integer-valued when an ML estimator was used (int in the reference), float otherwise
        gap = int(gap_d[j]) if flags[j] & 1 else gap_d[j]
        gaps[j] = int(gap)
        if flags[j] & 2:                      # -gap > len1 or -gap > len2
            scores[j] = 0
            continue
        std_dev_d_eq_0 = sd0_d[j] if flags[j] & 1 else 2 ** 32
        try:
            std_dev = ((obs_sq_l[j] - n * mean_ ** 2) / (n - 1)) ** 0.5
        except ZeroDivisionError:
            std_dev = 2 ** 32
        span_score = 0 if n < 5 else 1 - ks_h[j] * 1.0 / n
        try:
            std_dev_score = min(std_dev / std_dev_d_eq_0, std_dev_d_eq_0 / std_dev)
        except ZeroDivisionError:
            std_dev_score = 0
            sys.stderr.write(str(std_dev) + ' ' + str(std_dev_d_eq_0) + ' ' + str(span_score) + '\n')
        scores[j] = std_dev_score + span_score if std_dev_score > 0.5 and span_score > 0.5 else 0
        if score_file is not None:
            # --print_scores rows as the reference writes them (:622-651): the contig names of each scaffold, listed
            # from the far end towards the link, and one sign per contig - '+' throughout for an 'R' end on the left
            # and an 'L' end on the right, '-' throughout otherwise (the reference's `'+' if True else '-'`)
            u = (int(first[j]) >> 1, side[int(first[j]) & 1])
            v = (int(second[j]) >> 1, side[int(second[j]) & 1])
            objs1 = Scaffolds[u[0]].contigs if u[1] == 'R' else Scaffolds[u[0]].contigs[::-1]
            objs2 = Scaffolds[v[0]].contigs if v[1] == 'L' else Scaffolds[v[0]].contigs[::-1]
            print('{0}\t{1}\t{2}\t{3}\t{4}\t{5}\t{6}\t{7}'.format(
                ';'.join(c.name for c in objs1), ';'.join(('+' if u[1] == 'R' else '-') for _ in objs1),
                ';'.join(c.name for c in objs2), ';'.join(('+' if v[1] == 'L' else '-') for _ in objs2),
                gap, std_dev_score, span_score, n), file=score_file)
    if score_file is not None:
        score_file.close()
    print('Number of significantly spurious edges:', 0, file=Information)
    return idx, gaps, scores
