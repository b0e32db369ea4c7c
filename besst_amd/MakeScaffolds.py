"""Scaffold-graph linearisation (steps 1-4 of the reference's MakeScaffolds.Algorithm) on the MI355X.

Mirror of the reference's functions of the same names (BESST/MakeScaffolds.py):

    RemoveIsolatedContigs(G, Information)                                       :134-144   step 1 / 3
    RemoveAmbiguousRegionsUsingScore(G, G_prime, Information, param, plot)      :206-241   step 2 (+ remove_edges :156-204)
    RemoveLoops(G, G_prime, Scaffolds, Contigs, Information, param)             :248-274   step 4
    LinearizeGraph(...)   steps 1, 2, 3, 4 in the order of Algorithm (:75-82) with ONE device call
    NewContigsScaffolds(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, Information,
                        dValuesTable, param, already_visited)                   :270-341   step 5 (+ UpdateInfo :344-482)

They take and mutate the same ``networkx``-1.x style graphs CreateGraph.PE returns (``besst_amd.nxcompat.Graph``),
print the same lines to ``Information`` and leave the same graphs behind; the work itself runs in
``besst_linearize`` (besst_amd/csrc/linearize.hip) on arrays extracted from the graph.  There is no CPU path:
without libbesst_amd.so or a GPU the calls raise ``besst_amd._lib.BesstDeviceError``.

Not mirrored: the ``param.plots`` histograms of the decision scores (:231-234), and the text of the
'A cycle in the scaffold graph' lines lists a cycle's nodes in walking order rather than networkx's.
"""
from __future__ import print_function

import numpy as np

from . import _lib

STEP1, STEP2, STEP3, STEP4 = 1, 2, 4, 8


def linearize_arrays(n_scaffolds, a, b, score, steps=STEP1 | STEP2 | STEP3 | STEP4, device=0):
    """besst_linearize on host arrays.  Nodes are compact ids (scaffold k: 2k = 'L', 2k+1 = 'R'); a/b/score are the
    link edges in G.edges() order.  Returns a dict: alive2 (per edge), removed_by (per scaffold: 0 kept, else the
    step that removed it), present (= removed_by == 0), isolated [step 1, step 3], cycles, rounds, ambivalent
    [(top, second)] in the reference's visiting order."""
    lib = _lib.load()
    a = np.ascontiguousarray(a, np.int32)
    b = np.ascontiguousarray(b, np.int32)
    score = np.ascontiguousarray(score, np.float64)
    m, n = int(a.shape[0]), int(n_scaffolds)
    if b.shape[0] != m or score.shape[0] != m:
        raise ValueError('a, b and score must have one entry per edge')
    if m and (min(int(a.min()), int(b.min())) < 0 or max(int(a.max()), int(b.max())) >= 2 * n):
        raise ValueError('node id out of range')
    alive = np.zeros(max(m, 1), np.uint8)
    removed_by = np.zeros(max(n, 1), np.uint8)
    amb = np.zeros(max(2 * n, 1), np.uint8)
    top = np.zeros(max(2 * n, 1), np.float64)
    second = np.zeros(max(2 * n, 1), np.float64)
    best = np.zeros(max(2 * n, 1), np.uint32)
    counters = np.zeros(8, np.int64)
    _lib.check(lib.besst_linearize(device, steps, n, m, _lib.ptr(a), _lib.ptr(b), _lib.ptr(score), _lib.ptr(alive),
                                   _lib.ptr(removed_by), _lib.ptr(amb), _lib.ptr(top), _lib.ptr(second), _lib.ptr(best),
                                   _lib.ptr(counters)), 'besst_linearize')
    events = []
    nodes = np.nonzero(amb[:2 * n])[0]
    if nodes.shape[0]:
        # visiting order of the reference: the node's best edge by (score desc, edge index asc), edge[0] first
        be = best[nodes].astype(np.int64)
        is_b = (b[be] == nodes).astype(np.int64)
        order = np.lexsort((is_b, be, -score[be]))
        events = [(float(top[x]), float(second[x])) for x in nodes[order]]
    return dict(alive2=alive[:m].astype(bool), removed_by=removed_by[:n].copy(), present=removed_by[:n] == 0,
                isolated=[int(counters[0]), int(counters[1])], cycles=int(counters[2]), rounds=int(counters[4]),
                ambivalent=events)


class _GraphArrays(object):
    """Link edges of a graph as arrays.  Scaffold k = k-th distinct scaffold in node order."""

    def __init__(self, G, scored_only):
        cols = G.link_columns() if hasattr(G, 'link_columns') else None
        if cols is not None and self._from_columns(cols, scored_only):
            return
        self.index = {}
        for s, _ in G.nodes():
            self.index.setdefault(s, len(self.index))
        self.scaffolds = list(self.index)
        a, b, score, edges = [], [], [], []
        unscored = 0
        for u, v, d in G.edges(data=True):
            if d['nr_links'] is None:
                continue
            if 'score' not in d:
                unscored += 1
                if scored_only:
                    continue
            a.append(self.node(u))
            b.append(self.node(v))
            score.append(d.get('score', 1.0))
            edges.append((u, v))
        self.a, self.b, self.score, self.edges, self.unscored = a, b, score, edges, unscored

    def _from_columns(self, cols, scored_only):
        """A graph straight from CreateGraph.PE, not read or changed since: its link edges are still columns
        (CreateGraph.GraphColumns) - the arrays come from them, no neighbour or attribute dictionary is made.  A node's
        code here (2 x scaffold index + side) IS its slot in the columns' node order."""
        nodes = cols.nodes
        self.scaffolds = [n[0] for n in nodes[0::2]]
        self.index = dict(zip(self.scaffolds, range(len(self.scaffolds))))
        if len(self.index) != len(self.scaffolds):
            return False                                     # (a scaffold listed twice: the general way)
        a, b, pos = cols.link_arrays()
        vals = cols.link_scores()
        if vals is None:
            self.unscored = int(pos.shape[0])
            if scored_only:
                a, b, pos, score = a[:0], b[:0], pos[:0], []
            else:
                score = [1.0] * int(pos.shape[0])
        else:
            score_all = [vals[p] for p in pos.tolist()]
            if any(v.__class__ is object for v in score_all):
                return False                                 # (links without a score among scored ones: the general way)
            self.unscored, score = 0, score_all
        self.a, self.b, self.score = a.tolist(), b.tolist(), score
        self.edges = [(nodes[i], nodes[j]) for i, j in zip(self.a, self.b)]
        self.from_columns = True
        return True

    def node(self, n):
        return 2 * self.index[n[0]] + (n[1] == 'R')

    @property
    def n_scaffolds(self):
        return len(self.index)


def _remove_scaffolds(G, arrays, present):
    gone = [s for s, keep in zip(arrays.scaffolds, present) if not keep]
    for s in gone:
        G.remove_nodes_from([(s, 'L'), (s, 'R')])
    return gone


def RemoveIsolatedContigs(G, Information, device=0):
    print('Remove isolated nodes.', file=Information)
    arr = _GraphArrays(G, scored_only=False)
    res = linearize_arrays(arr.n_scaffolds, arr.a, arr.b, arr.score, STEP1, device)
    _remove_scaffolds(G, arr, res['present'])
    print(str(res['isolated'][0]) + ' isolated contigs removed from graph.', file=Information)
    return G


def _apply_step2(G, G_prime, Information, param, arr, res, nr_edges_before):
    for top, second in res['ambivalent']:
        print('SCORES AMBVIVALENT', top, second, file=Information)
    dropped = [e for e, keep in zip(arr.edges, res['alive2']) if not keep]
    G.remove_edges_from(dropped)
    if param.extend_paths:
        G_prime.remove_edges_from(dropped)                  # edges G_prime never had are ignored (:174-177)
    nr_edges_after = len(G.edges())
    print(' Number of edges in G before:', nr_edges_before, file=Information)
    print(' Number of edges in G after:', nr_edges_after, file=Information)
    try:
        print(' %-age removed edges:', 100 * (1 - (nr_edges_after / float(nr_edges_before))), file=Information)
    except ZeroDivisionError:
        pass


def _scored_arrays(G):
    arr = _GraphArrays(G, scored_only=True)
    if arr.unscored and arr.edges:
        # the reference reads G[node][nbr]['score'] of every link edge at a visited node (:161)
        raise KeyError('score')
    return arr


def RemoveAmbiguousRegionsUsingScore(G, G_prime, Information, param, plot, device=0):
    nr_edges_before = len(G.edges())
    print('Remove edges from node if more than two edges', file=Information)
    arr = _scored_arrays(G)
    res = linearize_arrays(arr.n_scaffolds, arr.a, arr.b, arr.score, STEP2, device)
    _apply_step2(G, G_prime, Information, param, arr, res, nr_edges_before)
    return ()


def _cycles_of(arr, alive, gone_scaffolds):
    """Node lists of the cycles among the removed scaffolds (for the log lines only)."""
    gone = set(gone_scaffolds)
    mate = {}
    for (u, v), keep in zip(arr.edges, alive):
        if keep and u[0] in gone and v[0] in gone:
            mate[u], mate[v] = v, u
    cycles, seen = [], set()
    for s in gone_scaffolds:
        if s in seen:
            continue
        cyc, x = [], (s, 'L')
        while x[0] not in seen:
            seen.add(x[0])
            other = (x[0], 'R' if x[1] == 'L' else 'L')
            cyc += [x, other]
            x = mate[other]
        cycles.append(cyc)
    return cycles


def _apply_step4(G, G_prime, Information, param, arr, alive, present_before, res):
    gone = [s for s, was, keep in zip(arr.scaffolds, present_before, res['present']) if was and not keep]
    cycles = _cycles_of(arr, alive, gone)
    for cycle in cycles:
        print('A cycle in the scaffold graph: ' + str(cycle) + '\n', file=Information)
        print('A cycle in the scaffold graph: ' + str(cycle), file=Information)
    for s in gone:
        G.remove_nodes_from([(s, 'L'), (s, 'R')])
        if param.extend_paths:
            G_prime.remove_nodes_from([(s, 'L'), (s, 'R')])
    print(str(res['cycles']) + ' cycles removed from graph.', file=Information)


def RemoveLoops(G, G_prime, Scaffolds, Contigs, Information, param, device=0):
    print('Contigs/scaffolds left:', len(G.nodes()) / 2, file=Information)
    print('Remove remaining cycles...', file=Information)
    arr = _GraphArrays(G, scored_only=False)
    res = linearize_arrays(arr.n_scaffolds, arr.a, arr.b, arr.score, STEP4, device)
    _apply_step4(G, G_prime, Information, param, arr, [True] * len(arr.edges), [True] * arr.n_scaffolds, res)
    return (G, Contigs, Scaffolds)


def LinearizeGraph(G, G_prime, Contigs, Scaffolds, Information, param, device=0):
    """Steps 1-4 as MakeScaffolds.Algorithm runs them when scoring is on (:75-82), with one device call.
    Returns (G, Contigs, Scaffolds) like RemoveLoops."""
    arr = _scored_arrays(G)
    res = linearize_arrays(arr.n_scaffolds, arr.a, arr.b, arr.score, STEP1 | STEP2 | STEP3 | STEP4, device)
    alive = res['alive2']
    # step 1
    print('Remove isolated nodes.', file=Information)
    after1 = (res['removed_by'] != 1).tolist()
    _remove_scaffolds(G, arr, after1)
    print(str(res['isolated'][0]) + ' isolated contigs removed from graph.', file=Information)
    # step 2
    nr_edges_before = len(G.edges())
    print('Remove edges from node if more than two edges', file=Information)
    _apply_step2(G, G_prime, Information, param, arr, res, nr_edges_before)
    # step 3
    print('Remove isolated nodes.', file=Information)
    after3 = [r != 1 and r != 3 for r in res['removed_by'].tolist()]
    _remove_scaffolds(G, arr, after3)
    print(str(res['isolated'][1]) + ' isolated contigs removed from graph.', file=Information)
    # step 4
    print('Contigs/scaffolds left:', len(G.nodes()) / 2, file=Information)
    print('Remove remaining cycles...', file=Information)
    _apply_step4(G, G_prime, Information, param, arr, alive, after3, res)
    return (G, Contigs, Scaffolds)


# -----------------------------------------------------------------------------------------------------------------
# step 5: the linear paths become the new scaffolds
# -----------------------------------------------------------------------------------------------------------------
def chain_arrays(n_scaffolds, link, gap, scaffold_length, node_order, device=0):
    """besst_chain_scaffolds on host arrays (include/besst_amd.h) -> (terminal, beyond, lowest_order, passes)."""
    lib = _lib.load()
    n = int(n_scaffolds)
    link = np.ascontiguousarray(link, np.int32)
    gap = np.ascontiguousarray(gap, np.int32)
    slen = np.ascontiguousarray(scaffold_length, np.int32)
    order = np.ascontiguousarray(node_order, np.int32)
    if link.shape[0] != 2 * n or gap.shape[0] != 2 * n or slen.shape[0] != n or order.shape[0] != 2 * n:
        raise ValueError('chain_arrays: array sizes do not match the scaffold count')
    terminal = np.zeros(max(2 * n, 1), np.int32)
    beyond = np.zeros(max(2 * n, 1), np.int64)
    lowest = np.zeros(max(2 * n, 1), np.int32)
    import ctypes as C
    passes = C.c_int32(0)
    _lib.check(lib.besst_chain_scaffolds(device, n, _lib.ptr(link), _lib.ptr(gap), _lib.ptr(slen), _lib.ptr(order),
                                         _lib.ptr(terminal), _lib.ptr(beyond), _lib.ptr(lowest), C.byref(passes)),
               'besst_chain_scaffolds')
    return terminal[:2 * n], beyond[:2 * n], lowest[:2 * n], int(passes.value)


def _edge_gap(data, c1_len, c2_len, dValuesTable, param):
    """The gap UpdateInfo adds when it crosses a link edge (:413-471) -> (avg_gap, value appended to
    param.gap_estimations or None).  Symmetric in the two scaffolds."""
    from . import mathstats_compat as GC
    if 'avg_gap' in data:
        return data['avg_gap'], data['avg_gap']
    if param.lognormal:
        return GC.lognormal_GapEstimator(param.lognormal_mean, param.lognormal_sigma, param.read_len,
                                         data['observations'], c1_len, c2_len=c2_len), None
    sum_obs, nr_links = data['obs'], data['nr_links']
    data_observation = (nr_links * param.mean_ins_size - sum_obs) / float(nr_links)
    mean_obs = sum_obs / float(nr_links)
    if param.std_dev_ins_size and nr_links >= 5:
        far = param.mean_ins_size + 4 * param.std_dev_ins_size
        near = param.std_dev_ins_size + param.read_len
        if c1_len > far and c2_len > far:
            try:
                return dValuesTable[int(round(data_observation, 0))], None
            except (KeyError, TypeError):                # TypeError: no table was built (dValuesTable is None)
                return GC.GapEstimator(param.mean_ins_size, param.std_dev_ins_size, param.read_len, mean_obs, c1_len,
                                       c2_len), None
        if c1_len > near and c2_len > near:
            return GC.GapEstimator(param.mean_ins_size, param.std_dev_ins_size, param.read_len, mean_obs, c1_len,
                                   c2_len), None
    return int(data_observation), int(data_observation)


def _relabel_path_ends(G_prime, name, start, end, old_nodes):
    """The new scaffold's two nodes in G_prime, with the link edges of the path's two ends (MakeScaffolds.py:309-339)."""
    import networkx as nx
    G_prime.add_node((name, 'L'))
    G_prime.add_node((name, 'R'))
    G_prime.add_edge((name, 'L'), (name, 'R'), nr_links=None)
    try:
        for new_side, old in (('L', start), ('R', end)):
            for nbr in G_prime.neighbors(old):
                d = G_prime[old][nbr]
                if d['nr_links']:
                    G_prime.add_edge((name, new_side), nbr, nr_links=d['nr_links'], obs=d['obs'],
                                     obs_sq=d['obs_sq'], observations=d['observations'])
        G_prime.remove_nodes_from(old_nodes)
    except (nx.exception.NetworkXError, KeyError):
        pass


def _extend_within_components(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, param, dValuesTable,
                              already_visited, within_scaffold):
    """What the reference's loop does per component apart from the walk (:272-339), for all components, in its order:
    the path search between the component's scaffolds (`within_scaffold` = BESST's PROWithinScaf: it moves small
    scaffolds into G and out of G_prime), then the relabelling of the path's two ends in G_prime to the scaffold the
    component becomes.  The searches only read G_prime, the scaffold lengths of their own component and
    `already_visited`; the walk of an earlier component (contig positions, the Scaffolds entries of ITS scaffolds) is
    nothing a later search looks at, so the walks can wait until every component has been extended.

    The callback's CONTRACT (what the equivalence with the reference's interleaved search / walk / relabel rests on; BESST's
    PROWithinScaf keeps it: its get_total_length only looks a scaffold up in small_scaffolds): it must not look up
    Scaffolds[id] for an id a previous component was renamed to - that object does not exist before the walks -, and it must
    not read param.scaffold_indexer, which advances only with the walks (the names handed to G_prime here count on from it
    in a local).  A component without a degree-1 node (a single scaffold whose two ends are not in G as a path) is left
    as it is; the reference would go on with the start / end of the component before - a state no test of BESST's shows."""
    import networkx as nx
    components = [G.subgraph(c) for c in nx.connected_components(G)]
    name = param.scaffold_indexer
    for component in components:
        name += 1
        within_scaffold(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, param, component, dValuesTable,
                        already_visited)
        ends = [node for node in component if len(G.neighbors(node)) == 1]
        if not ends:
            continue
        # (:287-293: the first such node is the start, the last other one the end; a single scaffold has both of its own)
        _relabel_path_ends(G_prime, name, ends[0], ends[-1] if len(ends) > 1 else ends[0], list(component))


def NewContigsScaffolds(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, Information, dValuesTable, param,
                        already_visited, device=0, within_scaffold=None):
    """Mirror of MakeScaffolds.NewContigsScaffolds (:270-341) with UpdateInfo (:344-482): every connected component of
    the linearised G - a path of scaffolds - becomes one new scaffold.  Same arguments, same mutations: contigs get the
    new scaffold id, their position along the path and (for scaffolds the walk enters through 'R') the flipped
    direction; the old scaffold objects are deleted, the new ones appended; G loses the nodes of every component; with
    param.extend_paths G_prime gets the new scaffold's two nodes with the link edges of the path's ends.  The walk
    itself - which end a path starts from, every scaffold's position and orientation - comes from the device
    (besst_chain_scaffolds, list ranking); the per-edge gap values follow the reference's rules on the host.

    PROWithinScaf (:283-285), the path search the reference runs per component when param.extend_paths, is BESST's
    own sequential code and is not rebuilt here: hand it in as ``within_scaffold`` (same signature) and it runs per
    component, in the reference's order, before the chains are extracted (_extend_within_components).  Without it the
    components are taken as they are - with param.extend_paths set (BESST's default) that is NOT what BESST computes,
    and the Information log says so."""
    from . import Scaffold
    relabelled = False
    if param.extend_paths:
        if within_scaffold is not None:
            _extend_within_components(G, G_prime, Contigs, small_contigs, Scaffolds, small_scaffolds, param, dValuesTable,
                                      already_visited, within_scaffold)
            relabelled = True
        else:
            print('WARNING: extend_paths is set but no within_scaffold (PROWithinScaf) was handed to NewContigsScaffolds: '
                  'small contigs are not placed inside the new scaffolds in this step.', file=Information)
    nodes = G.nodes()
    order_of = {n: i for i, n in enumerate(nodes)}
    index = {}
    for s, _ in nodes:
        index.setdefault(s, len(index))
    scaffolds = list(index)
    n = len(scaffolds)
    if n == 0:
        print('Nr of new scaffolds created in this step: 0', file=Information)
        return (Contigs, Scaffolds, param)

    def code(node):
        return 2 * index[node[0]] + (node[1] == 'R')
    link = np.full(2 * n, -1, np.int32)
    gap = np.zeros(2 * n, np.int32)
    appended = {}
    slen = np.array([Scaffolds[s].s_length for s in scaffolds], np.int64)
    order = np.zeros(2 * n, np.int32)
    for node, i in order_of.items():
        order[code(node)] = i
    for u, v, data in G.edges(data=True):
        if u[0] == v[0]:                                     # the edge inside a scaffold (the path search's insertions carry
            continue                                         # no 'nr_links' at all: the walk tells the two kinds apart by ends)
        avg_gap, app = _edge_gap(data, Scaffolds[u[0]].s_length, Scaffolds[v[0]].s_length, dValuesTable, param)
        if avg_gap <= 1:
            avg_gap = 1
        cu, cv = code(u), code(v)
        link[cu], link[cv] = cv, cu
        gap[cu] = gap[cv] = int(avg_gap)
        appended[cu] = appended[cv] = app
    terminal, beyond, lowest, _ = chain_arrays(n, link, gap, slen, order, device)
    k = np.arange(n)
    ord_l, ord_r = order[terminal[2 * k]], order[terminal[2 * k + 1]]
    from_l = ord_l < ord_r                                   # the path's start lies beyond this scaffold's 'L' end
    pos = np.where(from_l, beyond[2 * k], beyond[2 * k + 1])
    comp = np.minimum(np.minimum(lowest[2 * k], lowest[2 * k + 1]), np.minimum(order[2 * k], order[2 * k + 1]))
    comp_ids, comp_of = np.unique(comp, return_inverse=True)     # ascending first node = nx.connected_components order
    print('Nr of new scaffolds created in this step: ' + str(len(comp_ids)), file=Information)
    members = [[] for _ in comp_ids]
    for j in np.lexsort((pos, comp_of)).tolist():
        members[comp_of[j]].append(j)
    node_of = {}
    for node in nodes:
        node_of[code(node)] = node
    for path in members:
        param.scaffold_indexer += 1
        contig_list = []
        first, last = path[0], path[-1]
        start = node_of[2 * first + (0 if from_l[first] else 1)]       # the terminal the walk starts from
        end = node_of[2 * last + (1 if from_l[last] else 0)]
        for a, j in enumerate(path):
            s = scaffolds[j]
            obj = Scaffolds[s]
            p0 = int(pos[j])
            if from_l[j]:                                    # entered through 'L': same orientation (:363-378)
                for contig in obj.contigs:
                    contig.scaffold = param.scaffold_indexer
                    contig.position += p0
                    contig_list.append(contig)
            else:                                            # entered through 'R': flipped (:384-404)
                for contig in obj.contigs:
                    contig.scaffold = param.scaffold_indexer
                    contig.position = p0 + (obj.s_length - contig.position) - contig.length
                    contig.direction = bool(True - contig.direction)
                    contig_list.append(contig)
            if a + 1 < len(path):                            # the edge the walk crosses next
                app = appended[2 * j + (1 if from_l[j] else 0)]
                if app is not None:
                    param.gap_estimations.append(app)
            del Scaffolds[s]
        longest = max(contig_list, key=lambda c: c.position + c.length)
        S = Scaffold.scaffold(param.scaffold_indexer, contig_list, longest.position + longest.length)
        Scaffolds[S.name] = S
        old_nodes = [node_of[2 * j + side] for j in path for side in (0, 1)]
        old_nodes.sort(key=lambda nd: order_of[nd])
        G.remove_nodes_from(old_nodes)
        if param.extend_paths and not relabelled:
            _relabel_path_ends(G_prime, S.name, start, end, old_nodes)
    return (Contigs, Scaffolds, param)
