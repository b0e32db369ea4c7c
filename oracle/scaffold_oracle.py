"""CPU restatement of the scaffold-graph linearisation steps 1-4 of the reference.

TEST INFRASTRUCTURE ONLY - imported by tests/ (and bench.py's cpu_baseline leg); the product path
(besst_amd/MakeScaffolds.py -> besst_linearize in libbesst_amd.so) never touches this module.

Pinned against tests/golden/scaffold_steps.json.gz, which tests/golden/make_scaffold_golden.py captured by
running the reference's own functions (imported from /root/reference) on seeded graphs.

Array interface (shared with the device path)
  nodes      compact ids: scaffold k (in any fixed numbering) has the nodes 2k ('L') and 2k+1 ('R'); the
             sibling of x is x ^ 1.  Every scaffold has its intra-scaffold edge (nr_links=None in the reference).
  a, b       the link edges of G that carry a 'score', in ``G.edges()`` order, a = edge[0], b = edge[1]
  score      float64 score per edge
The functions follow the reference literally and sequentially; nothing here is vectorised on purpose.
"""


def remove_isolated(n_scaf, present, deg):
    """RemoveIsolatedContigs (MakeScaffolds.py:134-144): a node whose only neighbour is its sibling, whose only
    neighbour is the node, goes together with the sibling.  ``deg`` = live link edges per node.  Returns the
    number of removed scaffolds (the reference's counter) and updates ``present``."""
    removed = 0
    for k in range(n_scaf):
        if present[k] and deg[2 * k] == 0 and deg[2 * k + 1] == 0:
            present[k] = False
            removed += 1
    return removed


def remove_ambiguous(n_nodes, a, b, score):
    """RemoveAmbiguousRegionsUsingScore + remove_edges (MakeScaffolds.py:156-241).

    Edges are visited by descending score (Python's stable sort, reverse=True keeps G.edges() order among
    equals, :216-217); for every edge, remove_edges runs on edge[0] and then on edge[1] (:219-221) - whether or
    not the edge itself still exists.  At a node: the zero-scoring edges go (:164-172, 'zero' = not 0 < score);
    of the others, if there are at least two and second/top > 0.8 all of them go, otherwise all but the top one
    (:181-188).  Returns (alive flags per edge, ambivalent events [(top, second)] in visiting order)."""
    m = len(a)
    alive = [True] * m
    adj = [[] for _ in range(n_nodes)]
    for i in range(m):
        adj[a[i]].append(i)
        adj[b[i]].append(i)
    order = sorted(range(m), key=lambda i: score[i], reverse=True)
    ambivalent = []

    def visit(x):
        mine = [i for i in adj[x] if alive[i]]
        scored = sorted((score[i], i) for i in mine)            # ties: the reference sorts (score, nbr); the order
        non_zero = [t for t in scored if 0 < t[0]]              # among equal scores never changes the outcome
        for s, i in scored:
            if not 0 < s:
                alive[i] = False
        if len(non_zero) > 1:
            if non_zero[-2][0] / non_zero[-1][0] > 0.8:
                ambivalent.append((non_zero[-1][0], non_zero[-2][0]))
                drop = non_zero
            else:
                drop = non_zero[:-1]
            for _, i in drop:
                alive[i] = False

    for i in order:
        visit(a[i])
        visit(b[i])
    return alive, ambivalent


def cycle_scaffolds(n_scaf, present, a, b, alive):
    """RemoveLoops (MakeScaffolds.py:248-274): every scaffold with a node on a cycle of the cycle basis goes.
    After step 2 a node has at most one link edge, so a component is a path or one cycle; the union of the
    basis cycles' nodes is the set of nodes on non-bridge edges, independent of the traversal order.
    Returns (number of cycles, list of scaffold indices on cycles)."""
    mate = [-1] * (2 * n_scaf)
    for i in range(len(a)):
        if alive[i] and present[a[i] >> 1] and present[b[i] >> 1]:
            assert mate[a[i]] < 0 and mate[b[i]] < 0, 'step 2 leaves at most one link edge per node'
            mate[a[i]] = b[i]
            mate[b[i]] = a[i]
    seen = [False] * n_scaf
    cycles, on_cycle = 0, []
    for k in range(n_scaf):
        if not present[k] or seen[k]:
            continue
        # walk from scaffold k in one direction until an end or back at k
        comp, x, closed = [], 2 * k, False
        while True:
            comp.append(x >> 1)
            y = mate[x ^ 1]
            if y < 0:
                break
            if (y >> 1) == k:
                closed = (y == 2 * k)
                break
            x = y
        if closed:
            cycles += 1
            on_cycle.extend(comp)
            for c in comp:
                seen[c] = True
    return cycles, on_cycle


def linearize(n_scaf, a, b, score):
    """Steps 1-4 in the order of MakeScaffolds.Algorithm (:75-82).  Returns a dict with
      alive2       per-edge flag after step 2
      present      per-scaffold flag after step 4
      isolated     [removed by step 1, removed by step 3]
      cycles       number of cycles step 4 found
      ambivalent   [(top, second)] in visiting order"""
    m = len(a)
    present = [True] * n_scaf
    deg = [0] * (2 * n_scaf)
    for i in range(m):
        deg[a[i]] += 1
        deg[b[i]] += 1
    iso1 = remove_isolated(n_scaf, present, deg)
    alive, amb = remove_ambiguous(2 * n_scaf, a, b, score)
    deg = [0] * (2 * n_scaf)
    for i in range(m):
        if alive[i]:
            deg[a[i]] += 1
            deg[b[i]] += 1
    iso3 = remove_isolated(n_scaf, present, deg)
    cycles, on_cycle = cycle_scaffolds(n_scaf, present, a, b, alive)
    for k in on_cycle:
        present[k] = False
    return dict(alive2=alive, present=present, isolated=[iso1, iso3], cycles=cycles, ambivalent=amb)


def new_contigs_scaffolds(node_order, link, gap, appended, slen, contigs, scaffold_indexer):
    """NewContigsScaffolds + UpdateInfo (MakeScaffolds.py:270-341, 344-482) on arrays, walked sequentially like the
    reference (TEST INFRASTRUCTURE; the device path ranks the lists in parallel, besst_amd/csrc/chain.hip).

      node_order  nodes (2k / 2k+1 of scaffold k) in G.nodes() order
      link        per node: the node at the other end of its link edge, -1 without one (after steps 1-4: a set of paths)
      gap         per node: what crossing that edge adds to the position (max(1, int(avg_gap)), :468-471)
      appended    per node: the value that branch appends to param.gap_estimations, or None (:413-467)
      slen        per scaffold: s_length
      contigs     per scaffold: list of [name, position, direction, length]  (Scaffold.contigs order)
    Returns (new_scaffolds [(id, [[name, position, direction, length], ...], s_length)] in creation order,
             gap_estimations, scaffold_indexer afterwards)."""
    n_nodes = len(link)
    seen = [False] * n_nodes
    out, estimations = [], []
    for first in node_order:                                 # nx.connected_components: components by first node
        if seen[first]:
            continue
        comp, stack = [], [first]                            # the component's nodes
        seen[first] = True
        while stack:
            x = stack.pop()
            comp.append(x)
            for y in (x ^ 1, link[x]):
                if y >= 0 and not seen[y]:
                    seen[y] = True
                    stack.append(y)
        members = set(comp)
        in_order = [x for x in node_order if x in members]
        start = next(x for x in in_order if link[x] < 0)     # first node with one neighbour (:287-290)
        scaffold_indexer += 1
        pos = 0
        contig_list = []
        node = start
        while True:
            k, side = node >> 1, node & 1
            for name, cpos, cdir, clen in contigs[k]:
                if side == 0:                                # entered through 'L' (:363-378)
                    contig_list.append([name, cpos + pos, cdir, clen])
                else:                                        # entered through 'R' (:384-404)
                    contig_list.append([name, pos + (slen[k] - cpos) - clen, bool(True - cdir), clen])
            pos += slen[k]
            leave = node ^ 1
            if link[leave] < 0:                              # reached the end of the path (:348-357)
                break
            if appended[leave] is not None:
                estimations.append(appended[leave])
            pos += gap[leave]
            node = link[leave]
        length = max(c[1] + c[3] for c in contig_list)
        out.append((scaffold_indexer, contig_list, length))
    return out, estimations, scaffold_indexer
