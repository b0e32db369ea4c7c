"""networkx-1.x compatible graph facade.

The reference's downstream stages (MakeScaffolds.py, ExtendLargeScaffolds.py)
are written against networkx 1.10: ``G.edge[u][v]``, ``G.node[n]``,
``G.edges_iter()``, and list-returning ``neighbors()/edges()/nodes()`` which the
code mutates the graph under (CreateGraph.py:292-296,399-403,715-716;
MakeScaffolds.py:52,139).  :class:`Graph` restores that surface on top of
whatever networkx is installed so the emitted scaffold graph drops into the
unchanged reference code.
"""
import networkx as _nx
from networkx.classes.reportviews import DegreeView as _DegreeView
from networkx.classes.reportviews import EdgeView as _EdgeView
from networkx.classes.reportviews import NodeView as _NodeView

_BaseGraph = _nx.Graph      # bound at import: test harnesses may later rebind networkx.Graph to this facade


class LazyDict(dict):
    """A dictionary whose keys are all there from the start - in their final order - while a value is made the first
    time it is read.  ``Graph.adopt`` backs ``_adj`` and ``_node`` of a graph with two of these: CreateGraph.PE hands over
    200 k nodes and up to a million adjacency entries as columns, and what reads the graph afterwards (MakeScaffolds'
    walks, ExtendLargeScaffolds' searches) touches a node at a time; the containers of a node - its neighbour dictionary,
    the attribute dictionaries of its edges - are built when that node is first looked at, the whole graph at once (in
    bulk, as before) when something asks for every value.

    An unmade value is stored as the ``int`` token the source wants back (a legitimate value is always a dict).
    ``source.one(token) -> value`` makes one, ``source.rest(lazy_dict)`` makes every value still unmade.  Keys, length,
    membership and iteration over keys never make anything; assignment and deletion are the dictionary's own."""
    __slots__ = ('_source', '_untouched')

    def __init__(self, tokens, source):
        dict.__init__(self, tokens)
        self._source = source
        self._untouched = True                               # nothing made, set or deleted yet: the columns ARE the content

    def __getitem__(self, key):
        value = dict.__getitem__(self, key)
        if value.__class__ is int:
            value = self._source.one(value)
            dict.__setitem__(self, key, value)
            self._untouched = False
        return value

    def __setitem__(self, key, value):
        self._untouched = False
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self._untouched = False
        dict.__delitem__(self, key)

    def update(self, *args, **kw):
        self._untouched = False
        dict.update(self, *args, **kw)

    @property
    def columns(self):
        """The source behind the dictionary while it still IS the dictionary's whole content (no value made, nothing set or
        deleted), else None: a reader that wants every edge as arrays can take them from the columns and touch nothing."""
        return self._source if self._untouched else None

    def _all(self):
        source = self._source
        if source is not None:
            self._source = None
            self._untouched = False
            # (hundreds of thousands of small containers that reference nothing but numbers and each other: with the cyclic
            # collector running, every allocation burst re-walks the growing heap - 0.7 s instead of 0.15 for the two graphs
            # of a 100 k-contig assembly; CreateGraph.PE pauses it for the same reason)
            import gc
            was_enabled = gc.isenabled()
            gc.disable()
            try:
                source.rest(self)
            finally:
                if was_enabled:
                    gc.enable()

    # (defined so that C code which copies or merges a dictionary takes the generic path - keys() + __getitem__ -
    # instead of reading the raw slots)
    def __iter__(self):
        return dict.__iter__(self)

    def get(self, key, default=None):
        return self[key] if dict.__contains__(self, key) else default

    def items(self):
        self._all()
        return dict.items(self)

    def values(self):
        self._all()
        return dict.values(self)

    def pop(self, key, *default):
        self._untouched = False
        if dict.__contains__(self, key):
            self[key]
        return dict.pop(self, key, *default)

    def popitem(self):
        self._all()
        return dict.popitem(self)

    def setdefault(self, key, default=None):
        if dict.__contains__(self, key):
            return self[key]
        self._untouched = False
        dict.__setitem__(self, key, default)
        return default

    def copy(self):
        self._all()
        return dict(dict.items(self))

    def clear(self):
        self._source = None
        self._untouched = False
        dict.clear(self)

    def __eq__(self, other):
        self._all()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._all()
        return dict.__repr__(self)

    def __reduce__(self):
        self._all()
        return (dict, (dict(dict.items(self)),))


class Graph(_BaseGraph):
    @property
    def edge(self):
        return self._adj

    @property
    def node(self):
        return self._node

    def neighbors(self, n):
        try:
            return list(self._adj[n])
        except KeyError:
            raise _nx.NetworkXError('The node %s is not in the graph.' % (n,))

    def nodes(self, data=False):
        return list(_NodeView(self)(data=data))

    def edges(self, nbunch=None, data=False):
        if nbunch is None and (data is True or data is False):
            # the whole edge list in networkx's own order (every edge reported from its first endpoint in node order),
            # built straight from the adjacency: list(EdgeDataView) walks the graph twice (its __len__ is a full
            # iteration) through three layers of views - 3.8 of the drop-in's 9 host seconds on a 100 k-contig assembly
            out, seen = [], set()
            if data:
                for n, nbrs in self._adj.items():
                    out.extend([(n, nbr, dd) for nbr, dd in nbrs.items() if nbr not in seen])
                    seen.add(n)
            else:
                for n, nbrs in self._adj.items():
                    out.extend([(n, nbr) for nbr in nbrs if nbr not in seen])
                    seen.add(n)
            return out
        return list(_EdgeView(self)(nbunch=nbunch, data=data))

    def number_of_edges(self, u=None, v=None):
        """O(nodes) edge count (the base class goes through degree(), which this facade returns as a dict)."""
        if u is not None:
            return 1 if v in self._adj[u] else 0
        total = sum(len(nbrs) for nbrs in self._adj.values())
        loops = sum(1 for n, nbrs in self._adj.items() if n in nbrs)
        return (total + loops) // 2

    def subgraph(self, nodes):
        """networkx 1.x semantics: an independent COPY of the induced subgraph (MakeScaffolds.NewContigsScaffolds
        removes a component's nodes from G while it still iterates over / hands on the subgraph, :272-337; a 2.x/3.x
        view would empty itself underneath it).  The copy lists its nodes in this graph's node order, so code that picks
        "the first node with ..." of a component (:287-293) does not depend on set iteration order."""
        keep = set(nodes)
        H = self.__class__()
        for n in self._node:
            if n in keep:
                H.add_node(n, **self._node[n])
        for u in H._node:
            for v, d in self._adj[u].items():
                if v in keep:
                    H._adj[u][v] = d
        return H

    def adopt(self, node_tokens, node_source, adj_tokens, adj_source):
        """Back this (empty) graph's nodes and adjacency with values made on first touch (see LazyDict): the keys of both
        dictionaries are the nodes in their final order."""
        if self._node or self._adj:
            raise _nx.NetworkXError('adopt: the graph is not empty')
        self._node = LazyDict(node_tokens, node_source)
        self._adj = LazyDict(adj_tokens, adj_source)

    def link_columns(self):
        """The column source of a graph that CreateGraph.PE has filled and nobody has read or changed since (see LazyDict),
        else None."""
        adj, node = self._adj, self._node
        if isinstance(adj, LazyDict) and isinstance(node, LazyDict) and node._source is not None and \
                dict.__len__(node) == dict.__len__(adj):
            return adj.columns
        return None

    def add_scaffold(self, scaffold, length):
        """The two end nodes of a scaffold with their 'length' attribute and the intra-scaffold edge (nr_links=None):
        what InitializeGraph does per scaffold (CreateGraph.py:710-722) with three networkx calls."""
        left, right = (scaffold, 'L'), (scaffold, 'R')
        if left in self._node or right in self._node:
            self.add_edge(left, right, nr_links=None)
            self._node[left]['length'] = length
            self._node[right]['length'] = length
            return
        data = {'nr_links': None}
        self._node[left] = {'length': length}
        self._node[right] = {'length': length}
        self._adj[left] = {right: data}
        self._adj[right] = {left: data}

    def add_link(self, u, v, data):
        """add_edge(u, v, **data) for two nodes that are already in the graph and not yet adjacent (what
        CreateGraph.PE's bulk insertion of the device's edge rows guarantees): the attribute dict is stored as it is,
        once, under both endpoints - the work networkx's add_edge does per call (node checks, dict update, factory
        calls) is a third of the drop-in's host time on a 100 k-contig assembly."""
        self._adj[u][v] = data
        self._adj[v][u] = data

    def nodes_iter(self, data=False):
        return iter(_NodeView(self)(data=data))

    def edges_iter(self, nbunch=None, data=False):
        return iter(_EdgeView(self)(nbunch=nbunch, data=data))

    def degree(self, nbunch=None, weight=None):
        d = _DegreeView(self)(nbunch, weight)
        if isinstance(d, int):
            return d
        return dict(d)
