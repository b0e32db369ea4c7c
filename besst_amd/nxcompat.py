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


class Graph(_BaseGraph):
    @property
    def edge(self):
        return self._adj

    @property
    def node(self):
        return self._node

    def neighbors(self, n):
        return list(self._adj[n])

    def nodes(self, data=False):
        return list(_NodeView(self)(data=data))

    def edges(self, nbunch=None, data=False):
        return list(_EdgeView(self)(nbunch=nbunch, data=data))

    def number_of_edges(self, u=None, v=None):
        """O(nodes) edge count (the base class goes through degree(), which this facade returns as a dict)."""
        if u is not None:
            return 1 if v in self._adj[u] else 0
        total = sum(len(nbrs) for nbrs in self._adj.values())
        loops = sum(1 for n, nbrs in self._adj.items() if n in nbrs)
        return (total + loops) // 2

    def nodes_iter(self, data=False):
        return iter(_NodeView(self)(data=data))

    def edges_iter(self, nbunch=None, data=False):
        return iter(_EdgeView(self)(nbunch=nbunch, data=data))

    def degree(self, nbunch=None, weight=None):
        d = _DegreeView(self)(nbunch, weight)
        if isinstance(d, int):
            return d
        return dict(d)
