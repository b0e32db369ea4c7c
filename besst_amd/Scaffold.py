"""Boundary object: one scaffold (ordered list of contigs).

Mirrors BESST/Scaffold.py:23-36 of the reference.
"""


class scaffold(object):
    __slots__ = ('name', 'contigs', 's_length')

    def __init__(self, scaffold_name, scaffold_contigs, scaffold_length):
        self.name = scaffold_name          # running integer id (param.scaffold_indexer)
        self.contigs = scaffold_contigs    # contig objects, in scaffold order
        self.s_length = scaffold_length    # total scaffold length in bp

    def __repr__(self):
        return 'scaffold(%r, n_contigs=%d, len=%r)' % (self.name, len(self.contigs), self.s_length)
