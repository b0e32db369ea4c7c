"""Boundary object: one assembled contig.

Mirrors the attribute bag of the reference (BESST/Contig.py:23-38) so that the
dicts handed back by :func:`besst_amd.CreateGraph.PE` can be consumed by the
reference's unchanged downstream stages.
"""


class contig(object):
    __slots__ = ('name', 'scaffold', 'direction', 'position', 'length',
                 'coverage', 'repeat', 'is_haplotype', 'sequence')

    def __init__(self, contig_name, contig_scaffold=None, contig_direction=None,
                 contig_position=None, contig_length=None, contig_coverage=None,
                 contig_repeat=False, contig_haplotype=False, contig_sequence=None):
        self.name = contig_name
        self.scaffold = contig_scaffold      # name (int) of the owning scaffold
        self.direction = contig_direction    # True = forward inside the scaffold
        self.position = contig_position      # left-most coordinate inside the scaffold
        self.length = contig_length
        self.coverage = contig_coverage
        self.repeat = contig_repeat
        self.is_haplotype = contig_haplotype
        self.sequence = contig_sequence

    def __repr__(self):
        return 'contig(%r, scaffold=%r, dir=%r, pos=%r, len=%r)' % (
            self.name, self.scaffold, self.direction, self.position, self.length)
