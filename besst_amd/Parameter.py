"""Run-wide mutable parameter bag and per-library counters.

Field names follow the reference (BESST/Parameter.py:24-124): the hot path
mutates them in place and the reference's downstream stages read them, so the
names are part of the drop-in contract.
"""

_PARAM_DEFAULTS = dict(
    mean_ins_size=None, std_dev_ins_size=None, read_len=None, mean_coverage=None,
    lower_cov_cutoff=None, cov_cutoff=None, contig_index=None, score_cutoff=None,
    max_extensions=None, max_contig_overlap=None, std_dev_coverage=None,
    output_directory=None, bamfile=None, ins_size_threshold=None, contigfile=None,
    edgesupport=None, contig_threshold=None, scaffold_indexer=0, first_lib=None,
    tot_assembly_length=None, current_N50=None, current_L50=None, hapl_ratio=None,
    hapl_threshold=None, detect_haplotype=None, detect_duplicate=None, gff_file=None,
    information_file=None, extend_paths=None, development=None, plots=None,
    path_threshold=None, no_score=None, orientation=None, pass_number=None,
    print_scores=None, path_gaps_estimated=0, gap_estimations=None,
    contamination_ratio=0, contamination_mean=None, contamination_stddev=None,
    NO_ILP=None, FASTER_ILP=None,
)


class parameter(object):
    def __init__(self, **overrides):
        for key, value in _PARAM_DEFAULTS.items():
            setattr(self, key, value)
        self.gap_estimations = []
        for key, value in overrides.items():
            setattr(self, key, value)

    def get_params(self):
        rows = ['param\tvalue\n']
        for attr, value in self.__dict__.items():
            if callable(value):
                continue
            if value is None or type(value) in (bool, int, float) or \
                    (hasattr(value, '__len__') and len(value) < 5):
                rows.append('{0}\t{1}\n'.format(attr, value))
        return ''.join(rows)


class counters(object):
    """Per-library tallies of the record loop (reference: Parameter.py:113-124)."""

    def __init__(self, param_count=None, param_non_unique=None, param_non_unique_for_scaf=None,
                 param_nr_of_duplicates=None, param_prev_obs1=None, param_prev_obs2=None,
                 param_reads_with_too_long_insert=None):
        self.count = param_count
        self.non_unique = param_non_unique
        self.non_unique_for_scaf = param_non_unique_for_scaf
        self.nr_of_duplicates = param_nr_of_duplicates
        self.prev_obs1 = param_prev_obs1
        self.prev_obs2 = param_prev_obs2
        self.reads_with_too_long_insert = param_reads_with_too_long_insert
