"""CPU restatement of the reference's ScorePaths (ExtendLargeScaffolds.py:29-130) on CSR arrays.

TEST INFRASTRUCTURE ONLY - imported by tests/; the product path (besst_amd/ExtendLargeScaffolds.py ->
besst_score_paths in libbesst_amd.so) never touches this module.  Pinned against tests/golden/scorepaths.json.gz,
captured from the reference's own function by tests/golden/make_scorepaths_golden.py.

Array interface (shared with the device path)
  nodes          compact ids: scaffold k has the nodes 2k ('L') and 2k+1 ('R')
  row_ptr, col, weight   CSR of the LINK edges of G (both directions), weight = nr_links
  path_ptr, path_nodes   CSR of the paths
"""


def link_weights(row_ptr, col, weight, path, contamination):
    """(good, bad) link weights of one path: calculate_connectivity (:33-69) or, with contamination,
    calculate_connectivity_contamination (:72-105; its good weight is returned BEFORE the division by two)."""
    odd = set(path[1::2])
    even = set(path[0::2])
    visited = set()
    good = bad = 0
    for i, node in enumerate(path):
        for e in range(row_ptr[node], row_ptr[node + 1]):
            nbr, w = col[e], weight[e]
            if (nbr >> 1) == (node >> 1):
                continue
            if contamination:
                if nbr in (odd if i % 2 == 0 else even):
                    good += w
                else:
                    bad += w
            elif i % 2 == 0:
                if nbr in odd:
                    if nbr not in visited:
                        good += w
                else:
                    bad += w
            else:
                if nbr not in even:
                    bad += w
                elif nbr not in visited:
                    bad += w
        visited.add(node)
    return good, bad


def score_of(good, bad, contamination):
    """The score the reference derives from the two weights (:63-67, :94-103)."""
    g = good / 2 if contamination else good
    try:
        return g / float(bad)
    except ZeroDivisionError:
        return g


def score_paths(row_ptr, col, weight, path_ptr, path_nodes, contamination, no_score, score_cutoff):
    """The entries ScorePaths appends to all_paths (:118-128): [score, bad_link_weight, path index, len(path)]."""
    out = []
    for p in range(len(path_ptr) - 1):
        path = path_nodes[path_ptr[p]:path_ptr[p + 1]]
        good, bad = link_weights(row_ptr, col, weight, path, contamination)
        score = score_of(good, bad, contamination)
        if (no_score and score >= score_cutoff) or (len(path) > 2 and score >= score_cutoff):
            out.append([score, bad, p, len(path)])
    return out
