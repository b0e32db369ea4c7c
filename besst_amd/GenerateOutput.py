"""The four output helpers that the hot path calls from inside CreateGraph.PE.

Only the functions reached from the graph-construction path are provided (reference:
GenerateOutput.py:47-85, called at CreateGraph.py:432,803,1012-1013).  They matter for parity because
they DELETE the removed contigs from the caller's dicts.  FASTA/AGP/GFF writers of the scaffolding
result are downstream of this path and stay with the reference.
"""
from __future__ import print_function


def _write_fasta(handle, name, sequence):
    print('>' + name, file=handle)
    sequence = sequence or ''
    for i in range(0, len(sequence), 60):
        print(sequence[i:i + 60], file=handle)


def _forget(cont_obj, Contigs, small_contigs):
    try:
        del Contigs[cont_obj.name]
    except KeyError:
        del small_contigs[cont_obj.name]


def PrintOutRepeats(Repeats, Contigs, output_dest, small_contigs):
    handle = open(output_dest + '/repeats.fa', 'w') if output_dest else None
    for cont_obj in Repeats:
        if handle:
            _write_fasta(handle, cont_obj.name, cont_obj.sequence)
        _forget(cont_obj, Contigs, small_contigs)
    if handle:
        handle.close()
    return ()


def repeat_contigs_logger(Repeats, Contigs, output_dest, small_contigs, param):
    if not output_dest:
        return
    with open(output_dest + '/repeats_log.tsv', 'w') as handle:
        print('contig_accession\tlength\tcoverage\tcov/mean_cov(exp number of placements)\tlib_mean\tplacable',
              file=handle)
        for cont_obj in sorted(Repeats, key=lambda c: c.coverage, reverse=True):
            placable = 'Yes' if param.mean_ins_size > cont_obj.length else 'No'
            print('{0}\t{1}\t{2}\t{3}\t{4}\t{5}'.format(
                cont_obj.name, cont_obj.length, round(cont_obj.coverage, 1),
                round(cont_obj.coverage / param.mean_coverage, 0), round(param.mean_ins_size, 0), placable),
                file=handle)


def PrintOut_low_cowerage_contigs(low_coverage_contigs, Contigs, output_dest, small_contigs):
    handle = open(output_dest + '/low_coverage_contigs.fa', 'w') if output_dest else None
    for cont_obj in low_coverage_contigs:
        if handle:
            _write_fasta(handle, cont_obj.name, cont_obj.sequence)
        _forget(cont_obj, Contigs, small_contigs)
    if handle:
        handle.close()
    return ()


def ChangeToSmallContigs(Contigs, list_of_contigs, small_contigs):
    for cont_obj in list_of_contigs:
        del Contigs[cont_obj.name]
        small_contigs[cont_obj.name] = cont_obj
    return ()
