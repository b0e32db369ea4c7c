"""Thin command-line counterpart of ``runBESST`` for the accelerated path only.

    python -m besst_amd.cli -c contigs.fa -f lib1.bam [lib2.bam ...] -orientation fr [rf ...] -o outdir

Flag names and defaults follow runBESST:254-402 for everything the hot path reads (-m -s -T -k -r -e -z -z_min
--min_mapq -d -y --no_score).  Per library it runs the BAM front-end, ``libmetrics.get_metrics`` and
``CreateGraph.PE`` and writes Statistics.txt plus the scored edge tables of G and G' as TSV; scaffolding itself
(MakeScaffolds and later) stays with BESST - see INTEGRATION.md for plugging these calls into runBESST.

Several GPUs of one node: launch the same command line under torchrun, one process per GPU -

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m besst_amd.cli -c ... -f ...

Every rank ingests its slice of each BAM on its own GPU and takes part in the library scans and the graph build
(besst_amd.sharded); rank 0 writes the outputs.
"""
from __future__ import print_function

import argparse
import os
import sys
from time import time

from . import CreateGraph as CG
from . import MakeScaffolds as MS
from . import Parameter, bamio, libmetrics, session


def read_fasta(path):
    seqs, name, chunks = {}, None, []
    with open(path) as fh:
        for line in fh:
            if line.startswith('>'):
                if name is not None:
                    seqs[name] = ''.join(chunks)
                name, chunks = line[1:].strip().split()[0], []
            else:
                chunks.append(line.strip())
    if name is not None:
        seqs[name] = ''.join(chunks)
    return seqs


def build_parser():
    ap = argparse.ArgumentParser(prog='besst_amd.cli', description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('-c', dest='contigfile', required=True, help='contig FASTA')
    ap.add_argument('-f', dest='bamfiles', nargs='+', required=True, help='one BAM per library')
    ap.add_argument('-o', dest='output', default='.', help='output directory')
    ap.add_argument('-orientation', dest='orientation', nargs='+', choices=['fr', 'rf'], required=True)
    ap.add_argument('-r', dest='readlen', type=int, nargs='+')
    ap.add_argument('-m', dest='mean', type=float, nargs='+')
    ap.add_argument('-s', dest='stddev', type=float, nargs='+')
    ap.add_argument('-T', dest='threshold', type=int, nargs='+')
    ap.add_argument('-k', dest='minsize', type=int, nargs='+')
    ap.add_argument('-e', dest='edgesupport', type=int, nargs='+')
    ap.add_argument('-z', dest='covcutoff', type=int, default=None)
    ap.add_argument('-z_min', dest='lower_covcutoff', type=float, default=0.001)
    ap.add_argument('--min_mapq', dest='min_mapq', type=int, default=11)
    ap.add_argument('-d', dest='duplicate', action='store_false', help='switch duplicate detection off')
    ap.add_argument('-y', dest='extendpaths', action='store_false', help='switch path extension off')
    ap.add_argument('--no_score', dest='no_score', action='store_true')
    ap.add_argument('--threads', type=int, default=None, help='BAM inflate threads')
    ap.add_argument('--linearize', action='store_true',
                    help="also run steps 1-4 of MakeScaffolds.Algorithm on a copy of G (isolated scaffolds, "
                         "score-based ambiguity removal, cycles) and write the surviving link edges")
    return ap


def _per_lib(values, i):
    return values[i] if values is not None else None


def write_edges(path, G):
    with open(path, 'w') as fh:
        print('scaffold1\tside1\tscaffold2\tside2\tnr_links\tobs\tobs_sq\tgap\tscore', file=fh)
        for u, v in G.edges():
            d = G[u][v]
            if d['nr_links'] is None:
                continue
            print('\t'.join(str(x) for x in (u[0], u[1], v[0], v[1], d['nr_links'], d['obs'], d['obs_sq'],
                                             d.get('gap', ''), d.get('score', ''))), file=fh)


def join_process_group():
    """Under torchrun (WORLD_SIZE > 1): one rank per GPU over RCCL (backend 'nccl'; BESST_DIST_BACKEND=gloo for several
    ranks on one GPU).  -> (rank, whether this call initialised the group)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world < 2:
        return 0, False
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count()))
    from . import sharded
    if dist.is_initialized():
        sharded.enable()
        return dist.get_rank(), False
    # a rank that never arrives ends the job instead of holding it: every collective gives up after this long (gloo raises
    # in the waiting ranks; under RCCL the watchdog tears the process down)
    import datetime
    limit = datetime.timedelta(seconds=float(os.environ.get('BESST_COLLECTIVE_TIMEOUT', '1800')))
    dist.init_process_group(os.environ.get('BESST_DIST_BACKEND', 'nccl'), timeout=limit)
    sharded.enable()                                         # every rank makes the drop-in's calls together from here on
    return dist.get_rank(), True


def main(argv=None):
    args = build_parser().parse_args(argv)
    if len(args.orientation) != len(args.bamfiles):
        sys.exit('need one -orientation per BAM file')
    rank, joined = join_process_group()
    try:
        return _run(args, rank)
    finally:
        if joined:
            import torch.distributed as dist
            dist.destroy_process_group()


def _run(args, rank):
    out = os.path.join(args.output, 'BESST_output')
    os.makedirs(out, exist_ok=True)
    lead = rank == 0                                         # the followers compute, rank 0 also writes
    param = Parameter.parameter()
    param.scaffold_indexer = 1
    param.min_mapq = args.min_mapq
    param.cov_cutoff = args.covcutoff
    param.lower_cov_cutoff = args.lower_covcutoff
    param.no_score = args.no_score
    param.detect_duplicate = args.duplicate
    param.extend_paths = args.extendpaths
    param.detect_haplotype = False
    param.print_scores = False
    param.max_contig_overlap = 200
    param.output_directory = out
    param.first_lib = True
    Information = param.information_file = open(os.path.join(out, 'Statistics.txt') if lead else os.devnull, 'w')
    C_dict = read_fasta(args.contigfile) if lead else {}
    if lead:
        print('Number of initial contigs:', len(C_dict))
    Contigs, Scaffolds, small_contigs, small_scaffolds = {}, {}, {}, {}
    for i, bam in enumerate(args.bamfiles):
        param.pass_number = i + 1
        param.bamfile = bam
        param.orientation = args.orientation[i]
        param.mean_ins_size = _per_lib(args.mean, i)
        param.std_dev_ins_size = _per_lib(args.stddev, i)
        param.ins_size_threshold = _per_lib(args.threshold, i)
        param.contig_threshold = _per_lib(args.minsize, i)
        param.edgesupport = _per_lib(args.edgesupport, i)
        param.read_len = _per_lib(args.readlen, i)
        print('\nPASS ' + str(i + 1) + '\n\n', file=Information)
        t0 = time()
        # straight to HBM: inflate + record decode on the GPU (any BGZF block layout)
        # (under torchrun: this rank's slice of the file, on this rank's GPU)
        records = bamio.open_bam(bam, threads=args.threads)
        print('Time elapsed reading %s (%d records): %s' % (bam, len(records), time() - t0), file=Information)
        param.contig_index = dict(enumerate(records.references))
        t0 = time()
        libmetrics.get_metrics(records, param, Information)
        print('Time elapsed for getting libmetrics, iteration ' + str(i) + ': ' + str(time() - t0) + '\n',
              file=Information)
        t0 = time()
        G, G_prime = CG.PE(Contigs, Scaffolds, Information, C_dict, param, small_contigs, small_scaffolds, records)
        print('Total time for CreateGraph-module, iteration ' + str(i) + ': ' + str(time() - t0) + '\n',
              file=Information)
        session.close_session(records)
        records.close()
        param.first_lib = False
        if not lead:
            continue
        pass_dir = os.path.join(out, 'pass%d' % (i + 1))
        os.makedirs(pass_dir, exist_ok=True)
        write_edges(os.path.join(pass_dir, 'edges_G.tsv'), G)
        write_edges(os.path.join(pass_dir, 'edges_Gprime.tsv'), G_prime)
        if args.linearize and not param.no_score:
            # on copies: the graphs CreateGraph.PE returned stay as BESST's own MakeScaffolds expects them
            t0 = time()
            L, L_prime = G.copy(), G_prime.copy()
            L, _, _ = MS.LinearizeGraph(L, L_prime, Contigs, Scaffolds, Information, param)
            print('Time elapsed for the graph linearisation (steps 1-4), iteration ' + str(i) + ': ' + str(time() - t0)
                  + '\n', file=Information)
            write_edges(os.path.join(pass_dir, 'edges_G_linear.tsv'), L)
        print('pass %d: %d records, G %d link edges, G_prime %d link edges' % (
            i + 1, len(records), sum(1 for u, v in G.edges() if G[u][v]['nr_links'] is not None),
            sum(1 for u, v in G_prime.edges() if G_prime[u][v]['nr_links'] is not None)))
    Information.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
