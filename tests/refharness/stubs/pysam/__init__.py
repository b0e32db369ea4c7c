"""Empty stand-in: the reference hot path only needs `import pysam` to succeed (bam_parser.py:3)."""
