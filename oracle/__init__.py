"""TEST INFRASTRUCTURE: CPU restatements of the reference hot path (never imported by besst_amd)."""
