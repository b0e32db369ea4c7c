"""Harness-only shim: routes the reference's `mathstats` imports to besst_amd.mathstats_compat.

mathstats==0.2.6.5 is not installable here; golden vectors that
involve these functions therefore pin PLUMBING only, never the third-party arithmetic.
"""
