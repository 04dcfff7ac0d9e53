"""CPU oracle for the HMM forward log-likelihood.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Parity is UNPINNED against ziphmm itself (see forward_oracle.c header and DESIGN.md).
"""
