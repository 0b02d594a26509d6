"""CPU oracle for the PhantomFHE RNS hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (phantom-fhe_amd/) never does.  See oracle/oracle.h for the parity status
("parity unpinned against an executed reference") and how the oracle is pinned instead.
"""
from .oracle import *  # noqa: F401,F403
