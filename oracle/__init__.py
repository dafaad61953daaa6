"""CPU oracle for the GP hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (gp_amd / libgpx.so) never does.  PARITY UNPINNED: see oracle/gp_oracle.h.
"""
from .binding import *  # noqa: F401,F403
