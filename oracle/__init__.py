"""CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

ctypes binding over ``oracle/liboracle.so`` -- the plain-C restatement of the reference's BFV PolyRq/NTT hot
path (see ``he_oracle.h`` for the parity status and the reference citations).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.
"""
from . import pir  # noqa: F401
from .binding import *  # noqa: F401,F403
