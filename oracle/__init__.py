"""CPU oracle for the ocrs hot path — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never from ocrs_amd/.  See oracle/csrc/ocrs_oracle.c for the parity status.
"""
import os as _os

# Thread-pool hygiene for big hosts (the GPU box has 256 cores): libgomp (this oracle's C library) and
# torch's bundled OpenMP runtime are separate pools; at their defaults they oversubscribe and spin
# against each other.  Only defaults — anything the caller exported wins.  Must run before either loads.
_os.environ.setdefault("OMP_NUM_THREADS", str(min(16, _os.cpu_count() or 1)))
_os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
_os.environ.setdefault("GOMP_SPINCOUNT", "0")
_os.environ.setdefault("MKL_NUM_THREADS", _os.environ["OMP_NUM_THREADS"])
