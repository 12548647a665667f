"""CPU oracle for the ocrs hot path — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never from ocrs_amd/.  See oracle/csrc/ocrs_oracle.c for the parity status.
"""
