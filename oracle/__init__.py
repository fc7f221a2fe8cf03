"""CPU ORACLE package (test infrastructure only).

Importable ONLY from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under mcpower_amd/ imports it.
"""
