"""CPU oracle for the Speech2Text hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under neural_sp_amd/ may import this package; only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
"""
