"""CPU oracle for the th_rl hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (th_rl_amd) must never import this.
"""
