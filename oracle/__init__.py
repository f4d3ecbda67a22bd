"""TEST INFRASTRUCTURE ONLY: CPU oracle (scalar restatement of the reference HLSL).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
PARITY UNPINNED: the reference cannot execute here and ships no golden vectors.
"""
