"""CPU oracle for the hnsw-clj hot path -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
