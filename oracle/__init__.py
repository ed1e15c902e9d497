"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package.  PARITY UNPINNED: see
oracle_model.h."""
