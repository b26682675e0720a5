"""oracle/ -- TEST INFRASTRUCTURE.  CPU restatements of the reference's hot path.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
nothing under cslam_amd/ imports this package.
"""
