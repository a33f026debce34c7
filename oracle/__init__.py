"""oracle -- CPU restatements of the reference codecs.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (hipcomp-core_amd) never does.
"""
