"""CPU checkers for the bulk record-parsing path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package;
the product (dsp_amd) never does.
"""
