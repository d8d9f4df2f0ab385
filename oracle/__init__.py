"""CPU oracle for the variational-layer hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (bayesianneuralnetworks_amd/) never does.
"""
