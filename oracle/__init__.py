"""CPU oracle for the TriPlane / InfoInv ray-march path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
