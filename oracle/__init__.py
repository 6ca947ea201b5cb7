"""CPU oracle for the EnvGS render-and-trace hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker.  envgs_amd/ never does (tests/test_no_oracle_in_product.py).
Parity status: "parity unpinned" for the kernel-body arithmetic (extension sources are
absent from the reference tree); the in-tree boundary pieces are pinned by tests/golden/.
"""
