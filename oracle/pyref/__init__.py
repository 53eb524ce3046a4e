"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

Python big-int restatements of the reference's algorithms on the hot path and its callers (every module cites the
/root/reference files and lines it follows).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.

Parity status: the Rust reference cannot be built here (no cargo/rustc; un-vendored arkworks / liblasso / merlin) and its tests
hold no byte-level golden vectors.  This package is pinned by the reference's one integer KAT (COEFF_D, src/utils.rs:35), public
BLS12-381 / SHA-3 / merlin vectors, the algebraic identities the reference's own tests assert, agreement with the independent C
restatement (oracle/gkrmsm_oracle*.c), and prover-against-verifier consistency (verifier.py follows the reference's verify
functions).  Byte parity against the Rust binary itself is "parity unpinned" (DESIGN.md section 2).
"""
