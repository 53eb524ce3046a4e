"""GPU parity of the Knuckles opening (KnucklesOpeningProtocol::prove, opening.rs:39-98) through the C ABI vs the Python oracle,
with an SRS built from a known tau so that every KZG equation -- and the final deferred pairing pair <A, H0> = <B, H1>, i.e.
A = tau * B -- can be checked in the exponent; plus the verifier's algebraic identity (opening.rs:126-145).  Shaped like the
reference's own test (opening.rs:170-198: random polynomial shorter than 2^num_vars, random point)."""
import pytest

from gkr_msm_amd import codec, harness as H
from pyref import field as F
from pyref import g1 as G
from pyref import knuckles as K
from pyref import polys as PL
from pyref.sumcheck import TapeTranscript

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("num_vars,poly_len,k", [(1, 2, 2), (3, 6, 2), (4, 16, 2), (5, 23, 7), (7, 100, 2)])
def test_knuckles_open_matches_oracle(num_vars, poly_len, k):
    rng = F.SplitMix64(70 + num_vars)
    n = 1 << num_vars
    tau = rng.next_fr()
    basis, cur = [], G.GEN
    for _ in range(2 * n - 1):
        basis.append(cur)
        cur = G.mul(cur, tau)
    poly = [rng.next_fr() for _ in range(poly_len)]
    pt = [rng.next_fr() for _ in range(num_vars)]
    e = PL.eq_poly_sequence_last(pt)
    claimed = sum(a * b for a, b in zip(poly, e)) % F.P
    comm = K.kzg_commit(basis, poly)
    tape = [rng.next_bits(128) for _ in range(3)]
    inv = K.setup_inverses(k, num_vars)
    tr = TapeTranscript(tape)
    pts_out = []
    want_pair, want = K.knuckles_open(tr, pts_out, basis, inv, k, num_vars, comm, pt, claimed, poly)

    d_inv = H.knuckles_setup(k, num_vars)
    assert codec.from_mont_limbs(H.to_host(d_inv)) == inv
    d_basis = H.g1_aff_dev(basis)
    d_poly = H.to_dev(codec.to_mont_limbs(poly))
    got, pair = H.knuckles_open(d_basis, d_inv, k, num_vars, d_poly, poly_len, pt, claimed, comm, tape)
    for key in ("t_comm", "t_x", "p_x", "p_lt_x_proof", "t_kx", "t_kx_proof"):
        assert got[key] == want[key], key
    assert pair == want_pair
    assert pair[0] == G.mul(pair[1], tau)                      # the pairing check, in the exponent
    got.update(x=want["x"])
    assert K.verifier_identity(k, num_vars, pt, claimed, got)
    with pytest.raises(Exception):                             # a wrong opening claim is rejected like the reference's assert
        H.knuckles_open(d_basis, d_inv, k, num_vars, d_poly, poly_len, pt, (claimed + 1) % F.P, comm, tape)
