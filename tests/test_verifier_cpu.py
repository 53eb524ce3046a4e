"""The verifier runs on the host, so its tests run without a GPU: the library's Pippenger::verify and pairing
against the Python oracle (pyref/verifier.py is written from the reference's verify functions, pyref/pairing.py is an
independent pairing in a different Fq12 representation), on proofs made by the oracle prover.  GPU-made proofs are verified in
tests/test_verifier_gpu.py."""
import pytest

from gkr_msm_amd import verifier as VF
from pyref import field as F
from pyref import g1 as G
from pyref import gkr as GK
from pyref import knuckles as KN
from pyref import pairing as PR
from pyref import pippenger as PP
from pyref import verifier as V


def test_pairing_matches_the_independent_oracle_value_for_value():
    want = PR.pairing(G.GEN, PR.G2_GEN)
    got = VF.pairing(G.GEN, PR.G2_GEN)
    assert PR.tower_to_poly(got) == want
    a, b = 0x1234567, 0x7654321
    got_ab = VF.pairing(G.mul(G.GEN, a), PR.g2_mul(PR.G2_GEN, b))
    assert PR.tower_to_poly(got_ab) == PR.f12_pow(want, a * b)          # bilinear
    assert PR.f12_pow(want, PR.R_ORDER) == PR.ONE12 and want != PR.ONE12   # order r, non-degenerate


def test_kzg_verify_pair():
    tau = 0x1F3A9C7754BB21 * 3 + 11
    b = G.mul(G.GEN, 0xABCDEF12345)
    a = G.mul(b, tau)
    h1 = PR.g2_mul(PR.G2_GEN, tau)
    assert VF.kzg_verify_pair((a, b), PR.G2_GEN, h1)
    assert not VF.kzg_verify_pair((G.add(a, G.GEN), b), PR.G2_GEN, h1)
    assert not VF.kzg_verify_pair((a, b), PR.G2_GEN, PR.g2_mul(PR.G2_GEN, tau + 1))
    assert VF.kzg_verify_pair((None, None), PR.G2_GEN, h1)               # e(0, .) = 1 on both sides


def _oracle_proof(x_log, d_log, nbits, clm, seed):
    y_size = (nbits + d_log - 1) // d_log
    y_log = (y_size - 1).bit_length()
    n = 1 << x_log
    rng = F.SplitMix64(seed)
    pts = F.random_points(n, 2)
    sc = F.random_scalars(n, nbits, 3)
    nv = x_log + clm
    tau, k = rng.next_fr(), 2
    basis, cur = [], G.GEN
    for _ in range((2 << nv) - 1):
        basis.append(cur)
        cur = G.mul(cur, tau)
    st = PP.pippenger_wg(pts, sc, y_size, y_log, d_log, x_log, clm, basis)
    out = GK.pippenger_dense_output(st["wg"], y_log, d_log)
    r = [rng.next_fr() for _ in range(y_log)]
    claims = GK.pippenger_claims(out, r)
    tape = [rng.next_bits(512) for _ in range(8000)]
    tr = PP.Transcript(tape)
    pair = PP.pippenger_prove(tr, st, claims, y_size, y_log, d_log, x_log, clm, basis, KN.setup_inverses(k, nv), k)
    dev_tape = [t % F.P if i in tr.wide else t & ((1 << 128) - 1) for i, t in enumerate(tape[: tr.pos])]
    return dict(shape=(x_log, d_log, y_size, y_log, clm), claims=claims, scalars=[v for m in tr.msgs for v in m], points=list(tr.points),
                tape=dev_tape, raw_tape=tape, pair=pair, g0=basis[0], k=k, tau=tau)


@pytest.fixture(scope="module", params=[(3, 2, 8, 0), (3, 2, 8, 1), (4, 3, 12, 2), (3, 2, 40, 4)])
def proof(request):
    return _oracle_proof(*request.param, seed=123 + sum(request.param))


def test_verifier_accepts_the_oracle_proof_and_returns_its_pair(proof):
    p = proof
    # the Python restatement of the reference verifier reads exactly what the prover wrote
    rt = V.ReadTranscript(p["scalars"], p["points"], p["raw_tape"])
    assert V.pippenger_verify(rt, p["claims"], p["shape"][2], p["shape"][3], p["shape"][1], p["shape"][0], p["shape"][4], p["g0"],
                              p["k"]) == p["pair"]
    got = VF.pippenger_verify(*p["shape"], p["claims"][0], p["claims"][1], p["g0"], p["k"], p["scalars"], p["points"], p["tape"])
    assert got["pair"] == p["pair"] and got["tape_used"] == len(p["tape"]) == rt.pos
    assert VF.kzg_verify_pair(got["pair"], PR.G2_GEN, PR.g2_mul(PR.G2_GEN, p["tau"]))     # the real pairing check


def test_verifier_rejects_tampered_proofs(proof):
    p = proof
    args = lambda **kw: (*p["shape"], kw.get("cp", p["claims"][0]), kw.get("ce", p["claims"][1]), p["g0"], p["k"],
                         kw.get("scalars", p["scalars"]), kw.get("points", p["points"]), kw.get("tape", p["tape"]))
    n = len(p["scalars"])
    for idx in (0, 1, n // 3, n // 2, n - 6, n - 1):           # a round message, final evaluations, the opening's scalars
        bad = list(p["scalars"])
        bad[idx] = (bad[idx] + 1) % F.P
        with pytest.raises(VF.Rejected):
            VF.pippenger_verify(*args(scalars=bad))
        with pytest.raises(V.VerifyError):
            V.pippenger_verify(V.ReadTranscript(bad, p["points"], p["raw_tape"]), p["claims"], p["shape"][2], p["shape"][3],
                               p["shape"][1], p["shape"][0], p["shape"][4], p["g0"], p["k"])
    bad_claim = list(p["claims"][1])
    bad_claim[0] = (bad_claim[0] + 1) % F.P
    with pytest.raises(VF.Rejected):
        VF.pippenger_verify(*args(ce=bad_claim))
    with pytest.raises(VF.Rejected):                            # truncated
        VF.pippenger_verify(*args(scalars=p["scalars"][:-1]))
    with pytest.raises(VF.Rejected):                            # trailing garbage
        VF.pippenger_verify(*args(scalars=p["scalars"] + [1]))
    with pytest.raises(VF.Rejected):                            # a point off the curve
        VF.pippenger_verify(*args(points=[(p["points"][0][0], (p["points"][0][1] + 1) % G.Q)] + p["points"][1:]))
    # a different commitment passes the algebraic checks (they never open it) but not the pairing
    swapped = [G.add(p["points"][0], G.GEN)] + p["points"][1:]
    got = VF.pippenger_verify(*args(points=swapped))
    assert got["pair"] != p["pair"]
    assert not VF.kzg_verify_pair(got["pair"], PR.G2_GEN, PR.g2_mul(PR.G2_GEN, p["tau"]))


def _cofactor_torsion_point():
    """a point of the curve y^2 = x^3 + 4 outside the prime-order subgroup: [r] P for an arbitrary curve point P (order | h)"""
    x = 5
    while True:
        x += 1
        rhs = (x ** 3 + 4) % G.Q
        y = pow(rhs, (G.Q + 1) // 4, G.Q)
        if y * y % G.Q != rhs:
            continue
        acc, p, k = None, (x, y), G.R_ORDER      # plain double-and-add: G.mul reduces its scalar mod r
        while k:
            if k & 1:
                acc = G.add(acc, p)
            p = G.double(p)
            k >>= 1
        if acc is not None:
            return acc


def test_verifier_rejects_points_outside_the_prime_order_subgroup(proof):
    """read_points deserialises with Validate::Yes (proof_transcript.rs:59-69 -> ark-ec): a commitment moved by a cofactor-torsion
    point is still on the curve but must be refused, in the recorded form and through the merlin reader's decompression"""
    import ctypes as C
    import numpy as np
    from gkr_msm_amd import codec, ffi
    p = proof
    t = _cofactor_torsion_point()
    assert G.on_curve(t)
    bad_pt = G.add(p["points"][0], t)
    assert G.on_curve(bad_pt)
    with pytest.raises(VF.Rejected) as e:
        VF.pippenger_verify(*p["shape"], p["claims"][0], p["claims"][1], p["g0"], p["k"], p["scalars"], [bad_pt] + p["points"][1:],
                            p["tape"])
    assert "subgroup" in str(e.value)
    # the compressed encoding of that point through the built-in ProofTranscript2 reader
    L = ffi.lib()

    def compress(pt):
        x, y = pt
        b = bytearray(x.to_bytes(48, "big"))
        b[0] |= 0x80
        if y > (G.Q - 1) // 2:
            b[0] |= 0x20
        return bytes(b)
    for pt, ok in ((p["points"][0], True), (bad_pt, False), (t, False)):
        proof_bytes = compress(pt)
        h = C.c_void_p()
        ffi.check(L.gm_merlin_create_verifier(b"x", 1, proof_bytes, len(proof_bytes), C.byref(h)))
        rd = ffi.GmTranscriptReader()
        ffi.check(L.gm_merlin_reader(h, C.byref(rd)))
        out = np.zeros(12, dtype=np.uint64)
        rc = rd.read_points(rd.ctx, 1, out.ctypes.data_as(ffi.u64p))
        assert (rc == 0) == ok
        if ok:
            assert codec.g1_aff_from_limbs(out)[0] == pt
        L.gm_merlin_destroy(h)


@pytest.mark.parametrize("lp,lb", [(1, 1), (3, 2), (4, 3)])
def test_gen1_verifier_on_the_oracle_prover_stream(lp, lb):
    """gm_gkr_msm_verify (BintreeVerifier / SumcheckPolyMapVerifier / SplitVerifier) accepts what pyref's gkr_msm_prove writes,
    ends on the same claim, and rejects an altered stream"""
    from pyref import gen1 as G1
    pts = F.random_points(1 << lp, 3 + lp)
    rng = F.SplitMix64(40 + lb)
    bits = [[bool(rng.next() & 1) for _ in range(1 << lb)] for _ in range(1 << lp)]
    tape = [rng.next_fr() for _ in range(3000)]
    claim, out, tr = G1.gkr_msm_prove(bits, pts, lp, lb, tape)
    got = VF.gkr_msm_verify(lp, lb, tr.msgs, tape[: tr.pos])
    assert got["point"] == claim[0] and got["evs"] == claim[1] and got["tape_used"] == tr.pos
    base = G1.base_layer(bits, pts, lp, lb)
    assert [G1.evaluate(b, got["point"]) for b in base] == got["evs"]   # the claim the caller checks against its commitments
    n = len(tr.msgs)
    for idx in (0, 3 << lb, n // 2, n - 1):
        bad = list(tr.msgs)
        bad[idx] = (bad[idx] + 1) % F.P
        with pytest.raises(VF.Rejected):
            VF.gkr_msm_verify(lp, lb, bad, tape[: tr.pos])
    with pytest.raises(VF.Rejected):
        VF.gkr_msm_verify(lp, lb, tr.msgs[:-1], tape[: tr.pos])


def test_mock_vk_is_the_generator_and_its_tau_multiple():
    tau = 0x5A5A5A5A1234567890ABCDEF
    h0, h1 = VF.kzg_mock_vk(tau)
    assert h0 == PR.G2_GEN and h1 == PR.g2_mul(PR.G2_GEN, tau)


def _load_proof_fixture():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_x3_d2_n8_clm1.json")) as f:
        fx = json.load(f)
    h = lambda v: int(v, 16)
    pt = lambda p: None if p is None else (h(p[0]), h(p[1]))
    return dict(shape=(fx["x_logsize"], fx["d_logsize"], fx["y_size"], fx["y_logsize"], fx["commitment_log_multiplicity"]),
                claims=([h(v) for v in fx["claim_point"]], [h(v) for v in fx["claim_evs"]]), k=fx["k"], tau=h(fx["tau"]),
                scalars=[h(v) for v in fx["transcript_scalars"]], points=[pt(p) for p in fx["transcript_points"]],
                tape=[h(v) for v in fx["challenges"]], pair=(pt(fx["pair"][0]), pt(fx["pair"][1])),
                pts=[(h(p[0]), h(p[1])) for p in fx["points_xy"]], sc=[h(v) for v in fx["scalars_in"]], nbits=fx["nbits"])


def test_committed_proof_fixture_verifies():
    """tests/golden/proof_x3_d2_n8_clm1.json (scripts/make_golden.py): a whole proof as data -- both verifiers accept it, return
    the recorded pairing pair, and the pairing check passes under the recorded tau"""
    p = _load_proof_fixture()
    got = VF.pippenger_verify(*p["shape"], p["claims"][0], p["claims"][1], G.GEN, p["k"], p["scalars"], p["points"], p["tape"])
    assert got["pair"] == p["pair"] and got["tape_used"] == len(p["tape"])
    rt = V.ReadTranscript(p["scalars"], p["points"], p["tape"])
    assert V.pippenger_verify(rt, p["claims"], p["shape"][2], p["shape"][3], p["shape"][1], p["shape"][0], p["shape"][4], G.GEN,
                              p["k"]) == p["pair"]
    h0, h1 = VF.kzg_mock_vk(p["tau"])
    assert VF.kzg_verify_pair(p["pair"], h0, h1)
    assert p["pair"][0] == G.mul(p["pair"][1], p["tau"])


def test_a_third_party_reader_gets_the_subgroup_check_unless_it_opts_out(proof):
    """gm_transcript_reader::points_validated: a reader that delivers raw affine points (flag 0, the zero-initialised default)
    keeps the reference's Validate::Yes guarantee -- the verifier runs the membership test itself; only a reader that says it has
    validated (the built-in merlin reader does) skips it"""
    import ctypes as C
    import numpy as np
    from gkr_msm_amd import codec, ffi
    p = proof
    L = ffi.lib()
    h = C.c_void_p()
    ffi.check(L.gm_merlin_create_verifier(b"x", 1, b"\0", 1, C.byref(h)))
    rd0 = ffi.GmTranscriptReader()
    ffi.check(L.gm_merlin_reader(h, C.byref(rd0)))
    assert rd0.points_validated == 1
    L.gm_merlin_destroy(h)
    bad_pt = G.add(p["points"][0], _cofactor_torsion_point())

    def run(points, validated):
        sc = codec.to_mont_limbs(list(p["scalars"])).reshape(-1)
        pt = codec.g1_aff_to_limbs(points).reshape(-1)
        tp = codec.ints_to_limbs(p["tape"]).reshape(-1)
        pos = {"s": 0, "p": 0, "t": 0}

        def serve(arr, key, width):
            def cb(ctx, n, out):
                a, b = pos[key] * width, (pos[key] + n) * width
                if b > len(arr):
                    return 1
                for i in range(a, b):
                    out[i - a] = int(arr[i])
                pos[key] += n
                return 0
            return cb

        def chal(ctx, n, bits, out):
            a, b = pos["t"] * 4, (pos["t"] + n) * 4
            if b > len(tp):
                return 1
            for i in range(a, b):
                out[i - a] = int(tp[i])
            pos["t"] += n
            return 0
        rd = ffi.GmTranscriptReader(None, ffi.READ_CB(serve(sc, "s", 4)), ffi.CHALLENGE_CB(chal), ffi.READ_CB(serve(pt, "p", 12)),
                                    1 if validated else 0, 0)
        cp, ce = codec.to_mont_limbs(list(p["claims"][0])), codec.to_mont_limbs(list(p["claims"][1]))
        g0l, kk = codec.g1_aff_to_limbs([p["g0"]]), codec.to_mont_limbs([p["k"]])
        pair = np.zeros(24, dtype=np.uint64)
        rc = L.gm_pippenger_verify_tr(*p["shape"], cp.ctypes.data, ce.ctypes.data, g0l.ctypes.data, kk.ctypes.data, C.byref(rd),
                                      pair.ctypes.data)
        return rc, (L.gm_last_error() or b"").decode()
    rc, _ = run(p["points"], False)
    assert rc == 0
    rc, msg = run([bad_pt] + p["points"][1:], False)
    assert rc != 0 and "subgroup" in msg
    rc, msg = run([bad_pt] + p["points"][1:], True)      # the reader vouched for its points: no membership test behind it
    assert rc == 0 or "subgroup" not in msg             # (gm_last_error keeps the previous call's text after a success)
