"""Proofs by enumeration for the exact-by-construction float64 sequences the kernels use instead of IEEE divisions.

csrc/edge_exact.h (es_edge): skimage's Scharr magnitude ends in ``np.sqrt(out) / np.sqrt(2)``; the kernels divide by the
constant c = RN(sqrt 2) = 1.4142135623730951 with Markstein's three operations

    q0 = RN(g * r);   rem = RN(g - q0 * c)  (one fma);   q = RN(q0 + rem * r)  (one fma),      r = RN(1 / c).

Claim: q == RN(g / c) for EVERY normal float64 g (scaling by powers of two changes nothing, so g in [1, 2) suffices).
Proof: with u = 2^-53, q0 = (g/c)(1 + e1)(1 + e2), |e| <= u, so g - q0 c = -g (e1 + e2 + e1 e2); the fma rounds that to
rem = (g - q0 c)(1 + e3); the value the last fma rounds is therefore T = g/c + (g/c - q0)(e1 + e3 + e1 e3), i.e.
|T - g/c| <= (g/c)(2u + u^2)^2 < (g/c) * 4.001 * 2^-106.  q can differ from RN(g/c) only if a rounding boundary (the midpoint
m of two adjacent float64) lies between g/c and T, so only if |g/c - m| < (g/c) * 2^-103.99.  Writing g = G 2^-52,
c = C 2^-52 and m = M 2^-54 (g < c: quotient in [1/2, 1)) or M 2^-53 (g >= c: quotient in [1, 2)) with M odd, that is
|G 2^54 - M C| <= 8 resp. |G 2^53 - M C| <= 4: for each residue j the congruence M = -j C^-1 (mod 2^54 resp. 2^53) leaves at
most two candidates.  The test enumerates them with margin (|j| <= 10 resp. 6) and evaluates the three operations exactly
(rational arithmetic, round-to-nearest-even): all round correctly.  Same idea as tests/fma_quotient_check.c, which checks
pl_quot exhaustively because its divisors vary.
"""
from fractions import Fraction as F

C_FLOAT = 1.4142135623730951
R_FLOAT = float.fromhex("0x1.6a09e667f3bccp-1")          # the kernels' r16 * 16


def rn(q: F) -> float:
    """q (positive, in the normal range) rounded to the nearest float64, ties to even."""
    if q == 0:
        return 0.0
    e = q.numerator.bit_length() - q.denominator.bit_length()
    if F(2) ** e > q:
        e -= 1
    if F(2) ** (e + 1) <= q:
        e += 1
    scaled = q / F(2) ** (e - 52)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > F(1, 2) or (rem == F(1, 2) and n % 2 == 1):
        n += 1
    return float(n) * 2.0 ** (e - 52)


def three_ops(g: float) -> float:
    q0 = rn(F(g) * F(R_FLOAT))
    rem_exact = F(g) - F(q0) * F(C_FLOAT)
    rem = (1 if rem_exact >= 0 else -1) * rn(abs(rem_exact)) if rem_exact != 0 else 0.0
    t = F(q0) + F(rem) * F(R_FLOAT)
    return rn(t)


def test_reciprocal_constant_is_correctly_rounded():
    assert rn(1 / F(C_FLOAT)) == R_FLOAT
    assert float.fromhex("0x1.6a09e667f3bcdp+4") == 16 * C_FLOAT        # the kernels' c16
    assert float.fromhex("0x1.6a09e667f3bccp-5") == R_FLOAT / 16        # the kernels' r16


def test_division_by_sqrt2_constant_every_candidate_rounds_correctly():
    c_int = int(F(C_FLOAT) * 2**52)
    assert F(c_int, 2**52) == F(C_FLOAT) and c_int % 2 == 1
    checked = 0
    for shift, jmax, in_case in ((54, 10, lambda g: g < c_int), (53, 6, lambda g: g >= c_int)):
        mod = 2**shift
        cinv = pow(c_int, -1, mod)
        for j in range(-jmax, jmax + 1):
            m0 = (-j * cinv) % mod
            for m in range(m0, 2**54, mod):
                if m < 2**53 or m % 2 == 0:
                    continue
                num = m * c_int + j
                if num % mod:
                    continue
                g_int = num // mod
                if not (2**52 <= g_int < 2**53) or not in_case(g_int):
                    continue
                g = float(g_int) * 2.0**-52
                assert three_ops(g) == rn(F(g) / F(C_FLOAT)), g.hex()
                checked += 1
    assert checked >= 4                                                  # the candidates exist, and all of them pass


def test_division_by_sqrt2_constant_random_and_scaled():
    import random

    random.seed(1)
    for _ in range(3000):
        g = random.uniform(1, 2) * 2.0 ** random.randint(-3, 22)         # the range of RN(sqrt(K)), K = S0^2 + S1^2 < 2^43
        assert three_ops(g) == g / C_FLOAT
