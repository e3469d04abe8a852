#!/usr/bin/env python3
"""Coefficients of the exp2 polynomials in sipnet_amd/csrc/fast_math.h.

2^f on [-0.5, 0.5] as 1 + f*(c1 + c2 f + ... + cN f^(N-1)): the interpolant of
g(f) = (2^f - 1)/f through the N Chebyshev nodes of the interval, computed in 50-digit decimal
arithmetic (so that the printed doubles are the correctly rounded coefficients of THAT
interpolant), then evaluated in IEEE double Horner form, as the kernel does, against
a 50-digit 2^f on 20 001 points.

usage: fit_exp2.py [degree ...]      (default: 11 = the fp64 parity build; 8, 9 for comparison)
"""
import sys
from decimal import Decimal as D, getcontext

getcontext().prec = 50
LN2 = D(2).ln()


def g(f):
    # (2^f - 1)/f = ln2 * sum_k (f ln2)^k / (k+1)!   (no cancellation near f = 0)
    z, term, s, k = f * LN2, D(1), D(0), 0
    while abs(term) > D(10) ** -48:
        s += term
        k += 1
        term = term * z / (k + 1)
    return LN2 * s


def cheb_nodes(n):
    # cos((2k+1) pi / 2n) / 2 through a 50-digit cosine (Taylor series)
    pi = D("3.14159265358979323846264338327950288419716939937510")

    def cos(x):
        s, term, k = D(0), D(1), 0
        while abs(term) > D(10) ** -48:
            s += term
            k += 2
            term = -term * x * x / (k * (k - 1))
        return s
    return [cos((2 * k + 1) * pi / (2 * n)) / 2 for k in range(n)]


def fit(degree):
    n = degree                      # g has degree-1 ... coefficients c1..cN
    xs = cheb_nodes(n)
    ys = [g(x) for x in xs]
    # Newton divided differences -> monomial coefficients, all in Decimal
    coef = list(ys)
    for j in range(1, n):
        for i in range(n - 1, j - 1, -1):
            coef[i] = (coef[i] - coef[i - 1]) / (xs[i] - xs[i - j])
    mono = [D(0)] * n
    for i in range(n - 1, -1, -1):          # mono = mono * (x - xs[i]) + coef[i]
        new = [D(0)] * n
        for k in range(n - 1):
            new[k + 1] += mono[k]
        for k in range(n):
            new[k] -= mono[k] * xs[i]
        new[0] += coef[i]
        mono = new
    return [float(c) for c in mono]         # c1 .. cN


def max_rel_err(c):
    worst = 0.0
    for i in range(20001):
        f = -0.5 + i / 20000.0
        p = c[-1]
        for ck in reversed(c[:-1]):
            p = p * f + ck                    # Horner in double (fma differs by < 1 ulp)
        p = p * f + 1.0
        exact = (D(f) * LN2).exp()
        worst = max(worst, abs(float((D(p) - exact) / exact)))
    return worst


if __name__ == "__main__":
    for deg in [int(a) for a in sys.argv[1:]] or [11]:
        c = fit(deg)
        print(f"degree {deg}: max |rel err| on [-0.5, 0.5] = {max_rel_err(c):.2e}")
        print("  " + ", ".join(repr(x) for x in c))
