"""Parameter sets and a small noise model.

The reference never fixes cryptographic parameters: it hands (precision p, squared 2-norm of the
linear combinations) -- `stats()["norm2_linprod"]`, fbs_mapper/fbs_exec_env.py:245-276 -- to a patched
concrete-optimizer (experiments/add_exec_estimates.py:9-16, experiments/concrete.patch:21-27) that is
not available here.  This module is the stand-in: the BASELINE.md set P1024 and its N=2048 sibling,
plus textbook CGGI variance formulas so a caller can see how many standard deviations of margin a
(parameter set, p, norm2) combination has.  All variances are relative to q^2 (torus units).

The default noise (sigma = 2^6 on the 46-bit modulus, i.e. 2^-40 relative) is REDUCED NOISE: it makes
N=1024 correct for p=15 with generous margin but is far below what 128-bit security needs at these
dimensions (~2^-25 at N=1024, which would not leave room for p=15).  Throughput does not depend on it.
"""
from __future__ import annotations

import math

from ._native import MODULUS, Params

P1024 = Params()                                            # n=630 N=1024 k=1 l=3 beta=7 t=8 gamma=2
# p = 31 needs the wider accumulator, and its linear combinations (norm2 up to ~325) need a finer decomposition than
# P1024's 21 bits: 24 bits.  At the (reduced) key noise of the defaults a wide base costs nothing, so the 24 bits are
# 3 levels of 8 rather than 4 of 6 (one forward transform less per component and step; same margin in the model
# below, 6.3 sigma at norm2 = 325).
P2048 = Params(n=630, log_n_poly=11, l_bsk=3, beta_bsk=8)

def params_for(p: int, norm2: int | None = None) -> Params:
    """Default set for plaintext modulus p: the modulus switch alone (n = 630) leaves about
    N / (10.2 p) standard deviations between a value and the edge of its box, so p <= 16 fits
    N = 1024 (>= 6 sigma) and p <= 32 wants N = 2048."""
    base = P1024 if p <= 16 else P2048
    return base.replace(p_msg=p)


def variances(prm: Params):
    """(blind-rotate output, key switch, modulus switch) variances in torus units."""
    q = float(MODULUS)
    N, n, k, l, t = prm.N, prm.n, prm.k, prm.l_bsk, prm.t_ksk
    B, b2 = 2.0 ** prm.beta_bsk, 2.0 ** prm.gamma_ksk
    s_glwe, s_lwe = prm.sigma_glwe / q, prm.sigma_lwe / q
    # external product: (k+1) l N digits of variance (B^2+2)/12 against key noise in every step, plus the
    # rounding of the decomposition (half an ulp of q/B^l) seen through a binary GLWE key -- the latter only in the
    # steps whose LWE key bit is 1 (the CMUX output is s_i times the rounded difference): half of them.
    # Measured on the GPU this lands 15-35 % above the observed noise (tests/test_gpu_parity.py).
    v_br = n * ((k + 1) * l * N * (B * B + 2) / 12.0 * s_glwe ** 2 + 0.5 * (1 + k * N / 2.0) / (12.0 * B ** (2 * l)))
    # key switch: unsigned digits in [0, 2^gamma): E[d^2] = (2^g - 1)(2^(g+1) - 1)/6
    ed2 = (b2 - 1) * (2 * b2 - 1) / 6.0
    v_ks = k * N * (t * ed2 * s_lwe ** 2 + 0.5 / (12.0 * b2 ** (2 * t)))
    v_ms = (1 + n / 2.0) / (12.0 * (2.0 * N) ** 2)
    return v_br, v_ks, v_ms


def margin_sigmas(prm: Params, norm2: float = 1.0) -> float:
    """Half box width q/(4p) divided by the standard deviation of the phase that enters the
    blind rotation, when the inputs of the linear combination are bootstrap outputs."""
    v_br, v_ks, v_ms = variances(prm)
    sigma = math.sqrt(norm2 * v_br + v_ks + v_ms)
    return (1.0 / (4.0 * prm.p_msg)) / sigma


def bootstrap_cost(prm: Params) -> float:
    """Relative cost of one functional bootstrap, in the unit the kernels are bound by: n CMUX steps of (k+1)(l+1)
    transforms of N log N butterflies plus (k+1)^2 l N exact products, and the kN t (n+1) multiply-adds of the key
    switch weighted by their measured share (5 % of the P1024 step).  The reference ranks parameter sets by the
    optimizer's `boot_cost` (experiments/analyse_results.py:10); this is the same ranking for this executor."""
    N, n, k, l, t = prm.N, prm.n, prm.k, prm.l_bsk, prm.t_ksk
    blind = n * ((k + 1) * (l + 1) * N * prm.log_n_poly / 2.0 * 1.14 + (k + 1) ** 2 * l * N)
    switch = k * N * t * (n + 1)
    p1024_blind = 630 * (2 * 4 * 1024 * 5 * 1.14 + 4 * 3 * 1024)
    p1024_switch = 1024 * 8 * 631
    return 0.95 * blind / p1024_blind + 0.05 * switch / p1024_switch


def choose_params(p: int, norm2: float = 1.0, min_margin: float = 6.0, sigma: int | None = None) -> Params:
    """Cheapest gadget / key-switch shape whose modelled margin at (p, norm2) is at least `min_margin` standard
    deviations -- what the reference obtains from its patched optimizer for (precision, squared 2-norm)
    (experiments/add_exec_estimates.py:9-16, experiments/concrete.patch:21-27,66-74).  n and the noise are those of
    the default sets (`sigma` overrides both standard deviations); N is the smallest power of two whose modulus switch
    leaves room for p.  Falls back to the shape with the largest margin when none reaches `min_margin`."""
    base = params_for(p)
    if sigma is not None:
        base = base.replace(sigma_lwe=sigma, sigma_glwe=sigma)
    # conventional shapes only: 2..6 levels of 4..12 bits, key-switch digits of 1..4 bits (the executor itself takes
    # wider ones; with the reduced default noise the model would happily pick a single 19-bit level)
    gadgets = [(l, beta) for l in (2, 3, 4, 5, 6) for beta in range(4, 13) if 12 <= l * beta <= 30]
    switches = [(t, g) for t, g in ((4, 4), (5, 3), (6, 3), (8, 2), (10, 2), (16, 1), (20, 1))
                if 46 + g + math.log2(t * base.N) <= 63.9]
    best, best_key = None, None
    for l, beta in gadgets:
        for t, g in switches:
            cand = base.replace(l_bsk=l, beta_bsk=beta, t_ksk=t, gamma_ksk=g)
            m = margin_sigmas(cand, norm2)
            key = (0, bootstrap_cost(cand), -m) if m >= min_margin else (1, -m, bootstrap_cost(cand))
            if best_key is None or key < best_key:
                best, best_key = cand, key
    return best
