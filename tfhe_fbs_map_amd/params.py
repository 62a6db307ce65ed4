"""Parameter sets, the security floor on the noise, a noise model and the parameter selector.

The reference never fixes cryptographic parameters: it hands (precision p = number of plaintext values, squared
2-norm of the linear combinations) -- `stats()["norm2_linprod"]`, fbs_mapper/fbs_exec_env.py:245-276 -- to a patched
concrete-optimizer (experiments/add_exec_estimates.py:9-16) that returns `k, N, n, br_l, br_b, ks_l, ks_b, cost`
(experiments/concrete.patch:163) at a fixed security level and error probability (`--security-level`, default 128;
`--p-error`, default "4 sigma", concrete.patch:56,139-140), under the contract that the noise entering a bootstrap
stays inside half a box, q/(4p), at that probability (concrete.patch:21-27: 2^(log q - 2) / precision).  That
optimizer is Rust, un-vendored and unbuildable here.  `choose_params` is its stand-in for THIS executor: same inputs,
same outputs, the same contract, this executor's own noise formulas and cost unit.

All variances are relative to q^2 (torus units).

Security.  `sigma_min(dim)` is the smallest noise standard deviation for which an LWE instance of dimension `dim`
with a binary secret is estimated to resist 2^128 operations.  It restates the 128-bit line of the security curves
that concrete-optimizer itself uses (concrete-security-curves, generated with the lattice estimator; the optimizer's
`minimal_variance_lwe` evaluates log2(sigma/q) = slope * dim + bias and floors it at 2^2/q):
        slope = -0.0265994623, bias = 2.9815431841   (security level 128, binary keys)
[NOT IN REFERENCE: restated from the public source of zama-ai/concrete at the release the reference pins
(README.md:17); neither that source nor the estimator is available offline, so the two constants are unverifiable
here.  Cross-check against published TFHE-rs 128-bit parameter sets: n = 742 -> 2^-16.8 here against 2^-17.1
published; N = 2048 -> 2^-51.5 against 2^-51.6: this line is the more conservative of the two.]
The line does not depend on q; the floor does: with the 46-bit modulus of this executor no noise is ever below
2^-44 q, which is why N = 2048 keys are noisier here than under a 64-bit modulus (and still far inside the budget).

`P1024` -- BASELINE.md's synthetic benchmark shape n=630 N=1024 k=1 l=3 beta=7 t=8 gamma=2 -- is kept for the
throughput benchmark with REDUCED NOISE (sigma = 2^-40 q): at 128-bit noise an N = 1024 accumulator cannot carry
p = 15 (see `margin_sigmas`), so that set is a kernel benchmark shape, not a secure configuration, and says so
(`Params.security_bits`).
"""
from __future__ import annotations

import math

from ._native import Params
from .security import MODULUS, MODULUS_BITS

# ---- security floor (constants and sigma_min live in security.py, which has no dependencies) --------------------
from .security import MIN_LOG2_SIGMA_ABS, SECURITY_CURVES, log2_sigma_min, sigma_min      # noqa: E402,F401


def security_bits(prm: Params) -> float:
    """Rough security estimate of a parameter set: 128 (or more) when both noises sit on or above the 128-bit
    line, otherwise 128 scaled by dimension over the dimension that noise would need (security is close to
    linear in the dimension at fixed log(q/sigma))."""
    slope, bias = SECURITY_CURVES[128]
    worst = float("inf")
    for dim, sigma in ((prm.n, prm.sigma_lwe), (prm.k * prm.N, prm.sigma_glwe)):
        rel = math.log2(max(sigma, 1) / MODULUS)
        need = (rel - bias) / slope                       # dimension at which this sigma is 128-bit secure
        worst = min(worst, 128.0 * dim / max(need, 1.0))
    return worst


def with_secure_noise(prm: Params, security: int = 128) -> Params:
    return prm.replace(sigma_lwe=sigma_min(prm.n, security), sigma_glwe=sigma_min(prm.k * prm.N, security))


REDUCED_SIGMA = 1 << 6                                                    # 2^-40 q

# BASELINE.md's benchmark shape, REDUCED NOISE (not secure; see the module docstring)
P1024 = Params(n=630, log_n_poly=10, l_bsk=3, beta_bsk=7, t_ksk=8, gamma_ksk=2).reduced_noise(REDUCED_SIGMA)
# its N = 2048 sibling for p = 31 (24 gadget bits as three levels of eight), REDUCED NOISE as well
P2048 = P1024.replace(log_n_poly=11, l_bsk=3, beta_bsk=8)


def params_for(p: int, norm2: int | None = None) -> Params:
    """The reduced-noise benchmark set for plaintext modulus p: the modulus switch alone (n = 630) leaves about
    N / (10.2 p) standard deviations between a value and the edge of its box, so p <= 16 fits N = 1024 and p <= 32
    wants N = 2048.  For a secure set use `choose_params`."""
    base = P1024 if p <= 16 else P2048
    return base.replace(p_msg=p)


# ---- noise model -------------------------------------------------------------------------------------------------
def variances(prm: Params):
    """(blind-rotate output, key switch, modulus switch) variances in torus units."""
    q = float(MODULUS)
    N, n, k, l, t = prm.N, prm.n, prm.k, prm.l_bsk, prm.t_ksk
    B, b2 = 2.0 ** prm.beta_bsk, 2.0 ** prm.gamma_ksk
    s_glwe, s_lwe = prm.sigma_glwe / q, prm.sigma_lwe / q
    # external product: (k+1) l N balanced digits of variance (B^2+2)/12 against key noise in every step, plus the
    # rounding of the decomposition (half an ulp of q/B^l) seen through a binary GLWE key -- the latter only in the
    # steps whose LWE key bit is 1 (the CMUX output is s_i times the rounded difference): half of them.
    # Measured on the GPU this lands 15-35 % above the observed noise (tests/test_gpu_parity.py).
    if prm.bsk_group == 2:
        # two key bits per step: n/2 external products by a bundle of three keys, each times a monomial difference
        # X^e - 1 (squared norm 2): 6 sigma^2 of key noise per product where two single steps carry 2; the rounding of
        # the decomposition of ACC is seen through X^(a0 s0 + a1 s1) - 1, non-zero for three of the four key-bit pairs
        v_br = n / 2.0 * ((k + 1) * l * N * (B * B + 2) / 12.0 * 6.0 * s_glwe ** 2
                          + 1.5 * (1 + k * N / 2.0) / (12.0 * B ** (2 * l)))
    else:
        v_br = n * ((k + 1) * l * N * (B * B + 2) / 12.0 * s_glwe ** 2 + 0.5 * (1 + k * N / 2.0) / (12.0 * B ** (2 * l)))
    # key switch: kN t balanced digits in [-2^g/2, 2^g/2) against key noise, plus the rounding to t*gamma bits
    # seen through a binary key
    v_ks = k * N * (t * (b2 * b2 + 2) / 12.0 * s_lwe ** 2 + 0.5 / (12.0 * b2 ** (2 * t)))
    # modulus switch, mean-compensated (k_ms_body, fbs_kernels.hip): the body's own rounding, and the n mask roundings seen through
    # s_i - 1/2 (variance 1/4) because the expected value sum eps_i / 2 is taken off the body before it is rounded
    v_ms = (1 + n / 4.0) / (12.0 * (2.0 * N) ** 2)
    return v_br, v_ks, v_ms


def margin_sigmas(prm: Params, norm2: float = 1.0) -> float:
    """Half box width q/(4p) divided by the standard deviation of the phase that enters the blind rotation, when
    the inputs of the linear combination are bootstrap outputs (the reference's contract, concrete.patch:21-27:
    norm2 * bootstrap noise + key switch + modulus switch against 2^(log q - 2)/p)."""
    v_br, v_ks, v_ms = variances(prm)
    sigma = math.sqrt(norm2 * v_br + v_ks + v_ms)
    return (1.0 / (4.0 * prm.p_msg)) / sigma


def p_error(margin: float) -> float:
    """Probability that a Gaussian leaves +-margin standard deviations (one bootstrap)."""
    return math.erfc(margin / math.sqrt(2.0))


def bootstrap_cost(prm: Params) -> float:
    """Relative cost of one functional bootstrap, in the unit the kernels are bound by (FP64 instruction issue), with
    the default P1024 shape = 1.  Per CMUX step and coefficient, from the instruction counts of k_blind_rotate (DESIGN.md
    section 5): (k+1)(l+1) transforms of log N / 2 butterflies of 8 instructions, (k+1)^2 l exact products of 7 (accumulate
    included), 2 l for the digits and 12 for rotation, rounding and re-centring; times n steps, times what the multi-wave
    transforms of N >= 2048 cost on top -- and what they do NOT gain from the whole-CU workgroups and the priority
    hand-over that sped the benchmark shape up at the end of round 2 (measured: 1.19 at N = 2048, 1.22 at N = 4096).
    Against measurements (profiles/r02/selector_bench.jsonl, blind rotation per 1024-batch over P1024's 9.33 ms): N = 1024
    l = 2 0.67-0.76 modelled / 0.69-0.82 measured; N = 2048 l = 1 1.41 / 1.41; N = 2048 l = 2 2.32 / 2.30; N = 4096 l = 2
    5.5 / 5.5.  The kN t (n+1) multiply-adds of the key switch are weighted by their measured share
    (1.2 % of the P1024 step since it runs as an int8 GEMM on the matrix cores; 5 % before).  The reference ranks parameter
    sets by the optimizer's `boot_cost` (experiments/analyse_results.py:10); this is the same ranking for this executor."""
    N, n, k, l, t = prm.N, prm.n, prm.k, prm.l_bsk, prm.t_ksk

    def blind(n_, l_, N_, log_n, k=k):
        per_coefficient = (k + 1) * ((l_ + 1) * 4.0 * log_n + 7.0 * (k + 1) * l_ + 2.0 * l_ + 12.0)
        return n_ * N_ * per_coefficient * (1.22 if log_n >= 12 else 1.19 if log_n == 11 else 1.0)

    # two key bits per step (bsk_group = 2), MEASURED against one bit per step at the same shape (profiles/r02): 0.74 with one
    # gadget level (9.8 against 13.2 ms per 1024 bootstraps at N = 2048, n = 714 / 710).  With l levels the bundle costs
    # 6 l exact products per coefficient and pair of key bits where it saves l + 1 transforms: it does not pay beyond
    # l = 1 in the two-waves-per-polynomial kernel (P1024, l = 3: 13.5 against 10.8 ms).  N = 2048 with TWO levels runs on the
    # whole-CU kernel, which has the registers for it: 17.1 ms per 1024 bootstraps at n = 758 against 21.9 ms at n = 766 with one
    # bit per step (round 3: what the 128-bit sets for p = 31 now take)
    pairs = 1.0 if prm.bsk_group != 2 else (0.74 if l == 1 else 0.79 if (l == 2 and prm.log_n_poly == 11) else 1.25)
    # GLWE dimension k = 2 at N = 1024 (k_blind_rotate_pairs_k2, round 3: three wave-private 1024-point transforms each way on three
    # waves per bootstrap, twelve waves per CU), MEASURED: 7.50 ms per 1024 bootstraps at n = 760 against P1024's 9.26
    if k == 2 and prm.log_n_poly == 10 and prm.bsk_group == 2 and l == 1:
        pairs = 0.867
    elif k >= 2:
        # every other k >= 2 (k_blind_rotate_glwe: k + 1 waves per bootstrap): modelled instructions times what the kernel was MEASURED
        # to take per modelled instruction, shape by shape (profiles/r04/glwe_calibration.txt), over P1024's 9.2 ms per 1 024
        # (a further level costs a step more than its instructions -- its transform hangs on the one before: at p = 2, 566 steps of
        # two levels at k = 2 take 1.93 ms per launch of 64 where 570 steps of one level at k = 3 take 1.44, at the same throughput)
        return (glwe_instructions(prm) / 1e6 * glwe_ms_per_minstr(prm.log_n_poly, k) / 9.2 * (1.0 + 0.04 * (l - 1))
                + 0.012 * (k * N * t * (n + 1)) / (1024 * 8 * 631.0))
    return 0.988 * pairs * blind(n, l, N, prm.log_n_poly) / blind(630, 3, 1024, 10, 1) + 0.012 * (k * N * t * (n + 1)) / (1024 * 8 * 631.0)


def glwe_shape_built(k: int, log_n: int) -> bool:
    """Is there a blind-rotation kernel for GLWE dimension k >= 2 at N = 2^log_n (csrc/fbs_blind_rotate_glwe.hip: FBS_GLWE_SHAPES)?"""
    return (k in (2, 3, 4) and log_n in (8, 9)) or (k in (2, 3) and log_n == 10)


def glwe_instructions(prm: Params) -> float:
    """FP64 wave-instructions per bootstrap of k_blind_rotate_glwe, modelled: per step and coefficient (k+1) [(l+1) transforms of log N / 2
    butterflies of 8 + (k+1) l products of 7 -- four times that with two key bits per step, the bundle's three products included -- + 2 l
    for the digits + 12 for rounding and re-centring], n (or n / 2) steps, 64 lanes."""
    k, l, group = prm.k, prm.l_bsk, prm.bsk_group if prm.bsk_group == 2 else 1
    per_coef = (k + 1) * ((l + 1) * 4.0 * prm.log_n_poly + 7.0 * (k + 1) * l * (4 if group == 2 else 1) + 2.0 * l + 12.0)
    return prm.n / group * prm.N * per_coef / 64.0


# ms per 1 024 bootstraps per million modelled instructions, in full rounds of workgroups (tools/glwe_calibrate.py on the 128-bit sets
# for p = 4 of every shape; k_blind_rotate_pairs_k2<10,4> on the same scale: 2.48): by (log2 N, k), the slower of one / two key bits per step
GLWE_MS_PER_MINSTR = {(8, 4): 3.04, (9, 2): 2.47, (9, 3): 2.70, (9, 4): 3.04, (10, 2): 3.37, (10, 3): 2.69}


def glwe_ms_per_minstr(log_n: int, k: int) -> float:
    return GLWE_MS_PER_MINSTR.get((log_n, k), 3.1)


# ---- selector ------------------------------------------------------------------------------------------------------
DEFAULT_GLWE_DIMS = (1, 2, 3)    # the GLWE dimensions ExecConfig admits (fbs_exec_env.ExecConfig.glwe_dims)
REFERENCE_MARGIN = 4.0          # the optimizer's default p_error is "4 sigma" (concrete.patch:56: default_value_t = _4_SIGMA)

_GADGETS = [(l, beta) for l in (1, 2, 3, 4, 5, 6) for beta in range(3, 24) if 10 <= l * beta <= 30]
_SWITCHES = [(t, g) for g in (1, 2, 3, 4, 5, 6) for t in range(1, 24) if 8 <= t * g <= 30]


def _switch_fits(t, g, N):
    bits = MODULUS_BITS + g + math.log2(t * N)
    return bits <= 63.9 and bits - 32.0 <= 31.9          # the key-switch kernels' 64-bit accumulators (dev_supported)


def choose_params(p: int, norm2: float = 1.0, min_margin: float = 6.0, security: int | None = 128,
                  sigma: int | None = None, poly_sizes=(9, 10, 11, 12), n_range=(450, 1200, 4),
                  floor_margin: float | None = None, groups=(1, 2), glwe_dims=(1,)) -> Params:
    """Cheapest parameter set (n, N, l, beta, t, gamma and both noises) for plaintext modulus p and squared 2-norm
    `norm2` whose modelled margin is at least `min_margin` standard deviations -- what the reference obtains from its
    patched optimizer for (precision, sq_norm2) (experiments/add_exec_estimates.py:9-16, concrete.patch:21-27,163).

    security = 128: each noise is the smallest the security line allows for its dimension (`sigma_min`); n runs over
    `n_range`, N over 2^poly_sizes (512 .. 4096; k = 1: the kernels' shape), the gadget over 1..6 levels of 3..23 bits, the key
    switch over 1..23 levels of 1..6 bits.  security = None with `sigma`: the same search at a fixed noise (the
    reduced-noise benchmark setting).  `groups`: key bits per blind-rotation step to consider (2 = the multi-bit form: half
    the steps on bundles of three GGSW samples, 1.5x the key, more noise per step; built for N >= 1024, l <= 5, even n).
    `glwe_dims`: GLWE dimensions to consider.  k = 2 at N = 1024 with two key bits per step and one gadget level has kernels of its
    own (k_blind_rotate_pairs_k2, k_blind_rotate_cu_k2): the noise floor of k N = 2048 on 1024-point transforms, what p = 15 takes at
    ordinary norms.  Every other k = 2, 3, 4 at N = 256 / 512 (k <= 3 at N = 1024), any depth, one or two key bits per step, runs on
    k_blind_rotate_glwe (k + 1 waves per bootstrap) and is priced by that kernel's measured cost: k = 3 at N = 512 -- k N = 1536, a
    noise floor between the two k = 1 offers, on 512-point transforms -- is what p <= 7 takes (183-208 k FBS/s in full rounds against
    151-164 k at k = 2, 1.6-1.9 ms per launch of up to one bootstrap per CU against 2.0-2.4).
    Cost = `bootstrap_cost`.  Raises ValueError when nothing reaches `min_margin`
    (p too large for N <= 4096 at this security level); with `floor_margin` the requirement is first relaxed in steps of
    half a sigma down to that floor."""
    import numpy as np
    if floor_margin is not None and floor_margin < min_margin:
        m = min_margin
        while True:
            try:
                return choose_params(p, norm2, m, security, sigma, poly_sizes, n_range, None, groups, glwe_dims)
            except ValueError:
                if m <= floor_margin:
                    raise
                m = max(floor_margin, m - 0.5)
    q = float(MODULUS)
    ns = np.arange(*n_range, dtype=np.float64)
    if security is not None:
        s_lwe = np.array([sigma_min(int(n), security) for n in ns]) / q
    else:
        s_lwe = np.full(ns.shape, float(sigma if sigma is not None else REDUCED_SIGMA) / q)
    need = (1.0 / (4.0 * p) / min_margin) ** 2              # largest admissible variance
    best = None
    for k, log_n in [(k_, ln) for k_ in glwe_dims for ln in poly_sizes]:
        if k >= 2 and not glwe_shape_built(k, log_n):
            continue                                        # no kernel for this (k, N)
        N = 1 << log_n
        s_glwe = (sigma_min(k * N, security) if security is not None else (sigma if sigma is not None else REDUCED_SIGMA)) / q
        v_ms = (1 + ns / 4.0) / (12.0 * (2.0 * N) ** 2)
        if (v_ms >= need).all():
            continue
        for (l, beta), group in [((l_, b_), g_) for (l_, b_) in _GADGETS for g_ in groups]:
            if group == 2 and k == 1 and (log_n < 10 or l > 5):
                continue
            if (k + 1) * l > 20:
                continue
            B = 2.0 ** beta
            key_term = (k + 1) * l * N * (B * B + 2) / 12.0 * s_glwe ** 2
            round_term = (1 + k * N / 2.0) / (12.0 * B ** (2 * l))
            v_br = ns * (key_term + 0.5 * round_term) if group == 1 else ns / 2.0 * (6.0 * key_term + 1.5 * round_term)
            room = need - v_ms - norm2 * v_br
            if (room <= 0).all():
                continue
            for t, g in _SWITCHES:
                if not _switch_fits(t, g, k * N):
                    continue
                b2 = 2.0 ** g
                v_ks = k * N * (t * (b2 * b2 + 2) / 12.0 * s_lwe ** 2 + 0.5 / (12.0 * b2 ** (2 * t)))
                ok = np.nonzero(v_ks <= room)[0]
                if ok.size == 0:
                    continue
                n = int(ns[ok[0]])                          # cost grows with n: the smallest feasible n is the cheapest
                cand = Params(n=n, log_n_poly=log_n, k=k, l_bsk=l, beta_bsk=beta, t_ksk=t, gamma_ksk=g, p_msg=p,
                              sigma_lwe=int(round(s_lwe[ok[0]] * q)), sigma_glwe=int(round(s_glwe * q)), bsk_group=group)
                key = (bootstrap_cost(cand), n, l, t)
                if best is None or key < best[0]:
                    best = (key, cand)
    if best is None:
        raise ValueError("no parameter set with N <= %d reaches %.1f sigma at p = %d, norm2 = %g"
                         % (1 << max(poly_sizes), min_margin, p, norm2))
    return best[1]


# ---- execution estimates: the reference's cost unit, in this executor's terms -------------------------------------------
# One MI355X at the benchmark shape P1024 (cost 1.0) in whole rounds of 1 024 bootstraps: profiles/r03/batch_sweep.txt,
# 106-110 k FBS/s by the box.  Everything `exec_estimate` says about time is this figure scaled by `bootstrap_cost`.
MI355X_FBS_PER_S_AT_COST_1 = 107e3


def exec_estimate(p: int, norm2: float, nb_bootstrap: int, samples: int = 1, **choose) -> dict:
    """What the reference's flow computes for a mapped circuit with its patched optimizer (experiments/add_exec_estimates.py:9-16:
    `boot_cost` of (precision = fbs_size, sq_norm2 = norm2_linprod); experiments/analyse_results.py: `total_cost = nb_bootstrap x
    boot_cost`), restated for THIS executor: the parameter set `choose_params(p, norm2)` picks, its `bootstrap_cost` (issue slots of
    these kernels, P1024 = 1) as `boot_cost`, `total_cost = nb_bootstrap x boot_cost`, and what that is in time on one MI355X when
    the levels are wide enough to fill whole rounds (`seconds`; narrow levels pay one bootstrap's latency per level instead:
    distributed.launch_ms prices those)."""
    prm = choose_params(p, norm2, **choose)
    cost = bootstrap_cost(prm)
    total = nb_bootstrap * cost
    return dict(params=prm, boot_cost=cost, total_cost=total, margin_sigmas=margin_sigmas(prm, norm2),
                seconds=total * samples / MI355X_FBS_PER_S_AT_COST_1, samples=samples)
