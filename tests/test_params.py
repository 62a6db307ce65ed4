"""Noise model (tfhe_fbs_map_amd/params.py): the stand-in for the patched concrete-optimizer the reference
shells out to (experiments/add_exec_estimates.py:9-16)."""
from tfhe_fbs_map_amd import MODULUS_BITS, P1024, P2048, bootstrap_cost, choose_params, margin_sigmas, params_for
from tfhe_fbs_map_amd._native import MODULUS


def test_defaults_have_margin_and_secure_noise_would_not():
    assert MODULUS == (1 << 46) - 62 * (1 << 13) + 1 and MODULUS_BITS == 46
    # default (reduced) noise: mul16@15 has norm2 84, aes_sbox@15 281 -- comfortably inside the box
    assert margin_sigmas(P1024.replace(p_msg=15), norm2=84) > 6
    assert margin_sigmas(P1024.replace(p_msg=15), norm2=281) > 5.5
    # what 128-bit security would need at N=1024 (~2^-25 of q) leaves no room for p=15: documented, not hidden
    secure = P1024.replace(p_msg=15, sigma_lwe=1 << 21, sigma_glwe=1 << 21)
    assert margin_sigmas(secure, norm2=84) < 2
    # p=31 does not fit N=1024 (modulus switch alone) and is sent to N=2048
    assert margin_sigmas(P1024.replace(p_msg=31), norm2=1) < 4.5
    assert params_for(31).log_n_poly == 11 and params_for(15).log_n_poly == 10
    assert margin_sigmas(params_for(31), norm2=325) > 5


def test_bytes_per_fbs_is_baselines_figure():
    assert P1024.bytes_per_fbs() == 103_309_328


def test_parameter_choice_follows_the_model():
    """choose_params = the optimizer's role in the reference's flow: cheapest shape with the asked-for margin."""
    for p, norm2 in ((15, 84), (15, 281), (7, 50), (31, 325), (2, 6)):
        c = choose_params(p, norm2)
        assert margin_sigmas(c, norm2) >= 6.0 and c.p_msg == p
        assert c.log_n_poly == (10 if p <= 16 else 11)
        assert bootstrap_cost(c) <= bootstrap_cost(params_for(p)) + 1e-9     # never dearer than the fixed default
    # the reference cost unit: the default N = 1024 set is 1.0 by definition, N = 2048 with one more level costs more
    assert abs(bootstrap_cost(P1024) - 1.0) < 1e-9 and bootstrap_cost(P2048) > 1.5
    # more noise -> finer, dearer gadget; hopeless noise -> the best margin available, not an exception
    noisy = choose_params(3, 20, sigma=1 << 21)
    assert margin_sigmas(noisy, 20) >= 6.0 and bootstrap_cost(noisy) > bootstrap_cost(choose_params(3, 20))
    hopeless = choose_params(15, 84, sigma=1 << 21)
    assert margin_sigmas(hopeless, 84) < 6.0
    # a larger norm never buys a cheaper set
    assert bootstrap_cost(choose_params(15, 20000)) >= bootstrap_cost(choose_params(15, 84))
