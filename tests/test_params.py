"""Noise model (tfhe_fbs_map_amd/params.py): the stand-in for the patched concrete-optimizer the reference
shells out to (experiments/add_exec_estimates.py:9-16)."""
from tfhe_fbs_map_amd import MODULUS_BITS, P1024, margin_sigmas, params_for
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
