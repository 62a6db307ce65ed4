"""The ciphertext modulus and the security floor on the noise (no dependencies: imported by both `_native` and
`params`).  See `params` for where the numbers come from and what they are used for."""
from __future__ import annotations

import math

MODULUS = 0x3FFFFFF84001      # q = 2^46 - 62*2^13 + 1 (prime); ciphertext and NTT modulus, 64-bit words
MODULUS_BITS = 46

# log2(sigma/q) = slope * dimension + bias: the 128-bit line of the security curves concrete-optimizer uses for binary
# secrets (restated from the public source; provenance and cross-checks in params.py)
SECURITY_CURVES = {128: (-0.026599462343105267, 2.981543184145991)}
MIN_LOG2_SIGMA_ABS = 2.0      # no noise below 2^2 in absolute units (the optimizer's floor)


def log2_sigma_min(dim: int, security: int = 128, log_q: int = MODULUS_BITS) -> float:
    """log2 of the smallest secure noise standard deviation, relative to q, for LWE dimension `dim`."""
    slope, bias = SECURITY_CURVES[security]
    return max(slope * dim + bias, MIN_LOG2_SIGMA_ABS - log_q)


def sigma_min(dim: int, security: int = 128) -> int:
    """The same in absolute units of this executor's modulus, rounded up to an integer."""
    return max(1, math.ceil(2.0 ** (log2_sigma_min(dim, security) + math.log2(MODULUS))))
