"""tfhe_fbs_map_amd -- MI355X-native executor for the FBS programs of ssmiler/tfhe_fbs_map.

`fbs_exec_env.LutExecEnv` (alias `FbsExecEnv`) is the drop-in for the reference's
fbs_mapper/fbs_exec_env.py; `_native` binds libfbsexec.so (C ABI: include/fbs_exec.h), which holds
the hand-written gfx950 kernels.  Importing the package loads the shared library and fails loudly
if it has not been built -- there is no CPU execution path here.
"""
from . import _native
from ._native import MODULUS, MODULUS_BITS, Context, FbsError, Params, Program, TvSet
from .fbs_exec_env import ExecConfig, FbsExecEnv, LutExecEnv, min_fbs_size, parse_fbs, parse_lbf, table_is_valid
from .netlist import BitExecEnv, map_basic, parse_blif, parse_bristol
from .params import P1024, P2048, bootstrap_cost, choose_params, margin_sigmas, params_for, security_bits, sigma_min

__all__ = ["Context", "FbsError", "Params", "Program", "TvSet", "ExecConfig", "FbsExecEnv", "LutExecEnv",
           "min_fbs_size", "parse_fbs", "parse_lbf", "table_is_valid", "P1024", "P2048", "margin_sigmas",
           "params_for", "bootstrap_cost", "choose_params", "security_bits", "sigma_min", "BitExecEnv", "map_basic", "parse_blif", "parse_bristol"]
