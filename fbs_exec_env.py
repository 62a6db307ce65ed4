"""Top-level shim with the reference's module name.

Put the repository root on `sys.path` ahead of the reference's `fbs_mapper/` directory and
`from fbs_exec_env import *` (what fbs_mapper/map_to_fbs.py:2 does) picks up the MI355X executor:
the reference's mappers then build their program into this `LutExecEnv`, whose `eval` runs on ciphertexts.
"""
from tfhe_fbs_map_amd.fbs_exec_env import *  # noqa: F401,F403
from tfhe_fbs_map_amd.fbs_exec_env import ExecConfig, FbsExecEnv, LutExecEnv, parse_fbs, parse_lbf  # noqa: F401
