"""End-to-end circuit throughput through the drop-in API (encrypt + fbs_eval + decrypt timed separately)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import load_fixture, subsample
from tfhe_fbs_map_amd import P1024, Context, parse_fbs, _native as nat

for name, T in (("mul16__search_p15", 1000), ("adder128__search_p15", 1000), ("trivium_stream_short128__search_p15", 1000),
                ("trivium_stream_v2__search_p15", 64)):
    rec = load_fixture(name)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env.lower()
    ins, expect = subsample(rec, T)
    if os.environ.get("SECURE"):               # the 128-bit set of params.choose_params at the program's (p, norm2)
        from tfhe_fbs_map_amd import choose_params
        prm = choose_params(15, env.stats()["norm2_linprod"], glwe_dims=(1, 2))       # what ExecConfig() asks for
        print("  chosen: n=%d N=%d k=%d l=%d beta=%d t=%d gamma=%d, %d key bit(s) per step" % (
            prm.n, prm.N, prm.k, prm.l_bsk, prm.beta_bsk, prm.t_ksk, prm.gamma_ksk, prm.bsk_group))
    else:
        prm = P1024
    ctx = Context(prm, seed=1)
    tv = ctx.tvset(low["tables"])
    prog = nat.Program(ctx, tv, len(low["input_names"]), low["kind"], low["arg0"], low["arg1"], low["const_coef"],
                       low["term_coef"], low["term_src"], low["out_wire"])
    bits = np.stack([ins[n] for n in low["input_names"]])
    t0 = time.time(); cts = ctx.encrypt(bits); t_enc = time.time() - t0
    prog.eval(cts[:, :4].copy(), 4)            # warm-up (allocations)
    ctx.profile(True); ctx.profile_read()
    t0 = time.time(); out = prog.eval(cts, T); t_eval = time.time() - t0
    prof = ctx.profile_read()
    t0 = time.time(); dec = ctx.decrypt(out); t_dec = time.time() - t0
    ok = all(np.array_equal(dec[k], expect[n]) for k, n in enumerate(low["out_names"]) if low["out_wire"][k] >= 0)
    nfbs = prog.n_bootstrap * T
    print("%-40s depth %3d width<=%4d  %8d FBS  eval %.2fs -> %.0f FBS/s  (kernels: ks %.0f br %.0f lin %.0f ms)  enc %.2fs dec %.2fs ok=%s"
          % (name, prog.depth, prog.max_width, nfbs, t_eval, nfbs / t_eval, prof["keyswitch"]["ms"], prof["blind_rotate"]["ms"],
             prof["lincomb"]["ms"], t_enc, t_dec, ok), flush=True)
    ctx.close()
