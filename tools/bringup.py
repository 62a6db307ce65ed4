"""GPU bring-up: device NTT, keys, one small batch, P1024 batch vs the oracle."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tfhe_fbs_map_amd import _native as nat
from oracle import tfhe_oracle as orc

def check(params, B, tables, seed=1, label=""):
    t0 = time.time()
    ctx = nat.Context(params, seed=seed)
    print(label, ctx.device_info, "keygen %.2fs" % (time.time() - t0), flush=True)
    rng = np.random.default_rng(42)
    N = params.N
    a = rng.integers(0, nat.MODULUS, N, dtype=np.uint64); b = rng.integers(0, nat.MODULUS, N, dtype=np.uint64)
    c = ctx.debug_polymul(a, b)
    print(" polymul == oracle ntt:", np.array_equal(c, orc.polymul_ntt(a, b)), flush=True)
    o = orc.Oracle(params, seed=seed)
    keys = ctx.export_keys(); okeys = o.keys()
    print(" keygen identical:", all(np.array_equal(keys[k], okeys[k]) for k in keys), flush=True)
    msgs = rng.integers(0, params.p_msg, B)
    cts = ctx.encrypt(msgs, nonce0=7)
    print(" encrypt identical:", np.array_equal(cts, o.encrypt(msgs, nonce0=7)), flush=True)
    ids = (np.arange(B) % len(tables)).astype(np.uint32)
    tv = ctx.tvset(tables)
    t0 = time.time(); out = ctx.bootstrap_batch(tv, cts, ids); dt = time.time() - t0
    print(" gpu batch %d in %.3fs" % (B, dt), flush=True)
    nref = min(B, 16)
    ref, used = o.bootstrap_batch(cts[:nref], tables, ids[:nref])
    print(" ciphertexts bit-exact vs oracle (first %d):" % nref, np.array_equal(out[:nref], ref), flush=True)
    dec = ctx.decrypt(out)
    exp = np.array([tables[i][m] for i, m in zip(ids, msgs)])
    print(" decrypt == table[msg]:", np.array_equal(dec, exp), " sha", hashlib.sha256(out.tobytes()).hexdigest()[:12], flush=True)
    return ctx, tv

rng = np.random.default_rng(3)
toy = nat.Params(n=16, log_n_poly=10, p_msg=7, sigma_lwe=1 << 8, sigma_glwe=1 << 8)
check(toy, 8, [[0,1,1,0,1,0,0], [0,1,2,3,2,1,0], [0,1,1,0,1,0,0,1,0,0,1,0,1,1]], label="toy N=1024")
for logn in (8, 9, 11):
    check(toy.replace(log_n_poly=logn, n=12), 4, [[0,1,1,0,1,0,0]], label="toy N=%d" % (1 << logn))
tables = [[0] + list(rng.integers(0, 2, 14)) for _ in range(16)]
P = nat.Params()
ctx, tv = check(P, 64, tables, label="P1024")
# timing on device-resident buffers
import ctypes
B = 1024
msgs = rng.integers(0, 15, B); cts = ctx.encrypt(msgs)
ids = (np.arange(B) % 16).astype(np.uint32)
for rep in range(3):
    t0 = time.time(); out = ctx.bootstrap_batch(tv, cts, ids); dt = time.time() - t0
    print("P1024 host-buffer batch 1024: %.3fs -> %.0f FBS/s" % (dt, B / dt), flush=True)
ctx.profile(True)
out = ctx.bootstrap_batch(tv, cts, ids)
print(ctx.profile_read())
dec = ctx.decrypt(out); exp = np.array([tables[i][m] for i, m in zip(ids, msgs)])
print("decrypt ok:", np.array_equal(dec, exp))
