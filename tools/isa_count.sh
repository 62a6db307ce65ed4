#!/bin/bash
# instruction count of k_blind_rotate<10> (whole kernel; the CMUX loop is >99% of it) -- proxy for VALU time
cd /tmp/isa && hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s /root/repo/tfhe_fbs_map_amd/csrc/fbs_kernels.hip 2>&1 | grep -E "error" | head -5
python3 - <<'PY'
import re,collections
s=open('/tmp/isa/k.s').read()
m=re.search(r'^_ZN3fbs14k_blind_rotateILi10EEEvNS_6BrArgsE:(.*?)s_endpgm', s, re.S|re.M)
lines=[l.strip() for l in m.group(1).split('\n') if l.strip() and not l.strip().startswith((';','.'))]
c=collections.Counter(l.split()[0] for l in lines)
cost={'s_nop':1.0}
def w(op):
    if op.startswith('v_mad_u64'): return 1.8
    if op.startswith('s_'): return 0.3
    if op.endswith('_e64') or op in ('v_lshl_add_u64','v_mul_lo_u32','v_mul_hi_u32','v_lshlrev_b64','v_lshrrev_b64','v_alignbit_b32','v_add3_u32','v_bitop3_b32','v_lshl_add_u32','v_lshl_or_b32','v_and_or_b32') or op.startswith('v_cmp') and 'u64' in op: return 1.5
    return 1.0
tot=sum(c.values()); wt=sum(v*w(k) for k,v in c.items())
meta=s[m.end():m.end()+6000]
vg=re.search(r'NumVgprs: (\d+)', meta); sc=re.search(r'ScratchSize: (\d+)', meta)
print('BR<10> lines=%d weighted=%.0f nops=%d mads=%d vgpr=%s scratch=%s'%(tot,wt,c['s_nop'],c['v_mad_u64_u32'],vg and vg.group(1),sc and sc.group(1)))
print(c.most_common(14))
PY
