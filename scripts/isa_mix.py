"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc -save-temps ... -> <src>-hip-amdgcn-amd-amdhsa-gfx950.s).

usage: python scripts/isa_mix.py <file.s> [substring of the demangled kernel name ...]
Counts per kernel: VALU, v_mov, v_mov_dpp (an unfused DPP step), packed ops, global loads / stores, fp64 ops, divisions / rcp,
exp / log, ds_bpermute / ds_swizzle (a __shfl), MFMA, LDS ops, barriers.  Measurement aid, not product code.
"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
labels = [(m.start(), m.group(1)) for m in re.finditer(r'^(_ZN3cal\w+):', s, re.M)]
rows = []
for pos, name in labels:
    body = s[pos:s.find('s_endpgm', pos)]
    c = lambda p: len(re.findall(p, body, re.M))
    rows.append((name, c(r'^\s*v_'), c(r'v_mov_b32_e32'), c(r'v_mov_b32_dpp'), c(r'v_pk_'), c(r'global_load|buffer_load'), c(r'global_store'),
                 c(r'_f64'), c(r'v_div_|v_rcp'), c(r'v_exp_f32|v_log_f32'), c(r'ds_bpermute|ds_swizzle'), c(r'v_mfma'), c(r'^\s*ds_'), c(r's_barrier')))
names = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split('\n')
want = sys.argv[2:]
for n, r in zip(names, rows):
    if want and not any(w in n for w in want):
        continue
    n = re.sub(r'\(.*', '', n)
    print(f'{n[:52]:52s} valu {r[1]:5d} mov {r[2]:4d} mov_dpp {r[3]:3d} pk {r[4]:3d} ld {r[5]:3d} st {r[6]:3d} f64 {r[7]:4d} div {r[8]:3d} '
          f'exp {r[9]:2d} bperm {r[10]:3d} mfma {r[11]:3d} ds {r[12]:4d} bar {r[13]:2d}')
