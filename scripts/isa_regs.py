"""VGPR / AGPR / SGPR / spill / LDS per kernel from a gfx950 assembly file (hipcc -save-temps).

usage: python scripts/isa_regs.py <file.s> [substring of the demangled name]
Measurement aid: waves per SIMD = min(8, 512 // roundup(vgpr, 8)).
"""
import re,sys,subprocess
s=open(sys.argv[1]).read()
blocks=s.split('amdhsa.kernels:')[1]
rows=[]
for blk in blocks.split('  - .agpr_count:')[1:]:
    n=re.search(r'\.name:\s+(\S+)',blk).group(1)
    v=re.search(r'\.vgpr_count:\s+(\d+)',blk).group(1)
    sg=re.search(r'\.sgpr_count:\s+(\d+)',blk).group(1)
    sp=re.search(r'\.vgpr_spill_count:\s+(\d+)',blk).group(1)
    ag=blk.split('\n')[0].strip()
    lds=re.search(r'\.group_segment_fixed_size:\s+(\d+)',blk).group(1)
    rows.append((n,v,ag,sg,sp,lds))
dn=subprocess.run(['c++filt']+[r[0] for r in rows],capture_output=True,text=True).stdout.strip().split('\n')
for d,r in zip(dn,rows):
    d=re.sub(r'\(.*','',d)
    if len(sys.argv)>2 and sys.argv[2] not in d: continue
    print(f'{d[:80]:80s} vgpr {r[1]:>4s} agpr {r[2]:>3s} sgpr {r[3]:>3s} spill {r[4]} lds {r[5]}')
