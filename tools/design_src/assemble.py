"""DESIGN.md is assembled from the pieces in this directory; the @PLACEHOLDER@ numbers come from the committed bench line
(profiles/r03_bench_line.json) and size sweep (profiles/r03_msm_sizes.log).   python tools/design_src/assemble.py <CPU tests> <GPU tests>"""
import json, re, sys
import os; here=os.path.dirname(os.path.abspath(__file__)); root=os.path.dirname(os.path.dirname(here))+'/'
bench=json.loads(open(root+'profiles/r03_bench_line.json').read().strip().splitlines()[-1])
sizes={}
for line in open(root+'profiles/r03_msm_sizes.log'):
    m=re.match(r"2\^(\d+) auto .*device ([\d.]+) wall", line)
    if m: sizes[int(m.group(1))]=float(m.group(2))
ex=bench['extras']; sh=ex['shard_sizes']
def shard(lg): return sh['msm_ms_at_2^%d'%lg]['ms']
t1=bench['ms_per_step']
vals={
 'MSM26':'%.1f'%t1, 'MSM26PPS':'%.2f·10⁸'%(bench['value']/1e8),
 'ACC26':'%.1f'%bench['phases_ms']['accumulate'], 'PRE26':'%.1f'%bench['phases_ms']['before_first_accumulate'],
 'TAIL26':'%.1f'%(bench['phases_ms']['device_total']-bench['phases_ms']['accumulate']-bench['phases_ms']['before_first_accumulate']),
 'MSM25':'%.1f'%sizes.get(25,shard(25)), 'MSM24':'%.1f'%sizes.get(24,shard(24)), 'MSM23':'%.2f'%sizes.get(23,shard(23)), 'MSM22':'%.1f'%sizes.get(22,0),
 'MSM20':'%.2f'%sizes.get(20,shard(20)), 'MSM18':'%.2f'%sizes.get(18,0), 'MSM16':'%.2f'%sizes.get(16,shard(16)), 'MSM14':'%.2f'%sizes.get(14,0),
 'BN26':'%.2f·10⁸'%(ex['alt_bn128_g1_msm_points_per_s']/1e8),
 'HOST26':'%.3f'%ex['mult_pippenger_inf_host_buffers']['2^26']['seconds'],
 'NTTFWD':'%.3f'%bench['ntt']['forward_ms'], 'NTTINV':'%.3f'%bench['ntt']['inverse_ms'], 'NTTFRAC':'%.3f'%bench['ntt']['roofline']['frac'],
 'EFF2':'%.2f'%(t1/(2*shard(25))), 'EFF4':'%.2f'%(t1/(4*shard(24))), 'EFF8':'%.2f'%(t1/(8*shard(23))),
 'NCPU':sys.argv[1], 'NGPU':sys.argv[2],
}
parts=[open(here+'/design_new_head.md').read(), open(here+'/design_cov.md').read(), open(here+'/design_p1.md').read(), open(here+'/design_p2.md').read(),
       open(here+'/design_p3.md').read(), open(here+'/design_new_s4.md').read(), "\n", open(here+'/design_new_s5.md').read(), "\n", open(here+'/design_new_s6.md').read(), "\n",
       open(here+'/design_new_s7.md').read(), "\n", open(here+'/design_p8.md').read(), open(here+'/design_new_s9.md').read()]
out="".join(parts)
for k,v in vals.items(): out=out.replace('@%s@'%k, v)
left=re.findall(r'@[A-Z0-9]+@', out)
assert not left, left
open(root+'DESIGN.md','w').write(out)
print(len(out), vals)
