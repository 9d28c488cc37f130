"""Dev helper: the headline, roofline and legs of a bench.py line at a glance.   python3 scripts/show_bench.py <file with the JSON line>"""
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(d['value'], d['ms_per_step'], d['blocks'])
r=d['roofline']; print({k:r[k] for k in ('frac','f32_frac','bf16_frac','avg_launch_ms','launches_ms','dense_launch_ms','dense_frac','traffic','mfma_busy') if k in r})
lg=d.get('legs',{})
for k in lg:
    v=lg[k]
    if isinstance(v,dict):
        print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in('ms_per_step','sweep_kernel_ms','stats_kernels_ms','it_per_s')}, {kk: round(v['roofline'][kk],3) for kk in ('frac','full_evals_per_wave_tile','b3_evals_per_wave_tile') if 'roofline' in v and kk in v['roofline'] and v['roofline'][kk] is not None})
print(d.get('host_master',{}).get('it_per_s'), d.get('host_master',{}).get('ratio_to_headline'))
if 'growth' in d: print(d['growth']['it_per_s_whole_run'], d['growth'].get('moving_labels'))
print(d.get('cpu_baseline'))
