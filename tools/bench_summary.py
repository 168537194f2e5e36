"""Print the headline fields of a bench.py JSON line (last line of the given log)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.2f} ms/step  roofline[{r['kernel']}] {r['bound']} ({r.get('limited_by', r['bound'])}) frac {r['frac']:.3f} "
      f"avg {r['avg_launch_us']:.1f} us  cpu {(d.get('cpu_baseline') or {}).get('value')}")
for k, v in list(d.get('abi_ms_per_2_steps', {}).items())[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f'  {k:50s} {v}')
