"""Table of scripts/gpu_rank_emulation.sh: every emulated rank's ms per pass, the slowest rank per N, and the ratio
(N=1 pass / N) / slowest rank -- the per-rank COMPUTE share of a strong-scaling efficiency (no reduce, no xGMI in it)."""
import json
import sys

n1 = json.load(open(sys.argv[1]))
rows = [json.loads(l) for l in open(sys.argv[2]) if l.startswith('{')]
t1 = n1['ms_per_step']
print(f'# rank emulation on one MI355X: C4 3600x1800x75x12 float64, 65 transects; N=1 pass = {t1:.3f} ms '
      f'(value {n1["value"]:.4e} integrals/s, K1 {n1["roofline"]["avg_launch_ms"]} ms per launch)')
print('# each line: one process doing exactly what rank r of N does (its slabs, its launches), no reduce; NOT a scaling run')
print(f'{"N":>2} {"rank":>4} {"slabs":>11} {"steps":>7} {"launches":>8} {"ms/pass":>8} {"k_flux":>7} {"k_expand":>8} {"k3":>6}  partial-step planes')
groups = {}
for r in rows:
    e = r['emulated_rank']
    key = (e['of'], e['partial_step_planes'])
    groups.setdefault(key, []).append(e)
    print(f'{e["of"]:>2} {e["rank"]:>4} {e["slabs"][0]:>5}-{e["slabs"][1]:<5} {e["steps_touched"][0]:>3}-{e["steps_touched"][1]:<3} '
          f'{e["launches_per_pass"]:>8} {e["ms_per_pass"]:>8.3f} {e["k_flux_ms"]:>7.3f} {e["k_expand_ms"]:>8.3f} {e["k3_ms"]:>6.3f}  {e["partial_step_planes"]}')
print()
print('# per N: slowest rank, ideal = N=1 pass / N, compute-only ratio = ideal / slowest (the all-reduce of 220 KB and the barrier come on top)')
for (n, mode), es in sorted(groups.items()):
    worst = max(e['ms_per_pass'] for e in es)
    print(f'N={n} [{mode}]: ranks {len(es)}, slowest {worst:.3f} ms, mean {sum(e["ms_per_pass"] for e in es) / len(es):.3f} ms, '
          f'ideal {t1 / n:.3f} ms, ideal/slowest = {t1 / n / worst:.3f}')
