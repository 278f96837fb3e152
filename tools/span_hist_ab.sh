# A/B of the histogram-in-the-span-pass form on the headline bench (alternating runs, one gpurun call)
for rep in 1 2 3; do for off in 1 0; do echo -n "no_span_hist=$off "; GIQL_HIP_NO_SPAN_HIST=$off timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['config']['pairs_per_step'], r['phase_ms'])"; done; done
