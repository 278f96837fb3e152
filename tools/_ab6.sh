for rep in 1 2 3; do for lib in libv_early.so libv_late.so; do echo -n "$lib "; GIQL_HIP_LIB=$PWD/giql_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['phase_ms']; print(d['ms_per_step'], d['config']['pairs_per_step'], r['sort_scatter'])"; done; done
