# A/B of fill tile shapes (GIQL_FILL_NT / GIQL_FILL_ITEMS builds): usage: fill_ab.sh lib1.so lib2.so ...
for rep in 1 2; do for lib in "$@"; do echo -n "$lib "; GIQL_HIP_LIB=$PWD/giql_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['phase_ms']['fill'])"; done; done
