# A/B of experimental builds on the headline bench: usage: phase_ab.sh <phase> lib1.so lib2.so ...
ph="$1"; shift
for rep in 1 2; do for lib in "$@"; do echo -n "$lib "; GIQL_HIP_LIB=$PWD/giql_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | PH=$ph python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['config']['pairs_per_step'], r['phase_ms'][os.environ['PH']])"; done; done
