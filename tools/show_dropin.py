"""print the drop-in timings of a bench.py JSON line:  python tools/show_dropin.py gpurun_out/<file>.json"""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print('value', d['value'], 'ms/step', round(d['ms_per_step'], 2), 'roofline frac', round(d['roofline']['frac'], 3))
for k, c in d['dropin_e2e'].items():
    print(k)
    for n in ('cold', 'cold_no_disk_cache', 'second', 'warm'):
        if n in c:
            print('  ', n, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in c[n].items()})
    print('   get_sino floor', round(c['get_sino_floor_s'], 3), 'over floor', round(c['get_sino_over_floor'], 2))
