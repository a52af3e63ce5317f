"""profiles/traffic.json (what bench.py replays under --no-pmc) from the SAME passes bench.py makes live: tools/make_traffic_live.py <out.json> <label> <workload>:<photons> ..."""
import datetime, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out, label = sys.argv[1], sys.argv[2]
try:
    tj = json.load(open(out))
except Exception:
    tj = {}
for spec in sys.argv[3:]:
    work, nph = spec.split(':')
    t = bench.live_pmc(work, int(float(nph)))
    if not isinstance(t, dict):
        print(work, 'FAILED:', t); continue
    t['session'] = '%s, %s, tools/make_traffic_live.py (bench.live_pmc)' % (label, datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ'))
    tj[work] = t
    print(work, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items() if not isinstance(v, dict)})
json.dump(tj, open(out, 'w'), indent=1)
