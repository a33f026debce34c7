"""print_rows.py bench.json [other.json ...]: the headline and every row of bench.py's JSON line(s), side by side."""
import json, sys
runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
for f, d in zip(sys.argv[1:], runs):
    r = d.get("roofline", {})
    print(f"{f}: value {d['value']:.1f} {d['unit']}  ms_per_step {d['ms_per_step']:.2f}  compress {d.get('compress_GBps', 0):.1f} decompress {d.get('decompress_GBps', 0):.1f}"
          f"  frac {r.get('frac', 0):.4f} / {r.get('decompress_frac', 0):.4f}  cpu {d.get('cpu_baseline', {}).get('value')}")
rows = {}
for i, d in enumerate(runs):
    for r in d.get("extra_keys", []):
        if isinstance(r, dict) and "row" in r:
            rows.setdefault(r["row"], [None] * len(runs))[i] = r
for k, v in rows.items():
    print(f"  {k:52s}" + "".join(f" | {x.get('compress_GBps', 0):8.1f} {x.get('decompress_GBps', 0):8.1f}" if x else " |        -        -" for x in v))
