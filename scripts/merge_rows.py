"""Joins the parts written by scripts/collect_profiles.sh (ROWS=... PARTNAME=...) into one file:
   merge_rows.py profiles/r03_rows.json part1.json part2.json ...   (all parts must come from one kernel build)"""
import json, sys
out = None
for p in sys.argv[2:]:
    t = json.load(open(p))
    if out is None:
        out = t
    else:
        assert t["kernel_source_sha16"] == out["kernel_source_sha16"] or (
            t.get("device_asm_sha16") and t.get("device_asm_sha16") == out.get("device_asm_sha16")), "parts from different kernel builds"
        out["rows"].update(t["rows"])
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(sys.argv[1], sorted(out["rows"]))
