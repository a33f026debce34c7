"""The table of DESIGN.md section 9 from a bench line and the per-row rocprof summary:
   design_numbers.py profiles/r03_bench_all_configs.json.log profiles/r03_rows.json"""
import json, sys
bench = None
for ln in open(sys.argv[1]):
    ln = ln.strip()
    if ln.startswith("{") and '"metric"' in ln:
        bench = json.loads(ln)
rows = json.load(open(sys.argv[2]))
assert bench is not None
R = rows["rows"]
out = []
out.append(f"Headline (`value`): LZ4 round trip, 100 000 x 64 KiB uniform int32 as CHAR: **{bench['value']:.0f} GB/s** "
           f"(compress {bench['compress_ms']:.2f} ms = {bench['compress_GBps']:.0f} GB/s, decompress {bench['decompress_ms']:.2f} ms = "
           f"{bench['decompress_GBps']:.0f} GB/s); `roofline.frac` = {bench['roofline']['frac']:.3f} (compress), "
           f"{bench['roofline']['decompress_frac']:.2f} (decompress); geometric mean of the round trips of "
           f"{{uniform, harness, runs}} x {{CHAR, INT}}: **{bench['geomean_roundtrip_GBps']:.0f} GB/s**. "
           f"CPU baseline in the same run: {bench.get('cpu_baseline', {}).get('value', float('nan')):.1f} GB/s round trip "
           f"(liblz4, {bench.get('cpu_baseline', {}).get('cores', '?')} host threads).")
out.append("")
HAVE_REF = "reference_gpu" in bench
if HAVE_REF:
    out[0] += f" The reference's own kernels on the same GPU and buffers (`bench.py --ref`): {bench['reference_gpu']['roundtrip_GBps']:.1f} GB/s."
out.append("| row (bench `extra_keys[].row`) | ratio | compress GB/s | frac | decompress GB/s | frac | " + ("reference build on the same GPU: compress / decompress GB/s | " if HAVE_REF else "") + "compress kernel: rocprof avg ms, HBM-side traffic / algorithmic |")
out.append("|---|---|---|---|---|---|---|" + ("---|" if HAVE_REF else ""))


def ref_col(r):
    if not HAVE_REF:
        return ""
    g = r.get("reference_gpu")
    if g and "compress_GBps" not in g:   # (the headline: times only)
        nb = bench["compress_GBps"] * bench["compress_ms"] * 1e6
        g = {"compress_GBps": nb / g["compress_ms"] / 1e6, "decompress_GBps": nb / g["decompress_ms"] / 1e6}
    return (f"{g['compress_GBps']:.1f} / {g['decompress_GBps']:.0f} | " if g else "- | ")


def tag(e):
    return f" ({e['fetch_correction']})" if e.get("fetch_correction") else ""



def prof(key):
    e = R.get(key, {}).get("compress")
    if not e:
        return "-"
    return f"`{e['kernel']}` {e['avg_ms']:.2f} ms"


head_key = "lz4/uniform/char/100000"
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
e = R.get(head_key, {}).get("compress", {})
tr = e.get("traffic_bytes_per_launch")
out.append(f"| `{head_key}` (headline) | {bench['ratio']:.3f} | {bench['compress_GBps']:.0f} | {bench['roofline']['frac']:.3f} | "
           f"{bench['decompress_GBps']:.0f} | {bench['roofline']['decompress_frac']:.3f} | {ref_col(bench)}{prof(head_key)}"
           + (f", {tr / algo:.2f} x{tag(e)}" if tr else "") + " |")
for r in bench["extra_keys"]:
    key = r.get("row", "?")
    if "roofline" not in r:
        out.append(f"| `{key}` | {r['ratio']:.3f} | {r['compress_GBps']:.0f} | - | {r['decompress_GBps']:.0f} | - | {ref_col(r)}(host wall time; see section 7) |")
        continue
    rc = r["roofline"]["compress"]
    e = R.get(key, {}).get("compress", {})
    t = e["traffic_bytes_per_launch"] / rc["algorithmic_bytes_per_launch"] if e.get("traffic_bytes_per_launch") else None
    out.append(f"| `{key}` | {r['ratio']:.3f} | {r['compress_GBps']:.1f} | {r['hbm_frac_compress']:.3f} | {r['decompress_GBps']:.1f} | "
               f"{r['hbm_frac_decompress']:.3f} | {ref_col(r)}{prof(key)}" + (f", {t:.2f} x{tag(e)}" if t else "") + " |")
print("\n".join(out))
