"""profiles/<tag>/pmc_summary.txt -> profiles/pmc_traffic.json (HBM / LDS bytes per launch that bench.py reports as
roofline.traffic / roofline.lds).  usage: python scripts/pmc_traffic.py r02"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cur, d = None, {}
for l in open(os.path.join(ROOT, "profiles", tag, "pmc_summary.txt")):
    if not l.startswith(" "):
        cur = l.strip(); d[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+mean (\S+)", l)
        if m:
            d[cur][m.group(1)] = float(m.group(2))
k = lambda n: d["nvca::" + n]
KB = 1024
gray, lut = k("k_gray_fast4<3>"), k("k_lut")
integ = [k("k_colsum"), k("k_bandscan"), k("k_integral")]
band, group = k("k_band"), k("k_group")
deep = d.get("nvca::k_deep", {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_LDS": 0.0})      # round 4: the tile kernels walk the whole cascade, k_deep is not launched
per = {n: (x["FETCH_SIZE"] + x["WRITE_SIZE"]) * KB for n, x in (("k_band", band), ("k_deep", deep), ("k_group", group))}
sys.path.insert(0, ROOT)
import bench, time
out = {
    "_note": "HBM/fabric bytes per launch (32 x 1080p frames) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/%s/pmc_summary.txt), corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) x 1024 x 2 for the wide coalesced streaming reads of the gray and integral kernels (128-B requests tallied at 64 B); for the cascade group (tile staging + gathers, 64-B requests) FETCH_SIZE x 1024 equals TCC_EA0_RDREQ x 64 B, so it is taken as is. WRITE_SIZE (KB) x 1024." % tag,
    "gray_resize_hist": {"hbm_bytes_per_launch": round((gray["FETCH_SIZE"] * 2 + gray["WRITE_SIZE"]) * KB + (lut["FETCH_SIZE"] + lut["WRITE_SIZE"]) * KB, 1)},
    "integral": {"hbm_bytes_per_launch": round(sum((x["FETCH_SIZE"] * 2 + x["WRITE_SIZE"]) * KB for x in integ), 1)},
    "cascade": {"hbm_bytes_per_launch": round(sum(per.values()), 1), "per_kernel": {n: round(v, 1) for n, v in per.items()},
                "lds_bytes_per_launch": round((band["SQ_INSTS_LDS"] + deep["SQ_INSTS_LDS"]) * 64 * 3, 1),
                "_lds_note": "SQ_INSTS_LDS (wave instructions per launch: k_band %.1f M, k_deep %.1f M) x 64 lanes x 3 B mean access (u16 map look-ups and b32 sample reads in nearly equal numbers)" % (band["SQ_INSTS_LDS"] / 1e6, deep["SQ_INSTS_LDS"] / 1e6)},
    "_config": {"workload": "face1080p", "width": 1920, "height": 1080, "frames_per_launch": 32, "frames_resident": "hbm", "cascade": "calibrated",
                "src_sha16": bench.source_hash(), "collected": "profiles/%s/pmc_summary.txt, %s" % (tag, time.strftime("%Y-%m-%d"))},
}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("k_band: fetch %.3f GB, rdreq*64 %.3f GB; WAIT_ANY/WAVE_CYCLES %.3f; WAIT_INST_LDS/WAVE_CYCLES %.3f; BANK_CONFLICT/IDX_ACTIVE %.3f; LDS %.1f M, VALU %.1f M, VMEM_RD %.1f M wave instructions"
      % (band["FETCH_SIZE"] * KB / 1e9, band["TCC_EA0_RDREQ_sum"] * 64 / 1e9, band["SQ_WAIT_ANY"] / band["SQ_WAVE_CYCLES"], band["SQ_WAIT_INST_LDS"] / band["SQ_WAVE_CYCLES"],
         band["SQ_LDS_BANK_CONFLICT"] / band["SQ_LDS_IDX_ACTIVE"], band["SQ_INSTS_LDS"] / 1e6, band["SQ_INSTS_VALU"] / 1e6, band["SQ_INSTS_VMEM_RD"] / 1e6))
