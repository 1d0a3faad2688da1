"""profiles/pmc_traffic.json from a profile_bench.sh summary.json: HBM bytes per launch of each pipeline stage's kernel =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 (rocprofv3 reports KiB; gfx950 tallies 128-byte fetch requests as 64 bytes, so FETCH is
doubled: MI355X_MICROARCH.md, HBM / rocprofv3 section).  usage: make_pmc_traffic.py <summary.json> <tag> [out.json]"""
import json
import sys

STAGE_OF = {"gauss2d_mm": "gauss2d", "gauss_v_rw": "gauss_v", "gauss_h_rw": "gauss_h", "median3_threshold_colsum_kernel": "median3_threshold_colsum", "median3_oct_kernel": "median3", "otsu16_window_kernel": "median3_otsu16",
            "threshold_colsum_kernel": "threshold_colsum", "find_peaks_kernel": "find_peaks"}
summary = json.load(open(sys.argv[1]))
tag = sys.argv[2]
out = {"_comment": "HBM bytes per launch (256 frames 1024x1024 u16) = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 "
                   f"--pmc passes (FETCH doubled per MI355X_MICROARCH.md section HBM); see profiles/{tag}_rocprofv3_summary.txt"}
for name, rec in summary.items():
    for key, stage in STAGE_OF.items():
        if name.startswith(key) and "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            out[stage] = int(round((2 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024))
json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
