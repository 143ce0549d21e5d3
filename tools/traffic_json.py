"""profiles/<tag>/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh (run where gpurun_out/<tag>/ is):
NeRF-level launches of the featurisation kernel only (its FEW_LEVELS = false instantiation: a distinct kernel name since r04), bytes per launch."""
import csv, glob, json, sys
tag = sys.argv[1]
half = len(sys.argv) > 2 and sys.argv[2] == "half"          # the half-table gather of the mixed-precision render (bench.py --autocast)
tot = {"FETCH_SIZE": [0, 0.0], "WRITE_SIZE": [0, 0.0]}
for f in glob.glob(f"gpurun_out/{tag}/pmc_*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        is_half = "DF16_" in name or "_Float16" in name
        if r["Counter_Name"] in tot and "k_march_features" in name and ("ILj2E" in name or "<2" in name) and is_half == half and "bwd" not in name:
            if "Lb0E" in name or ", false>" in name:                      # the FEW_LEVELS name tag (r04): false = the NeRF-level launches
                tot[r["Counter_Name"]][0] += 1
                tot[r["Counter_Name"]][1] += float(r["Counter_Value"]) * 1024.0       # KB units
n = tot["FETCH_SIZE"][0]
out = {"kernel": "k_march_features<2, 256, float, false> (FEW_LEVELS = false: NeRF level)" + (", half tables, bf16 features" if half else ""), "launches": n, "rays_per_launch": 10240,
       "fetch_bytes_per_launch": round(tot["FETCH_SIZE"][1] / max(n, 1)),
       "write_bytes_per_launch": round(tot["WRITE_SIZE"][1] / max(tot["WRITE_SIZE"][0], 1)),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KB units) over `bench.py --steps 1 --warmup 1 "
                 "--no-cpu-baseline --no-train` (tools/pmc_passes.sh), NeRF-level launches only (the kernel's FEW_LEVELS = false instantiation); FETCH_SIZE = TCC_EA0_RDREQ x 64 B on gfx950 (a 128-byte request counts as "
                 "64: the true figure is between 1x and 2x); Infinity-Cache hits are included",
       "algorithmic_gather_bytes_per_launch": 10240 * 786432 // (2 if half else 1)}
json.dump(out, open(f"gpurun_out/{tag}/traffic{'_autocast' if half else ''}.json", "w"), indent=1)
print(out)
