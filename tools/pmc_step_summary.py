"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_step.sh for the gemm_conv kernel family: bytes per launch
over the LAST BrushNet+UNet forward of the run.  FETCH_SIZE is reported in KB and, on gfx950, counts wide (16 B/lane)
reads at half their size (MI355X_MICROARCH.md, HBM): it is doubled here.  WRITE_SIZE (KB) is exact for 16-byte stores."""
import csv, glob, json, sys


def family_sum(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # the last forward starts at the second-to-last timestep_embedding kernel (BrushNet's, then the UNet's)
    marks = [i for i, r in enumerate(rows) if "timestep_embedding" in r["Kernel_Name"]]
    rows = rows[marks[-2]:]
    fam = [r for r in rows if "gemm_conv_kernel" in r["Kernel_Name"] or "conv3x3_halo" in r["Kernel_Name"]]
    return sum(float(r["Counter_Value"]) for r in fam), len(fam)


fetch_kb, n = family_sum(sys.argv[1])
write_kb, n2 = family_sum(sys.argv[2])
out = {"kernel": "gemm_conv_kernel family (all tile instantiations), one denoise step, batch 4 x 512x512, bf16",
       "launches": n, "fetch_bytes_per_launch": round(2.0 * fetch_kb * 1024 / n), "write_bytes_per_launch": round(write_kb * 1024 / n2),
       "traffic_bytes_per_launch": round(2.0 * fetch_kb * 1024 / n + write_kb * 1024 / n2),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/profile_step.py; FETCH_SIZE x2 "
                 "(gfx950 wide-read correction); Infinity-Cache hits are included in both counters"}
print(json.dumps(out, indent=1))
