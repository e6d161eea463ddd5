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
    fam = [r for r in rows if "gemm_conv_kernel" in r["Kernel_Name"] or "conv3x3_halo" in r["Kernel_Name"] or "gemm_pers_kernel" in r["Kernel_Name"] or "gemm_nloop_kernel" in r["Kernel_Name"]]
    return sum(float(r["Counter_Value"]) for r in fam), len(fam)


def family_counters(d):
    """{counter: sum over the family's launches of the last forward} of a multi-counter pass ({} when the pass is missing / failed)"""
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        return {}
    rows = [r for r in csv.DictReader(open(fs[0]))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = sorted({int(r["Dispatch_Id"]) for r in rows if "timestep_embedding" in r["Kernel_Name"]})
    first = marks[-2] if len(marks) >= 2 else 0        # the last forward: BrushNet's timestep embedding, then the UNet's
    out = {}
    for r in rows:
        if int(r["Dispatch_Id"]) >= first and ("gemm_conv_kernel" in r["Kernel_Name"] or "conv3x3_halo" in r["Kernel_Name"] or "gemm_pers_kernel" in r["Kernel_Name"] or "gemm_nloop_kernel" in r["Kernel_Name"]):
            out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


fetch_kb, n = family_sum(sys.argv[1])
write_kb, n2 = family_sum(sys.argv[2])
out = {"kernel": "gemm_conv_kernel family (all tile instantiations, incl. the persistent tiles 69 / 70), one denoise step, batch 4 x 512x512, bf16",
       "launches": n, "fetch_bytes_per_launch": round(2.0 * fetch_kb * 1024 / n), "write_bytes_per_launch": round(write_kb * 1024 / n2),
       "traffic_bytes_per_launch": round(2.0 * fetch_kb * 1024 / n + write_kb * 1024 / n2),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/profile_step.py; FETCH_SIZE x2 "
                 "(gfx950 wide-read correction); Infinity-Cache hits are included in both counters"}
if len(sys.argv) >= 5:
    rd, wr = family_counters(sys.argv[3]), family_counters(sys.argv[4])
    if rd and wr:
        r_all, r32, r_dram = (rd.get(k, 0.0) for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum"))
        w_all, w64, w_dram = (wr.get(k, 0.0) for k in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_DRAM_sum"))
        out["l2_fabric_requests"] = {
            "read_requests_per_launch": round(r_all / n), "read_32B_share": round(r32 / max(r_all, 1.0), 4),
            "read_destined_for_memory_controllers_share": round(r_dram / max(r_all, 1.0), 4),
            "write_requests_per_launch": round(w_all / n2), "write_64B_share": round(w64 / max(w_all, 1.0), 4),
            "write_destined_for_memory_controllers_share": round(w_dram / max(w_all, 1.0), 4),
            "note": "TCC_EA0_*REQ_DRAM count L2 -> fabric requests routed to the memory controllers (as opposed to GMI / IO); the Infinity "
                    "Cache sits behind that port, so its hits are inside this share too. rocprofv3 -L on this image lists no counter of "
                    "Infinity Cache hits or misses (the one description that mentions MALL is a TCC stall counter): HBM bytes cannot "
                    "be separated from Infinity Cache bytes with the counters available here."}
print(json.dumps(out, indent=1))
