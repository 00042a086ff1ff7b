"""Post-process the two PMC passes of tools/pmc_traffic.sh into per-launch HBM bytes of the GEMM kernel.
Corrections per MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B per lane) coalesced read stream, which is what this kernel's
operand and C-tile loads are (the C tile is read 8 B per lane: uncalibrated, so the doubled figure
is an upper bound); WRITE_SIZE is exact for streaming stores."""
import csv, glob, hashlib, json, os, sys

def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "gemm_f64_kernel_b" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n

fetch, n1 = per_kernel(sys.argv[1], "FETCH_SIZE")
write, n2 = per_kernel(sys.argv[2], "WRITE_SIZE")
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "andvaranaut_amd", "csrc", "gemm_f64.hip")
out = {
    "kernel": "gemm_f64_kernel_b", "launches": n1, "gemm_src_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
    "fetch_kib_raw_per_launch": fetch / max(n1, 1), "write_kib_per_launch": write / max(n2, 1),
    "hbm_bytes_per_launch": (2.0 * fetch / max(n1, 1) + write / max(n2, 1)) * 1024.0,
    "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as is; average over every gemm_f64_kernel_b launch of the run",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
