"""Post-process the two PMC passes of tools/prof_round.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py) into
per-launch HBM bytes of the dominant GEMM kernel and of the covariance assembly kernel.
Corrections per MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B per lane) coalesced read stream, which is what the GEMM kernel's
operand and C-tile loads are (the C tile is read 8 B per lane: uncalibrated, so the doubled figure
is an upper bound); WRITE_SIZE is exact for streaming stores.  The assembly reads only X (8*N*d bytes, L2-resident
after the first tiles) and writes the lower triangle of K once: its WRITE_SIZE is the figure the north star asks for."""
import csv, glob, hashlib, json, os, sys

def per_kernel(d, counter, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n

fetch, n1 = per_kernel(sys.argv[1], "FETCH_SIZE", "gemm_f64_kernel_b")
write, n2 = per_kernel(sys.argv[2], "WRITE_SIZE", "gemm_f64_kernel_b")
afetch, a1 = per_kernel(sys.argv[1], "FETCH_SIZE", "assemble_kernel")
awrite, a2 = per_kernel(sys.argv[2], "WRITE_SIZE", "assemble_kernel")
here = os.path.dirname(os.path.abspath(__file__))
sha = lambda f: hashlib.sha256(open(os.path.join(here, "..", "andvaranaut_amd", "csrc", f), "rb").read()).hexdigest()[:16]
out = {
    "kernel": "gemm_f64_kernel_b", "launches": n1, "gemm_src_sha16": sha("gemm_f64.hip"),
    "fetch_kib_raw_per_launch": fetch / max(n1, 1), "write_kib_per_launch": write / max(n2, 1),
    "hbm_bytes_per_launch": (2.0 * fetch / max(n1, 1) + write / max(n2, 1)) * 1024.0,
    "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as is; average over every gemm_f64_kernel_b launch of the run",
    "assemble": {
        "kernel": "assemble_kernel", "launches": a1, "src_sha16": sha("assemble.hip"),
        "fetch_kib_raw_per_launch": afetch / max(a1, 1), "write_kib_per_launch": awrite / max(a2, 1),
        "hbm_bytes_per_launch": (2.0 * afetch / max(a1, 1) + awrite / max(a2, 1)) * 1024.0,
        "note": "bench.py workload (Matern-5/2, N=16384, d=16): lower-triangle 64x64 tiles written once, X read through L2",
    },
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
