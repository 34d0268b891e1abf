"""Summarises rocprofv3 --pmc passes into per-launch figures per kernel:
  * HBM bytes: FETCH_SIZE / WRITE_SIZE (separate passes).  MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
    FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (x2 is an upper bound for mixed patterns -> raw and
    corrected are both kept);
  * matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over the SIMDs: 64 per
    v_mfma_f64_16x16x4) against the kernel's duration x 1024 SIMDs (duration from the kernel trace of the same pass).
usage: python scripts/pmc_summary.py <workload> <fetch_dir> <write_dir> [<mfma_dir>] > profiles/r02_pmc_<workload>.json"""
import csv, glob, json, sys, collections
wl, fdir, wdir = sys.argv[1:4]
mdir = sys.argv[4] if len(sys.argv) > 4 else None
def short(k): return k.split("(")[0].replace("void ", "")
def load(d, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = short(r["Kernel_Name"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return acc
def durations(d):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][0] += 1; acc[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc
F, W = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
out = {}
for k in sorted(set(F) | set(W)):
    nf, vf = F.get(k, [0, 0.0]); nw, vw = W.get(k, [0, 0.0])
    out[k] = {"launches": max(nf, nw), "fetch_KiB_per_launch_raw": vf / max(nf, 1), "write_KiB_per_launch": vw / max(nw, 1),
              "hbm_bytes_per_launch_raw": 1024 * (vf / max(nf, 1) + vw / max(nw, 1)),
              "hbm_bytes_per_launch_fetch_x2": 1024 * (2 * vf / max(nf, 1) + vw / max(nw, 1))}
if mdir:
    B, M, T = load(mdir, "SQ_VALU_MFMA_BUSY_CYCLES"), load(mdir, "SQ_INSTS_VALU_MFMA_MOPS_F64"), durations(mdir)
    for k in B:
        n, v = B[k]
        if v <= 0: continue
        dur_ns = T[k][1] / max(T[k][0], 1)
        e = out.setdefault(k, {})
        e["mfma_busy_cycles_per_launch"] = v / n
        e["mfma_mops_f64_per_launch"] = M.get(k, [1, 0.0])[1] / max(M.get(k, [1, 0.0])[0], 1)
        e["avg_duration_us_in_pmc_pass"] = dur_ns / 1e3
        # 1024 SIMDs; the clock under this load is ~2.35 GHz (in-kernel s_memtime)
        e["mfma_utilisation"] = (v / n) / (1024 * dur_ns * 2.35) if dur_ns > 0 else None
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
# (the sources these kernels were built from: bench.py quotes roofline.traffic only from a pass of the SAME sources)
print(json.dumps({wl: out, "source_sha16": bench.kernel_source_sha16()}, indent=1))
