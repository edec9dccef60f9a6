"""Rebuild profiles/pmc_traffic.json from rocprofv3 --pmc passes (tools/pmc_run.sh) of the CURRENT kernel sources.
usage: python tools/update_pmc_traffic.py <pdtv-dir> <others-dir> <profile-name>
  <pdtv-dir>   gpurun_out/pmc_<tag> holding pdtv0_g0 (FETCH_SIZE) / pdtv0_g1 (WRITE_SIZE) of a 9-iteration prox
               (first + middle + last launch), and the same for pdtv0h (binary16 duals)
  <others-dir> the same for roftv, bpq, fpq (the fused pair as the FISTA loop runs it: residual epilogue of the forward
               projector, gradient-step epilogue of the back projector, residual handed over quad-interleaved)
FETCH_SIZE is doubled (MI355X_MICROARCH.md: 128-B requests are counted as 64 B); values are KiB."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pd_dir, ot_dir, profile = sys.argv[1:4]


def vals(d, what, g, key):
    f = glob.glob(os.path.join(d, f"{what}_g{g}", "*counter_collection.csv"))[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]


def traffic(fetch_kib, write_kib):
    return (2.0 * fetch_kib + write_kib) * 1024.0


out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only), per launch, tools/pmc_run.sh + "
               "tools/update_pmc_traffic.py; FETCH_SIZE doubled per MI355X_MICROARCH.md (128-B requests counted as 64 B); "
               "an entry is only reported by bench.py while the kernel sources hash to sources_sha16"}
f, w = vals(pd_dir, "pdtv0", 0, "xk_kernel"), vals(pd_dir, "pdtv0", 1, "xk_kernel")
assert len(f) == 3 and len(w) == 3, (f, w)
tr = [traffic(a, b) for a, b in zip(f, w)]
out["pdtv"] = {
    "workload": "1024^3 f32 duals, 3 iterations/launch (pd_zmarch_xk K=3, 8 rows, LDS lag); a 30-iteration prox is 10 launches: "
                "the first reads no duals, the last stores none",
    "fetch_kib_raw": [f[0], f[1], f[2]], "write_kib": [w[0], w[1], w[2]],
    "traffic_bytes": (tr[0] + tr[2] + 8 * tr[1]) / 10.0,
    "traffic_bytes_first_middle_last": tr,
    "sources_sha16": bench.source_hash("pdtv"), "profile": profile,
    "note": "traffic_bytes = (first + last + 8 x middle) / 10, the per-launch mean of the bench's 30-iteration prox"}
fh, wh = vals(pd_dir, "pdtv0h", 0, "xk_kernel"), vals(pd_dir, "pdtv0h", 1, "xk_kernel")
assert len(fh) == 3 and len(wh) == 3, (fh, wh)
trh = [traffic(a, b) for a, b in zip(fh, wh)]
out["pdtv_half"] = {
    "workload": "1024^3 binary16 duals, 3 iterations/launch (pd_zmarch_xk K=3, 8 rows, FMA-corrected exact roundings); "
                "first / middle / last launch of a prox",
    "fetch_kib_raw": [fh[0], fh[1], fh[2]], "write_kib": [wh[0], wh[1], wh[2]],
    "traffic_bytes": (trh[0] + trh[2] + 8 * trh[1]) / 10.0,
    "traffic_bytes_first_middle_last": trh,
    "sources_sha16": bench.source_hash("pdtv"), "profile": profile}
for name, what, keys, wl in (
        ("roftv", "roftv", ["rof_"], "1024^3, 1 iteration/launch (rof_zmarch, shipped build: FMA-corrected reference roundings); later launches"),
        ("bp", "bpq", ["bp_brick"], "1024^3, 75 angles (one subset), bp_brick_kernel with the FISTA gradient-step epilogue, "
                                    "quad-interleaved residual (what bench.py times)"),
        ("fp", "fpq", ["fp_tiled", "transpose"], "1024^3, 75 angles (one subset): 2 launches of fp_tiled_kernel<...,1024> with the "
                                                 "residual epilogue (quad-interleaved output) + the in-plane transpose")):
    fs = ws = 0.0
    for k in keys:
        fv, wv = vals(ot_dir, what, 0, k), vals(ot_dir, what, 1, k)
        if name == "fp":      # one call = two stepping-axis launches + one transpose; the probe makes two calls
            fs += sum(fv) / 2.0; ws += sum(wv) / 2.0
        else:                 # steady-state launch: the last one
            fs += fv[-1]; ws += wv[-1]
    out[name] = {"workload": wl, "fetch_kib_raw": fs, "write_kib": ws, "traffic_bytes": traffic(fs, ws),
                 "sources_sha16": bench.source_hash("pdtv" if name == "pdtv_half" else name), "profile": profile}
json.dump(out, open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, "%.2f GB" % (v["traffic_bytes"] / 1e9), v["sources_sha16"])
