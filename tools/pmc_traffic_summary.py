#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_step.py -> profiles/<tag>_pmc_hbm_traffic.csv and
profiles/pmc_traffic.json.  usage: pmc_traffic_summary.py <fetch_dir> <write_dir> <tag>
Units and corrections follow MI355X_MICROARCH.md (HBM section): counters are in KB; on gfx950
FETCH_SIZE under-reports by 2x (checked here against the 1 GiB calibration copy), WRITE_SIZE is exact."""
import csv, glob, json, os, sys
from collections import defaultdict
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fd, wd, tag = sys.argv[1:4]
# (kernel-name prefix, workgroups or None) -> key.  Prefixes stop before the closing '>' (k_segsum2<8, false> since round 4: the old
# "k_segsum2<8>" key matched nothing and the bench line carried traffic 0 for the scatter)
KEYS = {("k_mlp_fwd_x6<6, 4, 0", 512): "edge_fwd", ("k_mlp_bwd_x6<6, false, 0", 512): "edge_bwd", ("k_wgrad_x6", 512): "wgrad",
        ("k_edge_fwd_ppr<true", 256): "edge_fwd", ("k_edge_bwd_ppr", 256): "edge_bwd",   # [r6] the register-resident-weights generation
        ("k_wgrad_pc", 256): "wgrad", ("k_segsum2<8", None): "segsum", ("__amd_rocclr_copyBuffer", None): "calibration_copy"}
def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            for (k, g), key in KEYS.items():
                if name.startswith(k) and (g is None or wgs == g):
                    if key == "calibration_copy" and float(r["Counter_Value"]) < 400000:
                        continue  # small copies are not the calibration
                    acc[key].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
F, W = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
cal = F.get("calibration_copy", (0, 0))[0]
fcorr = (1 << 20) / cal if cal else 2.0   # KB that should have been read / KB reported
rows, out = [], {}
for key in ("calibration_copy", "edge_fwd", "edge_bwd", "wgrad", "segsum"):
    f, nf = F.get(key, (0.0, 0)); w, nw = W.get(key, (0.0, 0))
    rb, wb = f * 1024 * fcorr, w * 1024
    rows.append((key, nf, f, nw, w, int(rb), int(wb), int(rb + wb)))
    if key != "calibration_copy":
        out[key + "_bytes"] = int(rb + wb)
        out[key + "_read_bytes"], out[key + "_write_bytes"] = int(rb), int(wb)
PROF = os.environ.get("PROFILES_DIR") or os.path.join(REPO, "profiles")   # on the GPU box: a directory under gpurun_out/ (only that is merged back)
path = os.path.join(PROF, f"{tag}_pmc_hbm_traffic.csv")
with open(path, "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) of tools/pmc_step.py (real training steps); counters in KB.\n")
    fh.write(f"# FETCH_SIZE correction x{fcorr:.3f} from the 1 GiB calibration copy (gfx950 reports half, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact.\n")
    fh.write("kernel,fetch_launches,fetch_avg_KB,write_launches,write_avg_KB,read_bytes_corrected,write_bytes,hbm_bytes_per_launch\n")
    for r in rows:
        fh.write(",".join(str(x) for x in r) + "\n")
sys.path.insert(0, REPO)
from graph_physics_amd import _capi  # noqa: E402
out["csrc_hash"] = _capi.source_hash()   # bench.py refuses this file for any other build of csrc/
out["source"] = f"profiles/{tag}_pmc_hbm_traffic.csv (rocprofv3 PMC passes over real training steps, FETCH_SIZE x{fcorr:.2f} per the calibration copy)"
out["workload"] = "N=30160, E=180082 (bench default), per launch, averaged over the launches of 3 training steps"
# the 1M-node scatter-add figures (tools/profile_c4.sh -> <tag>_c4_pmc_hbm_traffic.csv) travel in the same file when they exist
c4csv = os.environ.get("C4_TRAFFIC_CSV")
if c4csv and os.path.exists(c4csv):
    for r in csv.DictReader(l for l in open(c4csv) if not l.startswith("#")):
        if r.get("kernel", "").startswith("k_segsum<8"):
            out["c4_segsum_read_bytes"], out["c4_segsum_write_bytes"] = int(float(r["read_bytes_corrected"])), int(float(r["write_bytes"]))
            out["c4_segsum_bytes"] = out["c4_segsum_read_bytes"] + out["c4_segsum_write_bytes"]
json.dump(out, open(os.path.join(PROF, "pmc_traffic.json"), "w"), indent=1)
print(open(path).read())
