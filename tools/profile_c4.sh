#!/bin/bash
# rocprofv3 evidence for the 1M-node kernels (run on the GPU box from the repo root through gpurun): kernel trace of
# tools/c4_kernels.py, then FETCH_SIZE / WRITE_SIZE passes over the scatter-add kernels alone (counters only, separate runs).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out/prof_c4_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 "$R/tools/c4_kernels.py" all > "$OUT/kt.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- python3 "$R/tools/c4_kernels.py" seg > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- python3 "$R/tools/c4_kernels.py" seg > "$OUT/write.log" 2>&1
DB=$(find "$OUT/kt" -name "*.db" | head -1)
[ -n "$DB" ] && python3 "$R/profiles/summarize_rocpd.py" "$DB" 40 by-grid > "$OUT/${TAG}_c4_kernel_stats.csv"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
out, tag = sys.argv[1:3]
def collect(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith("k_segsum") or (name.startswith("__amd_rocclr_copyBuffer") and float(r["Counter_Value"]) > 400000):
                acc.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
F, W = collect(os.path.join(out, "fetch"), "FETCH_SIZE"), collect(os.path.join(out, "write"), "WRITE_SIZE")
cal = [v for k, v in F.items() if k.startswith("__amd_rocclr_copyBuffer")]
fcorr = (1 << 20) / cal[0][0] if cal else 2.0
with open(os.path.join(out, f"{tag}_c4_pmc_hbm_traffic.csv"), "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) of tools/c4_kernels.py seg: N = 1 000 000, E = 5 999 924, 512-byte rows; counters in KB.\n")
    fh.write(f"# FETCH_SIZE correction x{fcorr:.3f} from the 1 GiB calibration copy (gfx950 reports half of a wide streaming read, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact.\n")
    fh.write("kernel,fetch_launches,fetch_avg_KB,write_launches,write_avg_KB,read_bytes_corrected,write_bytes,hbm_bytes_per_launch\n")
    for k in sorted(set(F) | set(W)):
        f, nf = F.get(k, (0.0, 0)); w, nw = W.get(k, (0.0, 0))
        fh.write(f"{k},{nf},{f:.0f},{nw},{w:.0f},{int(f * 1024 * fcorr)},{int(w * 1024)},{int(f * 1024 * fcorr + w * 1024)}\n")
print(open(os.path.join(out, f"{tag}_c4_pmc_hbm_traffic.csv")).read())
PY
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write"
du -sh "$OUT"; ls "$OUT"; tail -3 "$OUT/kt.log"
