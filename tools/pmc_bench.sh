#!/bin/bash
# On the GPU box: HBM traffic counters of one bench step, separate passes per counter as the microarch guide prescribes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass); per-kernel summaries to gpurun_out/<tag>_pmc_<counter>.txt and the
# fused pass's figures as JSON to gpurun_out/<tag>_pmc_traffic.json (copy to profiles/ to make bench.py report them).
# usage: tools/pmc_bench.sh <tag> [bench args...]
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d="/tmp/pmc_${tag}_${c}"
  rm -rf "$d"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -o "$tag" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 "$@" > /dev/null 2> "$R/gpurun_out/${tag}_pmc_${c}.err"
  f=$(find "$d" -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ] || [ ! -s "$f" ]; then echo "pmc_bench: no counter_collection.csv under $d (see ${tag}_pmc_${c}.err)" >&2; exit 1; fi
  : > "$R/gpurun_out/${tag}_pmc_${c}.txt"
  for k in fused_pass hgemm "cgemm_kernel" hgram jacobi2 lanczos; do
    python3 "$R/tools/pmc_summary.py" "$f" "$k" >> "$R/gpurun_out/${tag}_pmc_${c}.txt"
  done
  cp "$f" "$R/gpurun_out/${tag}_pmc_${c}.csv"
done
python3 - "$R" "$tag" <<'PY'
import csv, hashlib, json, os, subprocess, sys
R, tag = sys.argv[1], sys.argv[2]
def collect(match):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, disp, dur = 0.0, set(), 0.0
        for r in csv.DictReader(open("%s/gpurun_out/%s_pmc_%s.csv" % (R, tag, c))):
            if match in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in disp:
                    disp.add(r["Dispatch_Id"]); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        out[c] = (tot / max(len(disp), 1), len(disp), dur / max(len(disp), 1) / 1e3)
    fetch_kib, n, us = out["FETCH_SIZE"]
    write_kib = out["WRITE_SIZE"][0]
    if not n:
        return None
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md); both counters are in KiB
    return {"FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib, "dispatches": n, "avg_us_under_pmc": us,
            "hbm_read_bytes_per_launch": 2 * 1024 * fetch_kib, "hbm_write_bytes_per_launch": 1024 * write_kib,
            "hbm_bytes_per_launch": 2 * 1024 * fetch_kib + 1024 * write_kib}
# (the GPU box gets a snapshot without .git: the caller passes the commit it was taken from in JSTSP_GIT_SHA)
build = os.environ.get("JSTSP_GIT_SHA", "")
if not build:
    try:
        build = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "snapshot"
    except OSError:
        build = "snapshot"
fused_hash = hashlib.sha256(open("%s/jstsp19_amd/csrc/fused.hip" % R, "rb").read()).hexdigest()[:16]
doc = {"_how": "tools/pmc_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE, then in a separate pass --pmc WRITE_SIZE, of "
               "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path` on MI355X; averages per dispatch of the "
               "named kernel; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
       "build": build, "git_sha": build, "fused_hip_sha256_16": fused_hash}
for key, match in (("fused_pass64", "fused_pass64_kernel"), ("fused_pass", "fused_pass_kernel"), ("hgram3", "hgram3_kernel"),
                   ("hgram", "hgram_kernel"), ("hgemm", "hgemm_kernel"), ("jacobi2", "jacobi2_kernel")):
    v = collect(match)
    if v:
        doc[key] = v
json.dump(doc, open("%s/gpurun_out/%s_pmc_traffic.json" % (R, tag), "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k.startswith("fused_pass")}))
PY
