cd "${GRAFT_REPO_ROOT:-/root/repo}"; export FFT_TUNE_MERKLE=0 TMPDIR=/tmp; R=$PWD
timeout 200 python tools/fft_tune.py 22 128 4 fft.tile=0 fft.tile=1 fft.pipe=1
FFT_TUNE_ZEROS=1 timeout 200 python tools/fft_tune.py 22 128 4 fft.tile=0 fft.tile=1 fft.pipe=1
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
rocm-smi --showpower --showmaxpower 2>/dev/null | grep -i "power" | head -4
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/grbm -o g -- python $R/tools/fft_tune.py 22 64 2 fft.tile=0 > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/grbm/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("/tmp/grbm/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = {}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    name, d = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
    k = name.split("(")[0][:48]
    a = acc.setdefault(k, [0.0, 0])
    a[0] += float(r["Counter_Value"]); a[1] += d
for k, (cyc, ns) in acc.items():
    if ns and ("fft13" in k or "lde_mid" in k): print(k, "effective clock GHz (GRBM_GUI_ACTIVE / wall):", round(cyc / ns, 3))
PY
