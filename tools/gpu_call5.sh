# Call 5: XCD-aware attention (query block, key split) mapping: parity + probe per split factor, encode probe, bench
TAG=${1:-r1i}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
(timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_seqpar.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or attn or decode or band" 2>&1 | tail -15) > $O/pytest_attn_$TAG.log 2>&1
for V in MDTILE_ATTN_SPLIT=0 MDTILE_ATTN_SPLIT=2 MDTILE_ATTN_SPLIT=1 MDTILE_ATTN_SPLIT=3; do
  (env $V timeout 200 python probes/attn_probe.py 30000 71168 77284 2>&1 | grep -v amdgpu.ids | sed "s/^/$V /") >> $O/attn_probe_$TAG.log 2>&1
done
(timeout 300 python probes/encode_probe.py 2>&1 | grep -v amdgpu.ids) > $O/encode_probe_$TAG.log 2>&1
(timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1) > $O/bench_$TAG.log 2>&1
cd /tmp
(timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_attn_fetch_$TAG -o p -- python $R/probes/attn_probe.py --quick 77284 2>&1 | tail -2) > $O/pmc_attn_fetch_$TAG.log 2>&1
cd $R
python - <<'PY' > $O/pmc_attn_fetch_summary_$TAG.txt 2>&1
import csv, glob, collections, os
agg = collections.defaultdict(lambda: [0.0, set()])
for fn in glob.glob(os.path.join("gpurun_out", "pmc_attn_fetch_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "FETCH_SIZE":
            a = agg[r["Kernel_Name"][:70]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
for k, (v, d) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:5]:
    print(f"{k:70s} n={len(d)} fetch(x2) per launch = {2 * v / len(d) * 1024 / 1e9:.2f} GB")
PY
rm -rf $O/pmc_attn_fetch_$TAG
tail -4 $O/pytest_attn_$TAG.log; cat $O/attn_probe_$TAG.log; cat $O/encode_probe_$TAG.log; cat $O/pmc_attn_fetch_summary_$TAG.txt; cut -c1-1700 $O/bench_$TAG.log
