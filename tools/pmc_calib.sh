#!/bin/bash
# FETCH_SIZE / WRITE_SIZE and the raw TCC request counters of tools/ubench_traffic.hip's kernels, whose memory-side bytes are known by
# construction: the correction a kernel's counter traffic needs depends on its access pattern (MI355X_MICROARCH.md calibrates the wide
# streaming read only).  One counter group per rocprofv3 pass.   usage (GPU box): tools/pmc_calib.sh [round]   -> gpurun_out/<round>/traffic_calibration.json
round=${1:-r05}
bin=$GRAFT_REPO_ROOT/tools/bin/ubench_traffic
mkdir -p $(dirname $bin)
[ -x $bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $bin $GRAFT_REPO_ROOT/tools/ubench_traffic.hip || exit 1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$round
mkdir -p $out
$bin > $out/ubench_traffic_plain.txt 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rm -rf /tmp/calib_$i
  timeout -s KILL 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/calib_$i -o p -- $bin > /tmp/calib_$i.log 2>&1
  echo "group $i ($grp) rc=$?"
  cp /tmp/calib_$i/p_counter_collection.csv $out/calib_counters_$i.csv 2>/dev/null
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
known = {}
for line in open(out + "/ubench_traffic_plain.txt"):
    if line.startswith("{"):
        d = json.loads(line)
        known[d["kernel"]] = d
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(out + "/calib_counters_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
rows = []
for k, cs in sorted(acc.items()):
    c = {n: v[0] / v[1] for n, v in cs.items()}
    kb = known.get(k, {}).get("known_bytes")
    row = {"kernel": k, "known_bytes": kb, "known_gbs_plain_run": known.get(k, {}).get("known_gbs"), "counters_per_launch": c}
    if kb:
        if k in ("k_stream_read16", "k_gather12") and c.get("FETCH_SIZE"):
            row["FETCH_SIZE_bytes"] = c["FETCH_SIZE"] * 1024
            row["known_over_FETCH_SIZE"] = kb / (c["FETCH_SIZE"] * 1024)
        if k in ("k_stream_write16", "k_scatter4", "k_scatter16") and c.get("WRITE_SIZE"):
            row["WRITE_SIZE_bytes"] = c["WRITE_SIZE"] * 1024
            row["WRITE_SIZE_over_payload"] = c["WRITE_SIZE"] * 1024 / kb
    rows.append(row)
json.dump({"what": "rocprofv3 counters of tools/ubench_traffic.hip against byte counts known by construction (4 GB buffer, 32 Mi random requests)",
           "rows": rows}, open(out + "/traffic_calibration.json", "w"), indent=1)
for r in rows: print(json.dumps(r))
PY
rm -f $out/calib_counters_*.csv
