# developer tool: memory-side counters of the VN / CN kernels of one fixed-50 step (DT=f16|f32|i8)
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d /tmp/pm_$tag -- python3 $GRAFT_REPO_ROOT/tools/gap_trace.py run ${DT:-f16} fixed > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("/tmp/pm_%s/**/*counter_collection.csv"%tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in agg:
    if "vn_flood" in k or "cn_flood" in k:
        print(k, {c:(round(v/cnt[(k,c)])) for c,v in agg[k].items()}, "launches", max(cnt[(k,c)] for c in agg[k]))
PY
done
