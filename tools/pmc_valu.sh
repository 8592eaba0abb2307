cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc_list.txt 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/gap_trace.py run ${DT:-f16} > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv"%tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in agg:
    if "vn_flood" in k or "cn_flood" in k:
        print(k, {c:(round(v/cnt[(k,c)])) for c,v in agg[k].items()}, "launches", max(cnt[(k,c)] for c in agg[k]))
PY
done
