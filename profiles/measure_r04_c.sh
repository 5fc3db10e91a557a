#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $O
timeout 600 python profiles/raycast_split_probe.py depth > $O/split_depth.jsonl 2> $O/split_depth.err
timeout 600 python profiles/raycast_split_probe.py lidar > $O/split_lidar.jsonl 2> $O/split_lidar.err
cat $O/split_depth.jsonl $O/split_lidar.jsonl
cd /tmp && export TMPDIR=/tmp
for sp in 1 4 12; do
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $set | cut -d' ' -f1)
    AGX_PROBE_RAY_SPLIT=$sp timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_s${sp}_$tag -o p -- python $GRAFT_REPO_ROOT/profiles/pmc_raycast.py depth > $O/pmc_s${sp}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04c"
for d in sorted(glob.glob(O+"/pmc_s*_*/")):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_raycast" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d.rstrip("/")), {k: (sum(v)/len(v), len(v)) for k,v in acc.items()})
PY
find $O -name "*.csv" -size +2M -delete
