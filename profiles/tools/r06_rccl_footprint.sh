O=gpurun_out/r06_rccl; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/profiles/tools/rccl_kernel_footprint.py > $GRAFT_REPO_ROOT/$O/out.txt 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); head -1 $f; grep -i "nccl\|rccl" $f | head -8 | cut -c1-400
python3 - <<PY
import csv,sys
rows=list(csv.DictReader(open("$f")))
seen={}
for r in rows:
    k=r["Kernel_Name"][:70]
    seen.setdefault(k,(r.get("VGPR_Count"),r.get("Accum_VGPR_Count"),r.get("SGPR_Count"),r.get("LDS_Block_Size"),r.get("Scratch_Size"),r.get("Workgroup_Size"),r.get("Grid_Size"), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in seen.items(): print(k, "vgpr/agpr/sgpr/lds/scratch/wg/grid/ns", v)
PY
