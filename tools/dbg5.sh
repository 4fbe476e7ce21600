#!/bin/bash
O=$PWD/gpurun_out/dbg5; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/prof_cfg5.py > $O/prof.json 2> $O/prof.err ); echo "prof rc=$?"
cut -c1-250 $O/prof.json
python3 - <<'PY'
import csv,glob
kt=glob.glob('/root/repo/gpurun_out/dbg5/prof/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(kt)) if 'kf_lean' in r['Kernel_Name'] or 'k5_hist' in r['Kernel_Name']]
for r in rows: print(r['Kernel_Name'][:60], 'grid', r.get('Grid_Size'), r.get('Grid_Size_X'), 'wg', r.get('Workgroup_Size'), 'lds', r.get('LDS_Block_Size'), 'dur_us %.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
