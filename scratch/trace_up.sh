cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python3 bench.py --no-secondary --steps 2 --warmup 1 --no-cpu --no-profile > gpurun_out/tr.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/tr/tr_kernel_trace.csv')))
sel=[r for r in rows if 'k_hess_up_pad' in r['Kernel_Name'] or 'k_gram_partial' in r['Kernel_Name'] or 'k_lf_assemble' in r['Kernel_Name']]
for r in sel[-24:]:
    print(r['Kernel_Name'][:30], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us grid', r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'],'wg',r['Workgroup_Size_X'],'lds',r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), 'sgpr', r.get('SGPR_Count'))
PY
