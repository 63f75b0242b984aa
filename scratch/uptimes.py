import csv, sys
rows=list(csv.DictReader(open('/root/repo/gpurun_out/tr/tr_kernel_trace.csv')))
for name in sys.argv[1:] or ['k_hess_up_pad']:
    sel=[r for r in rows if name in r['Kernel_Name']]
    n=len(sel)//3 if len(sel)>=3 else len(sel)
    for r in sel[-n:]:
        print(name, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'])
