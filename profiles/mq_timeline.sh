R=$PWD; O=gpurun_out/mqk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $R/$O/p; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p -o t -- python3 $R/profiles/mq_probe.py > $R/$O/p.txt 2>&1
python3 - $(find $R/$O/p -name t_kernel_trace.csv | head -1) <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last burst: find last k_scan_multi_wg and print kernels from the preceding k_stage/first kernel within 5 ms before it to 3 ms after
idx=[i for i,r in enumerate(rows) if 'k_scan_multi_wg' in r['Kernel_Name']][-1]
t_wg=int(rows[idx]['Start_Timestamp'])
sel=[r for r in rows if t_wg-3_000_000 <= int(r['Start_Timestamp']) <= t_wg+4_000_000]
t0=int(sel[0]['Start_Timestamp'])
for r in sel:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  {r['Kernel_Name'][:90]}")
PY
grep burst $R/$O/p.txt | tail -2
rm -rf $R/$O/p
