"""One captured goku_step as the device ran it: kernel start offsets, durations and gaps from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --workload goku_step [--dtype mixed] --steps 30 --warmup 5
    python abl/step_timeline.py gpurun_out/tl"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_adamw_flux' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['End_Timestamp'])
prev_end = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('lde::', '')[:56]
    print('%8.1f %7.1f gap %6.1f  q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'), name))
    prev_end = max(prev_end, e)
print('kernels per step:', b - a, ' step (us):', (int(rows[b]['End_Timestamp']) - t0) / 1e3)
