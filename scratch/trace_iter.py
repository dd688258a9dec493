"""Per-kernel breakdown of one steady-state graph-replay iteration from a rocprofv3 kernel trace csv:
python scratch/trace_iter.py <ks_kernel_trace.csv> [list]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
n_it = 5
a, b = idx[-(3 * n_it + 10)], idx[-10]
seg = rows[a + 1:b + 1]
span = int(seg[-1]['End_Timestamp']) - int(rows[a]['End_Timestamp'])
print('launches/iter', len(seg) / n_it, 'span/iter ms', span / n_it / 1e6)
def short(n):
    n = n.replace('void ', '').replace('ali::', '')
    m = re.match(r'([\w:]+)(<[^>]*>)?', n)
    s = m.group(0) if m else n[:40]
    if 'at::native' in n:
        k = re.search(r'(\w+Functor|CatArray\w*|\w+_kernel_cuda)', n)
        s = 'AT:' + (k.group(1) if k else n[20:80])
    return s[:50]
if len(sys.argv) > 2:
    one = rows[idx[-13] + 1: idx[-10] + 1]
    for i, r in enumerate(one):
        t = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{i:3d} {short(r['Kernel_Name']):52s}{t:7.1f} g={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
else:
    d = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        n = short(r['Kernel_Name']); t = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        d[n][0] += t / n_it / 1e3; d[n][1] += 1 / n_it
    for k, v in sorted(d.items(), key=lambda x: -x[1][0]):
        print(f'{k:52s}{v[0]:8.1f} us {v[1]:6.1f} {v[0] / v[1]:7.1f}')
