"""Turn rocprofv3 output under gpurun_out/ into the small summaries committed under profiles/.

    python tools/summarize_profile.py stats   gpurun_out/r1b_stats/s_kernel_stats.csv  profiles/r01b_kernel_stats.csv
    python tools/summarize_profile.py traffic gpurun_out/r1b_fetch/f_counter_collection.csv \
           gpurun_out/r1b_write/w_counter_collection.csv profiles/r01b_hbm_traffic.json

`traffic` follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from two
separate --pmc passes, both are in KiB, and on gfx950 FETCH_SIZE tallies the 128-B requests of
16-B-per-lane streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact.  Both count the
L2's fabric-side requests, so Infinity-Cache hits are included: the figure is an upper bound on
true HBM bytes.
"""
import collections
import csv
import json
import re
import sys


def short_name(name):
    m = re.search(r'([A-Za-z_0-9]+)(<[^(]*>)?\(', name)
    if not m:
        return name[:60]
    return (m.group(1) + (m.group(2) or ''))[:80]


def stats(src, dst, top=60):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -int(r['TotalDurationNs']))
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'percent', 'min_us', 'max_us'])
        for r in rows[:top]:
            w.writerow([short_name(r['Name']), r['Calls'], '%.3f' % (int(r['TotalDurationNs']) / 1e6),
                        '%.2f' % (float(r['AverageNs']) / 1e3), r['Percentage'],
                        '%.2f' % (int(r['MinNs']) / 1e3), '%.2f' % (int(r['MaxNs']) / 1e3)])
    return rows


def per_kernel(path, counter):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = short_name(r['Kernel_Name'])
        e = out.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += float(r['Counter_Value'])
    return out


def traffic(fetch_csv, write_csv, dst, family=('gemm_kernel', 'gemm_wide_h_kernel', 'gemm_kmajor_wide_h_kernel')):
    f = per_kernel(fetch_csv, 'FETCH_SIZE')
    w = per_kernel(write_csv, 'WRITE_SIZE')
    kernels = {}
    fam = [0, 0.0, 0.0]
    for k, (n, kib) in f.items():
        wn, wkib = w.get(k, (0, 0.0))
        rd = 2.0 * kib * 1024.0 / n
        wr = wkib * 1024.0 / wn if wn else 0.0
        kernels[k] = {'launches': n, 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                      'hbm_bytes_per_launch': rd + wr}
        if k.startswith(family):
            fam[0] += n
            fam[1] += 2.0 * kib * 1024.0
            fam[2] += wkib * 1024.0
    doc = {'note': 'FETCH_SIZE (KiB, x2 gfx950 correction) + WRITE_SIZE (KiB) from separate --pmc '
                   'passes; L2 fabric-side requests, Infinity-Cache hits included',
           'family': ' / '.join(family), 'family_launches': fam[0],
           'family_bytes_per_launch': (fam[1] + fam[2]) / max(fam[0], 1),
           'family_read_bytes_per_launch': fam[1] / max(fam[0], 1),
           'family_write_bytes_per_launch': fam[2] / max(fam[0], 1),
           'kernels': dict(sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:40])}
    with open(dst, 'w') as fh:
        json.dump(doc, fh, indent=1)
    return doc


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'traffic':
        d = traffic(sys.argv[2], sys.argv[3], sys.argv[4])
        print(d['family'], d['family_launches'], '%.1f MB/launch' % (d['family_bytes_per_launch'] / 1e6))
    else:
        raise SystemExit(__doc__)
