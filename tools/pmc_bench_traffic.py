#!/usr/bin/env python3
"""Per-launch HBM traffic of every kernel of the bench, from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_bench_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json>

Corrections as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: both counters are reported in KiB; on gfx950 FETCH_SIZE
tallies a 128-byte request of a wide (16 bytes per lane) streaming read at 64 bytes, so it is doubled for the kernels whose
loads are 16 bytes per lane (all the kernels listed in WIDE below); WRITE_SIZE is taken as reported (a 512 MiB device copy
reads back exactly 524 288 KiB, profiles/pmc_r01.md).  Infinity-Cache hits are counted, so this is fabric-side traffic.
"""
import collections, csv, glob, json, re, sys

WIDE = ('conv_bf16x3_kernel', 'convt_fused_bf16x3_kernel', 'wgrad_bf16x3', 'fir44_tile_kernel', 'bias_act', 'pw_', 'plane_dot')


def bench_name(name):
    """rocprof's demangled template name -> the key bench.py / utils.profiling.conv_variant uses."""
    m = re.search(r'(conv_bf16x3_kernel)<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', name)
    if m:
        a = m.groups()
        return '%s<%s,%s,%s,%s>|up%s,down%s,k%s' % a
    m = re.search(r'(convt_fused_bf16x3_kernel)<(\d+), (\d+), (\d+), (\d+), \d+>', name)
    if m:
        return '%s<%s,%s,%s,%s>|up2,down1,k3' % m.groups()
    m = re.search(r'([A-Za-z_0-9]+)(<[^(]*>)?\(', name)
    return (m.group(1) + (m.group(2) or '')).replace(' ', '') if m else name[:80]


def collect(directory, counter):
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> dispatch -> value summed over the XCD instances
    for f in glob.glob(directory + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                per[r['Kernel_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return per


def main():
    fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, {}), write.get(name, {})
        key = bench_name(name)
        wide = any(t in name for t in WIDE)
        e = out.setdefault(key, {'launches': 0, 'fetch_kib': 0.0, 'write_kib': 0.0, 'wide_loads': wide})
        e['launches'] += max(len(f), len(w))
        e['fetch_kib'] += sum(f.values()) * (len(w) / len(f) if f and w and len(f) != len(w) else 1.0)
        e['write_kib'] += sum(w.values())
    for key, e in out.items():
        n = max(e['launches'], 1)
        read_b = e['fetch_kib'] * 1024 * (2 if e['wide_loads'] else 1) / n
        e.update(read_bytes_per_launch=read_b, write_bytes_per_launch=e['write_kib'] * 1024 / n)
        e['traffic_bytes_per_launch'] = e['read_bytes_per_launch'] + e['write_bytes_per_launch']
    json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 16 --warmup 0; FETCH_SIZE doubled for 16-byte-per-lane loads',
               'kernels': out}, open(sys.argv[3], 'w'), indent=1)
    top = sorted(out.items(), key=lambda kv: -kv[1]['traffic_bytes_per_launch'] * kv[1]['launches'])[:12]
    for k, e in top:
        print(f"{e['launches']:6d} launches  {e['traffic_bytes_per_launch'] / 2**20:9.1f} MiB/launch  (read {e['read_bytes_per_launch'] / 2**20:8.1f}, write {e['write_bytes_per_launch'] / 2**20:8.1f})  {k}")


if __name__ == '__main__':
    main()
