#!/usr/bin/env python
"""Per-launch HBM traffic of the cfl kernels from two rocprofv3 --pmc passes.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at
64 B, so it is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Launches of the
warm-up are included (same kernels, same shapes); the result is bytes per launch.
"""
import csv, glob, json, os, sys
from collections import defaultdict

KEYS = {'cfl_proj_kernel': 'proj', 'cfl_proj_bx3_kernel': 'proj', 'cfl_proj_stream_kernel': 'proj', 'cfl_proj_x3_kernel': 'proj', 'cfl_proj_x3_keep_kernel': 'proj', 'cfl_wplanes_kernel': 'wplanes', 'cfl_mid_row_kernel': 'mid', 'cfl_mid_kernel': 'mid',
        'cfl_grad_kernel': 'grad', 'cfl_grad_x3_kernel': 'grad', 'cfl_grad_x3_longrange_kernel': 'grad', 'cfl_grad_x3_half_kernel': 'grad', 'cfl_grad_x3_half_w8_kernel': 'grad', 'cfl_grad_x3_half_split_kernel': 'grad', 'cfl_finalize_kernel': 'finalize', 'cfl_adam_kernel': 'adam', 'cfl_adam_planes_kernel': 'adam'}


def collect(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            for pat, k in KEYS.items():
                if pat in r['Kernel_Name']:
                    tot[k] += float(r['Counter_Value'])
                    cnt[k] += 1
                    break
    return {k: tot[k] / cnt[k] for k in tot}, dict(cnt)


def main():
    fetch, nf = collect(sys.argv[1], 'FETCH_SIZE')
    write, nw = collect(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in fetch:
        out[k] = int(round((2.0 * fetch[k] + write.get(k, 0.0)) * 1024))
    detail = {k: {'fetch_kib_raw': round(fetch[k], 1), 'write_kib': round(write.get(k, 0.0), 1),
                  'launches': nf[k]} for k in fetch}
    print(json.dumps({'traffic_bytes_per_launch': out, 'detail': detail}, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
