"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof/) into the committed summaries
profiles/rNN_bench.json, rNN_bench_kernel_stats.csv, rNN_bench_under_rocprof.json, rNN_pmc_traffic.json.

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts exactly half of coalesced 16 B/lane
streaming reads (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the factor 2 on reads.
usage: python tools/summarize_profiles.py <round number>
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof')
KERNEL = 'k_morph_reduce<1, false, true, 0>'


def counter(path, name):
    vals, meta = [], {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERNEL in row['Kernel_Name'] and row['Counter_Name'] == name:
                vals.append(float(row['Counter_Value']))
                meta = dict(kernel_name=row['Kernel_Name'], grid=row['Grid_Size'], workgroup=row['Workgroup_Size'],
                            vgpr=row['VGPR_Count'], sgpr=row['SGPR_Count'], lds=row['LDS_Block_Size'])
    return dict(launches=len(vals), mean_KB=sum(vals) / len(vals), min_KB=min(vals), max_KB=max(vals), **meta)


def main(rnd):
    tag = 'r%02d' % rnd
    dst = os.path.join(ROOT, 'profiles')
    shutil.copy(os.path.join(SRC, 'bench.json'), os.path.join(dst, tag + '_bench.json'))
    shutil.copy(os.path.join(SRC, 'bench_under_rocprof.json'), os.path.join(dst, tag + '_bench_under_rocprof.json'))
    shutil.copy(os.path.join(SRC, 'kt', 'kt_kernel_stats.csv'), os.path.join(dst, tag + '_bench_kernel_stats.csv'))
    if os.path.exists(os.path.join(SRC, 'extras', 'extras_kernel_stats.csv')):
        shutil.copy(os.path.join(SRC, 'extras', 'extras_kernel_stats.csv'), os.path.join(dst, tag + '_extras_kernel_stats.csv'))
    bench = json.load(open(os.path.join(SRC, 'bench_under_rocprof.json')))
    algo = bench['roofline']['bytes_per_launch']
    fetch = counter(os.path.join(SRC, 'fetch', 'fetch_counter_collection.csv'), 'FETCH_SIZE')
    write = counter(os.path.join(SRC, 'write', 'write_counter_collection.csv'), 'WRITE_SIZE')
    rd = fetch['mean_KB'] * 1024 * 2
    wr = write['mean_KB'] * 1024
    out = {
        'round': rnd,
        'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 40 --warmup 4 '
                   '--no-cpu-baseline --no-extras   (tools/profile_round.sh)',
        'kernel': KERNEL.replace(', ', ','), 'workload': bench['config']['workload'],
        'counters': {'FETCH_SIZE': fetch, 'WRITE_SIZE': write},
        'correction': 'FETCH_SIZE x 1024 B x 2 (gfx950 reports exactly half of 16 B/lane coalesced streaming reads, '
                      'MI355X_MICROARCH.md section HBM); WRITE_SIZE x 1024 B exact',
        'hbm_read_bytes_per_launch': rd, 'hbm_write_bytes_per_launch': wr, 'traffic_bytes_per_launch': rd + wr,
        'algorithmic_bytes_per_launch': algo, 'traffic_over_algorithmic': (rd + wr) / algo,
    }
    with open(os.path.join(dst, tag + '_pmc_traffic.json'), 'w') as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(SRC, 'kt', 'kt_kernel_stats.csv')) as f:
        for row in csv.DictReader(f):
            if KERNEL in row['Name']:
                us = float(row['AverageNs']) / 1e3
                print('%s: %s launches, average %.2f us -> %.3f TB/s (%.1f %% of 8 TB/s); bench.py events %.2f us' % (
                    KERNEL, row['Calls'], us, algo / us / 1e6, algo / us / 1e6 / 8 * 100, bench['roofline']['avg_launch_us']))
    print('traffic / algorithmic = %.4f (read %.4f GB, write %.1f KB per launch)' % ((rd + wr) / algo, rd / 1e9, wr / 1e3))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
