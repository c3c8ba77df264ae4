"""gpurun_out/prof5/ (tools/profile_round5.sh) -> profiles/rNN_toy_points.json, rNN_bb_scan_pmc.json, rNN_grad_batch.json (+ the
kernel-stats CSVs of the three traces).  FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts exactly half of
coalesced 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the factor 2 on reads.
usage: python tools/summarize_round5.py <round>"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof5')


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def trace(sub, pick):
    out = {}
    with open(os.path.join(SRC, sub, 't_kernel_stats.csv')) as f:
        for row in csv.DictReader(f):
            if any(p in row['Name'] for p in pick):
                out[short(row['Name'])] = dict(calls=int(row['Calls']), average_us=float(row['AverageNs']) / 1e3,
                                               min_us=float(row['MinNs']) / 1e3, max_us=float(row['MaxNs']) / 1e3)
    return out


def counters(sub, pick):
    """per kernel: mean of every counter over its launches"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(SRC, sub, 't_counter_collection.csv')
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if any(p in row['Kernel_Name'] for p in pick):
                agg[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def sq_derived(per):
    mfma = per.get('SQ_INSTS_MFMA', 0.0)
    d = dict(lds_bank_conflict_share_of_wave_cycles=per['SQ_LDS_BANK_CONFLICT'] / per['SQ_WAVE_CYCLES'],
             lds_bank_conflict_cycles_per_lds_instruction=per['SQ_LDS_BANK_CONFLICT'] / max(per['SQ_INSTS_LDS'], 1.0),
             waves_waiting_share_of_wave_cycles=per['SQ_WAIT_INST_ANY'] / per['SQ_WAVE_CYCLES'])
    if mfma > 0:
        d.update(other_vector_instructions_per_mfma=(per['SQ_INSTS_VALU'] - mfma) / mfma,
                 lds_instructions_per_mfma=per['SQ_INSTS_LDS'] / mfma,
                 mfma_busy_fraction_of_SIMD_cycles=per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['GRBM_GUI_ACTIVE'] * 128.0))
    return d


def main(rnd):
    tag = 'r%02d' % rnd
    dst = os.path.join(ROOT, 'profiles')
    # ---- the toy-MC call over several hypotheses
    pick = ['_multi']
    toy = dict(round=rnd, workload='bi_eval_datasets_points: 10^4 device-drawn toys of C2 (configs[2]) x 4 hypotheses per call',
               command='tools/profile_round5.sh: rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (own runs) '
                       '-- python3 tools/profile/toy_points_trace.py N same|random',
               correction='FETCH_SIZE x 1024 B x 2 (gfx950 reports exactly half of 16 B/lane coalesced streaming reads); WRITE_SIZE x 1024 B',
               call_timings=json.load(open(os.path.join(SRC, 'toy_points.json')))['rows'])
    B, NS, T = 10 ** 6, 32, 10 ** 4
    entries = None
    for which in ('same', 'random'):
        kt = trace('toy_kt_' + which, pick)
        fetch, write = counters('toy_fetch_' + which, pick), counters('toy_write_' + which, pick)
        per = {}
        for k in kt:
            rd = fetch.get(k, {}).get('FETCH_SIZE', 0.0) * 1024 * 2
            wr = write.get(k, {}).get('WRITE_SIZE', 0.0) * 1024
            per[k] = dict(kt[k], hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr,
                          traffic_GB_per_s=(rd + wr) / (kt[k]['average_us'] * 1e-6) / 1e9)
            if 'logmu' in k:
                algo = 8.0 * NS * B * (1 if which == 'same' else 4) + 8.0 * 4 * B
                per[k].update(algorithmic_bytes_per_launch=algo, achieved_GB_per_s=algo / (kt[k]['average_us'] * 1e-6) / 1e9,
                              traffic_over_algorithmic=(rd + wr) / algo)
        toy['four_hypotheses_in_%s' % ('one_cell' if which == 'same' else 'random_cells')] = per
    sq = counters('toy_sq', pick)
    sq2 = counters('toy_sq2', pick)
    toy['counters_one_cell'] = {k: dict(counters_per_launch=dict(v, **sq2.get(k, {})), derived=sq_derived(v)) for k, v in sq.items()}
    with open(os.path.join(dst, tag + '_toy_points.json'), 'w') as f:
        json.dump(toy, f, indent=1)
    shutil.copy(os.path.join(SRC, 'toy_kt_same', 't_kernel_stats.csv'), os.path.join(dst, tag + '_toy_points_kernel_stats.csv'))
    # ---- Beeston-Barlow scan on the matrix cores
    kt = trace('bb_kt', ['k_scan_bb'])
    sq = counters('bb_sq', ['k_scan_bb'])
    fetch = counters('bb_fetch', ['k_scan_bb'])
    bb = dict(round=rnd, workload='256 Beeston-Barlow scan points in one grid cell of configs[4] (6 sources, 2^4 anchors, 50^4 bins): one launch',
              command='tools/profile_round5.sh: rocprofv3 --kernel-trace --stats / --pmc SQ_* / --pmc FETCH_SIZE (own runs) -- python3 '
                      'tools/profile/bb_scan_only.py', kernel_trace=kt,
              plain_runs=[ln.strip() for ln in open(os.path.join(SRC, 'bb_plain.txt')) if ln.startswith('Beeston')])
    for k, v in sq.items():
        bb['counters'] = dict(kernel=k, counters_per_launch=v, derived=sq_derived(v))
    for k, v in fetch.items():
        rd = v['FETCH_SIZE'] * 1024 * 2
        bb['hbm_read_bytes_per_launch'] = rd
        bb['rows_once_bytes'] = 8.0 * (16 * 6 + 16 + 1) * 50 ** 4
        bb['reads_over_rows_once'] = rd / bb['rows_once_bytes']
    if kt:
        k0 = list(kt.values())[0]
        bb['evaluations_per_s_by_kernel_trace'] = 256 / (k0['average_us'] * 1e-6)
        bb['three_products_TFLOPs'] = 2.0 * 112 * 50 ** 4 * 256 / (k0['average_us'] * 1e-6) / 1e12
    with open(os.path.join(dst, tag + '_bb_scan_pmc.json'), 'w') as f:
        json.dump(bb, f, indent=1)
    shutil.copy(os.path.join(SRC, 'bb_kt', 't_kernel_stats.csv'), os.path.join(dst, tag + '_bb_scan_kernel_stats.csv'))
    # ---- gradient batches
    kt = trace('grad_kt', ['k_grad_mfma'])
    sq = counters('grad_sq', ['k_grad_mfma<'])
    grad = json.load(open(os.path.join(SRC, 'grad_throughput.json')))
    grad['round'] = rnd
    grad['kernel_trace_131072_points'] = kt
    grad['counters_131072_points'] = {k: dict(counters_per_launch=v, derived=sq_derived(v)) for k, v in sq.items()}
    grad['counter_command'] = 'tools/profile_round5.sh: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ... -- python3 tools/profile/grad_only.py 2'
    with open(os.path.join(dst, tag + '_grad_batch.json'), 'w') as f:
        json.dump(grad, f, indent=1)
    shutil.copy(os.path.join(SRC, 'grad_kt', 't_kernel_stats.csv'), os.path.join(dst, tag + '_grad_batch_kernel_stats.csv'))
    print(json.dumps({k: v for k, v in bb.items() if k in ('counters', 'evaluations_per_s_by_kernel_trace', 'reads_over_rows_once')}, indent=1)[:1500])
    print(json.dumps(grad['counters_131072_points'], indent=1)[:1200])


if __name__ == '__main__':
    main(int(sys.argv[1]))
