"""gpurun_out/prof_scan_dense/ (tools/profile_scan_dense.sh) -> profiles/rNN_scan_dense_data_pmc.json +
rNN_scan_dense_data_kernel_stats.csv: the dense-data scan on count-sorted rows against rows in bin order.
usage: python tools/summarize_scan_dense.py <round>"""
import csv
import collections
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof_scan_dense')
POINTS = 131072


def counters(order):
    agg, launches, name = collections.defaultdict(float), 0, ''
    if not os.path.exists(os.path.join(SRC, 'pmc_%s' % order, 'pmc_counter_collection.csv')):
        return None
    with open(os.path.join(SRC, 'pmc_%s' % order, 'pmc_counter_collection.csv')) as f:
        for row in csv.DictReader(f):
            if 'k_scan_mfma' in row['Kernel_Name'] or 'k_scan_sorted' in row['Kernel_Name']:
                agg[row['Counter_Name']] += float(row['Counter_Value'])
                name = row['Kernel_Name']
                if row['Counter_Name'] == 'SQ_INSTS_MFMA':
                    launches += 1
    per = {k: v / launches for k, v in agg.items()}
    mfma = per['SQ_INSTS_MFMA']
    busy = per['SQ_VALU_MFMA_BUSY_CYCLES'] / (per['GRBM_GUI_ACTIVE'] * 128.0)        # as in r02_scan_dense_data_pmc.json
    return dict(kernel=name.replace('(anonymous namespace)::', ''), launches=launches, counters_per_launch=per, derived=dict(
        other_vector_instructions_per_mfma=(per['SQ_INSTS_VALU'] - mfma) / mfma,
        other_vector_instructions_per_matrix_element=2 * (per['SQ_INSTS_VALU'] - mfma) / mfma,
        lds_instructions_per_mfma=per['SQ_INSTS_LDS'] / mfma,
        lds_bank_conflict_cycles_per_lds_instruction=per['SQ_LDS_BANK_CONFLICT'] / per['SQ_INSTS_LDS'],
        lds_bank_conflict_share_of_wave_cycles=per['SQ_LDS_BANK_CONFLICT'] / per['SQ_WAVE_CYCLES'],
        mfma_busy_fraction_of_SIMD_cycles=busy))


def main(rnd):
    tag = 'r%02d' % rnd
    dst = os.path.join(ROOT, 'profiles')
    shutil.copy(os.path.join(SRC, 'kt', 'kt_kernel_stats.csv'), os.path.join(dst, tag + '_scan_dense_data_kernel_stats.csv'))
    trace = {}
    with open(os.path.join(SRC, 'kt', 'kt_kernel_stats.csv')) as f:
        for row in csv.DictReader(f):
            if 'k_scan_mfma' in row['Name'] or 'k_scan_sorted' in row['Name']:
                trace = dict(name=row['Name'], calls=int(row['Calls']), average_us=float(row['AverageNs']) / 1e3,
                             min_us=float(row['MinNs']) / 1e3, max_us=float(row['MaxNs']) / 1e3)
    out = dict(
        round=rnd,
        command='tools/profile_scan_dense.sh: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS '
                'SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE (one pass per row order) / --kernel-trace --stats '
                '(own run) -- python3 tools/profile/scan_only.py 2 dense [binorder]',
        workload='131 072-point scan of C2, sparse = 0, ~10 events in EVERY bin',
        kernel_trace_count_sorted_rows=trace,
        count_sorted_rows=counters('sorted'), rows_in_bin_order=counters('binorder'),
        plain_runs=[ln.strip() for ln in open(os.path.join(SRC, 'plain.txt')) if ln.startswith('scan of')])
    if trace:
        out['evaluations_per_s_by_kernel_trace'] = POINTS / (trace['average_us'] * 1e-6)
    with open(os.path.join(dst, tag + '_scan_dense_data_pmc.json'), 'w') as f:
        json.dump(out, f, indent=1)
    sparse = counters('sparse')
    if sparse:
        st = {}
        with open(os.path.join(SRC, 'kts', 'kt_kernel_stats.csv')) as f:
            for row in csv.DictReader(f):
                if 'k_scan_mfma' in row['Name'] or 'k_scan_sorted' in row['Name']:
                    st = dict(name=row['Name'], calls=int(row['Calls']), average_us=float(row['AverageNs']) / 1e3,
                              min_us=float(row['MinNs']) / 1e3, max_us=float(row['MaxNs']) / 1e3)
        shutil.copy(os.path.join(SRC, 'kts', 'kt_kernel_stats.csv'), os.path.join(dst, tag + '_sparse_scan_kernel_stats.csv'))
        with open(os.path.join(dst, tag + '_sparse_scan_pmc.json'), 'w') as f:
            json.dump(dict(round=rnd, command='tools/profile_scan_dense.sh: rocprofv3 --pmc <8 SQ counters> / --kernel-trace --stats (own run) '
                                              '-- python3 tools/profile/sparse_scan_only.py',
                           workload='10^6-point scan of C2 on the default path (non-empty-bin form): k_scan_sorted over the compacted rows, '
                                    'ordered by count (64-bin strips, one logarithm per four work items)',
                           kernel_trace=st, evaluations_per_s_by_kernel_trace=(1e6 / (st['average_us'] * 1e-6) if st else None),
                           previous_round='profiles/r03_sparse_scan_pmc.json: k_scan_mfma<2,8,false,2> 13.76 ms, 3.62 vector instructions per MFMA, matrix pipe busy 63.5 %',
                           **sparse), f, indent=1)
        print('default-path scan: %s' % st, sparse['derived'])
    d1, d0 = out['count_sorted_rows']['derived'], out['rows_in_bin_order']['derived']
    print('vector instructions per MFMA %.2f -> %.2f; LDS conflict share of wave cycles %.1f %% -> %.1f %%; matrix pipe busy %.1f %% -> %.1f %%' % (
        d0['other_vector_instructions_per_mfma'], d1['other_vector_instructions_per_mfma'],
        100 * d0['lds_bank_conflict_share_of_wave_cycles'], 100 * d1['lds_bank_conflict_share_of_wave_cycles'],
        100 * d0['mfma_busy_fraction_of_SIMD_cycles'], 100 * d1['mfma_busy_fraction_of_SIMD_cycles']))
    print(out['plain_runs'], trace)


if __name__ == '__main__':
    main(int(sys.argv[1]))
