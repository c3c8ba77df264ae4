"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof/) and tools/profile_scan.sh
(gpurun_out/prof_scan/, optionally gpurun_out/prof_scan_r1kernel/ = the same passes on the previous kernel) into the
committed summaries profiles/rNN_bench.json, rNN_bench_kernel_stats.csv, rNN_bench_under_rocprof.json,
rNN_pmc_traffic.json, rNN_bb.json + rNN_bb_kernel_stats.csv, rNN_scan_pmc.json + rNN_scan_kernel_stats.csv +
rNN_mfma_valu_overlap.txt.

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
                   '--no-cpu-baseline --no-extras --no-legs   (tools/profile_round.sh)',
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
    summarize_bb(tag, dst)
    summarize_unbinned(tag, dst)
    summarize_scan(tag, dst)


def _kernel_counters(path, needle):
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row['Kernel_Name']:
                acc.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}, (len(next(iter(acc.values()))) if acc else 0)


def _kernel_stats(path, needle):
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row['Name']:
                return dict(name=row['Name'], calls=int(row['Calls']), average_us=float(row['AverageNs']) / 1e3,
                            min_us=float(row['MinNs']) / 1e3, max_us=float(row['MaxNs']) / 1e3)
    return None


def summarize_bb(tag, dst):
    """The Beeston-Barlow pass (tools/profile/bb_only.py): kernel trace + FETCH_SIZE -> rNN_bb.json."""
    kt = os.path.join(SRC, 'bbkt', 'bb_kernel_stats.csv')
    if not os.path.exists(kt):
        return
    shutil.copy(kt, os.path.join(dst, tag + '_bb_kernel_stats.csv'))
    needle = 'k_morph_reduce<1, true, true, 0>'
    st = _kernel_stats(kt, needle)
    fetch, n = _kernel_counters(os.path.join(SRC, 'bbfetch', 'bb_counter_collection.csv'), needle)
    algo = 8 * (16 * 6 + 16 + 1) * 50 ** 4
    rd = fetch['FETCH_SIZE'] * 1024 * 2
    out = {'command': 'rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE (separate runs) -- python3 tools/profile/bb_only.py',
           'workload': 'Beeston-Barlow, one grid cell of configs[4]: 2^4 anchors, 6 sources, 50^4 bins, one evaluation per launch',
           'kernel': needle.replace(', ', ','), 'kernel_trace': st,
           'algorithmic_bytes_per_launch': algo, 'algorithmic_note': '8 * (16*5 plain + 16 BB-source + 16 MC-count rows + 1 counts row) * 50^4',
           'achieved_GBps': algo / st['average_us'] / 1e3, 'frac_of_8TBps': algo / st['average_us'] / 1e3 / 8000,
           'FETCH_SIZE_KB_mean': fetch['FETCH_SIZE'], 'fetch_launches': n, 'hbm_read_bytes_per_launch': rd,
           'traffic_over_algorithmic': rd / algo, 'hip_event_line': open(os.path.join(SRC, 'bb_plain.txt')).read().strip()}
    with open(os.path.join(dst, tag + '_bb.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('BB kernel: %.1f us -> %.0f GB/s (%.3f of 8 TB/s), traffic / algorithmic %.4f' % (
        st['average_us'], out['achieved_GBps'], out['frac_of_8TBps'], out['traffic_over_algorithmic']))


def summarize_unbinned(tag, dst):
    """The unbinned pass (tools/profile/unbinned_only.py): kernel trace + FETCH_SIZE -> rNN_unbinned.json."""
    kt = os.path.join(SRC, 'unbkt', 'unb_kernel_stats.csv')
    if not os.path.exists(kt):
        return
    shutil.copy(kt, os.path.join(dst, tag + '_unbinned_kernel_stats.csv'))
    needle = 'k_morph_reduce<1, false, true, 2>'
    st = _kernel_stats(kt, needle)
    score = _kernel_stats(kt, 'k_score_rows') or _kernel_stats(kt, 'k_score_events')     # (k_score_events: round 3's single kernel)
    fetch, n = _kernel_counters(os.path.join(SRC, 'unbfetch', 'unb_counter_collection.csv'), needle)
    algo = 8 * 8 * 32 * 10 ** 6
    rd = fetch['FETCH_SIZE'] * 1024 * 2
    out = {'command': 'rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE (separate runs) -- python3 tools/profile/unbinned_only.py',
           'workload': 'extended unbinned likelihood, C2 shape (4 sources, 5^3 anchors), 10^6 events, 8 evaluations per launch in disjoint grid cells',
           'kernel': needle.replace(', ', ','), 'kernel_trace': st, 'set_data_kernel_trace': score,
           'algorithmic_bytes_per_launch': algo, 'algorithmic_note': '8 evaluations * 8 B * 2^3 corners * 4 sources * 10^6 events (no counts row in this mode)',
           'achieved_GBps': algo / st['average_us'] / 1e3, 'frac_of_8TBps': algo / st['average_us'] / 1e3 / 8000,
           'FETCH_SIZE_KB_mean': fetch['FETCH_SIZE'], 'fetch_launches': n, 'hbm_read_bytes_per_launch': rd,
           'traffic_over_algorithmic': rd / algo, 'hip_event_line': open(os.path.join(SRC, 'unb_plain.txt')).read().strip()}
    with open(os.path.join(dst, tag + '_unbinned.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('unbinned kernel: %.1f us -> %.0f GB/s (%.3f of 8 TB/s), traffic / algorithmic %.4f; event scoring kernel %.1f us' % (
        st['average_us'], out['achieved_GBps'], out['frac_of_8TBps'], out['traffic_over_algorithmic'], score['average_us'] if score else -1))


def _scan_block(src, needle, points=131072, flop_per_eval=2.0 * 32 * 10 ** 6):
    c = {}
    for p in ('pmc1', 'pmc2', 'pmc3'):
        path = os.path.join(src, p, 'pmc_counter_collection.csv')
        if os.path.exists(path):
            c.update(_kernel_counters(path, needle)[0])
    st = _kernel_stats(os.path.join(src, 'kt', 'kt_kernel_stats.csv'), needle)
    if not c or not st:
        return None
    cycles = c['GRBM_GUI_ACTIVE'] / 8                          # the counter sums the 8 XCDs
    simd_cycles = 1024 * cycles
    n_mfma = c['SQ_INSTS_MFMA']
    valu_other = c['SQ_INSTS_VALU'] - n_mfma
    return {'kernel': needle, 'kernel_trace': st, 'counters_per_launch': c,
            'derived': {'effective_clock_GHz': cycles / (st['average_us'] * 1e3),
                        'mfma_busy_fraction_of_SIMD_cycles': c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles,
                        'cycles_per_mfma': c['SQ_VALU_MFMA_BUSY_CYCLES'] / n_mfma,
                        'valu_instructions_per_mfma': valu_other / n_mfma,
                        'valu_instructions_per_16point_item_and_64bin_strip': valu_other / n_mfma * 32,
                        'fp64_valu_instructions_per_mfma': (c.get('SQ_INSTS_VALU_ADD_F64', 0) + c.get('SQ_INSTS_VALU_FMA_F64', 0) + c.get('SQ_INSTS_VALU_MUL_F64', 0)) / n_mfma,
                        'mfma_valu_coexec_cycles': c.get('SQ_VALU_MFMA_COEXEC_CYCLES'),
                        'lds_bank_conflict_cycles_per_lds_instruction': c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_INSTS_LDS', 0), 1),
                        'waves': c.get('SQ_WAVES'),
                        'evaluations_per_s': points / (st['average_us'] * 1e-6),
                        'fp64_fma_TFLOPs': flop_per_eval * points / (st['average_us'] * 1e-6) / 1e12,
                        'frac_of_78.6_TFLOPs': flop_per_eval * points / (st['average_us'] * 1e-6) / 1e12 / 78.6}}


def summarize_scan(tag, dst):
    """Counter evidence for the dense scan (tools/profile_scan.sh) -> rNN_scan_pmc.json."""
    src = os.path.join(ROOT, 'gpurun_out', 'prof_scan')
    if not os.path.exists(os.path.join(src, 'pmc1')):
        return
    shutil.copy(os.path.join(src, 'kt', 'kt_kernel_stats.csv'), os.path.join(dst, tag + '_scan_kernel_stats.csv'))
    if os.path.exists(os.path.join(src, 'ktd', 'kt_kernel_stats.csv')):
        shutil.copy(os.path.join(src, 'ktd', 'kt_kernel_stats.csv'), os.path.join(dst, tag + '_scan_dense_data_kernel_stats.csv'))
    shutil.copy(os.path.join(src, 'overlap.txt'), os.path.join(dst, tag + '_mfma_valu_overlap.txt'))
    out = {'command': 'rocprofv3 --pmc <8 SQ counters [+ GRBM_GUI_ACTIVE]> (three separate runs) / --kernel-trace --stats (own run) '
                      '-- python3 tools/profile/scan_only.py   (tools/profile_scan.sh)',
           'workload': '131 072 parameter points over the C2 model (64 grid cells, 128 sixteen-point items per cell), sparse = 0 '
                       '(every bin visited), ~10^4 events in 10^6 bins',
           'units': 'SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, '
                    'GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md)',
           'plain_runs': open(os.path.join(src, 'plain.txt')).read().strip().splitlines(),
           'this_round': _scan_block(src, 'k_scan_valid<4, 8, false>')}
    summarize_scan_dense_data(tag, dst, src)
    prev = os.path.join(ROOT, 'gpurun_out', 'prof_scan_r1kernel')
    if os.path.exists(os.path.join(prev, 'pmc1')):
        out['previous_kernel'] = _scan_block(prev, 'k_scan_mfma<4, 8, false>')
        out['previous_kernel']['note'] = 'the round-1 kernel (per-bin Poisson terms inside the matrix-core kernel) on the same command, profiled before the change'
    with open(os.path.join(dst, tag + '_scan_pmc.json'), 'w') as f:
        json.dump(out, f, indent=1)
    for key in ('previous_kernel', 'this_round'):
        b = out.get(key)
        if b:
            d = b['derived']
            print('%s %s: %.1f ms, %.0f evals/s, MFMA busy %.3f, %.1f other VALU instr per MFMA, clock %.2f GHz' % (
                key, b['kernel'], b['kernel_trace']['average_us'] / 1e3, d['evaluations_per_s'], d['mfma_busy_fraction_of_SIMD_cycles'],
                d['valu_instructions_per_mfma'], d['effective_clock_GHz']))


def summarize_scan_dense_data(tag, dst, src):
    """The same scan over data with events in every bin (k_scan_mfma computes every logarithm) -> rNN_scan_dense_data_pmc.json."""
    path = os.path.join(src, 'pmcd', 'pmc_counter_collection.csv')
    ktd = os.path.join(src, 'ktd', 'kt_kernel_stats.csv')
    if not (os.path.exists(path) and os.path.exists(ktd)):
        return
    needle = 'k_scan_mfma<2, 8, false, false>'
    c, _ = _kernel_counters(path, needle)
    st = _kernel_stats(ktd, needle)
    if not c or not st:
        return
    simd_cycles = 1024 * c['GRBM_GUI_ACTIVE'] / 8
    n_mfma = c['SQ_INSTS_MFMA']
    plain = [l for l in open(os.path.join(src, 'plain.txt')).read().splitlines() if 'dense data' in l]
    out = {'command': 'rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT '
                      'SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE / --kernel-trace --stats (own run) -- python3 tools/profile/scan_only.py 2 dense',
           'workload': '131 072-point scan of C2, sparse = 0, ~10 events in EVERY bin (the split into a non-empty-bin pass and a '
                       'validity pass does not apply: every bin needs its logarithm)',
           'kernel': needle.replace(', ', ','), 'kernel_trace': st, 'counters_per_launch': c,
           'derived': {'evaluations_per_s': 131072 / (st['average_us'] * 1e-6),
                       'mfma_busy_fraction_of_SIMD_cycles': c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles,
                       'other_vector_instructions_per_mfma': (c['SQ_INSTS_VALU'] - n_mfma) / n_mfma,
                       'other_vector_instructions_per_matrix_element': (c['SQ_INSTS_VALU'] - n_mfma) / n_mfma * 8 / 4,
                       'lds_instructions_per_mfma': c['SQ_INSTS_LDS'] / n_mfma,
                       'lds_bank_conflict_cycles_per_lds_instruction': c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_INSTS_LDS'], 1),
                       'sum_of_parts_bound_evals_per_s': 1024 * (c['GRBM_GUI_ACTIVE'] / 8 / (st['average_us'] * 1e-6)) /
                                                         ((c['SQ_VALU_MFMA_BUSY_CYCLES'] + 4.0 * (c['SQ_INSTS_VALU'] - n_mfma)) / 131072),
                       'plain_run': plain}}
    with open(os.path.join(dst, tag + '_scan_dense_data_pmc.json'), 'w') as f:
        json.dump(out, f, indent=1)
    d = out['derived']
    print('dense-data scan %s: %.1f ms, %.0f evals/s, MFMA busy %.3f, %.1f other VALU per MFMA, bound if MFMA and VALU cycles simply add: %.0f evals/s' % (
        needle, st['average_us'] / 1e3, d['evaluations_per_s'], d['mfma_busy_fraction_of_SIMD_cycles'],
        d['other_vector_instructions_per_mfma'], d['sum_of_parts_bound_evals_per_s']))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
