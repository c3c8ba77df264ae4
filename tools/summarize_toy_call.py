"""gpurun_out/prof_toy/ (tools/profile_toy_call.sh) -> profiles/rNN_toy_call.json: per-kernel durations and gaps of the toy-MC
call of configs[2], counters of the dataset kernel (medians per launch) and the figures derived from them as in
profiles/r03_toy_call.json.   usage: python tools/summarize_toy_call.py <round>"""
import collections
import csv
import json
import os
import re
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof_toy')


def counters(sub):
    per = collections.defaultdict(list)
    path = os.path.join(SRC, sub, 'pmc_counter_collection.csv')
    if not os.path.exists(path):
        return {}
    by_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for row in csv.DictReader(f):
            if 'k_dataset_dot_tiled' in row['Kernel_Name']:
                by_dispatch[row['Dispatch_Id']][row['Counter_Name']] += float(row['Counter_Value'])
    for d in by_dispatch.values():
        for k, v in d.items():
            per[k].append(v)
    return {k: statistics.median(v) for k, v in per.items()}


def main(rnd):
    tl = open(os.path.join(SRC, 'timeline.txt')).read()
    trace = {}
    for m in re.finditer(r'(k_\w+)\s+median\s+([\d.]+) us\s+\(min\s+([\d.]+), (\d+) calls\)', tl):
        trace[m.group(1)] = dict(median_us=float(m.group(2)), min_us=float(m.group(3)), calls=int(m.group(4)))
    m = re.search(r'last kernel end: median ([\d.]+) us; call to call: median ([\d.]+) us', tl)
    trace['first_kernel_start_to_last_kernel_end_median_us'] = float(m.group(1))
    trace['call_to_call_median_us'] = float(m.group(2))
    c = counters('pmc1')
    c.update(counters('pmc2'))
    line = json.loads(open(os.path.join(SRC, 'plain.json')).read().strip().splitlines()[-1])
    leg = line.get('leg', {})
    entries = leg.get('nonempty_bins_this_rank') or 94226163
    entry_bytes = (leg.get('roofline') or {}).get('list_entry_bytes') or 4
    derived = {}
    if c:
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0                       # (the counter adds the 8 XCDs up)
        dur_us = trace['k_dataset_dot_tiled']['median_us']
        derived = {
            'kernel_cycles_per_XCD': cyc,
            'vector_alu_busy_fraction (4 cycles per wave instruction)': c['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * 256 * 4) if 'SQ_ACTIVE_INST_VALU' in c else None,
            'vector_instructions_per_64_entries': c['SQ_INSTS_VALU'] / (entries / 64.0),
            'lds_busy_fraction_of_CU_cycles': c['SQ_LDS_IDX_ACTIVE'] / (cyc * 256) if 'SQ_LDS_IDX_ACTIVE' in c else None,
            'lds_bank_conflict_share_of_lds_busy': c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] if 'SQ_LDS_IDX_ACTIVE' in c else None,
            'wave_time_waiting_on_counters': c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'] if 'SQ_WAIT_INST_ANY' in c else None,
            'list_entry_bytes': entry_bytes,
            'entry_stream_TBps': entries * entry_bytes / (dur_us * 1e-6) / 1e12,
        }
    out = dict(
        round=rnd,
        workload='bench.py --config C3: 10^4 toy datasets of C2 (~9 400 non-empty bins each), one parameter point per call (bi_eval_datasets)',
        commands=['tools/profile_toy_call.sh: rocprofv3 --kernel-trace -- python3 bench.py --config C3 --no-cpu-baseline --steps 30  (tools/profile/c3_call_timeline.py)',
                  'rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- (same, --steps 10)',
                  'rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU -- (same)'],
        bench_line_plain=dict(value=line.get('value'), unit=line.get('unit'), ms_per_step=line.get('ms_per_step')),
        kernel_trace=trace, k_dataset_dot_tiled_counters_median_per_launch=c, derived=derived,
        previous_round='profiles/r03_toy_call.json: log mu 46 us, dot 90 us, finish 15.6 us, call to call 221 us; 19.7 vector instructions per 64 entries; '
                       'earlier in round 4 (four-byte entries): dot 84.7 us, call to call 176 us',
        note='the bank-conflict share of LDS-busy time is a property of the access pattern (64 independent random 8-byte reads per instruction: '
             'expected worst bank load ~3.5 over 32 bank pairs).  Measured with the entry loads taken out the kernel runs in 24 us: the entry '
             'stream is its time, which is why the lists went to two-byte entries')
    dst = os.path.join(ROOT, 'profiles', 'r%02d_toy_call.json' % rnd)
    with open(dst, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(dict(kernel_trace=trace, derived=derived), indent=1))


if __name__ == '__main__':
    main(int(sys.argv[1]))
