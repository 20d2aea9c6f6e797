"""Per-kernel means of the counters collected by scripts/pmc_stalls.sh (kernels told apart by name and grid)."""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void pivp::', '').replace('pivp::', '')
            key = (name[:44], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))
            acc[key][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[key][r['Counter_Name']] += 1
names = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_VALU_MFMA_BUSY_CYCLES',
         'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_LDS']
rows = []
for k in acc:
    m = {n: acc[k][n] / max(1, cnt[k][n]) for n in names}
    rows.append((m['SQ_WAVE_CYCLES'] * cnt[k]['SQ_WAVE_CYCLES'], k, m, cnt[k]['SQ_WAVE_CYCLES']))
rows.sort(reverse=True)
print('per-launch means; fractions are of SQ_WAVE_CYCLES (wave-resident cycles, summed over waves) unless noted')
print('%-44s %-10s %5s %12s %8s %8s %8s %10s %8s %8s %8s %9s' % ('kernel', 'grid', 'n', 'wave_cyc', 'wait', 'wait_in', 'wait_lds', 'mfma/busy', 'a_lds', 'a_valu', 'a_vmem', 'bankconf'))
for _, k, m, n in rows[:28]:
    wc = max(m['SQ_WAVE_CYCLES'], 1.0)
    print('%-44s %-10s %5d %12.0f %8.3f %8.3f %8.3f %10.3f %8.3f %8.3f %8.3f %9.3f' % (
        k[0], k[1], n, wc, m['SQ_WAIT_ANY'] / wc, m['SQ_WAIT_INST_ANY'] / wc, m['SQ_WAIT_INST_LDS'] / wc,
        m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(m['SQ_BUSY_CYCLES'], 1.0), m['SQ_ACTIVE_INST_LDS'] / wc, m['SQ_ACTIVE_INST_VALU'] / wc,
        m['SQ_ACTIVE_INST_VMEM'] / wc, m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1.0)))
