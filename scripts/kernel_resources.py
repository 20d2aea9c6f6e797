"""Register / LDS / spill table of the kernels of one or more .hip sources (hipcc -Rpass-analysis=kernel-resource-usage; runs in the
build container, no GPU):   python scripts/kernel_resources.py igemm_f32 convlstm_bf16 [--grep pattern]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'physical-interaction-video-prediction_amd', 'csrc')
args = [a for a in sys.argv[1:] if not a.startswith('--')]
pat = None
if '--grep' in sys.argv:
    pat = sys.argv[sys.argv.index('--grep') + 1]
    args = [a for a in args if a != pat]
flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Rpass-analysis=kernel-resource-usage'] + os.environ.get('PIVP_EXTRA_FLAGS', '').split()
for name in args:
    src = os.path.join(CSRC, name if name.endswith('.hip') else name + '.hip')
    out = subprocess.run(['/opt/rocm/bin/hipcc'] + flags + ['-c', src, '-o', '/tmp/_kr.o'], stderr=subprocess.PIPE, text=True).stderr
    cur = None
    rows = {}
    for ln in out.splitlines():
        m = re.search(r'remark:\s+(.*?) \[-Rpass', ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:') or t.startswith('Name:'):
            cur = t.split(':', 1)[1].strip()
            rows[cur] = {}
        elif cur and ':' in t:
            k, v = t.split(':', 1)
            rows[cur][k.strip()] = v.strip()
    for fn, r in rows.items():
        dem = subprocess.run(['c++filt', fn], stdout=subprocess.PIPE, text=True).stdout.strip()
        dem = re.sub(r'\(.*', '', dem).replace('void pivp::', '').replace('pivp::', '')
        if pat and not re.search(pat, dem):
            continue
        print('%-52s VGPR %4s AGPR %4s spill %3s SGPR %3s LDS %6s occ %s' % (dem[:52], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('VGPRs Spill', '?'),
                                                                         r.get('TotalSGPRs', '?'), r.get('LDS Size [bytes/block]', '?'), r.get('Occupancy [waves/SIMD]', '?')))
