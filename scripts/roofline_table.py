"""Per-kernel roofline table of one config-2 rollout (CDNA, B = 32, T = 10, 64 x 64, fp32) from a rocprofv3 kernel-stats CSV:
    python scripts/roofline_table.py profiles/r02/v2_kernel_stats.csv [n_rollouts]
For every kernel family: launches per rollout, measured time, ALGORITHMIC work per rollout (flops for the contractions, compulsory
bytes for the streaming kernels: SURVEY.md App. E / DESIGN.md 5), the floor max(flops / 157.3 TF, bytes / 8 TB/s) and measured / floor.
The last line is SURVEY 8(d)'s  sum_k max(flops_k / peak, bytes_k / BW)  against the measured rollout."""
import csv
import sys

PEAK_TF, PEAK_TBS = 157.3, 8.0
B, T1 = 32, 9                      # sequences, predicted frames
MAC = dict(lstm=1678.0e6 * 0 + (209.7 * 2 + 157.3 + 209.7 + 157.3 + 314.6 + 419.4) * 1e6,      # per sample per step (App. E)
           enc56=(21.23 + 37.75) * 1e6, enc124=(2.36 + 2.36 + 9.44) * 1e6, heads=(2.88 + 0.79) * 1e6, enc0=2.46e6, enc3=0.30e6,
           lin=2.05e6, cdna=3.07e6)
# executed ConvLSTM flops: the first step skips the all-zero h half of K (976 of 1,015 GFLOP per rollout)
LSTM_GFLOP = 976.0
HW = 64 * 64
MB = 1e6
fam = [   # (label, substring(s) of the kernel name, GFLOP per rollout, MB per rollout)
    ('ConvLSTM gate conv (igemm_f32_kernel<..,true>)', ('igemm_f32_kernel', 'true'), LSTM_GFLOP, 0.0),
    ('enc5 / enc6 transposed conv (deconv3x3s2_tile)', ('deconv3x3s2_tile',), 2 * MAC['enc56'] * B * T1 / 1e9,
     T1 * B * (32 * 32 * 96 * 2 + 64 * 64 * 64 + 16 * 16 * 96) * 4 / MB),
    ('enc1 / enc2 / enc4 (igemm_small)', ('igemm_small',), 2 * MAC['enc124'] * B * T1 / 1e9, 0.0),
    ('LayerNorm apply x 4 (norm_enc0, hidden1, hidden3, hidden5; hidden2 / 4 / 6 / 7: inside enc1 / 2 / 5 / 6)', ('ln_apply',), 0.0, T1 * B * 2 * (2 * 32768 + 16384 + 8192) * 4 / MB),
    ('heads 1x1 + norm_enc6 + ReLU', ('heads_1x1',), 2 * MAC['heads'] * B * T1 / 1e9, T1 * (B * HW * 64 * 4 + 2 * HW * 64 * 4 + B * HW * 17 * 4) / MB),
    ('composite (softmax + CDNA transform + blend)', ('composite_kernel',), 2 * MAC['cdna'] * B * T1 / 1e9, T1 * B * HW * (3 + 11 + 3 + 3) * 4 / MB),
    ('kernel generator Linear(8192 -> 250)', ('skinny_linear_partials', 'cdna_kernels_finish'), 2 * MAC['lin'] * B * T1 / 1e9, T1 * 8192 * 256 * 4 / MB),
    ('enc0 5x5 s2 (+ LN partials)', ('conv_enc0_rows',), 2 * MAC['enc0'] * B * T1 / 1e9, T1 * B * (3 * HW + 32 * 32 * 32) * 4 / MB),
    ('enc3 1x1 + smear + state predictor', ('enc3_state_kernel',), 2 * MAC['enc3'] * B * T1 / 1e9, T1 * B * 64 * 64 * 2 * 4 / MB),
    ('loss / PSNR', ('sqerr_partials', 'loss_finalize'), 0.0, 8 * B * 3 * HW * 2 * 4 / MB),
]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    n_roll = float(sys.argv[2]) if len(sys.argv) > 2 else None
    if n_roll is None:     # infer from the composite kernel: one launch per predicted frame
        n_roll = sum(int(r['Calls']) for r in rows if 'composite_kernel' in r['Name']) / float(T1)
    used = set()
    print('| kernel family | launches / rollout | measured us / rollout | algorithmic GFLOP | algorithmic MB | bound | floor us | measured / floor |')
    print('|---|---|---|---|---|---|---|---|')
    tot_meas = tot_floor = 0.0
    for label, keys, gflop, mb in fam:
        sel = [r for r in rows if (all(k in r['Name'] for k in keys) if keys[0] == 'igemm_f32_kernel' else any(k in r['Name'] for k in keys))]
        for r in sel:
            used.add(r['Name'])
        calls = sum(int(r['Calls']) for r in sel) / n_roll
        us = sum(int(r['TotalDurationNs']) for r in sel) / 1e3 / n_roll
        f_us = gflop / PEAK_TF * 1e3                       # GFLOP / (TFLOP/s) = ms -> us
        b_us = mb / (PEAK_TBS * 1e6) * 1e6                 # MB / (TB/s) -> us
        floor = max(f_us, b_us)
        tot_meas += us; tot_floor += floor
        print('| %s | %.0f | %.0f | %.1f | %.0f | %s | %.0f | %.2f |' % (label, calls, us, gflop, mb, 'MFMA' if f_us >= b_us else 'HBM', floor, us / floor if floor else 0))
    rest = sum(int(r['TotalDurationNs']) for r in rows if r['Name'] not in used) / 1e3 / n_roll
    tot_meas += rest
    print('| (memsets, copies) | | %.0f | | | | 0 | |' % rest)
    print()
    print('sum of kernel time %.0f us per rollout; sum_k max(flops_k / %.1f TF, bytes_k / %.0f TB/s) = %.0f us; rollout / floor = %.2f (floor / rollout = %.2f)' % (
        tot_meas, PEAK_TF, PEAK_TBS, tot_floor, tot_meas / tot_floor, tot_floor / tot_meas))


if __name__ == '__main__':
    main()
