"""Per-kernel roofline table of one config-2 rollout (CDNA, B = 32, T = 10, 64 x 64, fp32) from a rocprofv3 kernel-stats CSV:
    python scripts/roofline_table.py profiles/r02/v2_kernel_stats.csv [n_rollouts]
    python scripts/roofline_table.py profiles/r04/v1_train_kernel_stats.csv --train        (the train step's families, per step)
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
# executed ConvLSTM flops, counted as the library counts them (pivp_plan_profile_read, the bench line's roofline object): 2 M 4C 25 (cx + C)
# per launch, and only the cx half at t = 0, where the all-zero h half of K is skipped: 8 x 107.4 + 60.4 = 919.4 GFLOP per rollout.
# (Round 3's table had 976 here and so under-stated this row's measured / floor: 1.12 instead of 1.19.)
LSTM_LAYERS = [(32, 32, 2), (32, 32, 2), (32, 64, 4), (64, 64, 4), (64, 128, 8), (128, 64, 4), (96, 32, 2)]     # (cx, C, map = 64 / level)
LSTM_FULL = sum(2.0 * B * (64 // lv) ** 2 * 4 * C * 25 * (cx + C) for cx, C, lv in LSTM_LAYERS) / 1e9
LSTM_FIRST = sum(2.0 * B * (64 // lv) ** 2 * 4 * C * 25 * cx for cx, C, lv in LSTM_LAYERS) / 1e9
LSTM_GFLOP = (T1 - 1) * LSTM_FULL + LSTM_FIRST
HW = 64 * 64
MB = 1e6
fam = [   # (label, substring(s) of the kernel name, GFLOP per rollout, MB per rollout)
    ('ConvLSTM gate conv (igemm_f32_kernel<..,true>)', ('igemm_f32_kernel', 'true'), LSTM_GFLOP, 0.0),
    ('enc5 / enc6 transposed conv (deconv3x3s2_tile)', ('deconv3x3s2_tile',), 2 * MAC['enc56'] * B * T1 / 1e9,
     T1 * B * (32 * 32 * 96 * 2 + 64 * 64 * 64 + 16 * 16 * 96) * 4 / MB),
    ('enc1 / enc2 / enc4 (igemm_small)', ('igemm_small',), 2 * MAC['enc124'] * B * T1 / 1e9, 0.0),
    ('LayerNorm apply x 4 (norm_enc0, hidden1, hidden3, hidden5; hidden2 / 4 / 6 / 7: inside enc1 / 2 / 5 / 6)', ('ln_apply',), 0.0, T1 * B * 2 * (2 * 32768 + 16384 + 8192) * 4 / MB),
    ('frame head: norm_enc6 + 1x1 heads + softmax + CDNA transform + blend, one launch (round 4; round 6: the kernel finisher rides behind enc5)', ('frame_head_kernel',),
     2 * (MAC['heads'] + MAC['cdna']) * B * T1 / 1e9, T1 * (B * HW * 64 * 4 + 2 * HW * 64 * 4 + B * HW * (3 + 3 + 3) * 4) / MB),
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
    n_roll = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else None
    if '--train' in sys.argv:
        return train_table(rows)
    if n_roll is None:     # infer from the enc0 kernel: one launch per predicted frame
        n_roll = sum(int(r['Calls']) for r in rows if 'conv_enc0_rows' in r['Name']) / float(T1)
    used = set()
    print('| kernel family | launches / rollout | measured us / rollout | algorithmic GFLOP | algorithmic MB | bound | floor us | measured / floor |')
    print('|---|---|---|---|---|---|---|---|')
    tot_meas = tot_floor = 0.0
    for label, keys, gflop, mb in fam:
        sel = [r for r in rows if (all(k in r['Name'] for k in keys) if keys[0] == 'igemm_f32_kernel' else any(k in r['Name'] for k in keys))]
        for r in sel:
            used.add(r['Name'])
        if not sel:
            continue               # (the separate heads / composite kernels, or the fused launch: whichever the trace does not contain)
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


def train_table(rows):
    """The fp32 train step (optimizer.update: forward + BPTT sweep + Adam) by kernel family, per step: kernel time summed over BOTH streams
    against the matrix-core floor of each family's algorithmic flops.  The backward sweep differentiates the same contractions: data
    gradients = the forward flops (t = 0 computes d x only), weight gradients = the forward flops (t = 0: the x half)."""
    steps = sum(int(r['Calls']) for r in rows if 'adam_kernel' in r['Name'])
    fwd = LSTM_GFLOP
    dgrad = (T1 - 1) * LSTM_FULL + LSTM_FIRST              # t = 0: only the cx columns of d [x, h]
    wgrad = (T1 - 1) * LSTM_FULL + LSTM_FIRST              # t = 0: h_{-1} = 0, its half of dW gets nothing
    small = 2 * (MAC['enc56'] + MAC['enc124']) * B * T1 / 1e9
    tfam = [
        ('forward ConvLSTM gate conv (igemm_f32_kernel<..,true>)', lambda n: 'igemm_f32_kernel' in n and 'true' in n, fwd),
        ('ConvLSTM data gradients (igemm_f32_kernel<..,false>)', lambda n: 'igemm_f32_kernel' in n and 'false' in n, dgrad),
        ('ConvLSTM weight gradients (wgrad5x5_kernel, side stream)', lambda n: 'wgrad5x5_kernel' in n, wgrad),
        ('3x3 conv / deconv: forward + data gradients (igemm_small, deconv3x3s2_tile)', lambda n: 'igemm_small' in n or 'deconv3x3s2_tile' in n, 2 * small),
        ('3x3 conv / deconv weight gradients (wgrad3x3s2_kernel: nine taps, batches of timesteps, + reduce; side stream)', lambda n: 'igemm_wgrad' in n or 'wgrad3x3s2' in n, small),
        ('gate math backward (lstm_gates_bwd*)', lambda n: 'lstm_gates_bwd' in n, 0.0),
        ('LayerNorm backward (ln_bwd_*)', lambda n: 'ln_bwd' in n, 0.0),
        ('LayerNorm apply', lambda n: 'ln_apply' in n, 0.0),
        ('heads / composite / kernel generator, forward and backward', lambda n: any(k in n for k in ('heads', 'composite', 'skinny_linear', 'cdna_kernels', 'mask_softmax', 'frame_head')), 0.0),
        ('enc0, enc3 forward and backward', lambda n: 'enc0' in n or 'enc3' in n, 0.0),
        ('relu masks, adds, repacks, loss, Adam', lambda n: any(k in n for k in ('relu_mask', 'add_strided', 'repack', 'sqerr', 'loss_', 'adam', 'scaled_diff', 'bias_grad')), 0.0),
    ]
    print('| kernel family | launches / step | kernel us / step (both streams) | algorithmic GFLOP | floor us at %.1f TF | measured / floor |' % PEAK_TF)
    print('|---|---|---|---|---|---|')
    used = set(); tot = totf = 0.0
    for label, pred, gflop in tfam:
        sel = [r for r in rows if pred(r['Name']) and r['Name'] not in used]
        for r in sel:
            used.add(r['Name'])
        if not sel:
            continue
        us = sum(int(r['TotalDurationNs']) for r in sel) / 1e3 / steps
        calls = sum(int(r['Calls']) for r in sel) / float(steps)
        floor = gflop / PEAK_TF * 1e3
        tot += us; totf += floor
        print('| %s | %.0f | %.0f | %.1f | %s | %s |' % (label, calls, us, gflop, ('%.0f' % floor) if floor else '-', ('%.2f' % (us / floor)) if floor else '-'))
    rest = sum(int(r['TotalDurationNs']) for r in rows if r['Name'] not in used) / 1e3 / steps
    print('| (everything else: memsets, copies) | | %.0f | | | |' % rest)
    print()
    print('%d steps in the trace; kernel time %.0f us per step over both streams; matrix-core floor of the contractions %.0f us (%.1f GFLOP at %.1f TF)' % (
        steps, tot + rest, totf, fwd + dgrad + wgrad + 3 * small, PEAK_TF))


if __name__ == '__main__':
    main()
