"""The train step's kernels STANDALONE (one stream: PIVP_SIDE_STREAM=0, durations additive) summed per step, next to the two-stream wall time.

    python scripts/train_standalone_table.py <single-stream kernel_stats.csv> <steps in that trace> <two-stream ms per step> [<single-stream ms per step>]

VERDICT r04 item 3b: if the two-stream wall time is within 1.05 x the standalone sum of the matrix-pipe kernels, the sweep's schedule has nothing
left to give and what remains is the kernels themselves.  Rows: kernel families (matrix-pipe ones first), launches and standalone microseconds
per step."""
import csv
import sys

FAMILIES = [   # (label, substrings (all must match), matrix pipe?)
    ('ConvLSTM gate conv, forward', ('igemm_f32_kernel', 'true'), True),
    ('ConvLSTM gate conv, forward (bf16 / pieces)', ('convlstm_bf16_kernel', ', true,'), True),
    ('ConvLSTM gate conv, forward (L2-direct pieces)', ('convlstm_x6g_kernel', 'true,'), True),
    ('ConvLSTM data gradient (fp32)', ('igemm_f32_kernel', 'false'), True),
    ('ConvLSTM data gradient (bf16 / pieces)', ('convlstm_bf16_kernel', ', false,'), True),
    ('ConvLSTM data gradient (L2-direct pieces)', ('convlstm_x6g_kernel', 'false,'), True),
    ('ConvLSTM weight gradient (fp32)', ('wgrad5x5_kernel',), True),
    ('ConvLSTM weight gradient (bf16 / pieces)', ('wgrad25_bf16_kernel',), True),
    ('ConvLSTM weight gradient (bf16, one timestep)', ('wgrad5x5_bf16_kernel',), True),
    ('3x3 conv / deconv forward + data gradients (igemm_small)', ('igemm_small_kernel',), True),
    ('3x3 deconv forward + data gradient (tile kernel)', ('deconv3x3s2_tile_kernel',), True),
    ('3x3 conv / deconv weight gradients: nine taps, batches of timesteps', ('wgrad3x3s2_kernel',), True),
    ('... their partial planes\' reduction', ('wgrad3x3s2_reduce',), False),
    ('3x3 conv / deconv weight gradients (per-tap kernel)', ('igemm_wgrad_kernel',), True),
    ('... their partial-sum reduction', ('igemm_wgrad_reduce',), False),
    ('gate backward (+ LayerNorm dx)', ('lstm_gates_bwd',), False),
    ('LayerNorm backward: sums + parameter planes', ('ln_bwd_sums_params',), False),
    ('LayerNorm backward: apply / reduce', ('ln_bwd_',), False),
    ('LayerNorm apply (forward)', ('ln_apply',), False),
    ('frame head (forward output side)', ('frame_head_kernel',), False),
    ('composite backward', ('composite_bwd',), False),
    ('heads backward + mask softmax backward', ('heads_bwd',), False),
    ('mask softmax backward', ('mask_softmax_bwd',), False),
    ('motion head Linear (fwd partials, bwd)', ('skinny_linear',), False),
    ('kernel generator backward', ('cdna_kernels_bwd',), False),
    ('enc0 (fwd, data / weight gradient)', ('enc0',), False),
    ('enc3 + state predictor (fwd, bwd)', ('enc3_state',), False),
    ('ReLU mask / strided add / loss gradient', ('relu_mask',), False),
    ('strided add', ('add_strided',), False),
    ('loss gradient', ('scaled_diff',), False),
    ('weight preparation (transposes, packs)', ('weight_prep', ), False),
    ('weight packs (per layer)', ('pack_lstm',), False),
    ('transposes (per layer)', ('repack_transpose',), False),
    ('absmax (fp16-piece scales)', ('absmax_partials',), False),
    ('Adam', ('adam_kernel',), False),
    ('loss', ('sqerr_partials',), False),
    ('loss finalize', ('loss_finalize',), False),
    ('memset / copies', ('__amd_rocclr',), False),
]


def main():
    path, steps, wall2 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
    wall1 = float(sys.argv[4]) if len(sys.argv) > 4 else None
    rows = list(csv.DictReader(open(path)))
    acc = [[0.0, 0.0] for _ in FAMILIES]
    other = [0.0, 0.0]
    for r in rows:
        name, calls, ns = r['Name'], float(r['Calls']), float(r['TotalDurationNs'])
        for k, (_, subs, _) in enumerate(FAMILIES):
            if all(s in name for s in subs):
                acc[k][0] += calls; acc[k][1] += ns
                break
        else:
            other[0] += calls; other[1] += ns
    print('| kernel family | launches / step | standalone us / step | matrix pipe |')
    print('|---|---|---|---|')
    tot = mp = 0.0
    nl = 0.0
    for (label, _, is_mp), (calls, ns) in zip(FAMILIES, acc):
        if calls == 0:
            continue
        us = ns / steps / 1e3
        tot += us; nl += calls / steps
        mp += us if is_mp else 0.0
        print('| %s | %.1f | %.1f | %s |' % (label, calls / steps, us, 'yes' if is_mp else ''))
    if other[0]:
        us = other[1] / steps / 1e3
        tot += us; nl += other[0] / steps
        print('| other | %.1f | %.1f | |' % (other[0] / steps, us))
    print('| **sum** | %.0f | **%.1f** | %.1f on the matrix pipe |' % (nl, tot, mp))
    print()
    print('two-stream wall time per step: %.3f ms = %.3f x the standalone sum of ALL kernels, %.3f x the matrix-pipe kernels\' sum' % (wall2, wall2 * 1e3 / tot, wall2 * 1e3 / mp))
    if wall1:
        print('one-stream wall time per step: %.3f ms (launch gaps: %.3f ms)' % (wall1, wall1 - tot / 1e3))


if __name__ == '__main__':
    main()
