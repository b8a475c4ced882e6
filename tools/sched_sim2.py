"""Replays real per-lane period sequences (from the CPU oracle) through the scheduler of
vs_synth_kernel AS IT IS NOW (room check on the actual period + spare slots, ready threshold) and
counts generator rounds and filter super-steps.  Cost model: a round or a super-step costs the
same whether 1 or 64 lanes take part.
Usage: python tools/sched_sim2.py [config index] [first lane]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po


def get_T(index, nl, lane0=0):
    specs, fs, dur, _ = configs.config_specs(index, nl, lane0)
    lanes, d = vs.lanes_from_specs(specs)
    n = vs.num_samples(fs, d)
    Ts = []
    for l in range(nl):
        f, recs, ncyc, nd = po.source_one(lanes[l], n, 4000)
        Ts.append(recs['T'][:ncyc].astype(np.int64))
    return Ts, n


def simulate(Ts, N, C, extra, ready_min, SS=24):
    L = len(Ts)
    g = np.zeros(L, int); n = np.zeros(L, int); k = np.zeros(L, int)
    T = np.array([t[0] for t in Ts])
    live = np.ones(L, bool)
    rounds = ssteps = gen_part = ss_part = 0
    while live.any():
        ready = live & ((g - n >= SS) | (g >= N))
        pend = live & (g < N)
        want = pend & (g - n + T + extra <= C)
        n_live, n_ready = live.sum(), ready.sum()
        filter_now = n_ready > 0 and (n_ready * 64 >= n_live * ready_min or not want.any())
        if not filter_now:
            idx = np.where(want)[0]
            assert len(idx)
            for i in idx:
                g[i] += T[i]; k[i] += 1
                T[i] = Ts[i][k[i]] if k[i] < len(Ts[i]) else T[i]
            rounds += 1; gen_part += len(idx)
            continue
        idx = np.where(ready)[0]
        n[idx] += SS; ssteps += 1; ss_part += len(idx)
        live &= n < N
    return rounds, ssteps, gen_part / (rounds * L), ss_part / (ssteps * L)


if __name__ == '__main__':
    index = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    lane0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    Ts, N = get_T(index, 64, lane0)
    cyc = [len(t) for t in Ts]
    print('cycles min/mean/max', min(cyc), np.mean(cyc), max(cyc), 'N', N, 'ideal super-steps', -(-N // 24))
    Rg, Rf = 3500.0, 24 * 59.0   # instructions per round / per super-step (order of magnitude)
    for C in (288, 312, 336):
        for extra in (0, 8):
            for rm in (32, 48, 58, 64):
                r, s, gp, sp = simulate(Ts, N, C, extra, rm)
                print('C=%d extra=%d ready_min=%2d: rounds %4d (attendance %.2f) super-steps %4d (attendance %.2f)  instr/sample %.1f'
                      % (C, extra, rm, r, gp, s, sp, (r * Rg + s * Rf) / N))
