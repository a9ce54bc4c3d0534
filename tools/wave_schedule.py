#!/usr/bin/env python3
"""What does the ORDER of k_march's waves cost?  From the per-wave durations tools/wave_timing.py saved (HZ_WT_SAVE=file.npy:
the launch grid, segment-major as the hardware dispatches it), list-schedule them onto the chip's 4096 wave slots (256 CUs x
4 SIMDs x 4 waves) in dispatch order, in reverse, longest first - against the bound sum / slots.

    python tools/wave_schedule.py gpurun_out/r5b27/wave_t.npy"""
import heapq
import sys

import numpy as np


def makespan(durations, slots=4096):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for d in durations:
        t = heapq.heappop(h) + d
        end = max(end, t)
        heapq.heappush(h, t)
    return end


def main():
    a = np.load(sys.argv[1])
    t = a[:, :, 0].astype(np.float64) / 2400.0           # us
    gy, gx = t.shape
    order = t.ravel()                                    # y-major: blockIdx.x fastest
    print("waves %d, sum %.1f ms, bound sum/4096 = %.1f us, longest wave %.1f us" % (order.size, order.sum()/1e3, order.sum()/4096, order.max()))
    print("dispatch order           : %.1f us" % makespan(order))
    print("reverse                  : %.1f us" % makespan(order[::-1]))
    print("longest first            : %.1f us" % makespan(np.sort(order)[::-1]))
    print("segments by mean duration: %.1f us" % makespan(t[np.argsort(-t.mean(axis=1))].ravel()))
    # where the long waves sit in dispatch order
    n = 10
    for k in range(n):
        blk = t[gy*k//n: gy*(k+1)//n]
        print("  segments %3d..%3d: mean %.1f us, p99 %.1f, max %.1f" % (gy*k//n, gy*(k+1)//n - 1, blk.mean(), np.percentile(blk, 99), blk.max()))


if __name__ == "__main__":
    main()
