#!/usr/bin/env python3
"""CPU model of the LDS-DMA protocol of the two-blocks-per-CU record conv kernels (csrc/vae_conv_rec2.hip).

The kernels request their operands with hand-issued `global_load_lds_dwordx4` pieces, count their completion by hand
(`s_waitcnt vmcnt(N)` in front of every block barrier) and re-use ring slots / input stages across steps, K-steps and items.
None of that is visible to a compiler or to a test that only runs small shapes on a GPU; a wrong N or slot index shows up as
wrong pixels on some launch.  This script re-states the request / wait / read schedule of both kernels (same loops, same
formulas as the .hip file) and checks it against an ADVERSARIAL memory model:

  * a piece is only known to have landed once the wave that requested it has executed a vmcnt wait that covers it
    (in-order retirement: vmcnt(N) leaves the N youngest requests of that wave in flight);
  * other waves may read it only after a block barrier that follows that wait;
  * a piece requested in a barrier interval may land at any time from its request on: reading its destination in the same
    interval (by any wave), or in a later one before it is published, is an error, as is reading a destination whose
    published content is not the operand the reader expects (ring slot / stage re-used too early or too late).

usage: python tools/rec2_protocol_sim.py      (exit code 0 = every configuration passes; also run by tests/test_rec2_protocol.py)
"""
from __future__ import annotations

import sys

NWV = 4


class Wave:
    def __init__(self, w):
        self.w = w
        self.fifo = []          # outstanding pieces, oldest first: (region, tag)
        self.landed = []        # pieces this wave knows to have landed since the last barrier


class Block:
    """4 waves + LDS regions.  A region is a 1 KB DMA destination; `pub[region]` = tag of the published content."""

    def __init__(self):
        self.waves = [Wave(w) for w in range(NWV)]
        self.pub = {}
        self.inflight = {}       # region -> tag requested and not yet published
        self.reads = []          # (region, expected tag, wave) of the current interval
        self.issued = set()      # regions requested in the current interval
        self.errors = []
        self.interval = 0

    def dma(self, w, region, tag):
        if region in self.inflight:
            self.errors.append(f"interval {self.interval}: wave {w} requests {region} <- {tag} while {self.inflight[region]} is still in flight")
        self.waves[w].fifo.append((region, tag))
        self.inflight[region] = tag
        self.issued.add(region)

    def wait(self, w, n):
        wv = self.waves[w]
        while len(wv.fifo) > n:
            wv.landed.append(wv.fifo.pop(0))

    def read(self, w, region, tag):
        self.reads.append((region, tag, w))

    def barrier(self):
        # reads of the interval that just ended: against the content published at its start
        for region, tag, w in self.reads:
            if region in self.issued or region in self.inflight:
                self.errors.append(f"interval {self.interval}: wave {w} reads {region} (expects {tag}) while a request to it is in flight ({self.inflight.get(region)})")
            elif self.pub.get(region) != tag:
                self.errors.append(f"interval {self.interval}: wave {w} reads {region}: expects {tag}, published {self.pub.get(region)}")
        self.reads = []
        self.issued = set()
        for wv in self.waves:
            for region, tag in wv.landed:
                self.pub[region] = tag
                if self.inflight.get(region) == tag:
                    del self.inflight[region]
            wv.landed = []
        self.interval += 1


def sim_conv3x3(NK, n_items, verbose=False):
    """k_conv3x3_rec2<2,2,4>: 9 NK steps per item, ring of 4 step chunks (2 pieces per wave), input stage of 22 pieces (6 / 5 per wave)"""
    B = Block()
    MT, MW, WM = 4, 2, 2
    IS_DMA, IS_PW = 22, 6
    N_OF_S = [2, 3, 4, 4, 4, 4, 3, 2, 2]

    def issue_input(item, k, stage, i):
        for w in range(NWV):
            di = w + NWV * i
            if di < IS_DMA:
                B.dma(w, ("in", stage, di), ("in", item, k, di))

    def issue_wstep(item, k, dy, dx, slot):
        for w in range(NWV):
            for i in range(2):
                p = w + NWV * i
                B.dma(w, ("w", slot, p), ("w", item, k, dy, dx, p))

    def issue_consts(item, par):
        for w in range(3):
            B.dma(w, ("ec", par, w), ("ec", item, w))

    def load_fw(slot, item, k, dy, dx):
        for w in range(NWV):
            wm = w % WM
            for m in range(MW):
                for hl in range(2):
                    p = hl * MT + wm * MW + m
                    B.read(w, ("w", slot, p), ("w", item, k, dy, dx, p))

    def load_fx(stage, item, k):
        # every wave's fragment rows lie somewhere in the stage: model = all pieces of the stage
        for w in range(NWV):
            for di in range(IS_DMA):
                B.read(w, ("in", stage, di), ("in", item, k, di))

    def read_consts(item, par):
        for w in range(NWV):
            for c in range(3):
                B.read(w, ("ec", par, c), ("ec", item, c))

    item = 0
    for i in range(IS_PW):
        issue_input(item, 0, 0, i)
    issue_wstep(item, 0, 0, 0, 0)
    issue_wstep(item, 0, 0, 1, 1)
    issue_wstep(item, 0, 0, 2, 2)
    issue_consts(item, 0)
    par, r0 = 0, 0
    while True:
        for w in range(NWV):
            B.wait(w, 0)
        B.barrier()
        load_fw(r0, item, 0, 0, 0)
        load_fx(0, item, 0)
        has_next = item + 1 < n_items
        nxt = item + 1
        for k2 in range(0, NK, 2):
            rb = (r0 + k2) & 3
            last_trip = k2 + 2 >= NK
            for u in range(18):
                kk, s = divmod(u, 9)
                dy, dx = divmod(s, 3)
                k = k2 + kk
                load_fx(kk, item, k)                      # half-step 1 fragments (same stage, same K-step)
                tail = kk == 1 and last_trip and not has_next
                for w in range(NWV):
                    B.wait(w, 0 if tail else N_OF_S[s])
                B.barrier()
                s3, slot3 = s + 3, (rb + u + 3) & 3
                into_next = kk == 1 and last_trip
                if s3 < 9:
                    issue_wstep(item, k, s3 // 3, s3 % 3, slot3)
                elif not into_next:
                    issue_wstep(item, k + 1, (s3 - 9) // 3, (s3 - 9) % 3, slot3)
                elif has_next:
                    issue_wstep(nxt, 0, (s3 - 9) // 3, (s3 - 9) % 3, slot3)
                if s < IS_PW:
                    if not into_next:
                        issue_input(item, k + 1, (kk + 1) & 1, s)
                    elif has_next:
                        issue_input(nxt, 0, 0, s)
                if s == 6 and into_next and has_next:
                    issue_consts(nxt, par ^ 1)
                if u < 17:
                    u1 = u + 1
                    kk1, s1 = divmod(u1, 9)
                    load_fw((rb + u1) & 3, item, k2 + kk1, s1 // 3, s1 % 3)
                    load_fx(kk1, item, k2 + kk1)
                elif not last_trip:
                    load_fw((rb + 18) & 3, item, k2 + 2, 0, 0)
                    load_fx(0, item, k2 + 2)
        read_consts(item, par)
        if not has_next:
            break
        item = nxt
        par ^= 1
        r0 = (r0 + NK) & 3
    for w in range(NWV):
        B.wait(w, 0)
    B.barrier()
    return B.errors


def wrap6(x):
    return x - 6 if x >= 6 else x


def sim_upconv(NK, n_items):
    """k_upconv_rec2: 8 NK steps per item, ring of 6 step chunks, input stage of 14 pieces (4 / 3 per wave), input piece BEFORE the chunk"""
    B = Block()
    MT, MW, WM, R = 4, 2, 2, 6
    IS_DMA, IS_PW = 14, 4
    N_OF_E = [6, 7, 8, 9, 8, 7, 6, 6]

    def issue_input(item, k, stage, i):
        for w in range(NWV):
            di = w + NWV * i
            if di < IS_DMA:
                B.dma(w, ("in", stage, di), ("in", item, k, di))

    def issue_wstep(item, k, u, c, slot):
        for w in range(NWV):
            for i in range(2):
                p = w + NWV * i
                B.dma(w, ("w", slot, p), ("w", item, k, u, c, p))

    def issue_consts(item, par):
        for w in range(3):
            B.dma(w, ("ec", par, w), ("ec", item, w))

    def load_fw(slot, item, k, u, c):
        for w in range(NWV):
            wm = w % WM
            for m in range(MW):
                for hl in range(2):
                    p = hl * MT + wm * MW + m
                    B.read(w, ("w", slot, p), ("w", item, k, u, c, p))

    def load_fx(stage, item, k):
        for w in range(NWV):
            for di in range(IS_DMA):
                B.read(w, ("in", stage, di), ("in", item, k, di))

    def read_consts(item, par):
        for w in range(NWV):
            for c in range(3):
                B.read(w, ("ec", par, c), ("ec", item, c))

    item = 0
    for i in range(IS_PW):
        issue_input(item, 0, 0, i)
    for t in range(R - 1):
        issue_wstep(item, t // 8, (t % 8) // 4, t % 4, t)
    issue_consts(item, 0)
    par, r0 = 0, 0
    while True:
        for w in range(NWV):
            B.wait(w, 0)
        B.barrier()
        load_fw(r0, item, 0, 0, 0)
        load_fx(0, item, 0)
        has_next = item + 1 < n_items
        nxt = item + 1
        rb = r0
        for k2 in range(0, NK, 2):
            last_trip = k2 + 2 >= NK
            for e in range(16):
                kk, u, c, e8 = e >> 3, (e >> 2) & 1, e & 3, e & 7
                k = k2 + kk
                tail = kk == 1 and last_trip and not has_next
                for w in range(NWV):
                    B.wait(w, 0 if tail else N_OF_E[e8])
                B.barrier()
                into_next = kk == 1 and last_trip
                if e8 < IS_PW:
                    if not into_next:
                        issue_input(item, k + 1, (kk + 1) & 1, e8)
                    elif has_next:
                        issue_input(nxt, 0, 0, e8)
                e5, slot5 = e8 + (R - 1), wrap6(rb + (e + R - 1) % 6)
                if e5 < 8:
                    issue_wstep(item, k, e5 >> 2, e5 & 3, slot5)
                elif not into_next:
                    issue_wstep(item, k + 1, (e5 - 8) >> 2, (e5 - 8) & 3, slot5)
                elif has_next:
                    issue_wstep(nxt, 0, (e5 - 8) >> 2, (e5 - 8) & 3, slot5)
                if e8 == 6 and into_next and has_next:
                    issue_consts(nxt, par ^ 1)
                if e < 15 or not last_trip:
                    e1 = (e + 1) & 15
                    kk1, u1, c1 = e1 >> 3, (e1 >> 2) & 1, e1 & 3
                    k1 = k2 + kk1 if e < 15 else k2 + 2
                    load_fw(wrap6(rb + (e + 1) % 6), item, k1, u1, c1)
                    if c1 != 2:
                        load_fx(kk1, item, k1)
            rb = wrap6(rb + 16 % 6)
        read_consts(item, par)
        if not has_next:
            break
        item = nxt
        par ^= 1
        r0 = rb
    for w in range(NWV):
        B.wait(w, 0)
    B.barrier()
    return B.errors


def main() -> int:
    bad = 0
    for NK in (2, 4, 6, 8, 10, 16, 32):
        for n_items in (1, 2, 3, 5):
            for name, fn in (("conv3x3_rec2", sim_conv3x3), ("upconv_rec2", sim_upconv)):
                errs = fn(NK, n_items)
                if errs:
                    bad += 1
                    print(f"{name} NK={NK} items={n_items}: {len(errs)} protocol errors, first: {errs[0]}")
    print("rec2 DMA protocol:", "FAIL" if bad else "ok (conv3x3_rec2 + upconv_rec2, NK 2..32, 1..5 items per block)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
