#!/usr/bin/env python3
"""CPU model of the request / wait / read schedule of the dripped-epilogue record conv (probes/csrc/vae_conv_recd.hip).

The kernel requests its operands with hand-issued `global_load_lds_dwordx4` pieces and interleaves them with the slot traffic of the
previous item's epilogue (global stores) and the next item's residual rows (global loads).  gfx9 counts ALL of these in ONE in-order
counter, and the wait in front of a phase's barrier is `vmcnt(N)` with N = a LOWER BOUND of the memory instructions the previous
phase's slot issued AFTER its chunk (15 / 2 / 0) or 5 (the input pieces of a dy = 0 phase).  What has to hold:
  * chunk ph+1 (requested in phase ph-1) has landed, as seen by every wave, before the first fragment read of phase ph+1's dx = 0 step
    (issued during phase ph's dx = 2 step), and its ring slot is not re-requested before every wave has read chunk ph-2... (3 slots);
  * the input stage of K-step k+1 (requested in pieces during phase (k, 0)) has landed before the first read of K-step k+1
    (during step (k, 2, 2)) and is not requested before every wave has finished K-step k-1;
  * the two wave groups pass a phase's barrier at different points of their stream (waves 0-3 BEHIND the dx = 0 block, waves 4-7 IN
    FRONT of it) -- the same schedule must hold for both;
  * whatever the slots really issued (any count >= the lower bound, or nothing: first item, rows past the image, no fp32 / record
    output, no residual), the wait never leaves a needed DMA piece in flight.
The model re-states the kernel's loops (slot trip of 4 K-steps + plain trips of 2, pieces 0 / 1 / 2 behind the three MFMA blocks of a
phase) on the adversarial memory model of tools/rec2_protocol_sim.py: a piece is only known to have landed once its wave executed a
covering vmcnt wait and a barrier followed; reading a destination with a request in flight, or with other content than expected, is an
error.  usage: python tools/recd_protocol_sim.py   (exit code 0 = every configuration passes; also run by tests/test_recd_protocol.py)"""
from __future__ import annotations

import importlib.util
import itertools
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("rec2_protocol_sim", os.path.join(os.path.dirname(_HERE), "tools", "rec2_protocol_sim.py"))
_r2 = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_r2)

NW = 8            # waves of the block
D_TK = 4          # K-steps of the slot trip
IN_PIECES = 40    # wave-instructions of one input stage: piece di = wave + 8 i, i < 5
W_PIECES = 12     # wave-instructions of one chunk: waves 0-7 piece w, waves 0-3 also piece w + 8
# lower bounds the kernel assumes per slot kind when the slot issued its traffic (see phase_wait in the .hip file)
BIG, SMALL = 15, 2


def slot_pos(p):
    return -1 if p % 3 == 0 else 2 * (p // 3) + p % 3 - 1


def slot_kind(p):
    return 0 if slot_pos(p) < 0 else 1 + slot_pos(p) % 2


class Block(_r2.Block):
    def __init__(self):
        super().__init__()
        self.waves = [_r2.Wave(w) for w in range(NW)]

    def other(self, w, n):
        """n memory instructions that are not DMA (slot stores / residual loads): they only occupy places in the wave's in-order counter"""
        for _ in range(n):
            self.waves[w].fifo.append((None, None))

    def wait(self, w, n):
        wv = self.waves[w]
        while len(wv.fifo) > n:
            r, t = wv.fifo.pop(0)
            if r is not None:
                wv.landed.append((r, t))


def sim(NK, n_items, flags, slack=0, verbose=False):
    """flags = (y32, yrec, res, rows_ok): what the slots issue.  slack: extra memory instructions a slot may issue beyond its lower bound
    (border cells).  Returns the list of protocol errors."""
    y32, yrec, res, rows_ok = flags
    B = Block()
    nph = NK * 3

    def chunk_pieces(w):
        return [w] + ([w + 8] if w < 4 else [])

    def issue_weights(w, item, ph, ring):
        for p in chunk_pieces(w):
            B.dma(w, ("w", ring, p), ("w", item, ph, p))

    def issue_input(w, item, k, stage, i0, i1):
        for i in range(i0, i1):
            B.dma(w, ("in", stage, w + 8 * i), ("in", item, k, w + 8 * i))

    def slot_counts(sp, piece, drip, has_next):
        """memory instructions piece `piece` of slot-trip phase sp really issues (upper end: + slack on the store pieces)"""
        sk = slot_kind(sp) if sp >= 0 else 0
        ok = drip and rows_ok
        if sk == 1:
            if piece == 0:
                return 16 if (ok and y32) else 0
            if piece == 2:
                return (2 + slack) if (ok and yrec) else 0
        if sk == 2:
            if piece == 1:
                return (2 + slack) if (ok and yrec) else 0
            if piece == 2:
                return 16 if (res and has_next) else 0
        return 0

    def wait_count(psp, dy, k, drip, has_next):
        """the kernel's phase_wait: N of the vmcnt in front of the barrier that follows a phase whose slot-trip index is psp (or -1)"""
        pk = slot_kind(psp) if psp >= 0 else 0
        if pk:
            yok = drip and rows_ok
            big = (yok and y32) if pk == 1 else (res and has_next)
            if big:
                return BIG
            if yok and yrec:
                return SMALL
        return 5 if (dy == 1 and k + 1 < NK) else 0

    # per-wave program position: the two groups differ only in WHERE the barrier of a phase sits; the simulation advances barrier
    # interval by barrier interval, each wave executing everything between its barrier instances.
    def wave_stream(w, item, drip, has_next):
        """generator of events of one item for wave w: ('bar', wait_n) | ('read_w', ring, ph) | ('read_in', stage, k) | ('dma_*' ...)"""
        ev = []
        def piece(sp, qdy, kq, j, can_be_last):
            qph = kq * 3 + qdy
            if j == 0 and qph + 2 < nph:
                ev.append(("wgt", item, qph + 2, (qdy + 2) % 3))
            n = slot_counts(sp, j, drip, has_next)
            if n:
                ev.append(("other", n))
            if qdy == 0 and kq + 1 < NK:
                ev.append(("inp", item, kq + 1, (kq + 1) & 1, 2 * j, 5 if j == 2 else 2 * j + 2))
            if j == 0 and qdy == 2 and can_be_last and kq + 1 == NK and has_next:
                ev.append(("inp", item + 1, 0, 0, 0, 5))
                ev.append(("wgt", item + 1, 0, 0))
                ev.append(("wgt", item + 1, 1, 1))
                ev.append(("other", 1))          # the constants piece of some waves (a DMA into its own LDS buffer: modelled as a counter place)
        trips = [(0, 9 * D_TK, True)] + [(k2, 18, False) for k2 in range(D_TK, NK, 2)]
        first_plain = True
        for kb, T, SLOTS in trips:
            for t in range(T):
                kk, dy, dx, pl = t // 9, (t // 3) % 3, t % 3, t // 3
                k = kb + kk
                if SLOTS:
                    psp = pl - 1
                else:
                    psp = (3 * D_TK - 1) if (pl == 0 and kb == D_TK) else -1
                if dx == 0 and w >= 4:
                    ev.append(("bar", wait_count(psp, dy, k, drip, has_next)))
                # fragments of the NEXT step are read at the start of this step
                more = (not SLOTS and kb + 2 < NK) or SLOTS
                t1 = t + 1
                if t1 < T:
                    kk1, dy1 = t1 // 9, (t1 // 3) % 3
                    ev.append(("read", item, (kb + kk1) * 3 + dy1, dy1, kb + kk1, (kb + kk1) & 1))
                elif more:
                    ev.append(("read", item, (kb + T // 9) * 3, 0, kb + T // 9, 0))
                if dx == 0 and w < 4:
                    ev.append(("bar", wait_count(psp, dy, k, drip, has_next)))
                piece(pl if SLOTS else -1, dy, k, dx, (not SLOTS) and kk == 1)
        return ev

    # build every wave's event list for all items, with the item top (vmcnt(0) + barrier + first fragment read) in front of each
    streams = []
    for w in range(NW):
        ev = []
        for it in range(n_items):
            if it == 0:                    # the prologue's requests
                ev.append(("inp", 0, 0, 0, 0, 5))
                ev.append(("wgt", 0, 0, 0))
                ev.append(("wgt", 0, 1, 1))
                ev.append(("other", 1 + (16 if res else 0)))
            ev.append(("bar", 0))
            ev.append(("read", it, 0, 0, 0, 0))
            ev += wave_stream(w, it, drip=it > 0, has_next=it + 1 < n_items)
        ev.append(("end",))
        streams.append(ev)
    pos = [0] * NW
    while True:
        done = 0
        for w in range(NW):
            ev = streams[w]
            while True:
                e = ev[pos[w]]
                if e[0] == "end":
                    done += 1
                    break
                pos[w] += 1
                if e[0] == "bar":
                    B.wait(w, e[1])
                    break
                if e[0] == "wgt":
                    issue_weights(w, e[1], e[2], e[3])
                elif e[0] == "inp":
                    issue_input(w, e[1], e[2], e[3], e[4], e[5])
                elif e[0] == "other":
                    B.other(w, e[1])
                elif e[0] == "read":
                    _, item, ph, ring, k, stage = e
                    for p in range(W_PIECES):
                        B.read(w, ("w", ring, p), ("w", item, ph, p))
                    for p in range(IN_PIECES):
                        B.read(w, ("in", stage, p), ("in", item, k, p))
        if done == NW:
            break
        assert done == 0, "the waves execute different numbers of barriers"
        B.barrier()
    if verbose:
        for e in B.errors[:10]:
            print(e)
    return B.errors


def main():
    bad = 0
    for NK in (8, 10, 16, 32):
        for items in (1, 2, 3):
            for flags in itertools.product((False, True), repeat=4):
                if not (flags[0] or flags[1]):
                    continue
                for slack in (0, 6):
                    errs = sim(NK, items, flags, slack)
                    if errs:
                        bad += 1
                        print(f"NK={NK} items={items} flags(y32, yrec, res, rows_ok)={flags} slack={slack}: {len(errs)} errors, first: {errs[0]}")
    print("recd protocol:", "FAIL" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
