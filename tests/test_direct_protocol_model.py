"""CPU: the push / flow-control protocol of the direct face carrier (tmlqcd_amd/csrc/hopping_split.inc `launch_direct`, DESIGN.md
section 7a) as a small executable MODEL -- ranks of a ring, each with a "main stream" and a "comm stream" thread, double-buffered receive
slots, arrival words, push numbers -- driven through random programs of stencils (packed now / chained on a push ahead / pushing ahead /
pushes ahead that nobody consumes) with random delays everywhere.  Checked on every step: a consumer reads exactly the faces it expects
(right push, right field, not overwritten while it reads), a slot is never overwritten before its previous push was consumed or
abandoned, and the arrival words only ever grow.

This is how the rule "a rank's push q + 1 is published only after its own push q" (HopArgs::pub_after) is pinned on the CPU: the same
model WITHOUT that rule must fail -- it is the bug that showed between two real processes on the GPU (profiles/r04_split_forms.md)."""
import random
import threading
import time

import pytest


class Violation(Exception):
    pass


class Rank:
    def __init__(self, r, n):
        self.r, self.n = r, n
        self.arr = [0, 0]                                  # arrival words: [0] written by the up neighbour, [1] by the down neighbour
        self.slot = [[None, None], [None, None]]           # [push parity][0: from up, 1: from down] = (push number, field tag)
        self.consumed = [[True, True], [True, True]]       # the push in that slot has been read by this rank (or nothing is in it yet)
        self.push_seq = 0
        self.pack_done = 0                                 # local word behind the pack kernel's publication (sync_flags[6])
        self.start = 0                                     # "main stream reached stencil n" (sync_flags[0])
        self.lock = threading.Lock()


def make_program(rng, nops):
    """[(chained, feed)]: chained only directly behind a feeding stencil; the last stencil of the program does not push ahead."""
    prog, prev_feed = [], False
    for i in range(nops):
        chained = prev_feed and rng.random() < 0.7        # (a push ahead that is NOT taken up is an abandoned push)
        feed = i + 1 < nops and rng.random() < 0.6
        prog.append((chained, feed))
        prev_feed = feed
    return prog


def pushes_of(prog):
    """push number -> (stencil that consumes it or None = abandoned, tag)"""
    out, seq, ahead = {}, 0, None
    for s, (chained, feed) in enumerate(prog):
        if chained:
            q = ahead
            out[q] = (s, out[q][1])
        else:
            seq += 1
            out[seq] = (s, ("in", s))
        if feed:
            seq += 1
            ahead = seq
            out[seq] = (None, ("out", s))
    return out


def run_model(n, prog, seed, pub_after=True, max_delay=0.0015):
    ranks = [Rank(r, n) for r in range(n)]
    table = pushes_of(prog)
    errors, stop = [], threading.Event()

    def nap(rng):
        if rng.random() < 0.5:
            time.sleep(rng.random() * max_delay)

    def wait(cond, what):
        t0 = time.time()
        while not cond():
            if stop.is_set():
                raise Violation("stopped")
            if time.time() - t0 > 5.0:
                raise Violation("deadlock: " + what)
            time.sleep(0.00005)

    def publish(me, rng, q, tag):
        """store the faces of push q into both neighbours' slots, then their words"""
        up, dn = ranks[(me.r + 1) % n], ranks[(me.r - 1) % n]
        for peer, side in ((up, 1), (dn, 0)):              # the up neighbour receives our faces "from down" (index 1), and vice versa
            with peer.lock:
                old = peer.slot[q & 1][side]
                if old is not None and not peer.consumed[q & 1][side] and table[old[0]][0] is not None:
                    raise Violation("rank %d overwrites push %d of rank %d's slot before it was read (writing %d)" % (me.r, old[0], peer.r, q))
                peer.slot[q & 1][side] = (q, tag)
                peer.consumed[q & 1][side] = False
            nap(rng)
        return up, dn

    def set_words(me, up, dn, q):
        for peer, side in ((up, 1), (dn, 0)):
            with peer.lock:
                if peer.arr[side] >= q:
                    raise Violation("rank %d: arrival word of rank %d runs backwards (%d -> %d)" % (me.r, peer.r, peer.arr[side], q))
                peer.arr[side] = q

    def comm_stream(me, jobs, rng):                        # the pack kernels, in order
        try:
            for (s, q) in jobs:
                wait(lambda: me.start >= s + 1, "pack of stencil %d waits for its start flag" % s)
                wait(lambda: me.arr[0] >= q - 1 and me.arr[1] >= q - 1, "pack of push %d: flow control" % q)
                nap(rng)
                up, dn = publish(me, rng, q, ("in", s))
                nap(rng)
                set_words(me, up, dn, q)
                me.pack_done = q
        except Violation as e:
            errors.append(str(e)); stop.set()

    def main_stream(me, rng):
        try:
            ahead = None
            for s, (chained, feed) in enumerate(prog):
                nap(rng)
                if chained:
                    q, packed = ahead, False
                else:
                    me.push_seq += 1
                    q, packed = me.push_seq, True
                    me.start = s + 1                       # (the stencil kernel's first thread)
                expect = table[q][1]
                for side in (0, 1):                        # boundary waves: wait for the neighbour's word, read its faces
                    wait(lambda: me.arr[side] >= q, "stencil %d waits for push %d" % (s, q))
                    nap(rng)
                    with me.lock:
                        got = me.slot[q & 1][side]
                    if got != (q, expect):
                        raise Violation("rank %d stencil %d read %r, expected push %d %r" % (me.r, s, got, q, expect))
                    nap(rng)
                    with me.lock:
                        if me.slot[q & 1][side] != (q, expect):
                            raise Violation("rank %d stencil %d: its faces were overwritten while it read them" % (me.r, s))
                        me.consumed[q & 1][side] = True
                if feed:                                   # the boundary waves push the faces of their output ahead
                    me.push_seq += 1
                    q2 = me.push_seq
                    up, dn = publish(me, rng, q2, ("out", s))
                    if pub_after and packed:
                        wait(lambda: me.pack_done >= q, "publication of push %d waits for this rank's own pack kernel" % q2)
                    set_words(me, up, dn, q2)
                    ahead = q2
        except Violation as e:
            errors.append(str(e)); stop.set()

    threads = []
    for me in ranks:
        seq, jobs = 0, []
        for s, (chained, feed) in enumerate(prog):
            if not chained:
                seq += 1
                jobs.append((s, seq))
            if feed:
                seq += 1
        threads.append(threading.Thread(target=comm_stream, args=(me, jobs, random.Random(seed * 1000 + 2 * me.r))))
        threads.append(threading.Thread(target=main_stream, args=(me, random.Random(seed * 1000 + 2 * me.r + 1))))
    for t in threads:
        t.start()
    for t in threads:
        t.join(30)
    return errors


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_protocol_model_random_programs(n):
    for seed in range(12):
        prog = make_program(random.Random(100 * n + seed), 14)
        errs = run_model(n, prog, seed)
        assert not errs, (n, seed, prog, errs[:2])


def test_without_the_publication_order_rule_the_model_fails():
    """The bug of profiles/r04_split_forms.md: a stencil's boundary waves publish push q + 1 while the same rank's pack kernel has
    not yet published q.  The model must see it (words running backwards, or faces read before they were written)."""
    seen = 0
    for seed in range(40):
        prog = [(False, True), (True, False)] * 6          # the benchmark loop: packed + pushing ahead, then chained
        if run_model(2, prog, seed, pub_after=False):
            seen += 1
    assert seen > 0, "the model without pub_after never failed: it does not exercise the race it is meant to pin"
