"""Randomised ordering check of the k-loops of bq_kernel<2,4> / <4,2>, bhq_kernel, bwq_kernel<2,4> / <4,2> (the shipped large-tile kernels) and bhq32_kernel (the experimental
BK = 32 variant; csrc/dpig_conv_bf16_q.hip, scripts/ubench/bhq32_probe.hip): eight waves run the kernel's instruction stream (fragment reads, LDS-DMA issues, counted vmcnt waits, the two barriers per k-tile, groups staggered by
one barrier) under a random scheduler; every DMA piece lands at a random later time, in issue order per wave (the only guarantee
vmcnt gives).  A fragment read must find the piece of ITS k-tile / chunk in the LDS region it reads, and never a region with a DMA
still in flight: that is the RAW / WAR argument of the kernels' headers, executed.     python scripts/ubench/simulate_kloop_hazards.py"""
import random
import sys

NP = 13
WAIT = [3, 4, 5, 5, 5, 5, 5, 4, 3]


def program(wave, nch):
    """Instruction list of one wave: ('dma', region, tag) / ('wait', n) / ('bar',) / ('read', [regions], tag_kind, tag)."""
    grp = wave >> 2
    wr, wc = wave >> 1, wave & 1
    nkt = 9 * nch
    prog = []

    def issueH(t, chunk):
        idp = 8 * t + wave
        region = ("H", chunk & 1, idp // NP, idp % NP) if idp < 4 * NP else ("PAD", wave)
        prog.append(("dma", region, chunk if chunk < nch else "dead"))

    def issueB(t):
        prog.append(("dma", ("B", t & 3, wave), t if t < nkt else "dead"))

    for t in range(7):
        issueH(t, 0)
    for t in range(3):
        issueB(t)
    prog.append(("wait", 2))
    prog.append(("bar",))
    if grp == 1:
        prog.append(("bar",))
    t = 0
    for c in range(nch):
        for tap in range(9):
            prog.append(("read", [("B", t & 3, 4 * wc + k) for k in range(4)], t))
            prog.append(("read", [("H", c & 1, wr, q) for q in range(NP)], c))
            issueB(t + 3)
            if tap < 7:
                issueH(tap, c + 1)
            prog.append(("wait", WAIT[tap]))
            prog.append(("bar",))
            prog.append(("mfma",))
            prog.append(("bar",))
            t += 1
    if grp == 0:
        prog.append(("bar",))
    prog.append(("wait", 0))
    return prog


def program_bhq(wave, nch, pb_wait=4, slack=False):
    """bhq_kernel: waves as 2 (pixel rows) x 4 (channel columns); per k-tile two phases PA / PB of 16 MFMAs; halo of a 64-channel chunk =
    2 wave rows x 23 pieces, fetched as piece 8 t + wave (< 46) in PA of taps 0..5; filter tile = 4 wave columns x 2 units x 4 pieces in two
    slots, tile t + 2 issued in PB(t) into the slot tile t was read from; PB waits with 4 pieces in flight."""
    grp = wave >> 2
    wr, wc = wave >> 2, wave & 3
    nkt = 9 * nch
    prog = []

    def issueH(t, chunk):
        idp = 8 * t + wave
        if idp < 46:
            prog.append(("dma", ("H", chunk & 1, idp // 23, idp % 23), chunk if chunk < nch else "dead"))

    def issueB(t, h):                                        # unit h of filter tile t: this wave's two pieces
        for j in range(2):
            prog.append(("dma", ("B", t & 1, (wave >> 2) + 2 * j, h, wave & 3), t if t < nkt else "dead"))

    for t in range(6):
        issueH(t, 0)
    issueB(0, 0); issueB(0, 1); issueB(1, 0); issueB(1, 1)
    prog.append(("wait", 4))
    prog.append(("bar",))
    if grp == 1:
        prog.append(("bar",))
    t = 0
    for c in range(nch):
        for tap in range(9):
            prog.append(("read", [("B", t & 1, wc, h, k) for h in range(2) for k in range(4)], t))
            prog.append(("read", [("H", c & 1, wr, q) for q in range(23)], c))                  # rows of M-half 0
            if tap < 6:
                issueH(tap, c + 1)
            prog.append(("bar",)); prog.append(("mfma",)); prog.append(("bar",))
            prog.append(("read", [("H", c & 1, wr, q) for q in range(23)], c))                  # rows of M-half 1
            issueB(t + 2, 0); issueB(t + 2, 1)
            # the <true> build (DPIG_BF16_QH=2): the halo piece issued in this k-tile's PA may stay in flight
            prog.append(("wait", pb_wait + (1 if slack and (tap < 5 or (tap == 5 and wave < 6)) else 0)))
            prog.append(("bar",)); prog.append(("mfma",)); prog.append(("bar",))
            t += 1
    if grp == 0:
        prog.append(("bar",))
    prog.append(("wait", 0))
    return prog


def program_bq(wave, nkt, WM=2, WN=4, relax=0):
    """bq_kernel<WM, WN> (256 x 256 / 512 x 128, operands gathered per tap): A = [wave row][2 slots][units UA0 / UA1 of 64 rows, 8 pieces each],
    B = [wave column][2 slots][units UB0 / UB1 of 32 columns, 4 pieces each]; PA(t) stages UA1(t + 1) into the other slot, PB(t) stages
    UA0, UB0, UB1 of (t + 2) into the current one; both phases wait with 2 (NA + NB) pieces in flight.  `nkt` k-tiles."""
    NA, NB = WM, WN // 2
    VMC = 2 * (NA + NB) + relax
    grp = wave >> 2
    wr, wc = wave // WN, wave % WN
    prog = []

    def issueA(t, h):
        for j in range(NA):
            prog.append(("dma", ("A", t & 1, j, h, wave), t if t < nkt else "dead"))

    def issueB(t, h):
        for j in range(NB):
            prog.append(("dma", ("B", t & 1, (wave >> 2) + 2 * j, h, wave & 3), t if t < nkt else "dead"))

    issueA(0, 0); issueB(0, 0); issueB(0, 1); issueA(0, 1)
    issueA(1, 0); issueB(1, 0); issueB(1, 1)
    prog.append(("wait", NA + 2 * NB))
    prog.append(("bar",))
    if grp == 1:
        prog.append(("bar",))
    for t in range(nkt):
        prog.append(("read", [("B", t & 1, wc, h, k) for h in range(2) for k in range(4)], t))
        prog.append(("read", [("A", t & 1, wr, 0, k) for k in range(8)], t))
        issueA(t + 1, 1)
        prog.append(("wait", VMC))
        prog.append(("bar",)); prog.append(("mfma",)); prog.append(("bar",))
        prog.append(("read", [("A", t & 1, wr, 1, k) for k in range(8)], t))
        issueA(t + 2, 0); issueB(t + 2, 0); issueB(t + 2, 1)
        prog.append(("wait", VMC))
        prog.append(("bar",)); prog.append(("mfma",)); prog.append(("bar",))
    if grp == 0:
        prog.append(("bar",))
    prog.append(("wait", 0))
    return prog


def program_bwq(wave, nkt, WM=2, WN=4, relax=0):
    """bwq_kernel<WM, WN> (wgrad, csrc/dpig_conv_bf16_wq.hip): k = pixels, 64 per k-tile; units are cut along k: X01 = pixels 0..31 and
    X23 = pixels 32..63 of every x block (WM items) and dy block (WN / 2 blocks of 128 co), 8 pieces of 4 pixel rows per block and unit.
    PA(t) reads X01(t) and stages X23(t + 1) into the other slot, PB(t) reads X23(t) and stages X01(t + 2) into its own."""
    NA, NB = WM, WN // 2
    VMC = 2 * (NA + NB) + relax
    grp = wave >> 2
    wr, wc = wave // WN, wave % WN
    prog = []

    def issue(t, u):
        tag = t if t < nkt else "dead"
        for j in range(NA):
            prog.append(("dma", ("X", t & 1, j, u, wave), tag))
        for j in range(NB):
            prog.append(("dma", ("Y", t & 1, j, u, wave), tag))

    issue(0, 0); issue(0, 1); issue(1, 0)
    prog.append(("wait", NA + NB))
    prog.append(("bar",))
    if grp == 1:
        prog.append(("bar",))
    for t in range(nkt):
        for u in range(2):
            prog.append(("read", [("X", t & 1, wr, u, k) for k in range(8)] + [("Y", t & 1, wc // 2, u, k) for k in range(8)], t))
            issue(t + 1, 1) if u == 0 else issue(t + 2, 0)
            prog.append(("wait", VMC))
            prog.append(("bar",)); prog.append(("mfma",)); prog.append(("bar",))
    if grp == 0:
        prog.append(("bar",))
    prog.append(("wait", 0))
    return prog


def run(nch, seed, lazy=0.5, make=None):
    rng = random.Random(seed)
    progs = [(make or program)(w, nch) for w in range(8)]
    pc = [0] * 8
    fifo = [[] for _ in range(8)]          # per wave: DMA ops in flight, in issue order
    lds = {}                               # region -> tag of the data it holds
    pending = {}                           # region -> number of DMA ops in flight to it
    at_bar = [False] * 8
    steps = 0
    while any(pc[w] < len(progs[w]) for w in range(8)) or any(fifo):
        steps += 1
        choices = []
        for w in range(8):
            if fifo[w]:
                choices.append(("land", w))
            if pc[w] < len(progs[w]) and not at_bar[w]:
                ins = progs[w][pc[w]]
                if ins[0] == "wait" and len(fifo[w]) > ins[1]:
                    continue                                   # blocked on vmcnt
                choices.append(("exec", w))
        if not choices:
            if all(at_bar[w] or pc[w] >= len(progs[w]) for w in range(8)) and any(at_bar):
                if not all(at_bar[w] for w in range(8) if pc[w] < len(progs[w])) or any(pc[w] >= len(progs[w]) for w in range(8)):
                    raise AssertionError("barrier deadlock")
            raise AssertionError("stuck")
        # landings are made rare (`lazy`), so that pieces stay in flight as long as the waits allow: the adversarial memory system
        execs = [ch for ch in choices if ch[0] == "exec"]
        lands = [ch for ch in choices if ch[0] == "land"]
        if execs and (not lands or rng.random() > lazy):
            kind, w = rng.choice(execs)
        else:
            kind, w = rng.choice(lands)
        if kind == "land":
            region, tag = fifo[w].pop(0)
            pending[region] -= 1
            lds[region] = tag
            continue
        ins = progs[w][pc[w]]
        if ins[0] == "dma":
            fifo[w].append((ins[1], ins[2]))
            pending[ins[1]] = pending.get(ins[1], 0) + 1
        elif ins[0] == "read":
            for region in ins[1]:
                assert pending.get(region, 0) == 0, ("read of a region with a DMA in flight", w, region, ins[2])
                assert lds.get(region) == ins[2], ("stale / overwritten data", w, region, lds.get(region), "wanted", ins[2])
        elif ins[0] == "bar":
            at_bar[w] = True
            if all(at_bar):
                at_bar = [False] * 8
                for v in range(8):
                    pc[v] += 1
                continue
            continue                                           # pc advances when the barrier releases
        pc[w] += 1
    return steps


if __name__ == "__main__":
    n = 0
    for nch in (1, 2, 3, 4):
        for seed in range(150):
            run(nch, seed, lazy=(0.98, 0.5, 0.1, 0.02)[seed % 4])
            n += 1
    print("bhq32 k-loop: %d random schedules (1..4 chunks), every fragment read saw the data of its own k-tile / chunk with no DMA in flight" % n)
    n = 0
    for nch in (1, 2, 3, 4):
        for seed in range(150):
            run(nch, seed, lazy=(0.98, 0.5, 0.1, 0.02)[seed % 4], make=program_bhq)
            n += 1
    for seed in range(200):
        run(3, seed, lazy=(0.98, 0.1, 0.02)[seed % 3], make=lambda w, c: program_bhq(w, c, slack=True))
        n += 1
    print("bhq   k-loop: %d random schedules (1..4 chunks, incl. the relaxed-wait build), every fragment read saw the data of its own k-tile / chunk with no DMA in flight" % n)
    try:
        for seed in range(300):
            run(3, seed, lazy=0.02, make=lambda w, c: program_bhq(w, c, pb_wait=5))
        print("WARNING: bhq's relaxed wait was not caught"); sys.exit(1)
    except AssertionError as e:
        print("bhq: a PB wait relaxed by one piece is caught as expected:", e.args[0][0])
    n = 0
    for (WM, WN) in ((2, 4), (4, 2)):
        for nkt in (2, 3, 9, 18):
            for seed in range(60):
                run(nkt, seed, lazy=(0.98, 0.5, 0.1, 0.02)[seed % 4], make=lambda w, k: program_bq(w, k, WM, WN))
                n += 1
    print("bq<2,4> / bq<4,2> k-loops: %d random schedules (2..18 k-tiles), every fragment read saw its own k-tile's data with no DMA in flight" % n)
    try:
        for seed in range(300):
            run(9, seed, lazy=0.02, make=lambda w, k: program_bq(w, k, 2, 4, relax=1))
        print("WARNING: bq's relaxed wait was not caught"); sys.exit(1)
    except AssertionError as e:
        print("bq: a wait relaxed by one piece is caught as expected:", e.args[0][0])
    n = 0
    for (WM, WN) in ((2, 4), (4, 2)):
        for nkt in (2, 3, 16):
            for seed in range(60):
                run(nkt, seed, lazy=(0.98, 0.5, 0.1, 0.02)[seed % 4], make=lambda w, k: program_bwq(w, k, WM, WN))
                n += 1
    print("bwq<2,4> / bwq<4,2> (wgrad) k-loops: %d random schedules, every fragment read saw its own k-tile's data with no DMA in flight" % n)
    try:
        for seed in range(300):
            run(9, seed, lazy=0.02, make=lambda w, k: program_bwq(w, k, 2, 4, relax=1))
        print("WARNING: bwq's relaxed wait was not caught"); sys.exit(1)
    except AssertionError as e:
        print("bwq: a wait relaxed by one piece is caught as expected:", e.args[0][0])
    # the check has teeth: a too-lax wait (one more piece allowed in flight) must be caught
    WAIT[4] += 1
    try:
        for seed in range(300):
            run(3, seed, lazy=0.02)
        print("WARNING: the relaxed wait was not caught"); sys.exit(1)
    except AssertionError as e:
        print("relaxed wait caught as expected:", e.args[0][0])
