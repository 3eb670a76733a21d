"""A small assembler front end for hand-scheduled gfx950 (CDNA4) kernels.

The sweeps of cpecan are written as Python programs that EMIT assembly: registers are named by the program, the
emitter keeps the books the hardware leaves to the programmer --
  * s_waitcnt: every LDS read / global load is tracked in issue order (the counters are in-order queues), and a wait
    with the exact count is emitted in front of the first instruction that touches a register a pending load
    writes;
  * the manually inserted wait states of the ISA that this code can run into: a DPP instruction reading a VGPR a
    VALU instruction wrote less than two instructions ago, and a VALU write of the data registers of a store wider
    than 64 bits in the two instructions after it;
  * temporaries: a pool of VGPR pairs handed out and taken back explicitly.
Nothing here knows about the alignment problem; see gen_sweeps.py.
"""


class V:
    """n consecutive VGPRs starting at i (64-bit and wider operands are even-aligned on gfx90a+)."""
    __slots__ = ("i", "n")

    def __init__(self, i, n=1):
        self.i, self.n = i, n

    def __str__(self):
        return "v%d" % self.i if self.n == 1 else "v[%d:%d]" % (self.i, self.i + self.n - 1)

    def regs(self):
        return range(self.i, self.i + self.n)

    def sub(self, k, n=1):
        assert k + n <= self.n
        return V(self.i + k, n)

    @property
    def lo(self):
        return V(self.i, 1)

    @property
    def hi(self):
        return V(self.i + 1, 1)


class S:
    __slots__ = ("i", "n")

    def __init__(self, i, n=1):
        self.i, self.n = i, n
        assert n == 1 or i % 2 == 0, "SGPR tuples are even-aligned"

    def __str__(self):
        return "s%d" % self.i if self.n == 1 else "s[%d:%d]" % (self.i, self.i + self.n - 1)

    def sub(self, k, n=1):
        return S(self.i + k, n)

    @property
    def lo(self):
        return S(self.i, 1)

    @property
    def hi(self):
        return S(self.i + 1, 1)


class Neg:
    """-operand (VOP3 input modifier)."""
    def __init__(self, x):
        self.x = x

    def __str__(self):
        return "-" + str(self.x)


def vregs_of(op):
    if isinstance(op, Neg):
        op = op.x
    if isinstance(op, V):
        return set(op.regs())
    return set()


class Emitter:
    def __init__(self):
        self.lines = []
        self.n = 0                 # instructions emitted (wait states are counted in instructions)
        self.lgkm = []             # pending LDS operations, oldest first: set of VGPRs each will write
        self.vm = []               # pending vector-memory operations, oldest first
        self.valu_w = {}           # VGPR -> index of the last VALU instruction that wrote it
        self.store_r = {}          # VGPR -> index of the last wide store that reads it as data
        self.sgpr_w = {}           # SGPR pair (its first register) -> index of the last VALU instruction that wrote it
        self.label_n = 0
        self.stats = {}

    # ------------------------------------------------------------------ text
    def raw(self, text):
        self.lines.append(text)

    def comment(self, text):
        self.lines.append("\t; " + text)

    def label(self, name):
        self.lines.append(name + ":")

    _labels = [0]

    def newlabel(self, stem):
        Emitter._labels[0] += 1
        return ".L_%s_%d" % (stem, Emitter._labels[0])

    def _ins(self, text, kind):
        self.lines.append("\t" + text)
        self.n += 1
        self.stats[kind] = self.stats.get(kind, 0) + 1

    def nop(self, k=0):
        self._ins("s_nop %d" % k, "nop")
        self.n += k

    # ------------------------------------------------------------------ counters
    def _need(self, regs):
        """Waits for every pending load that writes one of regs."""
        for q, name in ((self.lgkm, "lgkmcnt"), (self.vm, "vmcnt")):
            last = -1
            for k, dst in enumerate(q):
                if dst and (dst & regs):
                    last = k
            if last >= 0:
                left = len(q) - 1 - last
                self._ins("s_waitcnt %s(%d)" % (name, left), "wait")
                del q[: last + 1]

    def need(self, *vs):
        """One wait for several results that are about to be used one after the other (each wait is an instruction)."""
        regs = set()
        for v in vs:
            regs |= vregs_of(v)
        self._need(regs)

    def wait_lgkm(self, left=0):
        if len(self.lgkm) > left:
            self._ins("s_waitcnt lgkmcnt(%d)" % left, "wait")
            del self.lgkm[: len(self.lgkm) - left]

    def wait_vm(self, left=0):
        if len(self.vm) > left:
            self._ins("s_waitcnt vmcnt(%d)" % left, "wait")
            del self.vm[: len(self.vm) - left]

    def raw_wait_vm(self, left):
        """a wait for loads another copy of the loop issued: the queue here does not know them"""
        self._ins("s_waitcnt vmcnt(%d)" % left, "wait")

    def drain_lgkm(self):
        """(for what the queues here do not see: s_memtime and the like)"""
        self._ins("s_waitcnt lgkmcnt(0)", "wait")
        self.lgkm = []

    def wait_all(self):
        """An unconditional drain (entry of code reached from several places)."""
        self._ins("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait")
        self.lgkm, self.vm = [], []

    def forget(self):
        """Control flow joins here: nothing is known about pending operations (the caller drains or knows better)."""
        self.lgkm, self.vm = [], []
        self.valu_w, self.store_r, self.sgpr_w = {}, {}, {}

    # ------------------------------------------------------------------ instruction classes
    def _touch(self, reads, writes):
        self._need(reads | writes)

    def valu(self, op, dst, *srcs, dpp=None, extra=""):
        """A VALU instruction; dst may be a V, an S (compare to SGPRs), 'vcc', or a tuple of them."""
        dsts = dst if isinstance(dst, tuple) else (dst,)
        writes, reads = set(), set()
        for d in dsts:
            writes |= vregs_of(d)
        for s in srcs:
            reads |= vregs_of(s)
        self._touch(reads, writes)
        if dpp:
            worst = max([self.valu_w.get(r, -10) for r in reads] or [-10])
            gap = self.n - worst - 1  # instructions between the writer and this one
            if gap < 2:
                self.nop(1 - gap)
        worst = max([self.store_r.get(r, -10) for r in writes] or [-10])
        gap = self.n - worst - 1
        if gap < 2:
            self.nop(1 - gap)
        # a VALU instruction that reads an SGPR as an explicit operand needs two wait states after the VALU instruction
        # that wrote it (gfx940 and later; vcc read implicitly by the 32-bit encodings is interlocked by the hardware)
        worst = max([self.sgpr_w.get(q, -10) for x in srcs if isinstance(x, S) for q in range(x.i, x.i + x.n)] or [-10])
        gap = self.n - worst - 1
        if gap < 2:
            self.nop(1 - gap)
        text = "%s %s" % (op, ", ".join(str(x) for x in dsts + srcs))
        if dpp:
            text += " " + dpp
        if extra:
            text += " " + extra
        self._ins(text, "valu")
        for r in writes:
            self.valu_w[r] = self.n - 1
        for d in dsts:
            if isinstance(d, S):
                for q in range(d.i, d.i + d.n):
                    self.sgpr_w[q] = self.n - 1

    def salu(self, op, *ops):
        self._ins("%s %s" % (op, ", ".join(str(x) for x in ops)) if ops else op, "salu")

    def branch(self, op, target):
        self._ins("%s %s" % (op, target), "branch")

    def ds_read(self, bits, dst, addr, offset=0):
        self._touch(vregs_of(addr), vregs_of(dst))
        op = {32: "ds_read_b32", 64: "ds_read_b64", 128: "ds_read_b128"}[bits]
        self._ins("%s %s, %s%s" % (op, dst, addr, " offset:%d" % offset if offset else ""), "lds")
        self.lgkm.append(vregs_of(dst))

    def ds_write(self, bits, addr, data, offset=0):
        self._touch(vregs_of(addr) | vregs_of(data), set())
        op = {32: "ds_write_b32", 64: "ds_write_b64", 128: "ds_write_b128"}[bits]
        self._ins("%s %s, %s%s" % (op, addr, data, " offset:%d" % offset if offset else ""), "lds")
        self.lgkm.append(set())
        if bits > 64:
            for r in vregs_of(data):
                self.store_r[r] = self.n - 1

    mods = ""                                # cache policy modifiers of the loads and stores that follow (" nt", ...)

    def gload(self, dwords, dst, voff, sbase, offset=0):
        """global_load with a scalar base and a 32-bit vector offset (or a 64-bit vector address and 'off')."""
        self._touch(vregs_of(voff), vregs_of(dst))
        op = {1: "global_load_dword", 2: "global_load_dwordx2", 4: "global_load_dwordx4"}[dwords]
        self._ins("%s %s, %s, %s%s%s" % (op, dst, voff, sbase, " offset:%d" % offset if offset else "", self.mods), "vmem")
        self.vm.append(vregs_of(dst))

    def gload_lds(self, voff, sbase):
        """global_load_lds_dwordx4: every lane's 16 bytes at sbase + voff go to LDS at M0 + 16 * lane (counted by vmcnt)."""
        self._touch(vregs_of(voff), set())
        self._ins("global_load_lds_dwordx4 %s, %s%s" % (voff, sbase, self.mods), "vmem")
        self.vm.append(set())

    def gstore(self, dwords, voff, data, sbase, offset=0):
        self._touch(vregs_of(voff) | vregs_of(data), set())
        op = {1: "global_store_dword", 2: "global_store_dwordx2", 4: "global_store_dwordx4"}[dwords]
        self._ins("%s %s, %s, %s%s%s" % (op, voff, data, sbase, " offset:%d" % offset if offset else "", self.mods), "vmem")
        self.vm.append(set())
        if dwords > 2:
            for r in vregs_of(data):
                self.store_r[r] = self.n - 1

    def smem(self, op, dst, base, offset):
        self._ins("%s %s, %s, %s" % (op, dst, base, offset), "smem")
        self.lgkm.append(set())

    def text(self):
        return "\n".join(self.lines) + "\n"


class Pool:
    """VGPR pairs handed out from a fixed range; take(n) returns n consecutive registers, even-aligned."""

    def __init__(self, first, last):
        assert first % 2 == 0
        self.free = [True] * ((last + 1 - first) // 2)
        self.first = first
        self.high = 0

    def take(self, n=2):
        pairs = (n + 1) // 2
        run = 0
        for k, f in enumerate(self.free):
            run = run + 1 if f else 0
            if run == pairs:
                k0 = k - pairs + 1
                for q in range(k0, k + 1):
                    self.free[q] = False
                self.high = max(self.high, k + 1)
                return V(self.first + 2 * k0, n)
        raise RuntimeError("out of temporary VGPRs (%d pairs wanted)" % pairs)

    def give(self, *vs):
        for v in vs:
            for r in range(v.i, v.i + v.n, 2):
                k = (r - self.first) // 2
                assert not self.free[k], "register v%d freed twice" % r
                self.free[k] = True

    def hold(self, *vs):
        """registers that are live where the code being emitted runs, although this program has handed them back"""
        for v in vs:
            for r in range(v.i, v.i + v.n, 2):
                k = (r - self.first) // 2
                assert self.free[k], "register v%d is not free" % r
                self.free[k] = False

    def in_use(self):
        return sum(1 for f in self.free if not f)
