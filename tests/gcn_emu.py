"""A functional emulator of the gfx950 instruction subset the hand-scheduled sweeps use (test infrastructure).

One wave at a time, 64 lanes as numpy vectors, fp64 arithmetic with IEEE semantics (v_fma_f64 through libm's fma).
Besides computing what the kernel computes it checks what the hardware leaves to the programmer:
  * a register an LDS / SMEM / global load is still writing may not be touched before an s_waitcnt that covers it
    (the counters are modelled as in-order queues; loads commit when a wait retires them);
  * a DPP instruction may not read a VGPR a VALU instruction wrote in the two instructions before it;
  * a VALU instruction may not write the data registers of a store wider than 64 bits in the two instructions after it.
Addresses of instructions are 4 * index, which is all s_getpc / s_swappc / label differences need.
"""
import ctypes
import platform
import re
import struct

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3
_FE_TONEAREST, _FE_UPWARD = 0, 0x800       # <fenv.h> of x86-64 glibc
assert platform.machine() == "x86_64"

LANES = np.arange(64)
U64 = np.uint64
U32 = np.uint32


class EmuError(Exception):
    pass


class Memory:
    """Flat device memory: a bump allocator over one byte array (address 0 is never valid)."""

    def __init__(self, size=1 << 28):
        self.b = np.zeros(size, np.uint8)
        self.top = 1 << 16
        self.blocks = []

    def alloc(self, nbytes, name=None, align=256):
        a = (self.top + align - 1) // align * align
        self.top = a + int(nbytes) + 256
        if self.top > self.b.size:
            raise EmuError("out of emulated memory")
        self.blocks.append((a, a + int(nbytes), name or "block%d" % len(self.blocks)))
        return a

    def put(self, addr, arr):
        raw = np.ascontiguousarray(arr).view(np.uint8).ravel()
        self.b[addr: addr + raw.size] = raw

    def get(self, addr, dtype, count):
        n = np.dtype(dtype).itemsize * count
        return self.b[addr: addr + n].view(dtype).copy()

    def check(self, addr, n):
        for lo, hi, _ in self.blocks:
            if lo <= addr and addr + n <= hi:
                return
        raise EmuError("global access outside every allocation: 0x%x (+%d)" % (addr, n))


_REG = re.compile(r"^(-?)([vs])(\d+)$")
_RANGE = re.compile(r"^(-?)([vs])\[(\d+):(\d+)\]$")


class Op:
    __slots__ = ("kind", "i", "n", "neg", "val")

    def __init__(self, kind, i=0, n=1, neg=False, val=None):
        self.kind, self.i, self.n, self.neg, self.val = kind, i, n, neg, val


def parse_operand(tok, labels):
    tok = tok.strip()
    m = _REG.match(tok)
    if m:
        return Op(m.group(2), int(m.group(3)), 1, m.group(1) == "-")
    m = _RANGE.match(tok)
    if m:
        return Op(m.group(2), int(m.group(3)), int(m.group(4)) - int(m.group(3)) + 1, m.group(1) == "-")
    if tok in ("vcc", "exec", "off", "scc", "m0"):
        return Op(tok)
    if tok in ("exec_lo", "exec_hi", "vcc_lo", "vcc_hi"):
        return Op(tok)
    if re.match(r"^-?0x[0-9a-fA-F]+$", tok) or re.match(r"^-?\d+$", tok):
        return Op("int", val=int(tok, 0))
    if re.match(r"^-?\d+\.\d*$", tok):
        return Op("float", val=float(tok))
    if "-" in tok and not tok.startswith("-"):
        a, b = tok.split("-")
        return Op("labeldiff", val=(a.strip(), b.strip()))
    return Op("label", val=tok)


class Instr:
    __slots__ = ("op", "ops", "mods", "text", "line")


def parse(text):
    """-> (instructions, labels).  Directives and comments are skipped."""
    instrs, labels = [], {}
    for ln, line in enumerate(text.splitlines()):
        s = line.split(";")[0].split("//")[0].strip()
        if not s:
            continue
        if s.startswith(".amdgpu_metadata"):
            break
        if s.endswith(":"):
            labels[s[:-1]] = len(instrs)
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        ins = Instr()
        ins.op = parts[0]
        for suf in ("_e32", "_e64", "_dpp"):
            if ins.op.endswith(suf):
                ins.op = ins.op[: -len(suf)]
        ins.text, ins.line = s, ln + 1
        rest = parts[1] if len(parts) > 1 else ""
        mods = {}
        m = re.match(r"^hwreg\(HW_REG_MODE,\s*(\d+),\s*(\d+)\),\s*(\S+)$", rest) if ins.op == "s_setreg_imm32_b32" else None
        if m:                                        # (the only hardware register the sweeps write: bit offset, width, value)
            ins.ops, ins.mods = [m.group(1), m.group(2), m.group(3)], mods
            instrs.append(ins)
            continue
        # trailing modifiers: offset:N, wave_ror:1, row_mask:0xf, bank_mask:0xf, vmcnt(N), lgkmcnt(N)
        toks = []
        for piece in rest.split(","):
            piece = piece.strip()
            if not piece:
                continue
            words = piece.split()
            keep = []
            for w in words:
                if ":" in w and not w.startswith(("v[", "s[", "-v[", "-s[")):
                    kk, vv = w.split(":")
                    mods[kk] = int(vv, 0)
                elif "(" in w:
                    kk, vv = w.split("(")
                    mods[kk] = int(vv.rstrip(")"), 0)
                elif w in ("nt", "sc0", "sc1"):
                    mods[w] = 1                      # (cache policy: nothing to emulate)
                else:
                    keep.append(w)
            if keep:
                toks.append(" ".join(keep))
        ins.ops = toks
        ins.mods = mods
        instrs.append(ins)
    return instrs, labels


def _f64(u):
    return u.view(np.float64)


def _u64(f):
    return np.asarray(f, np.float64).view(np.uint64)


class Wave:
    def __init__(self, instrs, labels, mem, lds_bytes, kernarg_addr, wg_id, entry):
        self.I, self.labels, self.mem = instrs, labels, mem
        self.ops_cache = {}
        self.v = np.zeros((256, 64), U32)
        self.s = np.zeros(128, U32)
        self.vcc = 0
        self.exec = (1 << 64) - 1
        self.scc = 0
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.pc = labels[entry]
        self.lgkm, self.vmq = [], []       # pending operations: (regs: list of ('v'|'s', idx), commit closure)
        self.m0 = 0
        self.round64 = 0                   # MODE[3:2]: fp64 rounding, 0 to nearest even, 1 towards +inf (fma only)
        # values read from LDS bytes a load to LDS was still writing: any use of one but overwriting or deselecting it is an
        # error (the logAdd table is indexed without a clamp; what lies beyond it is read and thrown away)
        self.taint = np.zeros((256, 64), bool)
        self._tsrc = np.zeros(64, bool)
        self.lds_pending = []              # [lo, hi) byte ranges of LDS a load to LDS in flight will write
        self.inflight = set()
        self.count = 0
        self.last_valu = {}
        self.last_store = {}
        self.clock = 1000
        self.trace = None
        self.watch = {}                    # label -> callback(wave)
        self.s[0] = kernarg_addr & 0xFFFFFFFF
        self.s[1] = kernarg_addr >> 32
        self.s[2] = wg_id
        self.v[0] = LANES.astype(U32)
        self.stats = {}
        self.label_at = {}
        for name, idx in labels.items():
            self.label_at.setdefault(idx, []).append(name)

    # ---------------------------------------------------------------- operands
    def _ops(self, ins):
        o = self.ops_cache.get(id(ins))
        if o is None:
            o = [parse_operand(t, self.labels) for t in ins.ops]
            self.ops_cache[id(ins)] = o
        return o

    def _touch(self, op, n=None, write=False):
        if op.kind in ("v", "s"):
            for r in range(op.i, op.i + (n or op.n)):
                if (op.kind, r) in self.inflight:
                    raise EmuError("%s%d is still being written by a load (missing s_waitcnt)" % (op.kind, r))

    def mask(self):
        return ((self.exec >> LANES.astype(np.uint64)) & np.uint64(1)).astype(bool)

    def rd32(self, op):
        self._touch(op, 1)
        if op.kind == "v":
            self._tsrc |= self.taint[op.i]
            return self.v[op.i].copy()
        if op.kind == "s":
            return np.full(64, self.s[op.i], U32)
        if op.kind == "int":
            return np.full(64, op.val & 0xFFFFFFFF, U32)
        if op.kind == "float":
            return np.full(64, struct.unpack("<I", struct.pack("<f", op.val))[0], U32)
        if op.kind == "vcc_lo":
            return np.full(64, self.vcc & 0xFFFFFFFF, U32)
        raise EmuError("32-bit read of " + op.kind)

    def rd64(self, op):
        self._touch(op, 2)
        if op.kind == "v":
            self._tsrc |= self.taint[op.i] | self.taint[op.i + 1]
            u = self.v[op.i].astype(U64) | (self.v[op.i + 1].astype(U64) << U64(32))
        elif op.kind == "s":
            u = np.full(64, int(self.s[op.i]) | (int(self.s[op.i + 1]) << 32), U64)
        elif op.kind == "int":
            u = np.full(64, op.val & 0xFFFFFFFFFFFFFFFF, U64)
        elif op.kind == "float":
            u = np.full(64, struct.unpack("<Q", struct.pack("<d", op.val))[0], U64)
        elif op.kind == "vcc":
            u = np.full(64, self.vcc, U64)
        elif op.kind == "exec":
            u = np.full(64, self.exec, U64)
        else:
            raise EmuError("64-bit read of " + op.kind)
        if op.neg:
            u = u ^ U64(1 << 63)
        return u

    def rdf64(self, op):
        if op.kind == "int" and not -16 <= op.val <= 64:
            # a 32-bit literal as a 64-bit float operand is the value's high half
            return _f64(np.full(64, (op.val & 0xFFFFFFFF) << 32, U64))
        return _f64(self.rd64(op))

    def wr32(self, op, val, masked=True):
        self._touch(op, 1, True)
        val = np.asarray(val).astype(U32)
        if op.kind == "v":
            if masked:
                m = self.mask()
                self.v[op.i][m] = val[m] if val.ndim else val
                self.taint[op.i][m] = self._tsrc[m]
            else:
                self.v[op.i] = val
                self.taint[op.i] = self._tsrc
        elif op.kind == "s":
            self.s[op.i] = val if val.ndim == 0 else val[0]
        else:
            raise EmuError("32-bit write of " + op.kind)

    def wr64(self, op, u):
        self._touch(op, 2, True)
        u = np.asarray(u).astype(U64)
        if op.kind == "v":
            m = self.mask()
            self.v[op.i][m] = (u & U64(0xFFFFFFFF)).astype(U32)[m]
            self.v[op.i + 1][m] = (u >> U64(32)).astype(U32)[m]
            self.taint[op.i][m] = self._tsrc[m]
            self.taint[op.i + 1][m] = self._tsrc[m]
        else:
            raise EmuError("64-bit vector write of " + op.kind)

    def s_rd(self, op, bits=32):
        self._touch(op, bits // 32)
        if op.kind == "s":
            if bits == 32:
                return int(self.s[op.i])
            return int(self.s[op.i]) | (int(self.s[op.i + 1]) << 32)
        if op.kind == "int":
            return op.val & ((1 << bits) - 1)
        if op.kind == "vcc":
            return self.vcc
        if op.kind == "exec":
            return self.exec
        if op.kind == "m0":
            return self.m0
        if op.kind == "labeldiff":
            return ((self.labels[op.val[0]] - self.labels[op.val[1]]) * 4) & 0xFFFFFFFF
        raise EmuError("scalar read of " + op.kind)

    def s_wr(self, op, val, bits=32):
        self._touch(op, bits // 32, True)
        val &= (1 << bits) - 1
        if op.kind == "s":
            self.s[op.i] = val & 0xFFFFFFFF
            if bits == 64:
                self.s[op.i + 1] = val >> 32
        elif op.kind == "vcc":
            self.vcc = val
        elif op.kind == "exec":
            self.exec = val
        elif op.kind == "exec_lo":
            self.exec = (self.exec & ~0xFFFFFFFF) | (val & 0xFFFFFFFF)
        elif op.kind == "m0":
            self.m0 = val & 0xFFFFFFFF
        else:
            raise EmuError("scalar write of " + op.kind)

    # ---------------------------------------------------------------- queues
    def _issue(self, queue, regs, commit):
        for r in regs:
            if r in self.inflight:
                # (scalar loads return out of order: two of them in flight to one register leave either value)
                raise EmuError("%s%d is already the destination of a load in flight" % r)
            self.inflight.add(r)
        queue.append((regs, commit))

    def _retire(self, queue, left):
        while len(queue) > left:
            regs, commit = queue.pop(0)
            for r in regs:
                self.inflight.discard(r)
            if commit:
                commit()

    # ---------------------------------------------------------------- execution
    def run(self, max_instr=50_000_000):
        while True:
            if self.pc in self.label_at and self.watch:
                for name in self.label_at[self.pc]:
                    if name in self.watch:
                        self.watch[name](self)
            ins = self.I[self.pc]
            self.pc += 1
            self.count += 1
            if self.count > max_instr:
                raise EmuError("instruction limit reached (endless loop?) at line %d" % ins.line)
            try:
                if self.step(ins):
                    break
            except EmuError as e:
                raise EmuError("%s\n  at line %d: %s" % (e, ins.line, ins.text)) from None
        self._retire(self.lgkm, 0)
        self._retire(self.vmq, 0)

    def step(self, ins):
        op = ins.op
        self._tsrc = np.zeros(64, bool)
        self.stats[op[:2]] = self.stats.get(op[:2], 0) + 1
        h = getattr(self, "i_" + op, None)
        if h is None:
            raise EmuError("instruction not emulated: " + op)
        return h(ins, self._ops(ins))

    # ----- hazards
    def _valu_writes(self, regs):
        for r in regs:
            if self.count - self.last_store.get(r, -10) < 3:
                raise EmuError("VALU write of v%d within two instructions of a wide store that reads it" % r)
            self.last_valu[r] = self.count

    def _dst_regs(self, op, n):
        return range(op.i, op.i + n) if op.kind == "v" else ()

    # ----- scalar ALU
    def i_s_mov_b32(self, ins, o): self.s_wr(o[0], self.s_rd(o[1]))
    def i_s_mov_b64(self, ins, o):
        v = self.s_rd(o[1], 64)
        if o[1].kind == "int" and o[1].val < 0:
            v = o[1].val & ((1 << 64) - 1)
        self.s_wr(o[0], v, 64)

    def i_s_add_u32(self, ins, o):
        r = self.s_rd(o[1]) + self.s_rd(o[2])
        self.scc = 1 if r >> 32 else 0
        self.s_wr(o[0], r)

    def i_s_addc_u32(self, ins, o):
        r = self.s_rd(o[1]) + self.s_rd(o[2]) + self.scc
        self.scc = 1 if r >> 32 else 0
        self.s_wr(o[0], r)

    def i_s_sub_u32(self, ins, o):
        a, b = self.s_rd(o[1]), self.s_rd(o[2])
        self.scc = 1 if b > a else 0
        self.s_wr(o[0], a - b)

    def i_s_subb_u32(self, ins, o):
        a, b = self.s_rd(o[1]), self.s_rd(o[2]) + self.scc
        self.scc = 1 if b > a else 0
        self.s_wr(o[0], a - b)

    def i_s_add_i32(self, ins, o): self.i_s_add_u32(ins, o)
    def i_s_sub_i32(self, ins, o): self.i_s_sub_u32(ins, o)

    @staticmethod
    def _i32(x):
        x &= 0xFFFFFFFF
        return x - (1 << 32) if x >> 31 else x

    def i_s_mul_i32(self, ins, o): self.s_wr(o[0], self._i32(self.s_rd(o[1])) * self._i32(self.s_rd(o[2])))
    def i_s_mul_hi_u32(self, ins, o): self.s_wr(o[0], (self.s_rd(o[1]) * self.s_rd(o[2])) >> 32)
    def i_s_lshl_b32(self, ins, o): self.s_wr(o[0], self.s_rd(o[1]) << (self.s_rd(o[2]) & 31))
    def i_s_lshr_b32(self, ins, o): self.s_wr(o[0], self.s_rd(o[1]) >> (self.s_rd(o[2]) & 31))
    def i_s_lshl_b64(self, ins, o): self.s_wr(o[0], self.s_rd(o[1], 64) << (self.s_rd(o[2]) & 63), 64)
    def i_s_lshr_b64(self, ins, o): self.s_wr(o[0], self.s_rd(o[1], 64) >> (self.s_rd(o[2]) & 63), 64)

    def _logic(self, o, bits, f):
        r = f(self.s_rd(o[1], bits), self.s_rd(o[2], bits)) & ((1 << bits) - 1)
        self.scc = 1 if r else 0
        self.s_wr(o[0], r, bits)

    def i_s_and_b32(self, ins, o): self._logic(o, 32, lambda a, b: a & b)
    def i_s_and_b64(self, ins, o): self._logic(o, 64, lambda a, b: a & b)
    def i_s_or_b32(self, ins, o): self._logic(o, 32, lambda a, b: a | b)
    def i_s_or_b64(self, ins, o): self._logic(o, 64, lambda a, b: a | b)
    def i_s_andn2_b64(self, ins, o): self._logic(o, 64, lambda a, b: a & ~b)
    def i_s_xor_b64(self, ins, o): self._logic(o, 64, lambda a, b: a ^ b)

    def i_s_min_i32(self, ins, o):
        a, b = self._i32(self.s_rd(o[1])), self._i32(self.s_rd(o[2]))
        self.scc = 1 if a <= b else 0
        self.s_wr(o[0], min(a, b))

    def i_s_max_i32(self, ins, o):
        a, b = self._i32(self.s_rd(o[1])), self._i32(self.s_rd(o[2]))
        self.scc = 1 if a >= b else 0
        self.s_wr(o[0], max(a, b))

    def _cmp(self, o, signed, f):
        a, b = self.s_rd(o[0]), self.s_rd(o[1])
        if signed:
            a, b = self._i32(a), self._i32(b)
        self.scc = 1 if f(a, b) else 0

    def i_s_cmp_eq_u32(self, ins, o): self._cmp(o, False, lambda a, b: a == b)
    def i_s_cmp_lg_u32(self, ins, o): self._cmp(o, False, lambda a, b: a != b)
    def i_s_cmp_ge_u32(self, ins, o): self._cmp(o, False, lambda a, b: a >= b)
    def i_s_cmp_gt_u32(self, ins, o): self._cmp(o, False, lambda a, b: a > b)
    def i_s_cmp_lt_u32(self, ins, o): self._cmp(o, False, lambda a, b: a < b)
    def i_s_cmp_le_u32(self, ins, o): self._cmp(o, False, lambda a, b: a <= b)
    def i_s_cmp_eq_i32(self, ins, o): self._cmp(o, True, lambda a, b: a == b)
    def i_s_cmp_lg_i32(self, ins, o): self._cmp(o, True, lambda a, b: a != b)
    def i_s_cmp_ge_i32(self, ins, o): self._cmp(o, True, lambda a, b: a >= b)
    def i_s_cmp_gt_i32(self, ins, o): self._cmp(o, True, lambda a, b: a > b)
    def i_s_cmp_lt_i32(self, ins, o): self._cmp(o, True, lambda a, b: a < b)
    def i_s_cmp_le_i32(self, ins, o): self._cmp(o, True, lambda a, b: a <= b)
    def i_s_cmp_lg_u64(self, ins, o): self.scc = 1 if self.s_rd(o[0], 64) != self.s_rd(o[1], 64) else 0
    def i_s_cmp_eq_u64(self, ins, o): self.scc = 1 if self.s_rd(o[0], 64) == self.s_rd(o[1], 64) else 0
    def i_s_min_u32(self, ins, o):
        a, b = self.s_rd(o[1]), self.s_rd(o[2])
        self.scc = int(a <= b)
        self.s_wr(o[0], min(a, b))

    def i_s_bfm_b64(self, ins, o):
        width, offset = self.s_rd(o[1]) & 63, self.s_rd(o[2]) & 63
        self.s_wr(o[0], (((1 << width) - 1) << offset) & ((1 << 64) - 1), 64)

    def i_s_bcnt1_i32_b64(self, ins, o):
        r = bin(self.s_rd(o[1], 64)).count("1")
        self.scc = 1 if r else 0
        self.s_wr(o[0], r)

    def i_s_cselect_b32(self, ins, o): self.s_wr(o[0], self.s_rd(o[1]) if self.scc else self.s_rd(o[2]))

    def i_s_cselect_b64(self, ins, o):
        def rd(op):
            if op.kind == "int" and op.val < 0:
                return op.val & ((1 << 64) - 1)
            return self.s_rd(op, 64)
        self.s_wr(o[0], rd(o[1]) if self.scc else rd(o[2]), 64)

    def i_s_bitcmp1_b32(self, ins, o): self.scc = (self.s_rd(o[0]) >> (self.s_rd(o[1]) & 31)) & 1
    def i_s_bitcmp1_b64(self, ins, o): self.scc = (self.s_rd(o[0], 64) >> (self.s_rd(o[1]) & 63)) & 1
    def i_s_bitcmp0_b64(self, ins, o): self.scc = 1 - ((self.s_rd(o[0], 64) >> (self.s_rd(o[1]) & 63)) & 1)
    def i_s_nop(self, ins, o): self.count += o[0].val
    def i_s_setprio(self, ins, o): pass

    def i_s_setreg_imm32_b32(self, ins, o):
        off, width, val = o[0].val, o[1].val, o[2].val
        if (off, width) != (2, 2) or val not in (0, 1):
            raise NotImplementedError("s_setreg of MODE bits %d..%d to %d" % (off, off + width - 1, val))
        self.round64 = val
    def i_s_sleep(self, ins, o): pass

    def i_s_waitcnt(self, ins, o):
        if "lgkmcnt" in ins.mods:
            self._retire(self.lgkm, ins.mods["lgkmcnt"])
        if "vmcnt" in ins.mods:
            self._retire(self.vmq, ins.mods["vmcnt"])

    def i_s_endpgm(self, ins, o): return True

    def _jump(self, op): self.pc = self.labels[op.val]
    def i_s_branch(self, ins, o): self._jump(o[0])
    def i_s_cbranch_scc0(self, ins, o):
        if not self.scc: self._jump(o[0])
    def i_s_cbranch_scc1(self, ins, o):
        if self.scc: self._jump(o[0])
    def i_s_cbranch_vccz(self, ins, o):
        if self.vcc == 0: self._jump(o[0])
    def i_s_cbranch_vccnz(self, ins, o):
        if self.vcc != 0: self._jump(o[0])
    def i_s_cbranch_execz(self, ins, o):
        if self.exec == 0: self._jump(o[0])

    def i_s_getpc_b64(self, ins, o): self.s_wr(o[0], self.pc * 4, 64)

    def i_s_swappc_b64(self, ins, o):
        target = self.s_rd(o[1], 64)
        self.s_wr(o[0], self.pc * 4, 64)
        self.pc = target // 4

    def i_s_setpc_b64(self, ins, o): self.pc = self.s_rd(o[0], 64) // 4

    def i_s_memtime(self, ins, o):
        self.clock += 1000
        clk = self.clock
        op = o[0]
        self._issue(self.lgkm, [("s", op.i), ("s", op.i + 1)], lambda: self._commit_s(op, [clk & 0xFFFFFFFF, clk >> 32]))

    def i_s_memrealtime(self, ins, o): self.i_s_memtime(ins, o)

    def _commit_s(self, op, words):
        for k, w in enumerate(words):
            self.s[op.i + k] = w

    def _s_load(self, o, n):
        base = self.s_rd(o[1], 64)
        off = self.s_rd(o[2]) if o[2].kind != "int" else o[2].val
        addr = base + off
        self.mem.check(addr, 4 * n)
        words = self.mem.get(addr, U32, n)
        op = o[0]
        self._issue(self.lgkm, [("s", op.i + k) for k in range(n)], lambda: self._commit_s(op, words))

    def i_s_load_dword(self, ins, o): self._s_load(o, 1)
    def i_s_load_dwordx2(self, ins, o): self._s_load(o, 2)
    def i_s_load_dwordx4(self, ins, o): self._s_load(o, 4)
    def i_s_load_dwordx8(self, ins, o): self._s_load(o, 8)
    def i_s_load_dwordx16(self, ins, o): self._s_load(o, 16)

    # ----- vector ALU
    def _v64(self, ins, o, f, n_src):
        if self.round64 != 0:
            raise RuntimeError("%s under a rounding mode other than to nearest (line %d)" % (ins.op, ins.line))
        srcs = [self.rdf64(x) for x in o[1: 1 + n_src]]
        with np.errstate(all="ignore"):
            r = f(*srcs)
        self._valu_writes(self._dst_regs(o[0], 2))
        self.wr64(o[0], _u64(r))

    def i_v_add_f64(self, ins, o): self._v64(ins, o, lambda a, b: a + b, 2)
    def i_v_mul_f64(self, ins, o): self._v64(ins, o, lambda a, b: a * b, 2)
    def i_v_max_f64(self, ins, o): self._v64(ins, o, np.fmax, 2)
    def i_v_min_f64(self, ins, o): self._v64(ins, o, np.fmin, 2)
    def i_v_ceil_f64(self, ins, o): self._v64(ins, o, np.ceil, 1)

    def i_v_fma_f64(self, ins, o):
        a, b, c = (self.rdf64(x) for x in o[1:4])
        if self.round64 == 1:
            _libm.fesetround(_FE_UPWARD)
        try:
            r = np.array([_libm.fma(float(a[i]), float(b[i]), float(c[i])) for i in range(64)])
        finally:
            if self.round64 == 1:
                _libm.fesetround(_FE_TONEAREST)
        self._valu_writes(self._dst_regs(o[0], 2))
        self.wr64(o[0], _u64(r))

    def i_v_cvt_i32_f64(self, ins, o):
        a = self.rdf64(o[1])
        with np.errstate(all="ignore"):
            t = np.where(np.isnan(a), 0.0, np.clip(np.trunc(a), -2147483648.0, 2147483647.0))
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], t.astype(np.int64).astype(U32))

    def i_v_cvt_f32_f64(self, ins, o):
        with np.errstate(all="ignore"):
            r = self.rdf64(o[1]).astype(np.float32)
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], r.view(U32))

    def i_v_cvt_f64_f32(self, ins, o):
        r = self.rd32(o[1]).view(np.float32).astype(np.float64)
        self._valu_writes(self._dst_regs(o[0], 2))
        self.wr64(o[0], _u64(r))

    def _v32(self, ins, o, f, n_src=2):
        srcs = [self.rd32(x) for x in o[1: 1 + n_src]]
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], f(*srcs))

    def _f32(self, ins, o, f, n_src=2):
        srcs = [self.rd32(x).view(np.float32) for x in o[1: 1 + n_src]]
        with np.errstate(all="ignore"):
            r = np.asarray(f(*srcs), np.float32)
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], r.view(U32))

    def i_v_mul_f32(self, ins, o): self._f32(ins, o, lambda a, b: a * b)
    def i_v_add_f32(self, ins, o): self._f32(ins, o, lambda a, b: a + b)
    def i_v_max_f32(self, ins, o): self._f32(ins, o, np.fmax)
    def i_v_exp_f32(self, ins, o): self._f32(ins, o, np.exp2, 1)
    def i_v_log_f32(self, ins, o): self._f32(ins, o, np.log2, 1)

    def i_v_mov_b32(self, ins, o):
        src = self.rd32(o[1])
        if "wave_ror" in ins.mods or "wave_rol" in ins.mods:
            for r in (o[1].i,) if o[1].kind == "v" else ():
                if self.count - self.last_valu.get(r, -10) < 3:
                    raise EmuError("DPP read of v%d within two instructions of the VALU write" % r)
            src = np.roll(src, 1) if "wave_ror" in ins.mods else np.roll(src, -1)
            # (a disabled source lane would need bound_ctrl / the old value: the sweeps rotate with all lanes enabled)
            if self.exec != (1 << 64) - 1:
                raise EmuError("wave rotation with lanes disabled")
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], src)

    def i_v_lshlrev_b32(self, ins, o): self._v32(ins, o, lambda s, a: a << (s & U32(31)))
    def i_v_lshrrev_b32(self, ins, o): self._v32(ins, o, lambda s, a: a >> (s & U32(31)))
    def i_v_add_u32(self, ins, o): self._v32(ins, o, lambda a, b: a + b)
    def i_v_sub_u32(self, ins, o): self._v32(ins, o, lambda a, b: a - b)
    def i_v_subrev_u32(self, ins, o): self._v32(ins, o, lambda a, b: b - a)
    def i_v_and_b32(self, ins, o): self._v32(ins, o, lambda a, b: a & b)
    def i_v_or_b32(self, ins, o): self._v32(ins, o, lambda a, b: a | b)
    def i_v_xor_b32(self, ins, o): self._v32(ins, o, lambda a, b: a ^ b)
    def i_v_mul_u32_u24(self, ins, o): self._v32(ins, o, lambda a, b: ((a & U32(0xFFFFFF)).astype(U64) * (b & U32(0xFFFFFF)).astype(U64)).astype(U32))
    def i_v_mul_hi_u32(self, ins, o): self._v32(ins, o, lambda a, b: ((a.astype(U64) * b.astype(U64)) >> U64(32)).astype(U32))
    def i_v_min_i32(self, ins, o): self._v32(ins, o, lambda a, b: np.minimum(a.view(np.int32), b.view(np.int32)).view(U32))
    def i_v_max_i32(self, ins, o): self._v32(ins, o, lambda a, b: np.maximum(a.view(np.int32), b.view(np.int32)).view(U32))
    def i_v_min_u32(self, ins, o): self._v32(ins, o, np.minimum)
    def i_v_lshl_add_u32(self, ins, o): self._v32(ins, o, lambda a, s, c: (a << (s & U32(31))) + c, 3)
    def _mbcnt(self, o, shift):
        m = (self.s_rd(o[1]) if o[1].kind != "int" else o[1].val) & 0xFFFFFFFF
        below = np.array([bin(m & ((1 << max(0, min(32, int(l) - shift))) - 1)).count("1") for l in LANES], U32)
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], below + self.rd32(o[2]))

    def i_v_mbcnt_lo_u32_b32(self, ins, o): self._mbcnt(o, 0)
    def i_v_mbcnt_hi_u32_b32(self, ins, o): self._mbcnt(o, 32)

    def i_v_bcnt_u32_b32(self, ins, o):
        self._v32(ins, o, lambda a, b: np.array([bin(int(x)).count("1") for x in a], U32) + b)

    def _set_mask(self, dst, bits):
        """a compare's result: one bit per enabled lane, zero elsewhere"""
        m = self.mask()
        if (self._tsrc & m).any():
            raise EmuError("a compare reads a value fetched from LDS bytes a load to LDS was still writing")
        val = 0
        for i in np.nonzero(bits & m)[0]:
            val |= 1 << int(i)
        if dst.kind == "vcc":
            self.vcc = val
        else:
            self.s_wr(dst, val, 64)

    def _cmp64(self, o, f):
        a, b = self.rdf64(o[1]), self.rdf64(o[2])
        with np.errstate(all="ignore"):
            self._set_mask(o[0], f(a, b))

    def i_v_cmp_gt_f64(self, ins, o): self._cmp64(o, lambda a, b: a > b)
    def i_v_cmp_ge_f64(self, ins, o): self._cmp64(o, lambda a, b: a >= b)
    def i_v_cmp_lt_f64(self, ins, o): self._cmp64(o, lambda a, b: a < b)
    def i_v_cmp_le_f64(self, ins, o): self._cmp64(o, lambda a, b: a <= b)

    def _cmp32(self, o, signed, f):
        a, b = self.rd32(o[1]), self.rd32(o[2])
        if signed:
            a, b = a.view(np.int32), b.view(np.int32)
        self._set_mask(o[0], f(a, b))

    def i_v_cmp_le_i32(self, ins, o): self._cmp32(o, True, lambda a, b: a <= b)
    def i_v_cmp_lt_i32(self, ins, o): self._cmp32(o, True, lambda a, b: a < b)
    def i_v_cmp_gt_i32(self, ins, o): self._cmp32(o, True, lambda a, b: a > b)
    def i_v_cmp_ge_i32(self, ins, o): self._cmp32(o, True, lambda a, b: a >= b)
    def i_v_cmp_eq_u32(self, ins, o): self._cmp32(o, False, lambda a, b: a == b)
    def i_v_cmp_ne_u32(self, ins, o): self._cmp32(o, False, lambda a, b: a != b)
    def i_v_cmp_gt_u32(self, ins, o): self._cmp32(o, False, lambda a, b: a > b)
    def i_v_cmp_lt_u32(self, ins, o): self._cmp32(o, False, lambda a, b: a < b)
    def i_v_cmp_ge_u32(self, ins, o): self._cmp32(o, False, lambda a, b: a >= b)

    def i_v_cndmask_b32(self, ins, o):
        sel = self.s_rd(o[3], 64) if o[3].kind != "vcc" else self.vcc
        bits = ((sel >> LANES.astype(np.uint64)) & np.uint64(1)).astype(bool) if False else \
            np.array([(sel >> i) & 1 for i in range(64)], bool)
        a = self.rd32(o[1])
        ta, self._tsrc = self._tsrc, np.zeros(64, bool)
        b = self.rd32(o[2])
        self._tsrc = np.where(bits, self._tsrc, ta)
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], np.where(bits, b, a))

    def i_v_add_co_u32(self, ins, o):
        a, b = self.rd32(o[2]).astype(U64), self.rd32(o[3]).astype(U64)
        r = a + b
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], (r & U64(0xFFFFFFFF)).astype(U32))
        self._set_mask(o[1], (r >> U64(32)) != 0)

    def i_v_addc_co_u32(self, ins, o):
        cin = np.array([(self.vcc >> i) & 1 for i in range(64)], U64)
        r = self.rd32(o[2]).astype(U64) + self.rd32(o[3]).astype(U64) + cin
        self._valu_writes(self._dst_regs(o[0], 1))
        self.wr32(o[0], (r & U64(0xFFFFFFFF)).astype(U32))
        self._set_mask(o[1], (r >> U64(32)) != 0)

    def i_v_readlane_b32(self, ins, o):
        lane = self.s_rd(o[2]) & 63
        self.s_wr(o[0], int(self.rd32(o[1])[lane]))

    def i_v_readfirstlane_b32(self, ins, o):
        m = np.nonzero(self.mask())[0]
        self.s_wr(o[0], int(self.rd32(o[1])[m[0] if len(m) else 0]))

    # ----- LDS
    def _lds_read(self, o, ins, nbytes):
        addr = self.rd32(o[1]).astype(np.int64) + ins.mods.get("offset", 0)
        m = self.mask()
        out = np.zeros((64, nbytes), np.uint8)
        bad = np.zeros(64, bool)
        for i in np.nonzero(m)[0]:
            a = int(addr[i]) & 0xFFFFFFFF
            for lo, hi in self.lds_pending:
                if a < hi and a + nbytes > lo:
                    bad[i] = True
            if a % min(nbytes, 16) and nbytes >= 8 and a % 8:
                raise EmuError("misaligned LDS access 0x%x" % a)
            if a + nbytes <= self.lds.size:
                out[i] = self.lds[a: a + nbytes]
        words = out.view(U32)  # (64, nbytes / 4)
        op = o[0]
        n = nbytes // 4

        def commit():
            for k in range(n):
                self.v[op.i + k][m] = words[:, k][m]
                self.taint[op.i + k][m] = bad[m]
        self._issue(self.lgkm, [("v", op.i + k) for k in range(n)], commit)

    def i_ds_read_b32(self, ins, o): self._lds_read(o, ins, 4)
    def i_ds_read_b64(self, ins, o): self._lds_read(o, ins, 8)
    def i_ds_read_b128(self, ins, o): self._lds_read(o, ins, 16)

    def _lds_write(self, o, ins, n):
        addr = self.rd32(o[0]).astype(np.int64) + ins.mods.get("offset", 0)
        for k in range(n):
            self._touch(Op("v", o[1].i + k), 1)
            self._tsrc |= self.taint[o[1].i + k]
        m = self.mask()
        if (self._tsrc & m).any():
            raise EmuError("an LDS write uses a value fetched from LDS bytes a load to LDS was still writing")
        for i in np.nonzero(m)[0]:
            a = int(addr[i])
            if a + 4 * n > self.lds.size:
                raise EmuError("LDS write beyond the allocation: 0x%x" % a)
            self.lds[a: a + 4 * n] = np.array([self.v[o[1].i + k][i] for k in range(n)], U32).view(np.uint8)
        if n > 2:
            for k in range(n):
                self.last_store[o[1].i + k] = self.count
        self._issue(self.lgkm, [], None)

    def i_ds_write_b32(self, ins, o): self._lds_write(o, ins, 1)
    def i_ds_write_b64(self, ins, o): self._lds_write(o, ins, 2)
    def i_ds_write_b128(self, ins, o): self._lds_write(o, ins, 4)

    def i_ds_bpermute_b32(self, ins, o):
        addr = self.rd32(o[1])
        data = self.rd32(o[2])
        src = (addr >> U32(2)) & U32(63)
        val = data[src]
        m = self.mask()
        op = o[0]

        def commit():
            self.v[op.i][m] = val[m]
        self._issue(self.lgkm, [("v", op.i)], commit)

    # ----- global memory (saddr form: vdst, voffset, sbase  /  voffset, vdata, sbase)
    def _gaddr(self, voff, sbase, ins):
        base = self.s_rd(sbase, 64)
        return base + self.rd32(voff).astype(np.int64) + ins.mods.get("offset", 0)

    def _gload(self, ins, o, n):
        addr = self._gaddr(o[1], o[2], ins)
        m = self.mask()
        out = np.zeros((64, n), U32)
        for i in np.nonzero(m)[0]:
            a = int(addr[i])
            self.mem.check(a, 4 * n)
            out[i] = self.mem.b[a: a + 4 * n].view(U32)
        op = o[0]

        def commit():
            for k in range(n):
                self.v[op.i + k][m] = out[:, k][m]
                self.taint[op.i + k][m] = False
        self._issue(self.vmq, [("v", op.i + k) for k in range(n)], commit)

    def i_global_load_lds_dwordx4(self, ins, o):
        """voffset, sbase: every lane under EXEC moves 16 bytes from sbase + voffset + offset to LDS at M0 + offset +
        16 * lane (the instruction offset counts on both sides; measured: tools/ubench_glds.hip)"""
        addr = self._gaddr(o[0], o[1], ins)
        m = self.mask()
        dst0 = self.m0 + ins.mods.get("offset", 0)
        if dst0 + 1024 > self.lds.size:
            raise EmuError("load to LDS beyond the allocation: 0x%x" % dst0)
        got = {}
        for i in np.nonzero(m)[0]:
            a = int(addr[i])
            self.mem.check(a, 16)
            got[int(i)] = self.mem.b[a: a + 16].copy()
        rng = (dst0, dst0 + 1024)
        for lo, hi in self.lds_pending:
            if rng[0] < hi and rng[1] > lo:
                raise EmuError("two loads to LDS in flight to 0x%x" % dst0)
        self.lds_pending.append(rng)

        def commit():
            for i, b in got.items():
                self.lds[dst0 + 16 * i: dst0 + 16 * i + 16] = b
            self.lds_pending.remove(rng)
        self._issue(self.vmq, [], commit)

    def i_global_load_dword(self, ins, o): self._gload(ins, o, 1)
    def i_global_load_dwordx2(self, ins, o): self._gload(ins, o, 2)
    def i_global_load_dwordx4(self, ins, o): self._gload(ins, o, 4)

    def _gstore(self, ins, o, n):
        addr = self._gaddr(o[0], o[2], ins)
        for k in range(n):
            self._touch(Op("v", o[1].i + k), 1)
            self._tsrc |= self.taint[o[1].i + k]
        m = self.mask()
        if (self._tsrc & m).any():
            raise EmuError("a store uses a value fetched from LDS bytes a load to LDS was still writing")
        for i in np.nonzero(m)[0]:
            a = int(addr[i])
            self.mem.check(a, 4 * n)
            self.mem.b[a: a + 4 * n] = np.array([self.v[o[1].i + k][i] for k in range(n)], U32).view(np.uint8)
        if n > 2:
            for k in range(n):
                self.last_store[o[1].i + k] = self.count
        self._issue(self.vmq, [], None)

    def i_global_store_dword(self, ins, o): self._gstore(ins, o, 1)
    def i_global_store_dwordx2(self, ins, o): self._gstore(ins, o, 2)
    def i_global_store_dwordx3(self, ins, o): self._gstore(ins, o, 3)
    def i_global_store_dwordx4(self, ins, o): self._gstore(ins, o, 4)


def kernel_lds_bytes(text, name):
    m = re.search(r"\.amdhsa_kernel %s\s+\.amdhsa_group_segment_fixed_size (\d+)" % re.escape(name), text)
    return int(m.group(1))


def run_kernel(text, name, mem, kernarg_addr, n_workgroups, watch=None, parsed=None):
    """Runs workgroups 0..n-1 of a one-wave kernel, one after the other.  Returns the last wave (for its statistics)."""
    instrs, labels = parsed or parse(text)
    lds = kernel_lds_bytes(text, name)
    w = None
    for wg in range(n_workgroups):
        w = Wave(instrs, labels, mem, lds, kernarg_addr, wg, name)
        if watch:
            w.watch = watch
        w.run()
    return w
