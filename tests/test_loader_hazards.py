"""The GEMM kernels stage their operands with inline-asm `buffer_load_dwordx4` into registers and hand-counted
`s_waitcnt vmcnt(N)` (csrc/qv_gemm256.hip fetchA/swapA, csrc/qv_gemm.hip's LD = 1 loaders): hipcc's own bookkeeping
would drain both register sets at the loop head.  That is correct only as long as the COMPILER never touches a staged
register between its load and the wait that covers it (round 2 saw exactly that: v_mov copies of the asm outputs in
front of the hand-placed wait -- wrong results).  This test checks the shipped code objects instead of trusting the
register allocator: it disassembles the gfx950 device code of both GEMM translation units and walks every kernel's
control-flow graph with the queue of vector-memory operations in flight (vmcnt counts loads and stores in issue order
on gfx9); an instruction that reads or writes a VGPR which an outstanding load still owns is a hazard.

CPU only (hipcc cross-compiles; nothing runs on a GPU)."""

import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "offline-tarteel_amd" / "build"
OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^([0-9a-f]{8,16}) <([^>]+)>:$")
INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):")


def _vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _is_vmem(op):
    return op.startswith(("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic",
                          "flat_load", "flat_store", "flat_atomic", "scratch_load", "scratch_store"))


def disassemble(obj: Path):
    """{kernel name: [(addr, opcode, operand text)]} for the gfx950 code object bundled in `obj`."""
    subprocess.run([str(OBJDUMP), "--offloading", str(obj)], check=True, capture_output=True, cwd=str(obj.parent))
    co = [p for p in obj.parent.glob(obj.name + ".*gfx950")]
    assert len(co) == 1, co
    txt = subprocess.run([str(OBJDUMP), "-d", str(co[0])], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = LABEL.match(line)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            continue
        m = INSN.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return kernels


def _is_staged(args):
    """the inline-asm staging loads: `v[a:b], vOFF, s[rsrc], sK offen` -- the K-tile advances a SCALAR offset; the
    compiler's own loads (bias vectors, residual rows) use an immediate 0 there"""
    parts = [x.strip() for x in args.split(",")]
    return len(parts) >= 4 and re.match(r"s\d+ offen", parts[3]) is not None


# ---- a small scalar machine: SGPRs that derive from the integer kernel arguments are evaluated CONCRETELY -------------
# The staging protocol is driven by scalar loop arithmetic (kt + 2 < nk, kt + 1 < nk, ...); a path-insensitive walk would
# combine branch outcomes that no K produces ("nothing was requested" with "wait for all but eight").  So the walker
# executes the scalar ALU on known values (GemmArgs' M, N, K, lda, ldw, ldo from the kernarg segment, csrc/qv_kernels.h),
# follows a branch one way when its condition is known and both ways when it is not (anything derived from a VGPR, a
# pointer or a workgroup id), and is run once per K.
M32 = 0xFFFFFFFF
KERNARG_INTS = {0x28: "M", 0x2C: "N", 0x30: "K", 0x34: "lda", 0x38: "ldw", 0x3C: "ldo"}


def _s32(x):
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


class Scalar:
    def __init__(self, ints):
        self.r = {}            # sgpr index -> known 32-bit value (unsigned); absent = unknown
        self.scc = None
        self.vcc = None        # 64-bit masks are 0, -1 (all ones), "E" (= exec, known to be non-zero) or None
        self.ints = ints
        self.exec_nz = True    # exec != 0: true for running code except right after it has been narrowed

    def key(self):
        return (tuple(sorted(self.r.items())), self.scc, self.vcc, self.exec_nz)

    def copy(self):
        c = Scalar(self.ints)
        c.r, c.scc, c.vcc, c.exec_nz = dict(self.r), self.scc, self.vcc, self.exec_nz
        return c

    # operands -----------------------------------------------------------------------------------------------
    def get32(self, tok):
        tok = tok.strip()
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return self.r.get(int(m.group(1)))
        if re.fullmatch(r"-?\d+", tok):
            return int(tok) & M32
        if re.fullmatch(r"0x[0-9a-fA-F]+", tok):
            return int(tok, 16) & M32
        return None

    def get64(self, tok):
        tok = tok.strip()
        if tok == "exec":
            return "E" if self.exec_nz else None
        if tok == "vcc":
            return self.vcc
        if tok in ("0", "-1"):
            return int(tok)
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            lo, hi = self.r.get(int(m.group(1))), self.r.get(int(m.group(2)))
            if lo == 0 and hi == 0:
                return 0
            if lo == M32 and hi == M32:
                return -1
        return None

    def set32(self, tok, v):
        m = re.fullmatch(r"s(\d+)", tok.strip())
        if m:
            if v is None:
                self.r.pop(int(m.group(1)), None)
            else:
                self.r[int(m.group(1))] = v & M32

    def set64(self, tok, v):
        tok = tok.strip()
        if tok == "vcc":
            self.vcc = v
            return
        if tok == "exec":
            self.exec_nz = True if v in (-1, "E") else None
            return
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        if m:
            for i in range(int(m.group(1)), int(m.group(2)) + 1):
                if v is None or v == "E":
                    self.r.pop(i, None)
                else:
                    self.r[i] = M32 if v == -1 else 0

    def clobber(self, tok):
        tok = tok.strip()
        if tok == "vcc":
            self.vcc = None
        elif tok == "exec":
            self.exec_nz = None
        else:
            for i in _sregs(tok):
                self.r.pop(i, None)

    # one instruction ----------------------------------------------------------------------------------------
    def step(self, op, args):
        a = [x.strip() for x in args.split(",")] if args else []
        bin32 = {"s_add_i32": lambda x, y: x + y, "s_add_u32": lambda x, y: x + y, "s_sub_i32": lambda x, y: x - y,
                 "s_sub_u32": lambda x, y: x - y, "s_mul_i32": lambda x, y: _s32(x) * _s32(y),
                 "s_and_b32": lambda x, y: x & y, "s_or_b32": lambda x, y: x | y, "s_xor_b32": lambda x, y: x ^ y,
                 "s_lshl_b32": lambda x, y: x << (y & 31), "s_lshr_b32": lambda x, y: (x & M32) >> (y & 31),
                 "s_ashr_i32": lambda x, y: _s32(x) >> (y & 31),
                 "s_min_i32": lambda x, y: min(_s32(x), _s32(y)), "s_max_i32": lambda x, y: max(_s32(x), _s32(y)),
                 "s_min_u32": lambda x, y: min(x & M32, y & M32), "s_max_u32": lambda x, y: max(x & M32, y & M32)}
        cmp = {"eq": lambda x, y: x == y, "lg": lambda x, y: x != y, "lt": lambda x, y: x < y, "le": lambda x, y: x <= y,
               "gt": lambda x, y: x > y, "ge": lambda x, y: x >= y}
        if op in bin32 and len(a) == 3:
            x, y = self.get32(a[1]), self.get32(a[2])
            v = None if x is None or y is None else bin32[op](x, y) & M32
            self.set32(a[0], v)
            if op == "s_mul_i32":
                return     # (no SCC)
            logical = op in ("s_and_b32", "s_or_b32", "s_xor_b32", "s_lshl_b32", "s_lshr_b32", "s_ashr_i32")
            self.scc = (v != 0) if (logical and v is not None) else None
            if op.startswith(("s_min", "s_max")):
                self.scc = None
            return
        if op in ("s_addk_i32", "s_mulk_i32") and len(a) == 2:
            x, y = self.get32(a[0]), self.get32(a[1])
            self.set32(a[0], None if x is None else ((x + _s32(y)) if op == "s_addk_i32" else _s32(x) * _s32(y)) & M32)
            self.scc = None
            return
        m = re.fullmatch(r"s_cmpk?_(eq|lg|lt|le|gt|ge)_(i32|u32)", op)
        if m and len(a) == 2:
            x, y = self.get32(a[0]), self.get32(a[1])
            if x is None or y is None:
                self.scc = None
            elif m.group(2) == "i32":
                if op.startswith("s_cmpk") and 0x8000 <= y <= 0xFFFF:
                    y |= 0xFFFF0000      # simm16, sign-extended
                self.scc = cmp[m.group(1)](_s32(x), _s32(y))
            else:
                self.scc = cmp[m.group(1)](x & M32, y & M32)
            return
        if op.startswith(("s_cmp", "s_bitcmp")):
            self.scc = None
            return
        if op == "s_movk_i32" and len(a) == 2:
            y = self.get32(a[1])
            self.set32(a[0], None if y is None else (y | 0xFFFF0000 if 0x8000 <= y <= 0xFFFF else y))
            return
        if op == "s_mov_b32" and len(a) == 2:
            self.set32(a[0], self.get32(a[1]))
            return
        if op == "s_mov_b64" and len(a) == 2:
            self.set64(a[0], self.get64(a[1]))
            return
        if op == "s_cselect_b32" and len(a) == 3:
            self.set32(a[0], None if self.scc is None else self.get32(a[1] if self.scc else a[2]))
            return
        if op == "s_cselect_b64" and len(a) == 3:
            self.set64(a[0], None if self.scc is None else self.get64(a[1] if self.scc else a[2]))
            return
        if op in ("s_and_b64", "s_andn2_b64", "s_or_b64", "s_orn2_b64") and len(a) == 3:
            x, y = self.get64(a[1]), self.get64(a[2])
            if op.endswith("n2_b64"):
                y = {0: -1, -1: 0}.get(y)          # (~E is unknown)
            v = None
            if op.startswith("s_and"):
                if x == 0 or y == 0:
                    v = 0
                elif x == -1:
                    v = y
                elif y == -1:
                    v = x
                elif x == "E" and y == "E":
                    v = "E"
            else:
                if x == -1 or y == -1:
                    v = -1
                elif x == 0:
                    v = y
                elif y == 0:
                    v = x
                elif x == "E" and y == "E":
                    v = "E"
            if a[0] == "exec" and op == "s_or_b64" and a[1] == "exec":
                self.exec_nz = True       # reconvergence of a structured divergent region: the saved mask is back
                self.scc = None
                return
            self.set64(a[0], v)
            self.scc = None if v is None else v != 0
            return
        if op.startswith("s_load_dword") and len(a) >= 3 and a[1] == "s[0:1]":
            off = self.get32(a[2])
            dst = sorted(_sregs(a[0]))
            for k, reg in enumerate(dst):
                name = KERNARG_INTS.get((off or 0) + 4 * k) if off is not None else None
                if name is None:
                    self.r.pop(reg, None)
                else:
                    self.r[reg] = self.ints[name] & M32
            return
        # anything else: its first operand (if scalar) becomes unknown; VALU compares write vcc
        if op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop", "s_endpgm", "s_sleep", "s_setprio",
                          "s_sendmsg", "s_setreg", "s_dcache", "s_icache", "s_trap")):
            return
        if op.startswith("s_"):
            if a:
                self.clobber(a[0])
            if not op.startswith(("s_mov", "s_cmov", "s_cselect", "s_load", "s_buffer_load", "s_mul_", "s_sext", "s_bitset",
                                  "s_getpc", "s_brev", "s_pack", "s_bfm", "s_getreg", "s_memtime", "s_memrealtime")):
                self.scc = None
            if "saveexec" in op:
                self.exec_nz = None
            return
        if op.startswith("v_cmpx"):
            self.exec_nz = None
            return
        if op.startswith("v_cmp") and op.endswith("_e32"):
            self.vcc = None
            return
        if op.startswith(("v_cmp", "v_readfirstlane", "v_readlane", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co",
                          "v_mad_u64", "v_div_scale")) or "_e64" in op:
            for t in a[:2]:
                if re.fullmatch(r"s\d+|s\[\d+:\d+\]|vcc", t):
                    self.clobber(t)
            return


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def _sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def hazards(insns, ints=None, max_states=400000):
    """Execute every path of the kernel with the FIFO of outstanding vector-memory operations (destination registers of
    loads, empty for stores; `s_waitcnt vmcnt(N)` retires the oldest until N remain) next to the scalar machine above.
    An instruction that names a VGPR which an outstanding load still owns is reported.  Exploration of a path stops once
    it is past the last loop that stages by hand and nothing hand-staged is in flight (the rest is the compiler's own
    bookkeeping)."""
    ints = ints or {"M": 8064, "N": 2048, "K": 512, "lda": 512, "ldw": 512, "ldo": 2048}
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    staged_at = [a for a, op, args in insns if op == "buffer_load_dwordx4" and _is_staged(args)]
    loop_end = 0
    for a, op, args in insns:
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(args.split()[0])
            off = off - 65536 if off >= 32768 else off
            t = a + 4 + 4 * off
            if t <= a and any(t <= x <= a for x in staged_at):
                loop_end = max(loop_end, a)
    found, seen = {}, set()
    work = [(0, (), Scalar(ints), ())]
    while work:
        i, q, sc, trail = work.pop()
        while i < len(insns):
            key = (i, q, sc.key())
            if key in seen:
                break
            seen.add(key)
            assert len(seen) <= max_states, "state space exploded: a scalar loop counter the walker cannot bound?"
            addr, op, args = insns[i]
            if staged_at and addr > loop_end and not any(st for _, st in q):
                break
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", args)
                if m:
                    n = int(m.group(1))
                    q = q[len(q) - n:] if n < len(q) else q
                i += 1
                continue
            regs = _vregs(args)
            for owner, _ in q:
                if owner & regs:
                    found.setdefault((addr, op, args), (sorted(owner & regs), " ".join(trail[-12:])))
            if _is_vmem(op):
                to_lds = "_lds_" in op or args.rstrip().endswith(" lds")      # direct-to-LDS: no register destination
                dest = frozenset(_vregs(args.split(",")[0])) if ("load" in op and not to_lds) else frozenset()
                q = (q + ((dest, op == "buffer_load_dwordx4" and _is_staged(args)),))[-64:]
            if op.startswith("s_cbranch") or op == "s_branch":
                off = int(args.split()[0])
                off = off - 65536 if off >= 32768 else off
                tgt = index.get(addr + 4 + 4 * off)
                cond = None   # True = taken, False = not taken, None = unknown
                if op == "s_branch":
                    cond = True
                elif op in ("s_cbranch_scc0", "s_cbranch_scc1") and sc.scc is not None:
                    cond = sc.scc == (op == "s_cbranch_scc1")
                elif op in ("s_cbranch_vccz", "s_cbranch_vccnz") and sc.vcc is not None:
                    cond = (sc.vcc != 0) == (op == "s_cbranch_vccnz")
                elif op in ("s_cbranch_execz", "s_cbranch_execnz") and sc.exec_nz:
                    cond = op == "s_cbranch_execnz"
                if cond is not False and tgt is not None:
                    t2 = sc.copy()
                    if op == "s_cbranch_execnz":
                        t2.exec_nz = True
                    work.append((tgt, q, t2, trail + (f"{addr:x}:T",)))
                if cond is True:
                    break
                if op == "s_cbranch_execz":
                    sc.exec_nz = True      # not taken: lanes are active from here on
                trail = trail + (f"{addr:x}:F",)
            else:
                sc.step(op, args)
            i += 1
    return [(hex(a), op, args, regs, trail) for (a, op, args), (regs, trail) in sorted(found.items())]


def _kernels_of(stem):
    obj = BUILD / f"{stem}.o"
    if not obj.exists():
        import importlib.util

        spec = importlib.util.spec_from_file_location("_qv_build", str(ROOT / "offline-tarteel_amd" / "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    return disassemble(obj)


def test_the_walker_sees_a_planted_hazard():
    """self-check of the checker on a hand-written stream: a v_mov of a staged register before the wait is found, the
    same stream with the wait in front is clean, and a loop's back edge carries the outstanding loads around."""
    bad = [(0, "buffer_load_dwordx4", "v[4:7], v1, s[0:3], s8 offen"),
           (8, "v_mov_b32_e32", "v9, v5"),
           (12, "s_waitcnt", "vmcnt(0)"),
           (16, "s_endpgm", "")]
    assert hazards(bad) and hazards(bad)[0][3] == [5]
    good = [bad[0], (8, "s_waitcnt", "vmcnt(0)"), (12, "v_mov_b32_e32", "v9, v5"), (16, "s_endpgm", "")]
    assert hazards(good) == []
    # two loads in flight, vmcnt(1) frees only the older one
    two = [(0, "buffer_load_dwordx4", "v[4:7], v1, s[0:3], s8 offen"),
           (8, "buffer_load_dwordx4", "v[8:11], v1, s[0:3], s8 offen"),
           (16, "s_waitcnt", "vmcnt(1)"),
           (20, "ds_write_b128", "v2, v[4:7]"),
           (28, "ds_write_b128", "v2, v[8:11]"),
           (36, "s_endpgm", "")]
    h = hazards(two)
    assert len(h) == 1 and h[0][3] == [8, 9, 10, 11]
    # loop: the load issued at the bottom is still in flight at the top of the next iteration
    loop = [(0, "s_waitcnt", "vmcnt(0)"),
            (4, "v_add_u32_e32", "v3, v4, v4"),
            (8, "buffer_load_dwordx4", "v[4:7], v1, s[0:3], s8 offen"),
            (16, "s_cbranch_scc1", "65532"),     # back to address 4: reads v4 while the load owns it
            (20, "s_endpgm", "")]
    assert hazards(loop)


@pytest.mark.parametrize("stem", ["qv_gemm256", "qv_gemm"])
def test_no_instruction_touches_a_staged_register_before_its_wait(stem):
    kernels = _kernels_of(stem)
    gemm = {k: v for k, v in kernels.items() if "k_gemm" in k}
    assert len(gemm) >= 10, sorted(kernels)
    n_staged = 0
    for name, insns in gemm.items():
        n_staged += sum(1 for _, op, a in insns if op == "buffer_load_dwordx4" and _is_staged(a))
        # K-tile counts around every boundary of the staging protocol (1, 2, 3 tiles: no / one / two tiles staged ahead;
        # odd and even tails of the two-tile-unrolled loader loop) plus the model's own (K = 512, 2048, 2560)
        for nk in (1, 2, 3, 4, 5, 8, 32, 40):
            K = 64 * nk
            h = hazards(insns, {"M": 8064, "N": 2048, "K": K, "lda": K, "ldw": K, "ldo": 2048})
            assert not h, (name, nk, h[:4])
    assert n_staged > 100   # the hand-staged loads are really in there (not optimised into something else)


def test_build_info_names_the_compiler():
    import ctypes

    lib = ctypes.CDLL(str(ROOT / "offline-tarteel_amd" / "libqverse.so"))
    lib.qv_build_info.restype = ctypes.c_char_p
    info = lib.qv_build_info().decode()
    assert "gfx950" in info and "clang-" in info and "hip-" in info, info
