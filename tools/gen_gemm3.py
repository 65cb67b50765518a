#!/usr/bin/env python3
"""Generator of the hand-scheduled main loops of gemm_nt3 (csrc/gemm3_nt_loop.inc).

One workgroup = 4 waves (one per SIMD, the whole 512-register file each), 256x256x64 tile, wave tile 128x128 =
4x4 v_mfma_f32_32x32x16_bf16 accumulators (all 256 AGPRs), 2 LDS buffers of 64 KB filled by
`buffer_load_dwordx4 ... lds`.  Compared with the 8-wave kernel a wave reads 8 KB of fragments per 16 MFMA instead of
6 KB per 8 (one third fewer LDS bytes per flop) and there is one barrier per k tile instead of eight.

The compiler does not schedule a single in-order wave tightly enough (43 % cycle efficiency), so the
whole k loop is ONE inline-asm statement with literal registers, emitted by this script:

  k tile t lives in LDS buffer t & 1 and is consumed in 4 steps of 16 MFMA (k16 each).  Step s multiplies the
  fragment set F[s & 1] while the 8 ds_read_b128 of step s+1 fill F[(s + 1) & 1]; they sit in the first MFMA gaps of
  the step, one per gap, and are all back long before the step ends.
  The barrier of tile t stands between its steps 2 and 3: behind it buffer t & 1 is dead (step 3 runs from registers)
  and tile t+1 has landed, so step 3 reads the first fragments of tile t+1 and issues the A half (8 DMA) of tile t+2
  into the dead buffer; step 0 of the next tile issues its B half.  Every DMA therefore has >= 2 steps (1 000 cycles)
  before the wait that retires it, and a wave never waits for a load it has just issued.
  Each DMA is `s_add_u32 m0, ...` | MFMA | `buffer_load ... lds`: the MFMA is the wait state the M0 write needs.

Loop control: full bodies while t <= nk-3, then a peeled penultimate tile (no A-half DMA) and a last tile (no DMA, no
barrier); both LDS buffers are unrolled, so every LDS address is a register + immediate.

Operands (see gemm_nt3_kernel): %0..%15 acc[i][j] (AGPR tuples), %16 / %17 clock probe outputs (shader cycles and
100 MHz ticks spent in the k loop), %18 tid, %19 / %20 buffer descriptors of A / B, %21 m0, %22 n0, %23 lda, %24 ldb,
%25 nk, %26 LDS base.
"""
import sys

ACC = lambda i, j: f"%{i * 4 + j}"
TID, RSA, RSB, M0R, N0R, LDA, LDB, NK, LDS = (f"%{n}" for n in range(18, 27))

FRAG = [128, 160]                      # fragment sets: A_i at +4i, B_j at +16+4j
RADDR = lambda buf, op, kk: 192 + buf * 8 + (4 if op == "B" else 0) + kk
VOFF = {("A", 0): 208, ("A", 1): 209, ("B", 0): 210, ("B", 1): 211}     # DMA lane offsets, even / odd row group
SOFF = {"A": 36, "B": 44}              # DMA row-group offsets (8 each)
S_LDSW, S_M0SAVE, S_T, S_NK1, S_NK2 = 55, 56, 52, 53, 54
V_CLOBBER = range(128, 220)
S_CLOBBER = range(36, 72)
CYC, RT = "%16", "%17"                 # outputs: shader cycles / 100 MHz ticks spent in the k loop (clock probe)

# experiment switches (ablation variants; the shipped loop has all of them on)
OPT = dict(dma=True, read=True, barrier=True)


class Emit:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f"{name}_%=:")


def vreg4(base):
    return f"v[{base}:{base + 3}]"


def frag(fs, op, idx):
    return vreg4(FRAG[fs] + (16 if op == "B" else 0) + 4 * idx)


def dma_pair(op, g, buf):
    """(m0 write, DMA instruction) of row group g (8 rows) of operand op into LDS buffer buf."""
    const = buf * 65536 + (32768 if op == "B" else 0) + g * 1024
    rs = RSA if op == "A" else RSB
    return (f"s_add_u32 m0, s{S_LDSW}, 0x{const:x}",
            f"buffer_load_dwordx4 v{VOFF[(op, g & 1)]}, {rs}, s{SOFF[op] + g} offen lds")


def read_instr(fs, buf, kk, order):
    """ds_read_b128 of fragment `order`-th of step kk of buffer buf into fragment set fs."""
    op, idx = order
    return f"ds_read_b128 {frag(fs, op, idx)}, v{RADDR(buf, op, kk)} offset:{idx * 4096}"


READ_ORDER = [("A", 0), ("B", 0), ("B", 1), ("B", 2), ("B", 3), ("A", 1), ("A", 2), ("A", 3)]


def step(e, fs, reads, dmas, extras):
    if not OPT["dma"]:
        dmas, extras = [], []
    if not OPT["read"]:
        reads = []
    _step(e, fs, reads, dmas, extras)


def _step(e, fs, reads, dmas, extras):
    """16 MFMA on fragment set fs; `reads` (<= 8 ds_read strings) go in the first gaps, `dmas` (pairs) in the following
    ones, `extras` after them - one filler per gap."""
    pre = [None] * 16      # instruction placed BEFORE the MFMA of the gap (m0 writes)
    post = [[] for _ in range(16)]
    g = 0
    for r in reads:
        post[g].append(r)
        g += 1
    for m0w, ld in dmas:
        pre[g] = m0w
        post[g].append(ld)
        g += 1
    for x in extras:
        post[min(g, 15)].append(x)
        g += 1
    assert g <= 16 + len(extras)
    n = 0
    for i in range(4):
        for j in range(4):
            if pre[n]:
                e(pre[n])
            e(f"v_mfma_f32_32x32x16_bf16 {ACC(i, j)}, {frag(fs, 'B', j)}, {frag(fs, 'A', i)}, {ACC(i, j)}")
            for x in post[n]:
                e(x)
            n += 1


def tile_body(e, buf, kind):
    """kind: 'full' (t <= nk-3), 'penult' (t = nk-2), 'last' (t = nk-1)."""
    nb = buf ^ 1
    # step 0: fragments of step 1; B half of tile t+1 into the other buffer
    e("s_waitcnt lgkmcnt(0)")
    dm = [] if kind == "last" else [dma_pair("B", g, nb) for g in range(8)]
    ex = [] if kind == "last" else [f"v_add_u32 v{VOFF[('B', 0)]}, 0x80, v{VOFF[('B', 0)]}",
                                    f"v_add_u32 v{VOFF[('B', 1)]}, 0x80, v{VOFF[('B', 1)]}"]
    step(e, 0, [read_instr(1, buf, 1, o) for o in READ_ORDER], dm, ex)
    # step 1
    e("s_waitcnt lgkmcnt(0)")
    step(e, 1, [read_instr(0, buf, 2, o) for o in READ_ORDER], [], [])
    # step 2
    e("s_waitcnt lgkmcnt(0)")
    step(e, 0, [read_instr(1, buf, 3, o) for o in READ_ORDER], [], [])
    # barrier: this buffer is dead, tile t+1 has landed
    if kind == "last":
        e("s_waitcnt lgkmcnt(0)")
        step(e, 1, [], [], [])
        return
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if OPT["barrier"]:
        e("s_barrier")
    dm = [dma_pair("A", g, buf) for g in range(8)] if kind == "full" else []
    ex = [f"v_add_u32 v{VOFF[('A', 0)]}, 0x80, v{VOFF[('A', 0)]}",
          f"v_add_u32 v{VOFF[('A', 1)]}, 0x80, v{VOFF[('A', 1)]}"] if kind == "full" else []
    step(e, 1, [read_instr(0, nb, 0, o) for o in READ_ORDER], dm, ex)


def prologue(e):
    e("s_nop 4")
    e(f"s_mov_b32 s{S_M0SAVE}, m0")
    e(f"v_and_b32 v212, 63, {TID}")                    # lane
    e(f"v_lshrrev_b32 v213, 6, {TID}")
    e("s_nop 1")
    e("v_readfirstlane_b32 s57, v213")                 # wave id
    e("s_lshr_b32 s58, s57, 1")                        # wr
    e("s_and_b32 s59, s57, 1")                         # wc
    e("s_lshl_b32 s60, s57, 13")
    e(f"s_add_u32 s{S_LDSW}, {LDS}, s60")              # DMA destination base of this wave: rows wid*64..
    e(f"s_sub_u32 s{S_NK1}, {NK}, 1")
    e(f"s_sub_u32 s{S_NK2}, {NK}, 2")
    e(f"s_mov_b32 s{S_T}, 0")
    for op, row0, ld in (("A", M0R, LDA), ("B", N0R, LDB)):
        e("s_lshl_b32 s60, s57, 6")
        e(f"s_add_u32 s61, {row0}, s60")
        e(f"s_mul_i32 s61, s61, {ld}")
        e(f"s_lshl_b32 s{SOFF[op]}, s61, 1")           # bytes of row (row0 + wid*64)
        e(f"s_lshl_b32 s62, {ld}, 4")                  # 8 rows
        for g in range(1, 8):
            e(f"s_add_u32 s{SOFF[op] + g}, s{SOFF[op] + g - 1}, s62")
    # DMA lane offsets: lane -> row t8 = lane>>3 of the 8-row group, LDS chunk position p = lane&7 holds k-chunk
    # p ^ (t8>>1) ^ 4*(group & 1)   (nt2_swz: chunk c of row r sits at c ^ ((r>>1)&7))
    e("v_lshrrev_b32 v213, 3, v212")
    e("v_and_b32 v214, 7, v212")
    e("v_lshrrev_b32 v215, 1, v213")
    e("v_xor_b32 v214, v214, v215")
    e("v_xor_b32 v215, 4, v214")
    for op, ld in (("A", LDA), ("B", LDB)):
        e(f"v_mul_lo_u32 v216, v213, {ld}")
        e(f"v_lshl_add_u32 v217, v214, 3, v216")
        e(f"v_lshlrev_b32 v{VOFF[(op, 0)]}, 1, v217")
        e(f"v_lshl_add_u32 v217, v215, 3, v216")
        e(f"v_lshlrev_b32 v{VOFF[(op, 1)]}, 1, v217")
    # fragment read addresses: row (lane&31) of a 32-row tile, k-chunk kk*2 + (lane>>5), swizzled
    e("v_and_b32 v213, 31, v212")
    e("v_lshrrev_b32 v214, 5, v212")
    e("v_lshrrev_b32 v215, 1, v213")
    e("v_and_b32 v215, 7, v215")
    e("v_lshlrev_b32 v216, 7, v213")
    e("s_lshl_b32 s60, s58, 14")
    e(f"s_add_u32 s60, s60, {LDS}")                    # A rows of this wave: wr*128
    e("s_lshl_b32 s61, s59, 14")
    e(f"s_add_u32 s61, s61, {LDS}")
    e("s_add_u32 s61, s61, 0x8000")                    # B rows of this wave: wc*128, B region
    for kk in range(4):
        e(f"v_or_b32 v217, {kk * 2}, v214")
        e("v_xor_b32 v217, v217, v215")
        e("v_lshl_add_u32 v217, v217, 4, v216")
        e(f"v_add_u32 v{RADDR(0, 'A', kk)}, s60, v217")
        e(f"v_add_u32 v{RADDR(0, 'B', kk)}, s61, v217")
        e(f"v_add_u32 v{RADDR(1, 'A', kk)}, 0x10000, v{RADDR(0, 'A', kk)}")
        e(f"v_add_u32 v{RADDR(1, 'B', kk)}, 0x10000, v{RADDR(0, 'B', kk)}")
    # tile 0 -> buffer 0 (16 DMA), A half of tile 1 -> buffer 1
    for op in ("A", "B"):
        for g in range(8):
            m0w, ld = dma_pair(op, g, 0)
            e(m0w)
            e("s_nop 0")
            e(ld)
        e(f"v_add_u32 v{VOFF[(op, 0)]}, 0x80, v{VOFF[(op, 0)]}")
        e(f"v_add_u32 v{VOFF[(op, 1)]}, 0x80, v{VOFF[(op, 1)]}")
    e(f"s_cmp_lt_u32 {NK}, 2")
    e("s_cbranch_scc1 L_one_%=")
    for g in range(8):
        m0w, ld = dma_pair("A", g, 1)
        e(m0w)
        e("s_nop 0")
        e(ld)
    e(f"v_add_u32 v{VOFF[('A', 0)]}, 0x80, v{VOFF[('A', 0)]}")
    e(f"v_add_u32 v{VOFF[('A', 1)]}, 0x80, v{VOFF[('A', 1)]}")
    e("s_waitcnt vmcnt(8)")
    e("s_branch L_ready_%=")
    e.label("L_one")
    e("s_waitcnt vmcnt(0)")
    e.label("L_ready")
    e("s_barrier")
    for o in READ_ORDER:
        e(read_instr(0, 0, 0, o))


def main_loop(e):
    # t = tile index (even at L_loop).  kinds by t: last if t == nk-1, penult if t == nk-2, else full.
    e("s_memtime s[64:65]")
    e("s_memrealtime s[66:67]")
    e.label("L_loop")
    for buf in (0, 1):
        e(f"s_cmp_eq_u32 s{S_T}, s{S_NK1}")
        e(f"s_cbranch_scc1 L_last{buf}_%=")
        e(f"s_cmp_eq_u32 s{S_T}, s{S_NK2}")
        e(f"s_cbranch_scc1 L_pen{buf}_%=")
        tile_body(e, buf, "full")
        e(f"s_add_u32 s{S_T}, s{S_T}, 1")
    e("s_branch L_loop_%=")
    e.label("L_pen0")
    tile_body(e, 0, "penult")
    e.label("L_last1")
    tile_body(e, 1, "last")
    e("s_branch L_done_%=")
    e.label("L_pen1")
    tile_body(e, 1, "penult")
    e.label("L_last0")
    tile_body(e, 0, "last")
    e.label("L_done")
    e("s_memtime s[68:69]")
    e("s_memrealtime s[70:71]")
    e(f"s_mov_b32 m0, s{S_M0SAVE}")
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_sub_u32 {CYC}, s68, s64")
    e(f"s_sub_u32 {RT}, s70, s66")
    e("s_nop 15")                                      # MFMA result -> compiler-scheduled readers behind the statement
    e("s_nop 7")


def generate():
    e = Emit()
    prologue(e)
    main_loop(e)
    out = ["// GENERATED by tools/gen_gemm3.py - do not edit.  Main loop of gemm_nt3_kernel (one inline-asm statement)."]
    for l in e.lines:
        out.append('"' + l + '\\n\\t"')
    clob = ['"memory"', '"scc"', '"vcc"'] + [f'"v{r}"' for r in V_CLOBBER] + [f'"s{r}"' for r in S_CLOBBER]
    out.append("// clobbers")
    out.append("#define SPN_GEMM3_CLOBBERS " + ", ".join(clob))
    return "\n".join(out) + "\n"


VARIANTS = {           # name -> OPT overrides; "" is the shipped loop
    "": {},
    "_nodma": dict(dma=False),
    "_noread": dict(read=False),
    "_nobar": dict(barrier=False),
    "_mfma": dict(dma=False, read=False, barrier=False),
}


if __name__ == "__main__":
    # default: only the production loop (the one the shipped library compiles).  --ablation also writes the four
    # ablation loops (results wrong by design), used only by -DSPN_NT3_ABL experiment builds (tools/build_variant.sh);
    # they are generated files and are not kept in the repository.
    args = [a for a in sys.argv[1:] if a != "--ablation"]
    ablation = "--ablation" in sys.argv[1:]
    base = args[0] if args else "spn4cir_amd/csrc/gemm3_nt"
    for name, over in VARIANTS.items():
        if name and not ablation:
            continue
        OPT.update(dict(dma=True, read=True, barrier=True))
        OPT.update(over)
        text = generate()
        # the #define cannot live inside the asm string: split into two files
        body, clob = text.split("// clobbers\n")
        with open(f"{base}_loop{name}.inc", "w") as f:
            f.write(body)
        if not name:
            with open(f"{base}_clobbers.inc", "w") as f:
                f.write("// GENERATED by tools/gen_gemm3.py - do not edit.\n" + clob)
    print("wrote", base + "_loop*.inc")
