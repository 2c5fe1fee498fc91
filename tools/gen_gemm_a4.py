#!/usr/bin/env python3
"""Generates stonkgs_amd/csrc/gemm_a4_loop.inc: the instruction stream of the K loop of the four-wave NT GEMM
(gemm_a4.hip) as inline-assembly string literals, one set per tile width.

Why generated text and not C++: with one wave per SIMD the matrix pipe is busy only while the wave's single
instruction stream presents an MFMA every 16 cycles, and hipcc's scheduler would not keep the LDS reads, LDS-DMA
issues and scalar bookkeeping inside the MFMA shadows (gemm_w4.hip, compiled C++: 54-64 % of the pipe's cycles). Here
every instruction of the loop has its place; the schedule is DATA in this script (which gap after which MFMA carries
which memory instruction) and can be re-tuned without touching the kernel.

The pipeline:
  wave tile 128 x (16 NBJ), v_mfma_f32_16x16x32_bf16, accumulators a[0 : 32 NBJ), fragments in v[128:255] (two sets);
  K tile t lives in LDS stage t & 1 (A image 256 rows x 128 B, B image BN rows x 128 B, XOR-swizzled rows);
  phase 1 of a K tile multiplies k-half 0 (set 0) while reading k-half 1 (set 1) from the same stage; once every wave has
  read the stage dry - lgkmcnt(0), barrier B1 - the LDS-DMA of K tile t+2 into THAT stage starts (buffer_load ... lds,
  M0 = destination); phase 2 multiplies k-half 1 while the rest of those pieces are issued; then vmcnt(pieces) retires
  K tile t+1 (issued one K tile earlier), barrier B2, and k-half 0 of K tile t+1 is read from the other stage into set 0.
  Every wait is "all but this K tile's pieces", so anything older - the previous tile's stores included - only makes a
  wait stronger, never weaker. Two barriers per K tile; the operand stream runs two K tiles ahead and across output tiles.
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stonkgs_amd", "csrc", "gemm_a4_loop.inc")

IMG = 32768          # one operand image of one stage
B_BASE = 65536       # B images behind the two A images


class Cfg:
    def __init__(self, nbj):
        self.nbj = nbj                 # 16-column blocks per wave
        self.bn = 32 * nbj             # tile width (two wave columns)
        self.npb = self.bn // 32       # B pieces (8 rows x 128 B) per wave and K tile
        self.ndma = 8 + self.npb
        self.nm = 8 * nbj              # MFMAs per phase
        self.tag = str(self.bn)


def acc(c, i, j):
    b = 4 * (c.nbj * i + j)
    return f"a[{b}:{b + 3}]"


def fa(s, i):
    b = 128 + 64 * s + 4 * i
    return f"v[{b}:{b + 3}]"


def fb(s, j):
    b = 160 + 64 * s + 4 * j
    return f"v[{b}:{b + 3}]"


def mfma(c, h, i, j, first):
    cc = "0" if first else acc(c, i, j)
    # D[n][m]: srcA = the weight fragment (rows n), srcB = the activation fragment (rows m): a lane then holds
    # output row m = lane & 15 and four consecutive columns n = 4 (lane >> 4) .. +3 of the 16 x 16 block
    return f"v_mfma_f32_16x16x32_bf16 {acc(c, i, j)}, {fb(h, j)}, {fa(h, i)}, {cc}"


def frag_reads(c, h, stage):
    """ds_reads of k-half h of the K tile in `stage` into fragment set h: the weight fragments first (all of them are
    needed by the first NBJ MFMAs of the phase that uses them), then the activation fragments in the order of use."""
    r = []
    for j in range(c.nbj):
        r.append(f"ds_read_b128 {fb(h, j)}, %[raB{h}] offset:{stage * IMG + j * 2048}")
    for i in range(8):
        r.append(f"ds_read_b128 {fa(h, i)}, %[raA{h}] offset:{stage * IMG + i * 2048}")
    return r


def dma_ops(c, stage):
    """(m0 setup, LDS-DMA) pairs of one K tile into `stage`."""
    ops = []
    for q in range(8):
        ops.append((f"s_add_u32 m0, %[m0a], {stage * IMG + q * 1024}",
                    f"buffer_load_dwordx4 %[voffA{q}], s[36:39], 0 offen lds"))
    for q in range(c.npb):
        ops.append((f"s_add_u32 m0, %[m0b], {stage * IMG + q * 1024}",
                    f"buffer_load_dwordx4 %[voffB{q}], s[40:43], 0 offen lds"))
    return ops


ADVANCE = ["s_add_u32 s36, s36, 128", "s_addc_u32 s37, s37, 0", "s_sub_u32 s38, s38, 128",
           "s_add_u32 s40, s40, 128", "s_addc_u32 s41, s41, 0", "s_sub_u32 s42, s42, 128"]


def ktile(c, stage, first, sched):
    """One K tile: 2 * nm MFMAs with the side instructions placed in the gaps after them (gap g = after MFMA g). A gap
    carries at most ONE memory instruction (an LDS read or an LDS-DMA issue): `place` takes the first free gap at or
    after the one asked for. `ablate` (timing experiments only, results are garbage): "dma" / "reads" leave those out."""
    nm = c.nm
    ablate = sched.get("ablate", ())
    side = [[] for _ in range(2 * nm)]
    busy = [False] * (2 * nm)

    def place(g, ins):
        while busy[g]:
            g += 1
        busy[g] = True
        side[g].append(ins)
        return g

    # phase 1: k-half 1 of this K tile -> set 1
    g = sched["read1_start"]
    last_read = g
    for ins in frag_reads(c, 1, stage):
        if "reads" not in ablate:
            last_read = place(g, ins)
        g += sched["read_step"]
    b1 = sched["b1_gap"]
    assert last_read < b1, "reads must be issued before B1"
    side[b1].append("s_waitcnt lgkmcnt(0)")
    side[b1].append("s_barrier")
    # B2: K tile t+1 (issued one K tile earlier) has landed; behind it, k-half 0 of K tile t+1 from the other stage -> set 0
    # (placed first: its gaps are fixed, the LDS-DMA issues then take the free gaps around them)
    b2 = nm + sched["b2_gap"]
    g = b2 + 1
    step2 = sched.get("read2_step", sched["read_step"])
    for ins in frag_reads(c, 0, stage ^ 1):
        if "reads" not in ablate:
            last_read = place(g, ins)
        g += step2
    assert last_read <= 2 * nm - 2, (last_read, 2 * nm)
    side[2 * nm - 2].append("s_waitcnt lgkmcnt(0)")
    # LDS-DMA of K tile t+2 into this stage, from B1 on: piece n in a gap of its own, the next piece's M0 one gap later (an
    # MFMA between a piece's issue and the M0 write behind it, and at least one between that write and its use)
    g = b1 + 2
    ops = dma_ops(c, stage)
    assert sched["dma_step"] >= 2
    m0_gap = b1 + 1
    last_dma = b1
    for n, (m0, ld) in enumerate(ops):
        if "dma" in ablate:
            break
        side[m0_gap].append(m0)
        at = place(max(g, m0_gap + 1), ld)
        m0_gap = at + 1
        last_dma = at
        g = at + sched["dma_step"]
    assert last_dma + 2 < 2 * nm, last_dma
    side[last_dma + 1] += ADVANCE[0:3]       # (s_add / s_addc adjacent: nothing between them writes SCC)
    side[last_dma + 2] += ADVANCE[3:6]
    # the wait of B2 counts what THIS K tile has issued before it: all but those pieces must be done
    issued = sum(1 for gg in range(b2) for ins in side[gg] if ins.startswith("buffer_load"))   # (gap b2 itself: wait first)
    if "dma" not in ablate:
        side[b2].insert(0, f"s_waitcnt vmcnt({issued})")
    side[b2].insert(1 if "dma" not in ablate else 0, "s_barrier")
    out = []
    n = 0
    for h in range(2):
        for i in range(8):
            for j in range(c.nbj):
                out.append(mfma(c, h, i, j, first and h == 0))
                out.extend(side[n])
                n += 1
    return out


SWITCH = ["s_cmp_eq_u32 %[rem], 1",          # the last pair of K tiles prefetches the NEXT output tile's first two
          "s_cselect_b32 s36, %[nal], s36", "s_cselect_b32 s37, %[nah], s37", "s_cselect_b32 s38, %[nan], s38",
          "s_cselect_b32 s40, %[nbl], s40", "s_cselect_b32 s41, %[nbh], s41", "s_cselect_b32 s42, %[nbn], s42"]


def tile_asm(c, sched):
    t = []
    t += frag_reads(c, 0, 0)                 # K tile 0 landed and was waited for by the previous block / the prologue
    t += SWITCH
    t.append("s_waitcnt lgkmcnt(0)")
    t += ktile(c, 0, True, sched)
    t += ktile(c, 1, False, sched)
    t += ["s_sub_u32 %[rem], %[rem], 1", "s_cmp_eq_u32 %[rem], 0", "s_cbranch_scc1 L_a4_end_%="]
    t.append("L_a4_loop_%=:")
    t += SWITCH
    t += ktile(c, 0, False, sched)
    t += ktile(c, 1, False, sched)
    t += ["s_sub_u32 %[rem], %[rem], 1", "s_cmp_lg_u32 %[rem], 0", "s_cbranch_scc1 L_a4_loop_%="]
    t.append("L_a4_end_%=:")
    t += ["s_nop 7", "s_nop 7"]              # MFMA results -> the compiler's v_accvgpr_read (it cannot see into this block)
    return t


def prologue_asm(c):
    """First tile of a workgroup: K tiles 0 and 1 into stages 0 and 1, cursor left at K tile 2."""
    t = []
    for stage in range(2):
        for m0, ld in dma_ops(c, stage):
            t += [m0, "s_nop 0", ld]
        t += ADVANCE
    t += [f"s_waitcnt vmcnt({c.ndma})", "s_barrier"]
    return t


def emit(name, lines):
    body = "\n".join('  "' + ln + '\\n"' for ln in lines)
    return f"#define {name} \\\n" + body.replace("\n", " \\\n") + "\n"



# ===================================================================================== the weight-gradient (TN) form
# gemm_tn_a4.hip: C[M', N'] += A[T, M']^T . B[T, N'] - the contraction runs over the ROWS (tokens) of both operands. Same
# pipeline and register map as above (256 x 256 tiles, accumulators a[0:255], fragments v[128:255]); what differs:
#  * a K tile is 64 token rows x 512 B per operand; a piece (1 KiB) = TWO token rows, so the image is again lane-linear and
#    every piece reads eight full 128-byte lines; rows are 512 B apart (all rows alias the same banks), so the 32-byte
#    feature-block segments of a row are XOR-permuted by f(row) = (row & 3) | ((row >> 3) & 1) << 2 - on the source;
#  * a fragment = two transposed reads (ds_read_b64_tr_b16: 4 token rows x 16 features per 16-lane group, lane i receives
#    feature i): k group g of the 32-token k-half = tokens 8 g .. 8 g + 7, rows 8 g + {0-3} then 8 g + {4-7}; the half-wave's
#    eight rows land in eight different 32-byte bank slots. One per-lane address per feature block (the permutation is
#    per lane), everything else immediates;
#  * srcA = the A (dY) fragment: D[m][n], a lane holds output column n - what the atomic epilogue wants;
#  * the bias gradient (column sums of dY): one more MFMA per A block against an all-ones operand into VGPR accumulators
#    the compiler owns, on every ntn-th K tile (a second copy of each K-tile body, chosen by a scalar branch);
#  * the cursors carry 64-bit "bytes left" counts (an operand can exceed 4 GB): num_records = clamp(bytes left).
TN_IMG = 32768


def tn_frag_reads(h, stage):
    r = []
    for j in range(8):
        b = 160 + 64 * h + 4 * j
        off = stage * TN_IMG + h * 16384
        r.append([f"ds_read_b64_tr_b16 v[{b}:{b + 1}], %[taB{j}] offset:{off}",
                  f"ds_read_b64_tr_b16 v[{b + 2}:{b + 3}], %[taB{j}] offset:{off + 2048}"])
    for i in range(8):
        b = 128 + 64 * h + 4 * i
        off = stage * TN_IMG + h * 16384
        r.append([f"ds_read_b64_tr_b16 v[{b}:{b + 1}], %[taA{i}] offset:{off}",
                  f"ds_read_b64_tr_b16 v[{b + 2}:{b + 3}], %[taA{i}] offset:{off + 2048}"])
    return r


def tn_clamp(rlo, rhi, nrec):
    return [f"s_cmp_lt_i32 {rhi}, 0", f"s_cselect_b32 s48, 0, {rlo}", f"s_cmp_gt_i32 {rhi}, 0",
            f"s_cselect_b32 s48, 0xfffff000, s48", f"s_min_u32 {nrec}, s48, 0xfffff000"]


TN_ADVANCE_A = ["s_add_u32 s36, s36, %[stepa]", "s_addc_u32 s37, s37, 0", "s_sub_u32 %[ralo], %[ralo], %[stepa]",
                "s_subb_u32 %[rahi], %[rahi], 0"] + tn_clamp("%[ralo]", "%[rahi]", "s38")
TN_ADVANCE_B = ["s_add_u32 s40, s40, %[stepb]", "s_addc_u32 s41, s41, 0", "s_sub_u32 %[rblo], %[rblo], %[stepb]",
                "s_subb_u32 %[rbhi], %[rbhi], 0"] + tn_clamp("%[rblo]", "%[rbhi]", "s42")
# the last pair of K tiles of a work item prefetches the NEXT item's first two: cursor := the next item's
TN_SWITCH = (["s_cmp_eq_u32 %[rem], 1", "s_cbranch_scc0 L_tn_nosw_{u}_%=",
              "s_mov_b32 s36, %[nal]", "s_mov_b32 s37, %[nah]", "s_mov_b32 %[ralo], %[nralo]", "s_mov_b32 %[rahi], %[nrahi]",
              "s_mov_b32 s40, %[nbl]", "s_mov_b32 s41, %[nbh]", "s_mov_b32 %[rblo], %[nrblo]", "s_mov_b32 %[rbhi], %[nrbhi]"]
             + tn_clamp("%[ralo]", "%[rahi]", "s38") + tn_clamp("%[rblo]", "%[rbhi]", "s42") + ["L_tn_nosw_{u}_%=:"])


def tn_ktile(c, stage, first, sched, duty):
    nm = 64
    side = [[] for _ in range(2 * nm)]
    busy = [False] * (2 * nm)

    def place(g, ins):
        while busy[g]:
            g += 1
        busy[g] = True
        side[g] += ins if isinstance(ins, list) else [ins]
        return g

    g = sched["read1_start"]
    last_read = g
    for ins in tn_frag_reads(1, stage):
        last_read = place(g, ins)
        g += sched["read_step"]
    b1 = sched["b1_gap"]
    assert last_read < b1
    side[b1] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    b2 = nm + sched["b2_gap"]
    g = b2 + 1
    for ins in tn_frag_reads(0, stage ^ 1):
        last_read = place(g, ins)
        g += sched.get("read2_step", sched["read_step"])
    assert last_read <= 2 * nm - 2, (last_read, 2 * nm)
    side[2 * nm - 2].append("s_waitcnt lgkmcnt(0)")
    ops = dma_ops(c, stage)
    g = b1 + 2
    m0_gap = b1 + 1
    last_dma = b1
    for m0, ld in ops:
        side[m0_gap].append(m0)
        at = place(max(g, m0_gap + 1), ld)
        m0_gap = at + 1
        last_dma = at
        g = at + sched["dma_step"]
    assert last_dma + 3 < 2 * nm, last_dma
    side[last_dma + 1] += TN_ADVANCE_A
    side[last_dma + 2] += TN_ADVANCE_B
    issued = sum(1 for gg in range(b2) for ins in side[gg] if ins.startswith("buffer_load"))
    side[b2].insert(0, f"s_waitcnt vmcnt({issued})")
    side[b2].insert(1, "s_barrier")
    out = []
    n = 0
    for h in range(2):
        for i in range(8):
            for j in range(8):
                fa_, fb_ = fa(h, i), fb(h, j)
                cc = "0" if (first and h == 0) else acc(c, i, j)
                out.append(f"v_mfma_f32_16x16x32_bf16 {acc(c, i, j)}, {fa_}, {fb_}, {cc}")
                out.extend(side[n])
                n += 1
            if duty:   # column sums of dY: A fragment x ones, into the compiler's VGPR accumulators
                out.append(f"v_mfma_f32_16x16x32_bf16 %[bacc{i}], {fa(h, i)}, %[ones], %[bacc{i}]")
    return out


def tn_position(c, stage, first, sched, uid):
    """One K tile, in its two forms, and the bias-duty bookkeeping around them."""
    t = ["s_cmp_eq_u32 %[bph], 0", f"s_cbranch_scc1 L_tn_duty_{uid}_%="]
    t += tn_ktile(c, stage, first, sched, False)
    t += [f"s_branch L_tn_join_{uid}_%=", f"L_tn_duty_{uid}_%=:"]
    t += tn_ktile(c, stage, first, sched, True)
    t += [f"L_tn_join_{uid}_%=:", "s_add_u32 %[bph], %[bph], 1", "s_cmp_eq_u32 %[bph], %[ntn]", "s_cselect_b32 %[bph], 0, %[bph]"]
    return t


def tn_tile_asm(c, sched):
    t = []
    for ins in tn_frag_reads(0, 0):
        t += ins
    t += [x.replace("{u}", "a") for x in TN_SWITCH]
    t.append("s_waitcnt lgkmcnt(0)")
    t += tn_position(c, 0, True, sched, "p0")
    t += tn_position(c, 1, False, sched, "p1")
    t += ["s_sub_u32 %[rem], %[rem], 1", "s_cmp_eq_u32 %[rem], 0", "s_cbranch_scc1 L_tn_end_%="]
    t.append("L_tn_loop_%=:")
    t += [x.replace("{u}", "b") for x in TN_SWITCH]
    t += tn_position(c, 0, False, sched, "l0")
    t += tn_position(c, 1, False, sched, "l1")
    t += ["s_sub_u32 %[rem], %[rem], 1", "s_cmp_lg_u32 %[rem], 0", "s_cbranch_scc1 L_tn_loop_%="]
    t.append("L_tn_end_%=:")
    t += ["s_nop 7", "s_nop 7"]
    return t


def tn_prologue_asm(c):
    t = []
    for stage in range(2):
        for m0, ld in dma_ops(c, stage):
            t += [m0, "s_nop 0", ld]
        t += TN_ADVANCE_A + TN_ADVANCE_B
    t += [f"s_waitcnt vmcnt({c.ndma})", "s_barrier"]
    return t


TN_SCHED = {"read1_start": 1, "read_step": 1, "b1_gap": 20, "dma_step": 6, "b2_gap": 30}

# Where the side instructions sit (gap = after that MFMA of the K tile). Chosen on MI355X by tools/a4_sweep.py (round 4,
# profiles/r04_a4_sweep.md): an LDS-DMA issue is the expensive instruction of this loop (8192^3: 587 us with neither reads nor
# DMA, 619 with the reads, 820 with the DMA one gap in three, 750 one gap in six) - so the pieces are spread as far apart
# as the K tile allows; the reads cost little wherever they are; B1 as early as the k-half-1 reads allow.
SCHED = {
    8: {"read1_start": 1, "read_step": 1, "b1_gap": 20, "dma_step": 6, "b2_gap": 30},
    6: {"read1_start": 1, "read_step": 1, "b1_gap": 18, "dma_step": 4, "b2_gap": 28},
}


def generate(path, overrides=None):
    """Writes the include file; `overrides` = {8: {...}, 6: {...}} replaces schedule entries (tools/a4_sweep.py)."""
    parts = ["// GENERATED by tools/gen_gemm_a4.py - do not edit; the schedule is described there and in DESIGN.md 4.3.\n"
             "// Operands: %[voffA0..7] %[voffB0..] per-lane source offsets; %[raA0/1] %[raB0/1] fragment read addresses of\n"
             "// k-half 0 / 1; %[m0a] %[m0b] LDS destinations of this wave's first piece; %[rem] K-tile pairs left;\n"
             "// %[nal/nah/nan/nbl/nbh/nbn] the next output tile's buffer words; s[36:39] / s[40:43] the operand cursors.\n"]
    for nbj in (8, 6):
        c = Cfg(nbj)
        sched = dict(SCHED[nbj])
        sched.update((overrides or {}).get(nbj, {}))
        parts.append(emit(f"STONK_A4_PROLOGUE_{c.tag}", prologue_asm(c)))
        parts.append(emit(f"STONK_A4_TILE_{c.tag}", tile_asm(c, sched)))
    c = Cfg(8)
    tn_sched = dict(TN_SCHED)
    tn_sched.update((overrides or {}).get("tn", {}))
    parts.append("// ---- weight-gradient form (gemm_tn_a4.hip): %[taA0..7] %[taB0..7] transposed-read addresses per feature block,\n"
                 "// %[ralo/rahi/rblo/rbhi] bytes left in the operands from the cursors, %[stepa/stepb] bytes per K tile,\n"
                 "// %[nal/nah/nralo/nrahi/nbl/nbh/nrblo/nrbhi] the next work item's cursors, %[bph] / %[ntn] bias duty phase,\n"
                 "// %[bacc0..7] bias accumulators, %[ones] the all-ones operand; s48 scratch.\n")
    parts.append(emit("STONK_TN_A4_PROLOGUE", tn_prologue_asm(c)))
    parts.append(emit("STONK_TN_A4_TILE", tn_tile_asm(c, tn_sched)))
    text = "\n".join(parts)
    if path is None:
        return text
    with open(path, "w") as f:
        f.write(text)
    return text


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        ok = open(OUT).read() == generate(None)
        print("gemm_a4_loop.inc is", "up to date" if ok else "STALE")
        sys.exit(0 if ok else 1)
    text = generate(OUT)
    print("wrote", os.path.normpath(OUT), len(text), "bytes")


if __name__ == "__main__":
    main()
