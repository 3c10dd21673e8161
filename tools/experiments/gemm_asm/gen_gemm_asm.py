"""Generator of the hand-scheduled main loop of the 4-wave 256x256 GEMM probe (NT, k-major
operands): see ../gemm_asm_plan.md.  Writes gemm_asm_loop.inc = the body of one asm volatile
statement.  Register map (fixed by this file):
  a[0:255]            accumulators, block (ni, mi) at a[(ni*8+mi)*4 ..+3]
  v[128:159]/[160:191] fragment set X: A blocks mi=0..7 / B blocks ni=0..7 (4 VGPRs each)
  v[192:223]/[224:255] fragment set Y
  s[40:41] / s[42:43]  global base of the NEXT slab to stage (A / B), s44 slabs left to issue,
  s45 loop counter, s[46:47] scratch pointer, s48 scratch
operands: %0 vA01 %1 vA23 %2 vB01 %3 vB23 (fragment bases for ring stages 0,1 / 2,3)
          %4 voffA %5 voffB (per-lane DMA source offsets)
          %6 sA(64) %7 sB(64) %8 strideA16 %9 strideB16 %10 nslabs %11 ldsw (ring base + wave*4096)
"""
import os, sys
ORDER = os.environ.get("ORDER", "reads_first")      # reads_first | dma_first
NO_VM = os.environ.get("NO_VM") == "1"              # diagnostic (wrong results): no vmcnt wait in the loop
NO_BAR = os.environ.get("NO_BAR") == "1"            # diagnostic (wrong results): no barrier in the loop
NO_READS = os.environ.get("NO_READS") == "1"        # diagnostic (wrong results): no fragment reads in the loop
NO_DMA = os.environ.get("NO_DMA") == "1"            # diagnostic (wrong results): no LDS-DMA in the loop
STAGING = os.environ.get("STAGING", "dma")          # dma: global_load_lds | vgpr: global_load -> v[64:127] -> ds_write_b128
# STAGING=vgpr: operands %12 vW01 %13 vW23 = per-lane LDS store bases (ldsw + 16*lane, +65536)

STAGE = 32768
def fragreg(setn, op, blk):      # op 0 = A, 1 = B
    return 128 + setn * 64 + op * 32 + blk * 4

L = []
def e(s): L.append(s)

def dma(stage):
    """8 LDS-DMA instructions of one slab into ring stage `stage` (sources at s[40:41], s[42:43])."""
    out = []
    for op, (sb, voff, stride) in enumerate((("s[40:41]", "%4", "%8"), ("s[42:43]", "%5", "%9"))):
        out.append(f"s_mov_b64 s[46:47], {sb}")
        for i in range(4):
            out.append(f"s_add_u32 m0, %11, {stage * STAGE + op * 16384 + i * 1024}")
            out.append("s_nop 0")
            out.append(f"global_load_lds_dwordx4 {voff}, s[46:47]")
            if i < 3:
                out.append(f"s_add_u32 s46, s46, {stride}")
                out.append("s_addc_u32 s47, s47, 0")
    return out

def gloads(tset):
    """8 global loads of one slab into staging set tset (v[64+32*tset ..])"""
    out = []
    for op, (sb, voff, stride) in enumerate((("s[40:41]", "%4", "%8"), ("s[42:43]", "%5", "%9"))):
        out.append(f"s_mov_b64 s[46:47], {sb}")
        for i in range(4):
            r = 64 + tset * 32 + (op * 4 + i) * 4
            out.append(f"global_load_dwordx4 v[{r}:{r+3}], {voff}, s[46:47]")
            if i < 3:
                out.append(f"s_add_u32 s46, s46, {stride}")
                out.append("s_addc_u32 s47, s47, 0")
    return out

def lwrites(tset, stage):
    out = []
    vw = "%12" if stage < 2 else "%13"
    for op in range(2):
        for i in range(4):
            r = 64 + tset * 32 + (op * 4 + i) * 4
            out.append(f"ds_write_b128 {vw}, v[{r}:{r+3}] offset:{(stage & 1) * STAGE + op * 16384 + i * 1024}")
    return out

def bdma(stage):
    """STAGING=buf: the same 8 LDS-DMA instructions in MUBUF form: resource s[52:55] / s[56:59], the
    subtile's row offset + the slab's k offset in one scalar offset register each (s60..s67)"""
    out = []
    for op, (rs, voff, so) in enumerate((("s[52:55]", "%4", 60), ("s[56:59]", "%5", 64))):
        for i in range(4):
            out.append(f"s_add_u32 m0, %11, {stage * STAGE + op * 16384 + i * 1024}")
            out.append("s_nop 0")
            out.append(f"buffer_load_dwordx4 {voff}, {rs}, s{so + i} offen lds")
    return out

def bdma1(stage):
    """STAGING=buf1: ONE M0 per operand and slab; piece i rides on the instruction offset i*1024, which
    moves the LDS address AND the memory address, so its scalar offset register holds (row offset - 1024 i)"""
    out = []
    for op, (rs, voff, so) in enumerate((("s[52:55]", "%4", 60), ("s[56:59]", "%5", 64))):
        out.append(f"s_add_u32 m0, %11, {stage * STAGE + op * 16384}")
        out.append("s_nop 0")
        for i in range(4):
            out.append(f"buffer_load_dwordx4 {voff}, {rs}, s{so + i} offen offset:{i * 1024} lds")
    return out

def badvance():
    out = ["s_cmp_gt_i32 s44, 1", "s_cselect_b32 s48, 64, 0", "s_sub_i32 s44, s44, 1"]
    for r in range(60, 68): out.append(f"s_add_u32 s{r}, s{r}, s48")
    return out

def advance():
    """next slab's sources: +64 B while slabs remain to issue, else stay (harmless re-stage)."""
    return ["s_cmp_gt_i32 s44, 1", "s_cselect_b32 s48, 64, 0", "s_sub_i32 s44, s44, 1",
            "s_add_u32 s40, s40, s48", "s_addc_u32 s41, s41, 0", "s_add_u32 s42, s42, s48", "s_addc_u32 s43, s43, 0"]

def reads(setn, stage):
    """16 fragment reads of the slab in ring stage `stage` into set `setn`."""
    out = []
    va, vb = ("%0", "%2") if stage < 2 else ("%1", "%3")
    off = (stage & 1) * STAGE
    for blk in range(8):
        r = fragreg(setn, 0, blk)
        out.append(f"ds_read_b128 v[{r}:{r+3}], {va} offset:{off + blk * 1024}")
    for blk in range(8):
        r = fragreg(setn, 1, blk)
        out.append(f"ds_read_b128 v[{r}:{r+3}], {vb} offset:{off + blk * 1024}")
    return out

def mfmas(setn):
    out = []
    for mi in range(8):
        for ni in range(8):
            acc = (ni * 8 + mi) * 4
            a, b = fragreg(setn, 0, mi), fragreg(setn, 1, ni)
            out.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc+3}], v[{b}:{b+3}], v[{a}:{a+3}], a[{acc}:{acc+3}]")
    return out

def interleave(mf, aux):
    """spread the aux instructions (kept in order) over the MFMA stream, starting after the 2nd MFMA"""
    out, n, m = [], len(aux), len(mf)
    slots = m - 4
    k = 0
    for i, x in enumerate(mf):
        out.append(x)
        if i >= 1 and k < n:
            want = min(n, (i * n + slots - 1) // slots)
            while k < want:
                out.append(aux[k]); k += 1
    out += aux[k:]
    return out

def place(mf, units):
    """units: list of (gap index, [instructions]); emitted after MFMA number `gap` (kept in order per gap)"""
    by = {}
    for g, ins in units: by.setdefault(min(g, len(mf) - 1), []).extend(ins)
    out = []
    for i, x in enumerate(mf):
        out.append(x)
        out.extend(by.get(i, []))
    return out

def dma_units(stage):
    """the 8 LDS-DMA instructions of a slab as separate units (scalar preparation + the load)"""
    units = []
    for op, (sb, voff, stride) in enumerate((("s[40:41]", "%4", "%8"), ("s[42:43]", "%5", "%9"))):
        for i in range(4):
            u = []
            if i == 0: u.append(f"s_mov_b64 s[46:47], {sb}")
            u += [f"s_add_u32 m0, %11, {stage * STAGE + op * 16384 + i * 1024}", "s_nop 0", f"global_load_lds_dwordx4 {voff}, s[46:47]"]
            if i < 3: u += [f"s_add_u32 s46, s46, {stride}", "s_addc_u32 s47, s47, 0"]
            units.append(u)
    return units

STAGGER = os.environ.get("STAGGER") == "1"          # the four waves issue their LDS-DMAs in different MFMA gaps

def loop_body(wave):
    body = []
    for j in range(4):
        units = []
        du = dma_units((j + 3) & 3)
        for d, u in enumerate(du): units.append((1 + 8 * d + 2 * wave, u))          # gaps 1..63, one wave per gap
        units.append((60, advance()))
        rd = reads((j + 1) & 1, (j + 1) & 3)
        for r, x in enumerate(rd): units.append((4 * r + 3 if r < 15 else 62, [x]))
        body.extend(place(mfmas(j & 1), units))
        body += ["s_waitcnt vmcnt(8)", "s_waitcnt lgkmcnt(0)", "s_barrier"]
    return body

# ---- prologue
e("s_mov_b64 s[40:41], %6"); e("s_mov_b64 s[42:43], %7"); e("s_mov_b32 s44, %10"); e("s_lshr_b32 s45, %10, 2")
for i in range(256): e(f"v_accvgpr_write_b32 a{i}, 0")
if STAGING in ("buf", "buf1"):
    e("s_mov_b64 s[52:53], %6"); e("s_mov_b32 s54, -1"); e("s_mov_b32 s55, 0x00020000")
    e("s_mov_b64 s[56:57], %7"); e("s_mov_b32 s58, -1"); e("s_mov_b32 s59, 0x00020000")
    e("s_mov_b32 s60, 0"); e("s_mov_b32 s64, 0")
    for i in range(1, 4):
        e(f"s_add_u32 s{60 + i}, s{59 + i}, %8"); e(f"s_add_u32 s{64 + i}, s{63 + i}, %9")
    if STAGING == "buf1":
        for i in range(1, 4):
            e(f"s_sub_u32 s{60 + i}, s{60 + i}, {1024 * i}"); e(f"s_sub_u32 s{64 + i}, s{64 + i}, {1024 * i}")
    for st in range(3):
        L.extend((bdma1 if STAGING == "buf1" else bdma)(st)); L.extend(badvance())
    e("s_waitcnt vmcnt(8)"); e("s_barrier")
elif STAGING == "dma":
    for st in range(3):
        L.extend(dma(st)); L.extend(advance())
    e("s_waitcnt vmcnt(8)"); e("s_barrier")        # slabs 0 and 1 have landed, slab 2 may fly
else:
    L.extend(gloads(0)); L.extend(advance()); L.extend(gloads(1)); L.extend(advance())
    e("s_waitcnt vmcnt(0)")
    L.extend(lwrites(0, 0)); L.extend(lwrites(1, 1)); e("s_waitcnt lgkmcnt(0)")
    L.extend(gloads(1)); L.extend(advance())       # slab 2 waits in set 1 for iteration 0's stores
    e("s_barrier")
L.extend(reads(0, 0)); e("s_waitcnt lgkmcnt(0)")
if STAGGER:
    # operand %14 = wave index: four copies of the loop, the DMA gaps rotated per wave
    for wv in range(4):
        e(f"s_cmp_eq_u32 %14, {wv}"); e(f"s_cbranch_scc1 1{wv}f")
    for wv in range(4):
        e(f"1{wv}:")
        L.extend(loop_body(wv))
        e("s_sub_i32 s45, s45, 1"); e("s_cmp_gt_i32 s45, 0"); e(f"s_cbranch_scc1 1{wv}b"); e("s_branch 9f")
    e("9:")
e("1:")
# ---- four slabs per trip: slab j in stage j&3, set j&1
for j in range(0 if STAGGER else 4):
    if STAGING == "vgpr":
        # slab j+3 -> staging set j&1 (loads, early); slab j+2 (set (j+1)&1, loaded last iteration)
        # -> ring stage (j+2)&3 (stores, late: behind a wait that leaves this iteration's loads flying)
        aux = gloads(j & 1) + advance() + reads((j + 1) & 1, (j + 1) & 3) + ["s_waitcnt vmcnt(8)"] + lwrites((j + 1) & 1, (j + 2) & 3)
    elif STAGING == "buf":
        aux = bdma((j + 3) & 3) + badvance() + reads((j + 1) & 1, (j + 1) & 3)
    elif STAGING == "buf1":
        aux = bdma1((j + 3) & 3) + badvance() + reads((j + 1) & 1, (j + 1) & 3)
    elif ORDER == "dma_first":
        aux = dma((j + 3) & 3) + advance() + reads((j + 1) & 1, (j + 1) & 3)
    else:
        aux = reads((j + 1) & 1, (j + 1) & 3) + dma((j + 3) & 3) + advance()
    if os.environ.get("FILL"):                       # diagnostic: what does one scalar / vector filler per MFMA gap cost?
        aux = [os.environ["FILL"]] * int(os.environ.get("NFILL", "32"))
    if NO_READS: aux = [x for x in aux if not x.startswith("ds_read")]
    if NO_DMA: aux = [x for x in aux if x.startswith("ds_read")]
    if os.environ.get("BURST") == "1":               # diagnostic: all staging in one burst after the first MFMA
        mf = mfmas(j & 1)
        st = [x for x in aux if not x.startswith("ds_read")]
        rd = [x for x in aux if x.startswith("ds_read")]
        L.extend([mf[0]] + st + interleave(mf[1:], rd))
    else:
        L.extend(interleave(mfmas(j & 1), aux))
    if not NO_VM and STAGING != "vgpr": e("s_waitcnt vmcnt(8)")
    e("s_waitcnt lgkmcnt(0)")
    if not NO_BAR: e("s_barrier")
if not STAGGER:
    e("s_sub_i32 s45, s45, 1"); e("s_cmp_gt_i32 s45, 0"); e("s_cbranch_scc1 1b")
e("s_waitcnt vmcnt(0)"); e("s_nop 15"); e("s_nop 15")

# accumulator read-out: one statement per 16x16 block, C++ array acc[ni][mi][4]
with open("gemm_asm_readout.inc", "w") as f:
    for ni in range(8):
        for mi in range(8):
            b = (ni * 8 + mi) * 4
            body = "\\n\\t".join(f"v_accvgpr_read_b32 %{r}, a{b + r}" for r in range(4))
            outs = ", ".join(f'"=v"(acc[{ni}][{mi}][{r}])' for r in range(4))
            f.write(f'asm volatile("{body}" : {outs});\n')

with open(sys.argv[1] if len(sys.argv) > 1 else "gemm_asm_loop.inc", "w") as f:
    for s in L: f.write('"' + s + '\\n\\t"\n')
print(len(L), "lines")
