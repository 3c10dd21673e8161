"""Generator of the 32x32x16 variant of the hand-scheduled main loop (see ../gemm_asm_plan.md and
gen_gemm_asm.py).  A v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32 cycles and holds
the wave's issue for 8: 24 cycles per gap for the LDS reads / staging / scalar work that the
16x16x32 form (8 free cycles per gap) could not hide.
  a[0:255]   accumulators, block (ni, mi) (4 x 4 of 32x32) at a[(ni*4+mi)*16 ..+15]
  v[128:159] / v[160:191]  fragment set X: A (k16 step h, row block rb) at 128 + (h*4+rb)*4, B at 160 + ...
  v[192:255]               set Y
  LDS image: 16-row x 64-B subtiles (one LDS-DMA instruction each), the 16-B chunk c of row r
  (r = row inside its 32-row block) at chunk slot c ^ ((r >> 3) & 3): conflict-free for the
  four 16-lane groups of ds_read_b128 when lane l reads row l & 31, chunk 2h + (l >> 5).
operands: %0 vA0 %1 vA0hi %2 vA1 %3 vA1hi (A fragment bases, k16 step 0 / 1, ring stages 0,1 / 2,3)
          %4 vB0 %5 vB0hi %6 vB1 %7 vB1hi
          %8 voffAe %9 voffAo %10 voffBe %11 voffBo (DMA source offsets, even / odd subtile)
          %12 sA(64) %13 sB(64) %14 strideA16 %15 strideB16 %16 nslabs %17 ldsw
"""
import os, sys
STAGE = 32768
NO_READS = os.environ.get("NO_READS") == "1"
NO_DMA = os.environ.get("NO_DMA") == "1"
L = []
def e(s): L.append(s)

def frag(setn, op, h, rb): return 128 + setn * 64 + op * 32 + (h * 4 + rb) * 4

def dma(stage):
    out = []
    for op, (sb, voe, voo, stride) in enumerate((("s[40:41]", "%8", "%9", "%14"), ("s[42:43]", "%10", "%11", "%15"))):
        out.append(f"s_mov_b64 s[46:47], {sb}")
        for i in range(4):                      # this wave's subtiles 4w + i: parity = i & 1
            out.append(f"s_add_u32 m0, %17, {stage * STAGE + op * 16384 + i * 1024}")
            out.append("s_nop 0")
            out.append(f"global_load_lds_dwordx4 {voo if i & 1 else voe}, s[46:47]")
            if i < 3:
                out.append(f"s_add_u32 s46, s46, {stride}")
                out.append("s_addc_u32 s47, s47, 0")
    return out

def advance():
    return ["s_cmp_gt_i32 s44, 1", "s_cselect_b32 s48, 64, 0", "s_sub_i32 s44, s44, 1",
            "s_add_u32 s40, s40, s48", "s_addc_u32 s41, s41, 0", "s_add_u32 s42, s42, s48", "s_addc_u32 s43, s43, 0"]

def reads(setn, stage):
    out = []
    hi = stage >= 2
    off = (stage & 1) * STAGE
    for op, bases in enumerate(((("%0", "%1"), ("%2", "%3")), (("%4", "%5"), ("%6", "%7")))):
        for h in range(2):
            vb = bases[h][1 if hi else 0]
            for rb in range(4):
                r = frag(setn, op, h, rb)
                out.append(f"ds_read_b128 v[{r}:{r+3}], {vb} offset:{off + rb * 2048}")
    return out

def mfmas(setn):
    out = []
    for h in range(2):
        for mi in range(4):
            for ni in range(4):
                acc = (ni * 4 + mi) * 16
                a, b = frag(setn, 0, h, mi), frag(setn, 1, h, ni)
                out.append(f"v_mfma_f32_32x32x16_bf16 a[{acc}:{acc+15}], v[{b}:{b+3}], v[{a}:{a+3}], a[{acc}:{acc+15}]")
    return out

def interleave(mf, aux):
    out, n, m, k = [], len(aux), len(mf), 0
    slots = m - 2
    for i, x in enumerate(mf):
        out.append(x)
        if k < n:
            want = min(n, ((i + 1) * n + slots - 1) // slots)
            while k < want:
                out.append(aux[k]); k += 1
    out += aux[k:]
    return out

e("s_mov_b64 s[40:41], %12"); e("s_mov_b64 s[42:43], %13"); e("s_mov_b32 s44, %16"); e("s_lshr_b32 s45, %16, 2")
for i in range(256): e(f"v_accvgpr_write_b32 a{i}, 0")
for st in range(3):
    L.extend(dma(st)); L.extend(advance())
e("s_waitcnt vmcnt(8)"); e("s_barrier")
L.extend(reads(0, 0)); e("s_waitcnt lgkmcnt(0)")
e("1:")
for j in range(4):
    aux = dma((j + 3) & 3) + advance() + reads((j + 1) & 1, (j + 1) & 3)
    if NO_READS: aux = [x for x in aux if not x.startswith("ds_read")]
    if NO_DMA: aux = [x for x in aux if x.startswith("ds_read")]
    L.extend(interleave(mfmas(j & 1), aux))
    e("s_waitcnt vmcnt(8)"); e("s_waitcnt lgkmcnt(0)"); e("s_barrier")
e("s_sub_i32 s45, s45, 1"); e("s_cmp_gt_i32 s45, 0"); e("s_cbranch_scc1 1b")
e("s_waitcnt vmcnt(0)"); e("s_nop 15"); e("s_nop 15")

with open("gemm_asm32_readout.inc", "w") as f:
    for ni in range(4):
        for mi in range(4):
            for q in range(4):
                b = (ni * 4 + mi) * 16 + q * 4
                body = "\\n\\t".join(f"v_accvgpr_read_b32 %{r}, a{b + r}" for r in range(4))
                outs = ", ".join(f'"=v"(acc[{ni}][{mi}][{q}][{r}])' for r in range(4))
                f.write(f'asm volatile("{body}" : {outs});\n')
with open("gemm_asm32_loop.inc", "w") as f:
    for s in L: f.write('"' + s + '\\n\\t"\n')
print(len(L), "lines")
