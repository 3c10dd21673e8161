"""Generator, variant "B direct": the A operand goes through the LDS ring (4 LDS-DMA + 8 ds_read_b128 per
slab and wave), the B fragments come straight from global memory into registers (8 buffer_load_dwordx4 per
slab and wave, two slabs ahead, four register sets).  Halves the LDS reads (the power) and the LDS-DMAs of
gen_gemm_asm.py; costs 8 plain loads per slab.  See ../gemm_asm_plan.md.
  a[0:255] accumulators (ni*8+mi)*4;  A sets v[64:95] / v[96:127];  B sets v[128+32s ..], s = 0..3
operands: %0 vA01 %1 vA23 (A fragment bases, ring stages 0,1 / 2,3)  %2 voffA (DMA source offset)
          %3 voffBd (B fragment lane offset)  %4 sA(64) %5 sBw(64: B rows of this wave)  %6 strideA16
          %7 strideB16  %8 nslabs  %9 ldsw
"""
import os, sys
STAGE = 32768
NO_READS = os.environ.get("NO_READS") == "1"
NO_DMA = os.environ.get("NO_DMA") == "1"
NO_BL = os.environ.get("NO_BL") == "1"
L = []
def e(s): L.append(s)
def afrag(setn, blk): return 64 + setn * 32 + blk * 4
def bfrag(setn, blk): return 128 + setn * 32 + blk * 4

def dmaA(stage):
    out = ["s_mov_b64 s[46:47], s[40:41]"]
    for i in range(4):
        out += [f"s_add_u32 m0, %9, {stage * STAGE + i * 1024}", "s_nop 0", "global_load_lds_dwordx4 %2, s[46:47]"]
        if i < 3: out += ["s_add_u32 s46, s46, %6", "s_addc_u32 s47, s47, 0"]
    return out
def advA():
    return ["s_cmp_gt_i32 s44, 1", "s_cselect_b32 s48, 64, 0", "s_sub_i32 s44, s44, 1", "s_add_u32 s40, s40, s48", "s_addc_u32 s41, s41, 0"]
def loadsB(setn):
    return [f"buffer_load_dwordx4 v[{bfrag(setn, ni)}:{bfrag(setn, ni) + 3}], %3, s[56:59], s{64 + ni} offen" for ni in range(8)]
def advB():
    return ["s_cmp_gt_i32 s49, 1", "s_cselect_b32 s48, 64, 0", "s_sub_i32 s49, s49, 1"] + [f"s_add_u32 s{64 + ni}, s{64 + ni}, s48" for ni in range(8)]
def readsA(setn, stage):
    va = "%0" if stage < 2 else "%1"
    off = (stage & 1) * STAGE
    return [f"ds_read_b128 v[{afrag(setn, b)}:{afrag(setn, b) + 3}], {va} offset:{off + b * 1024}" for b in range(8)]
def mfmas(aset, bset):
    out = []
    for mi in range(8):
        for ni in range(8):
            acc = (ni * 8 + mi) * 4
            a, b = afrag(aset, mi), bfrag(bset, ni)
            out.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc+3}], v[{b}:{b+3}], v[{a}:{a+3}], a[{acc}:{acc+3}]")
    return out
def interleave(mf, aux):
    out, n, m, k = [], len(aux), len(mf), 0
    slots = m - 4
    for i, x in enumerate(mf):
        out.append(x)
        if i >= 1 and k < n:
            want = min(n, (i * n + slots - 1) // slots)
            while k < want:
                out.append(aux[k]); k += 1
    out += aux[k:]
    return out

e("s_mov_b64 s[40:41], %4"); e("s_mov_b32 s44, %8"); e("s_mov_b32 s49, %8"); e("s_lshr_b32 s45, %8, 2")
e("s_mov_b64 s[56:57], %5"); e("s_mov_b32 s58, -1"); e("s_mov_b32 s59, 0x00020000")
e("s_mov_b32 s64, 0")
for ni in range(1, 8): e(f"s_add_u32 s{64 + ni}, s{63 + ni}, %7")
for i in range(256): e(f"v_accvgpr_write_b32 a{i}, 0")
for st in range(3):
    L.extend(dmaA(st)); L.extend(advA())
L.extend(loadsB(0)); L.extend(advB()); L.extend(loadsB(1)); L.extend(advB())
e("s_waitcnt vmcnt(0)"); e("s_barrier")
L.extend(readsA(0, 0)); e("s_waitcnt lgkmcnt(0)")
e("1:")
for j in range(4):
    aux = loadsB((j + 2) & 3) + advB() + dmaA((j + 3) & 3) + advA() + readsA((j + 1) & 1, (j + 1) & 3)
    if NO_READS: aux = [x for x in aux if not x.startswith("ds_read")]
    if NO_DMA: aux = [x for x in aux if not ("global_load_lds" in x or x.startswith("s_add_u32 m0") or x == "s_nop 0")]
    if NO_BL: aux = [x for x in aux if not x.startswith("buffer_load")]
    L.extend(interleave(mfmas(j & 1, j & 3), aux))
    e("s_waitcnt vmcnt(12)"); e("s_waitcnt lgkmcnt(0)"); e("s_barrier")
e("s_sub_i32 s45, s45, 1"); e("s_cmp_gt_i32 s45, 0"); e("s_cbranch_scc1 1b")
e("s_waitcnt vmcnt(0)"); e("s_nop 15"); e("s_nop 15")
with open("gemm_asm_bd_loop.inc", "w") as f:
    for s in L: f.write('"' + s + '\\n\\t"\n')
print(len(L), "lines")
