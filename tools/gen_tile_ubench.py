#!/usr/bin/env python3
"""Generates tools/tile_ubench.hip: issue-rate microbenchmark of the block-scale tile body of gemm_q16.hip
(2 f16 MFMAs + 16 v_mul + 16 v_fmac per 32x32 tile and k-block) in several instruction orders, at 1-4 waves per SIMD.
Developer tool, not part of the product.  Registers are hard-coded inside one asm loop so that hipcc cannot reorder."""
import sys
RAND = "--rand" in sys.argv    # per-lane pseudo-random operand data and unit block scales: realistic toggling (power), not constants
ACC, T0, T1, DA, TMP, A0, A1, B0, B1, DW, LD = 0, 16, 32, 48, 64, 68, 72, 76, 80, 84, 88   # VGPR bases
P0, P1, PA, PB, NV = 48, 104, 64, 72, 120
PQ, ACC2 = 120, 152   # mxq kinds: 32 outer-product registers, second accumulator tile
SCL = LD + 15 if RAND else LD + 12          # E8M0 block-scale byte(s): 2^0 with --rand, 2^-127 (results ~0) otherwise   # outer-product variant: d0 (x) d1 by a bf16 MFMA, then 16 v_fmac


def mfma(t, first):
    a, b = (A0, B0) if first else (A1, B1)
    c = "0" if first else f"v[{t}:{t + 15}]"
    return f"v_mfma_f32_32x32x16_f16 v[{t}:{t + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], {c}"


def group(t, g):
    o = []
    for r in range(4):
        o.append(f"v_mul_f32 v{TMP + r}, v{t + 4 * g + r}, v{DA + 4 * g + r}")
    for r in range(4):
        o.append(f"v_fmac_f32 v{ACC + 4 * g + r}, v{TMP + r}, v{DW}")
    return o


def group_dpp(t, g):
    """the row scale of accumulator register r comes from lane r of the lane's own row of 16 (one register holds all 16 scales)"""
    o = []
    for r in range(4):
        o.append(f"v_mul_f32_dpp v{TMP + r}, v{DA}, v{t + 4 * g + r} row_newbcast:{4 * g + r} row_mask:0xf bank_mask:0xf")
    for r in range(4):
        o.append(f"v_fmac_f32 v{ACC + 4 * g + r}, v{TMP + r}, v{DW}")
    return o


def group_pk(t, g):
    o = []
    for r in (0, 2):
        o.append(f"v_pk_mul_f32 v[{TMP + r}:{TMP + r + 1}], v[{t + 4 * g + r}:{t + 4 * g + r + 1}], v[{DA + 4 * g + r}:{DA + 4 * g + r + 1}]")
    for r in (0, 2):
        o.append(f"v_pk_fma_f32 v[{ACC + 4 * g + r}:{ACC + 4 * g + r + 1}], v[{TMP + r}:{TMP + r + 1}], v[{DW}:{DW + 1}], v[{ACC + 4 * g + r}:{ACC + 4 * g + r + 1}] op_sel_hi:[1,0,1]")
    return o


def tile(kind, cur, nxt):
    """one tile: MFMAs produce `nxt`, VALU consumes `cur`"""
    g = [group(cur, i) for i in range(4)]
    m = [mfma(nxt, True), mfma(nxt, False)]
    if kind == "valu":
        return sum(g, [])
    if kind == "mfma":
        return m
    if kind == "mm_v":          # both MFMAs, then 32 VALU
        return m + sum(g, [])
    if kind == "m16m16":        # MFMA, 16 VALU, MFMA, 16 VALU
        return [m[0]] + g[0] + g[1] + [m[1]] + g[2] + g[3]
    if kind == "m8m24":
        return [m[0]] + g[0] + [m[1]] + g[1] + g[2] + g[3]
    if kind == "pk":
        gp = [group_pk(cur, i) for i in range(4)]
        return [m[0]] + gp[0] + gp[1] + [m[1]] + gp[2] + gp[3]
    if kind == "real":          # m16m16 + 6 extra VALU (weight expand) + 3 ds_read_b128 + wait
        ex = [f"v_and_b32 v{TMP}, 0xf000f, v{LD + (i % 4)}" if i % 2 == 0 else f"v_pk_add_f16 v{B1 + (i % 4)}, v{TMP}, v{DW}" for i in range(6)]
        ld = [f"ds_read_b128 v[{LD}:{LD + 3}], v{LD + 12}", f"ds_read_b128 v[{LD + 4}:{LD + 7}], v{LD + 12} offset:1024",
              f"ds_read_b128 v[{LD + 8}:{LD + 11}], v{LD + 12} offset:2048"]
        return [m[0]] + ld + g[0] + g[1] + [m[1]] + g[2] + ["s_waitcnt lgkmcnt(0)"] + g[3] + ex
    if kind in ("mx", "mx_only", "mx_real"):
        mm = f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{nxt}:{nxt + 15}], v[{A0}:{A0 + 5}], v[{B0}:{B0 + 5}], 0, v{SCL}, v{SCL} op_sel_hi:[0,0,0] cbsz:3 blgp:3"
        if kind == "mx_only":
            return [mm]
        if kind == "mx":
            return [mm] + sum(g, [])
        ld = [f"ds_read_b128 v[{LD}:{LD + 3}], v{LD + 12}", f"ds_read_b128 v[{LD + 4}:{LD + 7}], v{LD + 12} offset:1024",
              f"ds_read_b64 v[{LD + 8}:{LD + 9}], v{LD + 12} offset:2048"]
        return [mm] + ld + g[0] + g[1] + g[2] + ["s_waitcnt lgkmcnt(0)"] + g[3]
    if kind in ("mxk", "mxk_real"):      # MX MFMA + PACKED f32 epilogue (8 v_pk_mul + 8 v_pk_fma): half the VALU instructions, each twice as wide
        mm = f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{nxt}:{nxt + 15}], v[{A0}:{A0 + 5}], v[{B0}:{B0 + 5}], 0, v{SCL}, v{SCL} op_sel_hi:[0,0,0] cbsz:3 blgp:3"
        gp = [group_pk(cur, i) for i in range(4)]
        if kind == "mxk":
            return [mm] + sum(gp, [])
        ld = [f"ds_read_b128 v[{LD}:{LD + 3}], v{LD + 12}", f"ds_read_b128 v[{LD + 4}:{LD + 7}], v{LD + 12} offset:1024",
              f"ds_read_b64 v[{LD + 8}:{LD + 9}], v{LD + 12} offset:2048"]
        return [mm] + ld + gp[0] + gp[1] + gp[2] + ["s_waitcnt lgkmcnt(0)"] + gp[3]
    if kind in ("mxd", "mxd_real"):
        mm = f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{nxt}:{nxt + 15}], v[{A0}:{A0 + 5}], v[{B0}:{B0 + 5}], 0, v{SCL}, v{SCL} op_sel_hi:[0,0,0] cbsz:3 blgp:3"
        gd = [group_dpp(cur, i) for i in range(4)]
        if kind == "mxd":
            return [mm] + sum(gd, [])
        ld = [f"ds_read_b128 v[{LD}:{LD + 3}], v{LD + 12}", f"ds_read_b64 v[{LD + 8}:{LD + 9}], v{LD + 12} offset:2048", f"ds_read_b32 v{LD + 10}, v{LD + 12} offset:1024"]
        return [mm] + ld + gd[0] + gd[1] + gd[2] + ["s_waitcnt lgkmcnt(0)"] + gd[3]
    if kind in ("mxp_b2", "mxp_b22", "mxp_n"):
        # register-bank variants of mxp (all three v_fmac operands sit at the same index mod 4 there): _b2 = P two registers up,
        # _b22 = S and P two registers up; _n = mxp's registers with the kernel's order (MX, wait, 16 v_fmac, P MFMA; ONE P buffer)
        sb = 2 if kind == "mxp_b22" else 0
        pb = (PQ + 2, PQ + 18) if kind != "mxp_n" else (P1, P1)
        ps, pn = pb if cur == T0 else (pb[1], pb[0])
        mm = f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{nxt + sb}:{nxt + sb + 15}], v[{A0}:{A0 + 5}], v[{B0}:{B0 + 5}], 0, v{SCL}, v{SCL} op_sel_hi:[0,0,0] cbsz:3 blgp:3"
        pp = f"v_mfma_f32_32x32x16_bf16 v[{pn}:{pn + 15}], v[{PA}:{PA + 3}], v[{PB}:{PB + 3}], 0"
        f = [f"v_fmac_f32 v{ACC + r}, v{cur + sb + r}, v{ps + r}" for r in range(16)]
        if kind == "mxp_n":
            return [mm, "s_nop 7"] + f + [pp]
        return [mm] + f[:8] + [pp] + f[8:]
    if kind in ("mxp", "mxp_real", "mxp_only"):
        # S = MX MFMA; P = (weight scales) (x) (activation scales) as a K=6 bf16 MFMA; acc += S * P: 16 VALU instead of 32
        ps, pn = (P0, P1) if cur == T0 else (P1, P0)
        mm = f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{nxt}:{nxt + 15}], v[{A0}:{A0 + 5}], v[{B0}:{B0 + 5}], 0, v{SCL}, v{SCL} op_sel_hi:[0,0,0] cbsz:3 blgp:3"
        pp = f"v_mfma_f32_32x32x16_bf16 v[{pn}:{pn + 15}], v[{PA}:{PA + 3}], v[{PB}:{PB + 3}], 0"
        f = [f"v_fmac_f32 v{ACC + r}, v{cur + r}, v{ps + r}" for r in range(16)]
        if kind == "mxp_only":
            return [mm, pp]
        if kind == "mxp":
            return [mm] + f[:8] + [pp] + f[8:]
        ld = [f"ds_read_b128 v[{LD}:{LD + 3}], v{LD + 12}", f"ds_read_b128 v[{LD + 4}:{LD + 7}], v{LD + 12} offset:1024",
              f"ds_read_b64 v[{LD + 8}:{LD + 9}], v{LD + 12} offset:2048"]
        return [mm] + ld + f[:8] + [pp] + f[8:12] + ["s_waitcnt lgkmcnt(0)"] + f[12:]
    if kind in ("mxq", "mxq_real"):
        # per PAIR of tiles (the wave's two m-tiles of one n-tile and k-block): 2 MX MFMAs -> S0, S1; ONE v_mfma_f32_32x32x1_2b_f32
        # = two exact f32 outer products (activation scales) (x) (weight scales of m-tile 0 | 1) -> P (32 registers); acc += S * P:
        # 16 VALU per tile.  Single-buffered (the kernel's register budget): the partner wave fills the MFMA latency.
        s0, s1 = T0, T1
        mm = [f"v_mfma_scale_f32_32x32x64_f8f6f4 v[{t}:{t + 15}], v[{A0}:{A0 + 5}], v[{B0}:{B0 + 5}], 0, v{SCL}, v{SCL} op_sel_hi:[0,0,0] cbsz:3 blgp:3" for t in (s0, s1)]
        pq = f"v_mfma_f32_32x32x1_2b_f32 v[{PQ}:{PQ + 31}], v{DW}, v{DW + 1}, 0"
        f0 = [f"v_fmac_f32 v{ACC + r}, v{s0 + r}, v{PQ + r}" for r in range(16)]
        f1 = [f"v_fmac_f32 v{ACC2 + r}, v{s1 + r}, v{PQ + 16 + r}" for r in range(16)]
        ld = [f"ds_read_b128 v[{LD}:{LD + 3}], v{LD + 12}", f"ds_read_b64 v[{LD + 4}:{LD + 5}], v{LD + 12} offset:1024",
              f"ds_read_b32 v{LD + 8}, v{LD + 12} offset:2048"] if kind == "mxq_real" else []
        wt = ["s_waitcnt lgkmcnt(0)"] if kind == "mxq_real" else []
        # order: P first (64 cycles), the MX pair behind it; the fmacs of S0 need S0 (issued 2 MFMAs ago) and P
        return [pq] + mm + ld + ["s_nop 7", "s_nop 7"] + f0 + wt + f1
    raise ValueError(kind)


KINDS = ["valu", "mfma", "mx_only", "mx", "mx_real", "mxk", "mxk_real", "mxd", "mxd_real", "mxp_only", "mxp", "mxp_real", "mxp_b2", "mxp_b22", "mxp_n", "mxq", "mxq_real"]
src = ['// generated by tools/gen_tile_ubench.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>', '']
for k in KINDS:
    body = tile(k, T0, T1) if k.startswith("mxq") else tile(k, T0, T1) + tile(k, T1, T0)
    init = [f"v_cvt_f32_u32 v{r}, v{LD + 13}" for r in range(ACC, ACC + 16)]
    init += [f"v_mov_b32 v{r}, 0x3c004000" for r in range(A0, B1 + 4)]          # f16 pairs (2.0, 1.0)
    init += [f"v_mov_b32 v{r}, 0x3f7fe000" for r in range(DA, DA + 16)]          # ~0.9998
    init += [f"v_mov_b32 v{DW}, 0x3f801000", f"v_mov_b32 v{DW + 1}, 0x3f801000", f"v_mov_b32 v{LD + 12}, 0"]  # LD+12: LDS address 0 and (as an E8M0 byte 0) a tiny block scale
    init += [f"v_mov_b32 v{r}, 0x40400000" for r in range(T0, T1 + 16)] + [f"v_mov_b32 v{r}, 0x12345" for r in range(LD, LD + 12)]
    init += [f"v_mov_b32 v{r}, 0x3f801000" for r in list(range(P0, P0 + 16)) + list(range(P1, P1 + 16))]
    if k.startswith("mxq"):
        init += [f"v_cvt_f32_u32 v{r}, v{LD + 13}" for r in range(ACC2, ACC2 + 16)] + [f"v_mov_b32 v{r}, 0x3f801000" for r in range(PQ, PQ + 32)]
    # bf16 scale fragments as the kernel would hold them: k slots 0..5 in lanes 0..31, everything else zero
    init += [f"v_and_b32 v{LD + 14}, 63, v{LD + 13}", f"v_cmp_gt_u32 vcc, 32, v{LD + 14}", f"v_mov_b32 v{LD + 15}, 0x3c123f81"]
    init += [f"v_cndmask_b32 v{r}, 0, v{LD + 15}, vcc" for r in (PA, PA + 1, PA + 2, PB, PB + 1, PB + 2)] + [f"v_mov_b32 v{PA + 3}, 0", f"v_mov_b32 v{PB + 3}, 0"]
    if RAND:
        # hash(lane, register) -> operand bits; floats forced into [1, 2), bf16 pairs likewise, bf6 codes are any 6 bits
        def rnd(r, andm, orm, mult):
            return [f"v_mov_b32 v{LD + 14}, {mult:#x}", f"v_mul_lo_u32 v{r}, v{LD + 13}, v{LD + 14}",
                    f"v_xor_b32 v{r}, {0x9E3779B1 ^ (r * 0x85EBCA6B) & 0xFFFFFFFF:#x}, v{r}", f"v_mov_b32 v{LD + 14}, 0x2c1b3c6d",
                    f"v_mul_lo_u32 v{r}, v{r}, v{LD + 14}", f"v_and_b32 v{r}, {andm:#x}, v{r}", f"v_or_b32 v{r}, {orm:#x}, v{r}"]
        for r in list(range(A0, A0 + 6)) + list(range(B0, B0 + 6)):
            init += rnd(r, 0xFFFFFFFF, 0, 0x61c88647 + 2 * r)
        for r in list(range(DA, DA + 16)) + [DW, DW + 1]:
            if k.startswith("mxp") and DA <= r < DA + 16:
                continue                                   # P0 aliases DA in the mxp kinds
            init += rnd(r, 0x007FFFFF, 0x3F800000, 0x61c88647 + 2 * r)
        if k.startswith("mxp"):
            for r in (PA, PA + 1, PA + 2, PB, PB + 1, PB + 2):
                init += rnd(r, 0x007F007F, 0x3F803F80, 0x61c88647 + 2 * r) + [f"v_cndmask_b32 v{r}, 0, v{r}, vcc"]
        init += [f"v_mov_b32 v{SCL}, 0x7f7f7f7f"]
    clob = ", ".join(f'"v{i}"' for i in range(0, 184 if k.startswith("mxq") or k.startswith("mxp_") else NV)) + ', "vcc"'
    asm = "\\n\\t".join(init + ["s_mov_b32 s20, %1", "s_memtime s[22:23]", "s_waitcnt lgkmcnt(0)", "1:"] + body +
                        ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 15", "s_nop 15", "s_memtime s[24:25]",
                         "s_waitcnt lgkmcnt(0)", "s_sub_u32 %0, s24, s22", f"v_mov_b32 %2, v{ACC}"])
    lb = "__launch_bounds__(512) " if k.startswith("mxq") or k.startswith("mxp_") else ""
    src += [f'__global__ void {lb}k_{k}(float *out, unsigned *cyc, int n) {{', '    __shared__ float lds[1024];', '    lds[threadIdx.x & 1023] = 1.0f;',
            '    __syncthreads();', '    unsigned dt; float r;',
            f'    asm volatile("v_mov_b32 v{LD + 13}, %3\\n\\t{asm}" : "=s"(dt), "+s"(n), "=v"(r) : "v"(threadIdx.x) : {clob}, "s20", "s22", "s23", "s24", "s25", "scc", "memory");',
            '    out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds[(threadIdx.x * 7) & 1023];', '    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = dt;', '}', '']
# role split: same SIMD, one wave only MFMAs (2 per iteration-tile), its partner only the 32 VALU
body_m = tile("mfma", T0, T1) + tile("mfma", T1, T0)
body_v = tile("valu", T0, T1) + tile("valu", T1, T0)
init = [f"v_cvt_f32_u32 v{r}, v{LD + 13}" for r in range(ACC, ACC + 16)]
init += [f"v_mov_b32 v{r}, 0x3c004000" for r in range(A0, B1 + 4)] + [f"v_mov_b32 v{r}, 0x3f7fe000" for r in range(DA, DA + 16)]
init += [f"v_mov_b32 v{DW}, 0x3f801000", f"v_mov_b32 v{DW + 1}, 0x3f801000", f"v_mov_b32 v{LD + 12}, 0"]
init += [f"v_mov_b32 v{r}, 0x40400000" for r in range(T0, T1 + 16)] + [f"v_mov_b32 v{r}, 0x12345" for r in range(LD, LD + 12)]
clob = ", ".join(f'"v{i}"' for i in range(0, LD + 14))
loop = lambda b: ["s_mov_b32 s20, %1", "1:"] + b + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 15"]
for name, roles in (("split", ("m", "v")), ("split_mm", ("m", "m")), ("split_vv", ("v", "v"))):
    bodies = {"m": body_m, "v": body_v}
    a0 = "\\n\\t".join(init + loop(bodies[roles[0]]) + [f"v_mov_b32 %2, v{ACC}"])
    a1 = "\\n\\t".join(init + loop(bodies[roles[1]]) + [f"v_mov_b32 %2, v{ACC}"])
    src += [f'__global__ void k_{name}(float *out, unsigned *cyc, int n) {{', '    float r; unsigned dt = 0;',
            '    if (threadIdx.x < 256) {',
            f'        asm volatile("v_mov_b32 v{LD + 13}, %3\\n\\t{a0}" : "=s"(dt), "+s"(n), "=v"(r) : "v"(threadIdx.x) : {clob}, "s20", "scc", "memory");',
            '    } else {',
            f'        asm volatile("v_mov_b32 v{LD + 13}, %3\\n\\t{a1}" : "=s"(dt), "+s"(n), "=v"(r) : "v"(threadIdx.x) : {clob}, "s20", "scc", "memory");',
            '    }', '    out[blockIdx.x * blockDim.x + threadIdx.x] = r;', '    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = dt;', '}', '']
src += ['template <typename K> void run(const char *name, K kern, int w, float *o, unsigned *c) {',
        '    const int iters = 4000;', '    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);',
        '    kern<<<256, 256 * w>>>(o, c, 100); hipDeviceSynchronize();',
        '    hipEventRecord(e0); kern<<<256, 256 * w>>>(o, c, iters); hipEventRecord(e1); hipEventSynchronize(e1);',
        '    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned cyc; hipMemcpy(&cyc, c, 4, hipMemcpyDeviceToHost);',
        '    // one loop iteration = 2 tiles; per SIMD there are w waves',
        '    printf("%-8s waves/SIMD %d: %8.1f us  %7.1f cyc/tile/wave  %7.1f cyc/tile/SIMD  %6.2f ns/tile/SIMD  clock %.2f GHz\\n", name, w, ms * 1e3,',
        '           cyc / (2.0 * iters), cyc / (2.0 * iters) / w, ms * 1e6 / (2.0 * iters) / w, cyc / (ms * 1e6));', '}', '',
        'int main() {', '    float *o; unsigned *c; hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&c, 4);', '    for (int w = 1; w <= 4; ++w) {']
src += [(f'        if (w <= 2) run("{k}", k_{k}, w, o, c);' if k.startswith("mxq") or k.startswith("mxp_") else f'        run("{k}", k_{k}, w, o, c);') for k in KINDS]
src += ['    }',
        '    { const int iters = 4000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;',
        '      auto t = [&](const char *nm, auto kern) { kern<<<256, 512>>>(o, c, 100); hipDeviceSynchronize(); hipEventRecord(e0); kern<<<256, 512>>>(o, c, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);',
        '        printf("%-10s 2 waves/SIMD, roles split by wave: %8.1f us  %6.2f ns per (2 MFMA | 32 VALU) step per SIMD\\n", nm, ms * 1e3, ms * 1e6 / (2.0 * iters)); };',
        '      t("mfma|valu", k_split); t("mfma|mfma", k_split_mm); t("valu|valu", k_split_vv); }',
        '    return 0;', '}', '']
open(__file__.replace("gen_tile_ubench.py", "tile_ubench_rnd.hip" if RAND else "tile_ubench.hip"), "w").write("\n".join(src))
