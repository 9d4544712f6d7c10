"""The kernel plan (csrc/plan.cpp, ggml_hip_mm_plan) without a GPU: ONE decision per product, and the invariant the multi-GPU row split
stands on -- the order of an element's additions (tree_id) is a function of (type, K, N), never of the number of weight rows M -- checked
over a sweep of type x K x N x M that includes vocabulary-sized matrices and the stated 32-bit-offset exception (the
three shard-identity violations of round 3 were found by GPU sweeps only; each of them fails this test in the build container)."""
import ctypes as C

import pytest

from ggmlsharp_amd import _lib

F32, F16, Q4_0, Q4_1, Q4_2, Q5_0, Q5_1, Q8_0, Q5_K, Q4_K, Q6_K = 0, 1, 2, 3, 4, 6, 7, 8, 113, 112, 114
QUANT = (Q4_0, Q4_1, Q4_2, Q5_0, Q5_1, Q8_0)
WIDE, EPI, PERSIST, Q8K, NEEDS_WORK, MIN_PIECES = 1, 2, 4, 8, 16, 32
FAM = {"gemv_fused": 1, "gemv_rows": 2, "k3s_mx": 3, "k3s_i8": 4, "k3p_mx": 5, "k3p_i8": 6, "mx": 7, "f16": 8, "i8": 9, "dense": 10,
       "dense_gemv": 11, "dense16": 12, "dense32": 13}
KS = (32, 64, 256, 512, 1024, 2048, 2304, 4096, 4352, 11008, 16384, 20480, 22016, 32768)
NS = (1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65, 128, 129, 256, 257, 300, 384, 512, 513, 640, 768, 1023, 1024, 2048, 4096)
MS = (1, 31, 32, 100, 256, 512, 1000, 2048, 4000, 4096, 8192, 11008, 16384, 24576, 26000, 32000, 33000, 65536, 262144)


def plan(t, M, K, N):
    out = _lib.ggml_hip_mm_plan_t()
    rc = _lib.lib().ggml_hip_mm_plan(t, M, K, N, C.byref(out))
    assert rc == 0, (t, M, K, N, rc)
    return out


@pytest.mark.parametrize("t", QUANT + (F16, F32, Q5_K, Q4_K, Q6_K))
def test_summation_tree_is_a_function_of_type_K_and_N_only(t):
    checked = wide_seen = 0
    for K in KS:
        if t in (Q5_K, Q4_K, Q6_K) and K % 256:
            continue
        for N in NS:
            trees = {}
            for M in MS:
                p = plan(t, M, K, N)
                assert p.family in FAM.values(), (t, M, K, N, p.family)
                assert p.ksplit >= 1 and p.tile_m > 0 and p.tile_n > 0 and p.workgroups > 0
                if p.flags & WIDE:
                    # the stated exception: only where a plane of the weight really is beyond 32-bit offsets (> 4 GiB), and then a family
                    # that addresses with 64 bits: the int8 staged kernel / the mat-vec (quantized), dense.hip (dense)
                    wide_seen += 1
                    rows = (M + 255) // 256 * 256
                    plane = rows * (K + 1024) * (6 if t == F32 else 2) if t in (F16, F32) else rows * (K // 32 + 18) * 32   # (F32: three bf16 pieces)
                    assert plane > 0xFFFFFFFF, (t, M, K, N)
                    assert p.family in (FAM["i8"], FAM["gemv_rows"], FAM["gemv_fused"], FAM["dense"], FAM["dense_gemv"]), (t, M, K, N, p.family)
                    continue
                trees.setdefault(p.tree_id, []).append((M, p.family, p.form, p.ksplit, p.kstyle, p.kunit, p.arith))
                checked += 1
            assert len(trees) <= 1, f"type {t} K {K} N {N}: the summation tree follows M: {trees}"
    assert checked > 1000
    if t in QUANT:
        assert wide_seen > 0            # (262144 rows x K = 32768: 4 GiB of nibbles -- the sweep does reach the exception)


def test_the_image_of_a_plan_is_the_image_of_its_type_K_and_N():
    L = _lib.lib()
    for t in QUANT:
        for K in KS:
            for N in NS:
                for M in (100, 4096, 32000):
                    p = plan(t, M, K, N)
                    if p.flags & WIDE:
                        continue
                    if p.family == FAM["gemv_fused"]:
                        assert p.image_kind == -1 and not (p.flags & NEEDS_WORK)
                    else:
                        assert p.image_kind == L.ggml_hip_act_image_kind(t, K, N), (t, M, K, N)
                        assert p.flags & NEEDS_WORK


def test_baseline_configs_get_the_kernels_design_md_names():
    p = plan(Q4_0, 4096, 4096, 4096)                                # the headline: K3m 256 x 128, one chain over K
    assert (p.family, p.form, p.tile_m, p.tile_n, p.ksplit, p.image_kind) == (FAM["mx"], 0, 256, 128, 1, 3) and p.workgroups == 512
    assert plan(Q4_0, 4096, 4096, 1).family == FAM["gemv_fused"]    # config 2
    for M in (4096, 4000, 32000):                                   # configs 3 and 5 (total and the 4000-row shard): K3p, eight K ranges of 16 k-blocks
        p = plan(Q4_0, M, 4096, 512)
        assert (p.family, p.ksplit, p.kstyle, p.kunit, p.tile_m, p.tile_n) == (FAM["k3p_mx"], 8, 2, 16, 128, 64)
    for t in (Q8_0, Q5_0, Q5_1, Q4_1, Q5_K):                        # config 4 and its min-term relatives: K3p on the int8 cores
        p = plan(t, 4096, 11008, 512)
        assert (p.family, p.ksplit, p.kunit) == (FAM["k3p_i8"], 8, 44), t
        assert bool(p.flags & Q8K) == (t == Q5_K)
        assert bool(p.flags & EPI) == (t != Q5_K)
        # the min-term types: INIT writes image 0 AND the bf16 piece planes of d * sum (kind 0 + 64); the others plain image 0
        assert bool(p.flags & MIN_PIECES) == (t in (Q5_1, Q4_1, Q5_K)) and p.image_kind == (64 if t in (Q5_1, Q4_1, Q5_K) else 0), t
    a, b = plan(Q4_K, 4096, 11008, 512), plan(Q5_K, 4096, 11008, 512)     # r4: Q4_K lives in Q5_K's resident form and gets its plans
    assert (a.family, a.form, a.tree_id, a.flags, a.image_kind) == (b.family, b.form, b.tree_id, b.flags, b.image_kind)
    # r4: the k-quants of the planar Q5_1 form have a fused mat-vec (the Q8_K rule in the kernel) up to 4 rows and K = 32768; beyond: two steps
    assert plan(Q5_K, 4096, 4096, 1).family == FAM["gemv_fused"] and plan(Q4_K, 4096, 4096, 4).family == FAM["gemv_fused"]
    assert plan(Q5_K, 4096, 4096, 5).family == FAM["gemv_rows"] and plan(Q5_K, 4096, 33024, 1).family == FAM["gemv_rows"]
    assert plan(Q5_K, 4096, 4096, 1).tree_id == plan(Q5_K, 100, 4096, 1).tree_id
    # r4: Q6_K lives in the planar Q4_2 form on int8 planes alone: its own mat-vec up to 4 rows (fused with the Q8_K rule up to K = 32768), the
    # batched-decode form at 5..64 rows, the staged int8 form's two-scale instantiation elsewhere
    assert plan(Q6_K, 4096, 4096, 1).family == FAM["gemv_fused"] and plan(Q6_K, 4096, 1024, 4).family == FAM["gemv_fused"] and plan(Q6_K, 4096, 33024, 1).family == FAM["gemv_rows"]
    assert plan(Q6_K, 4096, 4096, 5).family == FAM["k3s_i8"] and plan(Q6_K, 4096, 4096, 64).family == FAM["k3s_i8"]
    # (r4: the two-scale types -- Q4_2 and Q6_K in its form -- stayed on the batched-decode form up to 256 rows with only the staged int8 kernel behind it;
    # r5: K3p has a two-scale form and takes them from 129 rows at every size, like Q5_1)
    assert plan(Q6_K, 4096, 4096, 128).family == FAM["k3s_i8"] and plan(Q4_2, 4096, 11008, 128).family == FAM["k3s_i8"] and plan(Q8_0, 4096, 4096, 129).family != FAM["k3s_i8"]
    for t in (Q6_K, Q4_2):
        for (M, K, N) in ((4096, 4096, 129), (4096, 4096, 256), (4096, 11008, 512), (4096, 4096, 4096), (32000, 4096, 8192)):
            assert plan(t, M, K, N).family == FAM["k3p_i8"], (t, M, K, N)
        assert plan(t, 4096, 4096, 512).tree_id not in {plan(u, 4096, 4096, 512).tree_id for u in (Q8_0, Q5_0, Q5_1, Q4_1)}   # (two scale-accumulates per k-block: a tree of its own)
    assert plan(Q6_K, 4096, 4096, 512).tree_id != plan(Q4_2, 4096, 4096, 512).tree_id              # (activations by the Q8_K rule)
    assert plan(Q6_K, 4096, 768, 5).family == FAM["i8"] and plan(Q4_2, 4096, 1024, 512).family == FAM["i8"]   # (K < 1024, or K < 2048 beyond 64 rows: the staged int8 kernel)
    # r5: the batched-decode forms from K = 1024 (it was 2048) up to 64 src1 rows -- no K3p behind such a K, so no shared range there
    for t, fam in ((Q8_0, "k3s_i8"), (Q5_0, "k3s_i8"), (Q4_0, "k3s_mx"), (Q4_1, "k3s_mx"), (Q6_K, "k3s_i8")):
        for K in (1024, 1536, 1792):                          # (multiples of 256: the k-quants' super-block)
            assert plan(t, 4096, K, 32).family == FAM[fam] and plan(t, 32000, K, 64).family == FAM[fam], (t, K)
            assert plan(t, 4096, K, 65).family not in (FAM["k3s_i8"], FAM["k3s_mx"], FAM["k3p_i8"], FAM["k3p_mx"]), (t, K)
        assert plan(t, 4096, 768, 32).family not in (FAM["k3s_i8"], FAM["k3s_mx"]), t
    assert plan(Q6_K, 4096, 4096, 8).tree_id != plan(Q4_2, 4096, 4096, 32).tree_id              # (activations by the Q8_K rule: another tree)
    # r4: behind a long K (>= 11008, a down projection) the one-scale int8 types stay on the batched-decode form up to 128 rows
    assert plan(Q8_0, 4096, 11008, 128).family == FAM["k3s_i8"] and plan(Q5_1, 5120, 13824, 96).family == FAM["k3s_i8"] and plan(Q5_K, 4096, 11008, 128).family == FAM["k3s_i8"]
    # (... and, with the XCD-aware tile order, whatever K for the int8 types and Q4_1; Q4_0 keeps the K rule: its staged forms are better)
    assert plan(Q8_0, 4096, 11008, 129).family == FAM["k3p_i8"] and plan(Q8_0, 4096, 8192, 128).family == FAM["k3s_i8"] and plan(Q5_0, 4096, 4096, 65).family == FAM["k3s_i8"]
    # (r5: Q4_0 at 65..128 rows runs its stage-free forms whatever K -- K3s and K3p are one tree, the family follows M: test_k3s_and_k3p_mx_are_one_tree...)
    assert plan(Q4_0, 4096, 11008, 128).family == FAM["k3s_mx"] and plan(Q4_1, 8192, 28672, 64).family == FAM["k3s_mx"] and plan(Q4_0, 4096, 4096, 128).family == FAM["k3s_mx"]
    # (r5: Q4_1 from 65 rows on the int8 pair -- K3s-int8 on a short matrix, K3p-int8 on a tall one; up to 64 rows its MX form)
    assert plan(Q4_1, 4096, 4096, 128).family == FAM["k3s_i8"] and plan(Q4_1, 8192, 8192, 96).family == FAM["k3p_i8"] and plan(Q4_1, 4096, 4096, 64).family == FAM["k3s_mx"]
    assert plan(Q4_1, 8192, 28672, 96).family == FAM["k3s_mx"] and plan(Q4_1, 4096, 11008, 128).family == FAM["k3s_mx"] and plan(Q4_1, 1024, 11008, 129).family == FAM["k3s_i8"]   # (behind K >= 11008 its MX form keeps 65..128 rows)
    assert plan(Q8_0, 4096, 4096, 32).family == FAM["k3s_i8"] and plan(Q4_0, 4096, 4096, 32).family == FAM["k3s_mx"]
    # r4: Q5_0 (on its int8 operand planes) and Q5_1 (+ the min-term product: INIT writes the piece planes) run the batched-decode form too
    assert plan(Q5_0, 4096, 4096, 32).family == FAM["k3s_i8"] and plan(Q5_0, 4096, 4096, 32).image_kind == 0
    p = plan(Q5_1, 4096, 4096, 32)
    assert p.family == FAM["k3s_i8"] and p.image_kind == 64 and (p.flags & MIN_PIECES)
    assert plan(Q5_1, 4096, 4096, 8).family == FAM["gemv_fused"]                    # (up to 8 rows its fused mat-vec is as fast)
    assert plan(Q4_2, 4096, 4096, 32).family == FAM["k3s_i8"] and plan(Q4_2, 4096, 4096, 9).family == FAM["k3s_i8"] and plan(Q4_2, 4096, 4096, 8).family == FAM["gemv_fused"]   # (r4: from 9 rows; its mat-vec keeps up to 8, and 9..16 where K < 2048)
    assert plan(Q4_2, 4096, 768, 16).family == FAM["gemv_rows"] and plan(Q4_2, 4096, 1024, 16).family == FAM["k3s_i8"]   # (r5: the batched-decode form from K = 1024)
    assert len({plan(t, 4096, 4096, 32).tree_id for t in (Q8_0, Q5_0, Q5_1, Q4_2)}) == 4   # four different arithmetics, four trees
    # r4: K3p serves Q8_0 / Q5_0 up to 3072 rows, Q4_1 up to 1024, Q5_1 / Q5_K without bound; Q4_0 (MX) up to 512
    assert plan(Q8_0, 4096, 11008, 3072).family == FAM["k3p_i8"] and plan(Q8_0, 4096, 11008, 3073).family != FAM["k3p_i8"]
    assert plan(Q4_1, 4096, 4096, 1024).family == FAM["k3p_i8"] and plan(Q4_1, 4096, 4096, 1025).family == FAM["mx"]
    assert plan(Q4_0, 4096, 4096, 513).family == FAM["mx"]
    assert plan(Q8_0, 4096, 4096, 129).family == FAM["k3p_i8"] and plan(Q8_0, 4096, 4096, 128).family != FAM["k3p_i8"]   # (from 129 rows: Q8_0 / Q5_0 / Q5_1)
    assert plan(Q4_1, 4096, 4096, 64).family == FAM["k3s_mx"] and plan(Q4_1, 4096, 4096, 192).family == FAM["k3p_i8"]  # (r5: Q4_1 from 129 too; it was 257 -- then from 65 with K3s-int8 beside it)
    assert plan(Q5_1, 4096, 4096, 4096).family == FAM["k3p_i8"] and plan(Q5_K, 4096, 11008, 8192).family == FAM["k3p_i8"]   # (Q5_1 / Q5_K: no upper bound)
    # r4: K > 19968 -- the eight scale tables no longer fit LDS whole: K3p refills them in slices (up to four: K <= 79872), beyond that the staged forms
    assert plan(Q8_0, 4096, 22016, 512).family == FAM["k3p_i8"] and plan(Q4_0, 4096, 28672, 512).family == FAM["k3p_mx"]
    assert plan(Q8_0, 4096, 79872, 512).family == FAM["k3p_i8"] and plan(Q8_0, 4096, 79904, 512).family == FAM["f16"]
    assert plan(F16, 4096, 4096, 4096).family == FAM["dense16"] and plan(F32, 4096, 4096, 4096).family == FAM["dense32"]
    assert plan(F32, 64, 128, 256).family == FAM["dense"]           # config 1 (Test1-style f32 64 x 128 x 256)
    # r4: F32 at 5..256 rows, K % 256 == 0 from 1024 on: K split over the workgroup's eight waves (one tree whatever M is)
    a, b = plan(F32, 4096, 4096, 64), plan(F32, 100, 4096, 64)
    assert (a.family, a.form, a.ksplit, a.tile_m, a.tile_n) == (FAM["dense"], 2, 8, 32, 32) and a.tree_id == b.tree_id
    assert plan(F32, 4096, 4096, 4).family == FAM["dense_gemv"] and plan(F32, 4096, 4000, 64).form != 2
    assert plan(F32, 4096, 4096, 257).family == FAM["dense32"] and plan(F32, 4096, 4096, 256).family == FAM["dense"]


def test_geometry_may_follow_M_and_the_sweep_can_see_it():
    """Not vacuous: the plan DOES change with M -- in the fields that are allowed to (tile height, tiles per workgroup, form, family
    between forms of one tree) -- while tree_id stays put; so an M-dependent tree would be seen just as well."""
    a, b = plan(Q4_1, 4096, 1024, 100), plan(Q4_1, 32000, 1024, 100)          # four-way tree of the staged MX forms (K < 2048) on 32-row / 64-row tiles
    assert (a.tile_m, b.tile_m) == (32, 64) and a.tree_id == b.tree_id and a.ksplit == b.ksplit == 4
    a, b = plan(Q4_0, 4096, 4096, 100), plan(Q4_0, 32000, 4096, 100)          # r5: one tree for Q4_0's stage-free forms -- K3s on a short matrix, K3p on a tall one
    assert (a.family, b.family) == (FAM["k3s_mx"], FAM["k3p_mx"]) and a.tree_id == b.tree_id and a.ksplit == b.ksplit == 8
    a, b = plan(Q4_0, 512, 4096, 4096), plan(Q4_0, 4096, 4096, 4096)          # a row shard of the headline: 64 x 64 tiles, the whole: 256 x 128
    assert (a.tile_m, a.tile_n, b.tile_m, b.tile_n) == (64, 64, 256, 128) and a.tree_id == b.tree_id
    a, b, c = plan(Q8_0, 4096, 4096, 32), plan(Q8_0, 32000, 4096, 32), plan(Q8_0, 8192, 4096, 32)   # K3s on the int8 cores: r5 16-row tiles while they fit one round of the chip, else one / two 32-row tiles per workgroup
    d = plan(Q8_0, 12000, 4096, 32)                                                                  # (r5: four 32-row tiles beyond 512 tile groups, two up to there)
    assert (a.tile_m, a.tile_n, a.workgroups, b.tile_m, c.tile_m, d.tile_m) == (16, 32, 256, 128, 32, 64) and a.tree_id == b.tree_id == c.tree_id == d.tree_id
    assert plan(Q5_1, 32000, 4096, 32).tile_m == 64 and plan(Q4_2, 32000, 4096, 32).tile_m == 64         # (the min-term and two-scale types: two at most)
    # r5: the batched-decode forms on 16-row tiles (Q4_0 / Q4_1 on the MX cores, every other type on the int8 cores): 16 / 32 columns per workgroup by N,
    # the tree of the 32-row form (Q5_1 and Q5_K / Q4_K in its form: the min-term product stays the 32-row form's instruction, its result is handed to the 16 x 16 tiles' lanes)
    for t in (Q4_0, Q4_1, Q8_0, Q5_0, Q4_2, Q5_1):
        for N, tn in ((9, 16), (16, 16), (17, 32), (32, 32)):
            a, b = plan(t, 4096, 4096, N), plan(t, 16384, 4096, N)
            assert (a.tile_m, a.tile_n, a.workgroups) == (16, tn, 256) and b.tile_m >= 32 and a.tree_id == b.tree_id and a.family == b.family, (t, N)
        assert plan(t, 4096, 4096, 33).tile_m == 32 and plan(t, 4096, 4096, 64).tile_m == 32       # (33..64 rows: two 32-row workgroups per weight tile measured faster)
        a, b = plan(t, 2048, 4096, 64), plan(t, 1024, 4096, 64)                                       # (r5: 33..64 rows on a short matrix -- column groups of 32 / 16 columns, one round at most)
        assert (a.tile_m, a.tile_n, a.workgroups) == (16, 32, 256) and (b.tile_m, b.tile_n, b.workgroups) == (16, 16, 256) and a.tree_id == b.tree_id == plan(t, 32000, 4096, 64).tree_id, t
        a, b = plan(t, 2048, 4096, 32), plan(t, 2049, 4096, 32)                                       # (17..32 rows on a short matrix: one 16-column slice per workgroup while two workgroups per tile fit one round)
        assert (a.tile_m, a.tile_n, a.workgroups) == (16, 16, 256) and (b.tile_m, b.tile_n) == (16, 32) and a.tree_id == b.tree_id, t
        assert plan(t, 4112, 4096, 32).tile_m == 32                                                   # (more than one round of 16-row tiles: the 32-row form)
    for t in (Q5_K, Q4_K, Q6_K):                                                                          # (the k-quants live in the Q5_1 / Q4_2 forms: the same geometry rule)
        a, b = plan(t, 4096, 4096, 32), plan(t, 16384, 4096, 32)
        assert a.tile_m == 16 and b.tile_m >= 32 and a.tree_id == b.tree_id and a.family == b.family, t
    assert plan(Q6_K, 4096, 4096, 32).tile_m == 16 and plan(Q6_K, 4096, 4096, 5).tile_m == 16              # (Q6_K lives in Q4_2's form)
    # r5: K3p (both kernels) on 64-row wave tiles where a grid of 128-row tiles leaves CUs idle (<= 256 workgroups of 64 rows): the tree of the 128-row form
    for t, N in ((Q8_0, 192), (Q5_0, 129), (Q5_1, 256), (Q4_1, 300), (Q4_0, 512), (Q5_K, 256)):
        Ms = 4096 if N <= 256 and t != Q4_0 else 2048 if t == Q4_0 else 3072    # (short matrices run K3s up to 512 src1 rows, see below: these have 192+ workgroups of K3p)
        a, b = plan(t, Ms, 4096, N), plan(t, 32768, 4096, N)
        assert (a.tile_m, b.tile_m) == (64, 128) and a.tree_id == b.tree_id and a.family == b.family and a.family in (FAM["k3p_mx"], FAM["k3p_i8"]), (t, N)
    assert plan(Q8_0, 4096, 4096, 256).tile_m == 64 and plan(Q8_0, 4096, 4096, 257).tile_m == 128       # (256 workgroups of 64 rows fit one round, 320 do not)
    # ... and, int8 kernel, on grids of a fractional number of rounds: 258 workgroups of 128 rows (two rounds for two tiles) run as 516 of 64 (three half-size rounds)
    assert plan(Q8_0, 11008, 4096, 192).tile_m == 64 and plan(Q8_0, 11008, 4096, 256).tile_m == 64 and plan(Q8_0, 8192, 8192, 192).tile_m == 128
    assert plan(Q8_0, 14336, 4096, 256).tile_m == 128 and plan(Q4_0, 11008, 4096, 320).tile_m == 128                                 # (the MX kernel: one round only)
    for t, M in ((Q4_0, 4096), (Q4_0, 4000), (Q4_0, 32000), (Q8_0, 4096), (Q5_0, 4096), (Q5_K, 4096)):                              # the contract configs keep their 128-row tiles
        assert plan(t, M, 4096 if t == Q4_0 else 11008, 512).tile_m == 128, (t, M)
    assert plan(Q8_0, 11008, 4096, 512).tile_m == 128
    # and trees DO differ where they should: across N classes and across types
    assert plan(Q4_0, 4096, 4096, 512).tree_id != plan(Q4_0, 4096, 4096, 513).tree_id
    assert plan(Q4_0, 4096, 4096, 512).tree_id != plan(Q4_1, 4096, 4096, 512).tree_id
    assert plan(Q5_1, 4096, 4096, 512).tree_id != plan(Q5_0, 4096, 4096, 512).tree_id     # the min term is part of the tree


def test_bad_arguments_are_reported():
    out = _lib.ggml_hip_mm_plan_t()
    L = _lib.lib()
    assert L.ggml_hip_mm_plan(Q4_0, 4096, 4096, 16, None) == -4
    assert L.ggml_hip_mm_plan(9, 4096, 4096, 16, C.byref(out)) == -2      # Q8_1 is not a weight type (Ggml.cs:278-282)
    assert L.ggml_hip_mm_plan(Q4_0, 4096, 4100, 16, C.byref(out)) == -3    # K % 32
    assert L.ggml_hip_mm_plan(Q5_K, 4096, 4128, 16, C.byref(out)) == -3    # K % 256
    assert L.ggml_hip_mm_plan(Q4_0, 0, 4096, 16, C.byref(out)) == -3


def test_k3s_and_k3p_int8_are_one_tree_and_the_family_follows_M():
    """r5: the stage-free int8 families -- batched decode K3s and prompt-sized K3p -- share their summation tree (the same eight K
    ranges, statement, wave-order sum; GPU: test_k3s_and_k3p_int8_compute_the_same_bits), so between 33 and 512 src1 rows the plan picks the FAMILY by
    M: K3p once its grid of 64-row tiles has 160 workgroups (192 behind K >= 11008), K3s below.  tree_id may not move with that choice."""
    for t in (Q8_0, Q5_0, Q5_1, Q4_1, Q4_2, Q5_K, Q6_K):
        for K in (4096, 11008, 2048 + (256 if t in (Q5_K, Q6_K) else 64)):
            for N in (33, 64, 65, 100, 128, 129, 200, 256, 384, 512):
                ids, fams = set(), set()
                if t == Q4_1 and (N <= 64 or (K >= 11008 and N <= 128)):
                    continue                                   # (its MX form there)
                for M in (512, 1024, 2048, 4096, 8192, 11008, 32000):
                    p = plan(t, M, K, N)
                    assert p.family in (FAM["k3s_i8"], FAM["k3p_i8"]), (t, M, K, N, p.family)
                    want = FAM["k3p_i8"] if -(-M // 64) * -(-N // 64) >= (192 if K >= 11008 else 160) else FAM["k3s_i8"]
                    assert p.family == want, (t, M, K, N)
                    ids.add(p.tree_id); fams.add(p.family)
                assert len(ids) == 1 and len(fams) == 2, (t, K, N)
    # outside the shared range nothing moved: up to 32 rows K3s whatever M, from 513 K3p whatever M
    for M in (512, 4096, 32000):
        assert plan(Q8_0, M, 4096, 32).family == FAM["k3s_i8"] and plan(Q8_0, M, 4096, 513).family == FAM["k3p_i8"]
    # ... and the two families label the same arithmetic the same way below and above it too (one K range rule: an even number of k-blocks per wave)
    assert plan(Q8_0, 4096, 11008, 32).kunit == plan(Q8_0, 4096, 11008, 512).kunit == 44
    assert plan(Q8_0, 4096, 11008, 32).tree_id == plan(Q8_0, 4096, 11008, 512).tree_id
    # Q4_1: its MX batched-decode form up to 64 rows (another arithmetic), from 65 the int8 pair like the others
    assert plan(Q4_1, 32000, 4096, 64).family == FAM["k3s_mx"] and plan(Q4_1, 32000, 4096, 128).family == FAM["k3p_i8"] and plan(Q4_1, 512, 4096, 129).family == FAM["k3s_i8"]
    assert plan(Q4_1, 32000, 4096, 128).tree_id == plan(Q4_1, 512, 4096, 128).tree_id != plan(Q4_1, 512, 4096, 64).tree_id


def test_k3s_and_k3p_mx_are_one_tree_and_the_family_follows_M():
    """r5: the same for Q4_0 on the MX cores between 33 and 512 src1 rows, whatever K (the staged K-split forms that served up to 256 there are left with K < 2048)."""
    for K in (2048, 4096, 11008, 2048 + 64):
        for N in (33, 64, 65, 100, 128, 129, 200, 256, 384, 512):
            ids, fams = set(), set()
            for M in (512, 2048, 4096, 8192, 11008, 32000):
                p = plan(Q4_0, M, K, N)
                want = FAM["k3p_mx"] if -(-M // 64) * -(-N // 64) >= (192 if K >= 11008 else 160) else FAM["k3s_mx"]
                assert p.family == want, (M, K, N, p.family)
                ids.add(p.tree_id); fams.add(p.family)
            assert len(ids) == 1 and len(fams) == 2, (K, N)
    assert plan(Q4_0, 32000, 4096, 32).family == FAM["k3s_mx"] and plan(Q4_0, 512, 4096, 513).family == FAM["mx"] and plan(Q4_0, 4096, 1024, 129).family == FAM["mx"]      # outside the range nothing moved
    assert plan(Q4_0, 4096, 11008, 32).kunit == plan(Q4_0, 4096, 11008, 512).kunit == 44 and plan(Q4_0, 4096, 11008, 32).tree_id == plan(Q4_0, 4096, 11008, 512).tree_id
    assert plan(Q4_1, 32000, 4096, 64).family == FAM["k3s_mx"]                                                            # (Q4_1: its K3p is the int8 kernel -- another arithmetic; from 65 rows the int8 pair)
