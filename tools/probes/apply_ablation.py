#!/usr/bin/env python3
"""Timing ablations and diagnostic forms of the trunk kernels, kept OUT of the product sources.

Every entry below is a list of (file, search, replace) edits applied to a COPY of
othello_reinforcement_learning_test_amd/csrc (tools/build_variant.sh --ablation NAME ... does the copy, the edit and the
build).  Most of them give WRONG results by construction -- they exist to price one term of a kernel's time (DESIGN_HISTORY.md,
sections K3w / K3 / K3d) -- so none of them is a `-D` away from a shipping library any more.

usage: apply_ablation.py NAME SRCDIR      (NAME = one of the keys of ABLATIONS; `list` prints them)
"""
import os
import sys

ABLATIONS = {
    # ---- k_trunk_w (net_wino.hip), round 3: what is left is power, not issue slots -------------------------------------
    "wino_halflds": ("every second conv step reuses stale operand registers instead of reading LDS (WRONG results)", [
        ("net_wino.hip", "                if (q + PD < QT) xh[psl] = *(const half8*)src_of(q + PD);",
         "                if (q + PD < QT && ((q + PD) & 1) == 0) xh[psl] = *(const half8*)src_of(q + PD);"),
        ("net_wino.hip", "                if (q + PD < QT) xl[psl] = *(const half8*)(src_of(q + PD) + 256);",
         "                if (q + PD < QT && ((q + PD) & 1) == 0) xl[psl] = *(const half8*)(src_of(q + PD) + 256);"),
    ]),
    "wino_nowrite": ("no V stores in the epilogue; the values stay live through a dummy use (WRONG results)", [
        ("net_wino.hip", "                        *(uint2*)dst = hi;\n                        *(uint2*)(dst + 256) = lo;",
         '                        asm volatile("" :: "v"(hi), "v"(lo), "v"(dst));'),
    ]),
    "wino_halfw": ("every second weight fragment keeps its stale registers: half the L2 -> CU stream (WRONG results)", [
        ("net_wino.hip", "                if (step < 8)\n                    wq[(grp + RING - 1) % RING][step] =",
         "                if (step < 8 && (step & 1) == 0)\n                    wq[(grp + RING - 1) % RING][step] ="),
    ]),
    "wino_nosat": ("no saturation tracking in the epilogue (clamped activations would go unreported)", [
        ("net_wino.hip", "                sat_bits = max(sat_bits, max(max(__float_as_uint(v0[0].x), __float_as_uint(v0[0].y)),\n"
                         "                                             max(__float_as_uint(v0[1].x), __float_as_uint(v0[1].y))));\n"
                         "                sat_bits = max(sat_bits, max(max(__float_as_uint(v1[0].x), __float_as_uint(v1[0].y)),\n"
                         "                                             max(__float_as_uint(v1[1].x), __float_as_uint(v1[1].y))));\n", ""),
    ]),
    # ---- k_trunk16 (net_mfma.hip), round 3 -----------------------------------------------------------------------------
    "mfma_taps": ("2 of 3 column taps: 1.5x fewer MFMAs, LDS reads and weight bytes, epilogue unchanged (WRONG results)", [
        ("net_mfma.hip", "            for (int dxi = 0; dxi < 3; ++dxi) {\n                const int tap = (DY + 1) * 3 + dxi;",
         "            for (int dxi = 0; dxi < 2; ++dxi) {\n                const int tap = (DY + 1) * 3 + dxi;"),
    ]),
    "mfma_halfw": ("the lo weight fragments are not loaded: half the L2 -> CU weight stream (WRONG results)", [
        ("net_mfma.hip", "                        if (neww) wq[wslot][1] = wl[(size_t)nstep * 1024 + 64];",
         "                        if (neww) wq[wslot][1] = wq[wslot][0];"),
        ("net_mfma.hip", "                        if (neww) wq[wslot][3] = wl[(size_t)nstep * 1024 + 192];",
         "                        if (neww) wq[wslot][3] = wq[wslot][2];"),
    ]),
    # ---- k_trunk_w6 (net_wino6.hip), round 4: the LDS image of V was the cost ------------------------------------------
    "w6_linear": ("every operand read a plain lane-linear 1 KB row instead of the real addresses (WRONG results)", [
        ("net_wino6.hip", "            const int xi = q & 3, nt = (q >> 2) % k6NT, grp = q / GS, kk = grp & 1, d = grp >> 1;\n"
                          "            return rd_base[nt / 3][d] + (uint32_t)k6_run(nt % 3, xi, kk, half);",
         "            return (uint32_t)(wave * 32768 + (q & 15) * 2048 + half * 1024 + lane * 16);"),
    ]),
    "w6_nolds": ("no operand reads; the registers are made opaque so that the MFMAs stay (WRONG results)", [
        ("net_wino6.hip", "                if (q + PD < QT) xh[psl] = *(const half8*)(lds + src_of(q + PD, 0));",
         '                asm volatile("" : "+v"(xh[psl]));'),
        ("net_wino6.hip", "                if (q + PD < QT) xl[psl] = *(const half8*)(lds + src_of(q + PD, 1));",
         '                asm volatile("" : "+v"(xl[psl]));'),
    ]),
    "w6_now": ("no weight loads; the ring registers are made opaque so that the MFMAs stay (WRONG results)", [
        ("net_wino6.hip", "                if (step < 8) wq[(grp + 1) & 1][step] = wl[(size_t)(grp + 1) * k6GroupU4 + (size_t)step * 64];",
         '                if (step < 8) asm volatile("" : "+v"(wq[(grp + 1) & 1][step]));'),
    ]),
    # ---- k_trunk_w6 experiment (b) of round 5 (CORRECT results): VERDICT r4 item 4b -- read the finished accumulators into
    #      VGPR copies BETWEEN the last group's MFMAs (each three steps after its last MFMA) instead of after the convolution.
    #      ISA of the build: the allocator has no room for 144 more live VGPRs and parks the copies back in AGPRs (80
    #      v_accvgpr_write + 68 v_mov in the convolution, the epilogue's reads stay): more VALU, not less.
    "w6_exp_early_acc_reads": ("finished accumulators read between the last group's MFMAs (correct results; measured slower)", [
        ("net_wino6.hip", "    f32x4 res[k6NT][2];   // [N-tile][x parity]: the residual in the spatial domain, fp32, x act_scale\n",
         "    f32x4 res[k6NT][2];   // [N-tile][x parity]: the residual in the spatial domain, fp32, x act_scale\n"
         "    f32x4 accv[4][k6NT];  // VGPR copies of the finished accumulators\n"),
        ("net_wino6.hip", "                        const f32x2 a0 = whalf(acc[0][nt], h), a1 = whalf(acc[1][nt], h), a2 = whalf(acc[2][nt], h),\n"
                          "                                    a3 = whalf(acc[3][nt], h);",
         "                        const f32x2 a0 = whalf(accv[0][nt], h), a1 = whalf(accv[1][nt], h), a2 = whalf(accv[2][nt], h),\n"
         "                                    a3 = whalf(accv[3][nt], h);"),
        ("net_wino6.hip", "    // V addressing (see the header): run (j, xi, k-step, half)",
         "#pragma unroll\n    for (int xi = 0; xi < 4; ++xi)\n#pragma unroll\n        for (int nt = 0; nt < k6NT; ++nt) accv[xi][nt] = acc[xi][nt];\n"
         "    // V addressing (see the header): run (j, xi, k-step, half)"),
        ("net_wino6.hip", "                acc[xi][nt] = w6mfma(wlo, xh[sl], acc[xi][nt]);\n                OTH_W6SB;\n            }\n        };",
         "                acc[xi][nt] = w6mfma(wlo, xh[sl], acc[xi][nt]);\n                OTH_W6SB;\n"
         "                if (grp == k6Groups - 1 && step >= 3) {\n                    const int s2 = step - 3;\n"
         "                    f32x4 tcopy = acc[s2 & 3][s2 >> 2];\n                    asm volatile(\"\" : \"+v\"(tcopy));\n"
         "                    accv[s2 & 3][s2 >> 2] = tcopy;\n                    OTH_W6SB;\n                }\n            }\n        };"),
        ("net_wino6.hip", "        conv_d(std::integral_constant<int, 2>{});\n    }\n",
         "        conv_d(std::integral_constant<int, 2>{});\n#pragma unroll\n        for (int s2 = GS - 3; s2 < GS; ++s2) {\n"
         "            f32x4 tcopy = acc[s2 & 3][s2 >> 2];\n            asm volatile(\"\" : \"+v\"(tcopy));\n"
         "            accv[s2 & 3][s2 >> 2] = tcopy;\n        }\n    }\n"),
    ]),
    # ---- k_trunk_w6 experiment (a') of round 5 (CORRECT results, bit-identical): the heads' two FC weight arrays staged through LDS
    #      once per workgroup instead of six exposed L2 round trips per wave (VERDICT r4 item 4a asked for the heads out of the
    #      trunk; a head kernel cannot co-reside with a trunk workgroup, so the in-trunk latency is what can be attacked).
    "w6_exp_fc_lds": ("heads' FC weights through LDS (correct, bit-identical results)", [
        ("net_wino6.hip", "                for (int r = 0; r < 4; ++r) planes[(ch0 + r) * NCO + ci] = v[r] * us;\n            }\n    __syncthreads();",
         "                for (int r = 0; r < 4; ++r) planes[(ch0 + r) * NCO + ci] = v[r] * us;\n            }\n"
         "    constexpr int kFcFloats = 2 * k6Cells * k6NP + k6Cells * 256;          // 2664 + 9216 = 11880 floats = 2970 float4\n"
         "    constexpr int kFcOff = k6F * NCO * 4 + 4 * 2 * 192 * 4;                 // behind the planes (73 728 B) and the scratch rows\n"
         "    static_assert(kFcOff + kFcFloats * 4 <= k6Lds && (2 * k6Cells * k6NP) % 4 == 0, \"FC weights must fit behind the planes\");\n"
         "    float* fc_lds = (float*)(lds + kFcOff);\n"
         "    {\n        float4 fcw[12];\n#pragma unroll\n        for (int i = 0; i < 12; ++i) {\n"
         "            const int q = tid + 256 * i;                                   // float4 index: policy FC first, then value FC1\n"
         "            const float4* src = q < 2 * k6Cells * k6NP / 4 ? (const float4*)a.pfc_wt + q : (const float4*)a.vfc1_wt + (q - 2 * k6Cells * k6NP / 4);\n"
         "            fcw[i] = q < kFcFloats / 4 ? *src : make_float4(0.f, 0.f, 0.f, 0.f);\n        }\n#pragma unroll\n"
         "        for (int i = 0; i < 12; ++i)\n            if (tid + 256 * i < kFcFloats / 4) ((float4*)fc_lds)[tid + 256 * i] = fcw[i];\n    }\n"
         "    __syncthreads();"),
        ("net_wino6.hip", "heads_wave_n<k6F, k6BS, 2, true>(a.heads, a.pfc_wt, a.vfc1_wt, srcs, NCO, scratch, lane, lps, vs, live);",
         "heads_wave_n<k6F, k6BS, 2, true>(a.heads, fc_lds, fc_lds + 2 * k6Cells * k6NP, srcs, NCO, scratch, lane, lps, vs, live);"),
    ]),
    # ---- k_trunk_w6 experiment of round 6 (CORRECT results, bit-identical): VERDICT r5 item 3 -- the epilogue VALU of one lane group
    #      in the MFMA gaps of another lane group's convolution.  The whole layer loop is swapped for tools/probes/w6_lgpipe_loop.inc
    #      (its header explains why the unit is the lane group, not a half-group of four positions, and the schedule); build with
    #      `-mllvm -pragma-unroll-threshold=4000000` (the 96-step loops with the micro-op chain are past hipcc's default limit for a
    #      `#pragma unroll`), optionally -DOTH_W6_RING=3.  Measured and closed: DESIGN.md section 7, profiles/r06_w6_lgpipe_*.log.
    "w6_exp_lgpipe": ("lane-group pipeline: epilogue VALU interleaved with another lane group's MFMAs (correct results)", [
        ("net_wino6.hip", ("REGION", "    // [layer loop: begin]", "    // [layer loop: end]\n"),
         ("FILE", "w6_lgpipe_loop.inc")),
    ]),
    # ---- k_trunk_w6 heads, cooperative form (round 6; CORRECT results, bit-identical): the value FC1 once per workgroup (thread =
    #      output, all eight positions) and every FC weight requested up front -- tools/probes/w6_coop_heads.inc.
    "w6_exp_coop_heads": ("cooperative heads of k_trunk_w6 (correct, bit-identical results)", [
        ("net_wino6.hip", ("REGION", "    // [heads: begin]", "    // [heads: end]\n"), ("FILE", "w6_coop_heads.inc")),
    ]),
}


def apply(name, srcdir):
    desc, edits = ABLATIONS[name]
    for fname, search, replace in edits:
        path = os.path.join(srcdir, fname)
        text = open(path).read()
        if isinstance(search, tuple):   # ("REGION", begin marker, end marker): everything from the first to the end of the second
            _, m0, m1 = search
            if text.count(m0) != 1 or text.count(m1) != 1:
                raise SystemExit("apply_ablation %s: region markers not found exactly once in %s" % (name, fname))
            a, b = text.index(m0), text.index(m1) + len(m1)
            search = text[a:b]
        if isinstance(replace, tuple):  # ("FILE", name): the content of a file next to this script
            replace = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), replace[1])).read()
        if text.count(search) != 1:
            raise SystemExit("apply_ablation %s: the text to replace occurs %d times in %s (the product source moved on: "
                             "update tools/probes/apply_ablation.py)" % (name, text.count(search), fname))
        open(path, "w").write(text.replace(search, replace))
    print("applied ablation %s: %s" % (name, desc))


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "list":
        for k, (d, _) in ABLATIONS.items():
            print("%-14s %s" % (k, d))
    elif len(sys.argv) == 3 and sys.argv[1] in ABLATIONS:
        apply(sys.argv[1], sys.argv[2])
    else:
        raise SystemExit(__doc__)
