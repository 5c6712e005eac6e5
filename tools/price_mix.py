#!/usr/bin/env python3
"""Prices K3's vector-instruction mix with the measured issue rates (profiles/r03_valu_calibration.json, tools/valu_calib.hip).

For every render permutation: the kernel's ISA (hipcc -S, no GPU needed) is cut into basic blocks, the blocks are sorted
into node round / leaf round / triangle loop / shade pass by the loops they sit in, every block's vector instructions are
priced per class (cycles per wave64 instruction per SIMD at the kernel's occupancy), and the blocks are weighted with the
wave-level counters of the counting build on that workload (profiles/r04_wave_stats.json: node rounds, leaf rounds,
passes per 64 rays).  Result: the average issue cost of one vector instruction of THAT kernel on THAT workload, hence the
issue ceiling in wave-instructions per second — the denominator of bench.py's `valu` roofline — written to
profiles/r04_valu_mix.json.  The predicted dynamic instruction count per ray is printed next to the counters' own
(SQ_INSTS_VALU / rays, profiles/pmc_summary.json) as a check of the weighting.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pooraytracer_amd", "csrc")
N_SIMD = 256 * 4

# mnemonic -> calibration entry (profiles/r03_valu_calibration.json).  Anything not listed is reported and priced as v_cndmask_b32.
CLASS = [
    (r"v_(fma|fmac|fmamk|fmaak|mul|add|sub|subrev|mac|mad)_f32|v_(trunc|floor|fract|rndne)_f32", "v_fma_f32"),
    (r"v_div_(scale|fmas|fixup)_f32|v_rcp_iflag_f32", "v_rcp_f32"),
    (r"v_mbcnt_|v_bitop3_b32|v_trig_preop_f64", "v_alignbit_b32"),
    (r"v_pk_", "v_pk_fma_f32"),
    (r"v_(fma|fmac)_f64", "v_fma_f64"),
    (r"v_mul_f64", "v_mul_f64"),
    (r"v_(add|max|min)_f64|v_ldexp_f64|v_frexp_(mant|exp_i32)_f64|v_rndne_f64|v_floor_f64|v_trunc_f64|v_fract_f64|v_div_(scale|fmas|fixup)_f64", "v_add_f64"),
    (r"v_(rcp|rsq|sqrt)_f64", "v_rcp_f64"),
    (r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", "v_rcp_f32"),
    (r"v_cvt_f32_u32_sdwa|v_cvt_f32_ubyte", "v_cvt_f32_u32_sdwa"),
    (r"v_cvt_f64_(f32|i32|u32)", "v_cvt_f64_f32"),
    (r"v_cvt_(f32|i32|u32)_f64", "v_cvt_f32_f64"),
    (r"v_cvt_", "v_cvt_f32_u32"),
    (r"v_alignbit_b32|v_alignbyte_b32|v_perm_b32|v_bfe_|v_bfi_|v_lshl_or_b32|v_and_or_b32|v_or3_b32|v_xad_u32|v_lshl_add_u32|v_add_lshl_u32|v_add3_u32|v_mad_u32_u24|v_mad_i32_i24", "v_alignbit_b32"),
    (r"v_(max|min)3_|v_med3_", "v_max3_f32"),
    (r"v_(max|min)_(f32|u32|i32)", "v_max_f32"),
    (r"v_cndmask_b32", "v_cndmask_b32"),
    (r"v_cmp_.*_f64|v_cmpx_.*_f64|v_cmp_class_f64", "v_cmp_lt_f64"),
    (r"v_cmp|v_cmpx", "v_cmp_le_f32"),
    (r"v_mul_(lo|hi)_(u32|i32)|v_mul_u32_u24|v_mul_i32_i24", "v_mul_lo_u32"),
    (r"v_mad_(u64_u32|i64_i32)", "v_mad_u64_u32"),
    (r"v_lshl_add_u64|v_(lshlrev|lshrrev|ashrrev)_b64|v_(lshlrev|lshrrev)_u64", "v_lshl_add_u64"),
    (r"v_mov_b64", "v_mov_b64"),
    (r"v_mov_b32|v_accvgpr", "v_mov_b32"),
    (r"v_readlane_b32|v_readfirstlane_b32", "v_readlane_b32"),
    (r"v_writelane_b32", "v_writelane_b32"),
    (r"v_(xor|and|or|not|bfrev)_b32|v_(lshlrev|lshrrev|ashrrev)_(b32|i32)", "v_xor_b32"),
    (r"v_(add|sub|subrev)(_co)?_u32|v_(addc|subb|subbrev)_co_u32|v_(add|sub)_i32", "v_add_u32"),
]

# permutation -> (FEAT template argument, workload that runs it, resident waves per SIMD of the fp64 / fp32 kernel)
PERMS = {"cornell-box": (0, 3, 4), "bathroom2": (1, 3, 4), "veach-mis": (2, 3, 4), "cornell-ct": (4, 2, 4)}


def asm_of(f32):
    out = f"/tmp/_prt_mix_{'f32' if f32 else 'f64'}.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-function", "-w",
           "-o", out, os.path.join(CSRC, "prt_kernels_f32.hip" if f32 else "prt_kernels.hip")]
    subprocess.check_call(cmd)
    return open(out).read().split("\n")


def kernel_body(src, feat, f32):
    # <COUNT false, FEAT, LLDS true, PAD false>: veach-mis has light tables, so it runs the PRT_FEAT_EXTRA (| 8) kernel
    pat = re.compile(r"^_ZN.*k_renderILb0ELi%dELb1ELb0E.*:" % (feat | 8 if feat == 2 else feat))
    start = next(i for i, l in enumerate(src) if pat.match(l))
    end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
    return src[start:end]


def blocks_of(body):
    blocks, cur = [], {"name": "entry", "ins": [], "notes": []}
    blocks.append(cur)
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = {"name": m.group(1), "ins": [], "notes": []}
            blocks.append(cur)
            t = t[m.end():].strip()
        if not t or t.startswith("."):
            continue
        if t.startswith(";"):
            cur["notes"].append(t)
            continue
        cur["ins"].append(t.split(";")[0].strip())
    return blocks


def loop_depth(b):
    d = 0
    for n in b["notes"]:
        m = re.search(r"Depth=(\d+)", n) or re.search(r"Depth (\d+)", n)
        if m and ("in Loop" in n or "Inner Loop Header" in n or "Loop Header" in n or "Parent Loop" in n):
            d = max(d, int(m.group(1)))
    return d


def classify(op, cal, w, unknown):
    for pat, key in CLASS:
        if re.match(pat + r"(_e32|_e64|_sdwa|_dpp)?$", op) or re.match(pat, op):
            return cal[key]["w%d" % w]["cycles_per_instruction"]
    unknown[op] = unknown.get(op, 0) + 1
    return cal["v_cndmask_b32"]["w%d" % w]["cycles_per_instruction"]


def main():
    cal = json.load(open(os.path.join(ROOT, "profiles", "r03_valu_calibration.json")))["ops"]
    stats = json.load(open(os.path.join(ROOT, "profiles", "r04_wave_stats.json")))
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    except Exception:
        pmc = {}
    clock = 2.4
    result = {"source": "tools/price_mix.py: ISA of k_render x profiles/r03_valu_calibration.json x profiles/r04_wave_stats.json",
              "clock_ghz": clock, "simds": N_SIMD, "workloads": {}}
    for f32 in (False, True):
        src = asm_of(f32)
        for wl, (feat, w64, w32) in PERMS.items():
            w = w32 if f32 else w64
            name = wl + ("-f32" if f32 else "")
            st = stats[name]
            blocks = blocks_of(kernel_body(src, feat, f32))
            # Blocks by signature, not by the loop depth LLVM prints (the traversal do-while is merged into the wave loop, so the
            # node visit shows at depth 1): the NODE VISIT is the block that rotates the packed ranges (>= 10 v_alignbit_b32) and
            # issues the four 16-byte node loads, plus the stack push / pop blocks that follow it up to the next block with
            # fp64 work; the TRIANGLE LOOP is the depth-3 loop; the LEAF step is the depth-2 code around it (it has global
            # loads); everything else inside the wave loop — fp64 helper loops included — is the shade pass.
            unknown = {}
            tot = {"node": [0, 0.0], "leaf": [0, 0.0], "tri": [0, 0.0], "pass": [0, 0.0], "once": [0, 0.0]}
            node_at = next(i for i, b in enumerate(blocks)
                           if sum(1 for x in b["ins"] if x.startswith("v_alignbit")) >= 10 and sum(1 for x in b["ins"] if x.startswith("global_load_dwordx4")) >= 4)
            node_set = {node_at}
            for i in range(node_at + 1, len(blocks)):
                o_ = [x.split()[0] for x in blocks[i]["ins"] if x.startswith("v_")]
                if any("f64" in o for o in o_) or loop_depth(blocks[i]) >= 2 or len(o_) > 40:
                    break
                node_set.add(i)
            tri_idx = [i for i, b in enumerate(blocks) if loop_depth(b) >= 3]
            leaf_lo, leaf_hi = (min(tri_idx), max(tri_idx)) if tri_idx else (0, -1)
            while leaf_lo - 1 >= 0 and loop_depth(blocks[leaf_lo - 1]) == 2 and any(x.startswith("global_load") or x.startswith("v_") for x in blocks[leaf_lo - 1]["ins"]) and leaf_lo - 1 not in node_set:
                leaf_lo -= 1
            while leaf_hi + 1 < len(blocks) and loop_depth(blocks[leaf_hi + 1]) == 2 and leaf_hi + 1 not in node_set:
                leaf_hi += 1
            for bi, b in enumerate(blocks):
                d = loop_depth(b)
                ops = [i.split()[0] for i in b["ins"] if i.startswith("v_")]
                if d == 0:
                    kind = "once"
                elif bi in node_set:
                    kind = "node"
                elif d >= 3:
                    kind = "tri"
                elif leaf_lo <= bi <= leaf_hi:
                    kind = "leaf"
                else:
                    kind = "pass"
                cost = sum(classify(o, cal, w, unknown) for o in ops)
                tot[kind][0] += len(ops)
                tot[kind][1] += cost
            weights = {"node": st["inner_rounds"], "leaf": st["leaf_rounds"], "tri": st["leaf_rounds"] * 2.5, "pass": st["passes"], "once": 0.0}
            # A pass skips the blocks none of its lanes needs (s_cbranch_execz): how much of the pass body runs is taken from the
            # counters when they exist for this build (SQ_INSTS_VALU per ray minus the traversal's share) at the pass's own mix
            pm = pmc.get(name) or {}
            cpl = pm.get("counters_per_launch", {})
            fresh = str(pm.get("round", "")).startswith("r04")
            pmc_per_ray = cpl["SQ_INSTS_VALU"] / pm["rays_per_launch"] if fresh and "SQ_INSTS_VALU" in cpl and pm.get("rays_per_launch") else None
            pmc_cost = 4.0 * cpl["SQ_ACTIVE_INST_VALU"] / cpl["SQ_INSTS_VALU"] if pmc_per_ray and "SQ_ACTIVE_INST_VALU" in cpl else None
            trav = sum(tot[k][0] * weights[k] for k in ("node", "leaf", "tri"))
            if pmc_per_ray is not None and tot["pass"][0]:
                weights["pass"] = max(0.0, pmc_per_ray * st["rays"] - trav) / tot["pass"][0]
            n_dyn = sum(tot[k][0] * weights[k] for k in tot)
            c_dyn = sum(tot[k][1] * weights[k] for k in tot)
            avg = c_dyn / n_dyn
            per_ray = n_dyn / st["rays"]
            entry = {"permutation_feat": feat, "waves_per_simd": w, "static_valu": {k: tot[k][0] for k in tot},
                     "static_issue_cycles": {k: round(tot[k][1], 1) for k in tot},
                     "weights_per_64_rays": {"node_rounds": round(st["inner_rounds_per_64_rays"], 2), "leaf_rounds": round(st["leaf_rounds_per_64_rays"], 2),
                                             "passes": round(st["passes_per_64_rays"], 2), "triangle_loop_iterations_per_leaf_round": 2.5},
                     "predicted_valu_per_ray": round(per_ray, 1),
                     "pmc_valu_per_ray": round(pmc_per_ray, 1) if pmc_per_ray else None,
                     "pass_body_executed_fraction": round(weights["pass"] / st["passes"], 3) if st["passes"] else None,
                     "pmc_cycles_per_valu_instruction": round(pmc_cost, 3) if pmc_cost else None,  # 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
                     "avg_cycles_per_valu_instruction": round(avg, 3),
                     "issue_ceiling_ginstr_per_s": round(N_SIMD * clock / avg, 1),
                     "share_of_issue_cycles": {k: round(tot[k][1] * weights[k] / c_dyn, 3) for k in tot if weights[k]},
                     "unclassified": unknown}
            result["workloads"][name] = entry
            print(f"{name:16s} feat {feat} w{w}: static VALU node {tot['node'][0]} leaf {tot['leaf'][0]} tri {tot['tri'][0]} pass {tot['pass'][0]} | "
                  f"predicted {per_ray:.1f} VALU/ray (PMC {entry['pmc_valu_per_ray']}) | avg {avg:.2f} cycles/instr -> ceiling {entry['issue_ceiling_ginstr_per_s']} G/s"
                  + (f" | unclassified {unknown}" if unknown else ""))
    json.dump(result, open(os.path.join(ROOT, "profiles", "r04_valu_mix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
