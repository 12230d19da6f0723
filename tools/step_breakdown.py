#!/usr/bin/env python3
"""Per-step time by kernel group from a profiles/rocprof_*_kernel_stats.md summary (the numbers quoted in DESIGN.md section 8
and profiles/README.md).   usage: tools/step_breakdown.py profiles/rocprof_r04_warp_c2_kernel_stats.md [steps=4]"""
import re
import sys

GROUPS = [
    ("pre-cut ring GEMMs", r"^conv_fwd_pc_kernel"),
    ("weight-gradient ring GEMMs", r"^conv_wgrad_dma_kernel"),
    ("split reductions", r"reduce_kernel|slab_sum"),
    ("Winograd filter refresh (side stream)", r"winog_filter_pc|wino_s2_filter_kernel|tailw_filter_kernel|^winog_filter_kernel|^wino_filter_kernel"),
    ("weight-side amax / pre-cut / re-pack (side stream)", r"amax_partials|conv_precut|repack_dgrad|head_pack"),
    ("strided Winograd transforms", r"wino_s2_input|wino_s2_fold|winog_dy_kernel<F42|winog_patch_kernel<F42|winog_output_kernel<F42|wino_s2_filter_grad"),
    ("stride-1 / tail Winograd transforms", r"winog_|wino_|tailw_"),
    ("InstanceNorm / activation", r"in_fused|in_partial|in_finalize|norm_act|ew_kernel|colsum"),
    ("AdamW", r"adamw"),
    ("other GEMM kernels (narrow, first layer, register-staged)", r"conv_fwd|conv_wgrad|tail_"),
    ("losses / gather / misc", r".*"),
]
SETUP = re.compile(r"fillBuffer|copyBuffer|nchw_to_nhwc|pack_kernel|init_|rng|philox")


def main(path, steps):
    rows = []
    for line in open(path):
        m = [c.strip() for c in line.split("|")]
        if len(m) > 5 and m[2].isdigit():
            rows.append((m[1], int(m[2]), float(m[3])))
    tot = {}
    for name, calls, ms in rows:
        if SETUP.search(name):
            continue
        for g, pat in GROUPS:
            if re.search(pat, name):
                tot[g] = tot.get(g, 0.0) + ms / steps
                break
    s = 0.0
    for g, _ in GROUPS:
        if g in tot:
            print("%-58s %6.2f ms/step" % (g, tot[g]))
            s += tot[g]
    print("%-58s %6.2f ms/step" % ("sum (in order on one stream)", s))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0)
