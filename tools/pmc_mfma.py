#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes of tools/pmc_mfma.sh per kernel.

  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)      share of the CUs' busy cycles with a matrix pipe busy
  mfma_chip  = SQ_VALU_MFMA_BUSY_CYCLES / (4 x 256 CUs x GRBM_GUI_ACTIVE / 8)  the same against the whole chip for the kernel's duration
               (GRBM_GUI_ACTIVE is summed over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back)
  wait / issue-stall / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of a wave's life)
Counters are hardware sums over all dispatches of a kernel name in the run (autotuning launches included)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(run_dir):
    files = glob.glob(os.path.join(run_dir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {run_dir}")
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("fid::", "").split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


def durations(run_dir):
    """kernel name -> summed device duration in ns (kernel_trace.csv of the same pass)"""
    files = glob.glob(os.path.join(run_dir, "**", "*kernel_trace.csv"), recursive=True)
    d = defaultdict(float)
    if files:
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("fid::", "").split("(")[0]
                d[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return d


def main():
    out = sys.argv[1]
    # optional: tools/klog_map.py's JSON + the number of net runs the profiled program made -> executed / algorithmic MFMA work per kernel
    alg, runs = {}, 0
    if len(sys.argv) > 3:
        import json
        alg = json.load(open(sys.argv[2]))["gflop_per_run_by_kernel"]
        runs = int(sys.argv[3])
    a, na = load(os.path.join(out, "a"))
    dur = durations(os.path.join(out, "a"))
    b, _ = load(os.path.join(out, "b"))
    rows = []
    for k, c in a.items():
        if c.get("SQ_WAVE_CYCLES", 0) <= 0:
            continue
        busy_cu = c.get("SQ_BUSY_CU_CYCLES", 0)
        gui = c.get("GRBM_GUI_ACTIVE", 0)
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        wc = c["SQ_WAVE_CYCLES"]
        d = b.get(k, {})
        rows.append((gui, k, na[k], mf / (4 * busy_cu) if busy_cu else 0, mf / (4 * 256 * gui / 8) if gui else 0,
                     c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                     d.get("SQ_WAIT_INST_LDS", 0) / wc, d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, d.get("SQ_LDS_IDX_ACTIVE", 0)),
                     c.get("SQ_INSTS_MFMA", 0), (gui / 8 / dur[k]) if dur.get(k) else 0.0, dur.get(k, 0.0) / max(1, na[k]) / 1e3,
                     d.get("SQ_ACTIVE_INST_VALU", 0) / wc, d.get("SQ_ACTIVE_INST_LDS", 0) / wc, d.get("SQ_ACTIVE_INST_SCA", 0) / wc))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("clk_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (kernel trace of the same pass): the shader clock the chip held while the kernel ran")
    if alg:
        print(f"executed_over_algorithmic = SQ_INSTS_MFMA x 16 384 flop (v_mfma_f32_16x16x32_f16; kernels that also issue 16x16x16 steps -- ir_stem_block, scrfd_stem_rows -- are over-counted by those) "
              f"/ (algorithmic GFLOP per run of the kernel's ops, tools/klog_map.py, x {runs} runs of the net in the profiled program)")
    print(f"{'kernel':56s} {'launches':>8s} {'time%':>6s} {'avg_us':>8s} {'clk_ghz':>7s} {'mfma_util':>9s} {'mfma_chip':>9s} {'wait':>6s} {'istall':>6s} {'active':>6s} {'lds_st':>6s} {'bankcf':>6s} {'mfma_insts':>12s} {'valu':>6s} {'lds':>6s} {'salu':>6s} {'exec/alg':>8s} {'alg_TF/s':>8s}   (valu / lds / salu = SQ_ACTIVE_INST_* of the second pass over SQ_WAVE_CYCLES of the first)")
    busy_w, t_w = 0.0, 0.0
    for gui, k, n, mu, mc, w, ws, ac, wl, bc, ni, clk, avg, va, la, sa in rows[:40]:
        ea, tf = "", ""
        if alg.get(k) and runs:
            ea = f"{ni * 16384 / (alg[k] * 1e9 * runs):8.3f}"
            tf = f"{alg[k] * runs / (avg * n * 1e-6) / 1e3:8.1f}"
        if ni > 0:
            busy_w += mu * gui; t_w += gui
        print(f"{k[:56]:56s} {n:8d} {100 * gui / tot:6.1f} {avg:8.1f} {clk:7.2f} {100 * mu:9.1f} {100 * mc:9.1f} {100 * w:6.1f} {100 * ws:6.1f} {100 * ac:6.1f} {100 * wl:6.1f} {100 * bc:6.1f} {ni:12.0f} {100 * va:6.1f} {100 * la:6.1f} {100 * sa:6.1f} {ea:>8s} {tf:>8s}")
    if t_w:
        print(f"time-weighted MFMA-busy of the kernels that issue MFMAs: {100 * busy_w / t_w:.1f} %")


if __name__ == "__main__":
    main()
