#!/usr/bin/env python3
"""HBM traffic of the MFMA conv kernels per bench step, from rocprofv3 PMC passes.

Collect (each its own run, --kernel-trace only, as MI355X_MICROARCH.md prescribes):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/f2 -o p -- python3 bench.py --steps 2 --warmup 1 ...
    ... --steps 6 ...   (OUT/f6),   --pmc WRITE_SIZE --steps 2 / 6   (OUT/w2, OUT/w6)
then:  python tools/pmc_traffic.py OUT > profiles/rNN/pmc_traffic.json

Per step = (6-step run - 2-step run) / 4, so tuning / warm-up launches cancel.  FETCH_SIZE and WRITE_SIZE count
kilobytes; on gfx950 FETCH_SIZE counts a 128-byte request of a wide coalesced read as 64 B (guide, HBM section): x2.
"""
import csv
import glob
import json
import os
import sys

CONV = ("conv_mfma_kernel", "conv_mfma_dma_kernel", "conv3x3_direct", "conv3x3_chunked", "conv3x3_pp", "conv3x3_pc", "conv_mfma_pc_kernel", "conv3x3_wr", "conv3x3_s2", "conv_gw",
        "scrfd_stem_fused", "stem_conv_mfma", "conv_bb", "conv3x3_ks", "scrfd_stem_rows", "dwpw_kernel", "mbf_block", "ir_stem_block", "lat_fpn")


def total(run_dir, counter):
    files = glob.glob(os.path.join(run_dir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {run_dir}")
    tot, launches = 0.0, set()
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter or not any(k in row["Kernel_Name"] for k in CONV):
                continue
            tot += float(row["Counter_Value"])
            launches.add(row["Dispatch_Id"])
    return tot, len(launches)


def main():
    out = sys.argv[1]
    f2, n2 = total(os.path.join(out, "f2"), "FETCH_SIZE")
    f6, n6 = total(os.path.join(out, "f6"), "FETCH_SIZE")
    w2, _ = total(os.path.join(out, "w2"), "WRITE_SIZE")
    w6, _ = total(os.path.join(out, "w6"), "WRITE_SIZE")
    fetch = (f6 - f2) / 4 * 1024
    write = (w6 - w2) / 4 * 1024
    launches = (n6 - n2) / 4
    hbm = 2 * fetch + write
    print(json.dumps({
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), "
                  + (sys.argv[2] if len(sys.argv) > 2 else "python3 bench.py --steps {2,6} --warmup 1 --streams 1")
                  + "; per-step = (6-step run - 2-step run)/4; MFMA conv kernel families only (tools/pmc_traffic.py)",
        "fetch_bytes_per_step_raw": fetch,
        "write_bytes_per_step": write,
        "fetch_correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) coalesced reads "
                            "(MI355X_MICROARCH.md, HBM): x2",
        "hbm_bytes_per_step_corrected": hbm,
        "launches_per_step": launches,
        "hbm_bytes_per_launch_corrected": hbm / launches if launches else None,
    }, indent=1))


if __name__ == "__main__":
    main()
