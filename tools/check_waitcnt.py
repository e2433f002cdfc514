#!/usr/bin/env python3
"""Build-time ISA check of the kernels whose `s_waitcnt vmcnt(N)` are counted by hand (csrc/Makefile HANDCOUNTED).

A counted wait is only right while exactly the vector-memory operations the source counted sit between the awaited operation and the
wait.  The scratch guard catches spills; it cannot see the compiler merging two epilogue stores, sinking a load below a wait or
hoisting one across it -- silent wrong data, not a fault.  This script disassembles the device code of an object file
(llvm-objdump), reduces every kernel to its stream of vector-memory events

    W<n>  s_waitcnt vmcnt(n)        L<k>  k loads to registers        D<k>  k loads to LDS (`... lds`)
    S<k>  k stores                  B     s_barrier                   |     a branch (the next run may be the other arm; forward
                                                                            skips over plain ALU code leave no token)

and compares the part between the first and the last counted (n > 0) wait with the signature recorded for that kernel in
csrc/waitcnt.sig (one line per kernel: name <TAB> signature).  A signature is the declared count table of the kernel: the per-step
operation counts the source's N_TOP / N_COL / E1 / E2 constants are made of can be read off it directly (docs in each .hip file).

  check_waitcnt.py build/conv_wr.o ...            compare; exit 1 on any difference (the Makefile runs this)
  check_waitcnt.py --update build/conv_wr.o ...   rewrite the signatures of these objects' kernels after a REVIEWED kernel change
"""
import os
import re
import subprocess
import sys

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
HERE = os.path.dirname(os.path.abspath(__file__))
SIG = os.environ.get("FID_WAITCNT_SIG") or os.path.join(HERE, "..", "scrfd_arcface_facerecognition_amd", "csrc", "waitcnt.sig")


def device_disassembly(obj):
    """disassembly text of the gfx950 code object bundled in a hipcc object file"""
    obj = os.path.abspath(obj)
    out = subprocess.run([OBJDUMP, "--offloading", obj], capture_output=True, text=True, cwd=os.path.dirname(obj))
    co = None
    for line in out.stdout.splitlines():
        m = re.search(r"Extracting offload bundle: (\S*amdgcn\S*)", line)
        if m:
            co = os.path.join(os.path.dirname(obj) or ".", os.path.basename(m.group(1)))
    if co is None or not os.path.exists(co):
        raise SystemExit(f"{obj}: no gfx code object found ({out.stdout} {out.stderr})")
    try:
        return subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True, check=True).stdout
    finally:
        for f in os.listdir(os.path.dirname(co) or "."):
            if f.startswith(os.path.basename(obj) + ".0."):
                os.unlink(os.path.join(os.path.dirname(co) or ".", f))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
        return out.stdout.splitlines() if out.returncode == 0 else names
    except FileNotFoundError:
        return names


def kernel_streams(text):
    """{mangled kernel name: token list}.  A FORWARD branch whose skipped range holds no vector-memory operation, barrier or wait (the
    compiler wraps per-lane address arithmetic in `s_cbranch_execz` skips) is not a fork of the event stream and leaves no token:
    every path through it issues the same operations.  Backward branches (loops) and forward branches over events stay
    as `|`: the runs on both sides of such a token are alternative arms and must be compared arm by arm when a signature is reviewed."""
    kernels, cur, base = {}, None, 0
    pending = []                                   # (index in cur of a forward-branch token, target address)
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            base = int(m.group(1), 16)
            pending = []
            continue
        if cur is None:
            continue
        parts = line.strip().split("//")
        ins = parts[0].strip()
        if not ins:
            continue
        addr = None
        if len(parts) > 1:
            ma = re.match(r"\s*([0-9A-Fa-f]+):", parts[1])
            if ma:
                addr = int(ma.group(1), 16)
        if addr is not None:                       # forward branches that land here: drop the ones that skipped nothing
            for idx, tgt in [p for p in pending if p[1] <= addr]:
                if all(k == "|" or k == "_" for k, _ in cur[idx + 1:]):
                    cur[idx] = ("_", 0)
            pending = [p for p in pending if p[1] > addr]
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                cur.append(("W", int(m.group(1))))
        elif op == "s_barrier":
            cur.append(("B", 0))
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur.append(("|", 0))
            mt = re.search(r"<[^>+]+\+0x([0-9a-fA-F]+)>", parts[1] if len(parts) > 1 else "")
            if mt and addr is not None and base + int(mt.group(1), 16) > addr:
                pending.append((len(cur) - 1, base + int(mt.group(1), 16)))
        elif re.match(r"(buffer|global|flat|scratch)_(load|store|atomic)", op):
            kind = "S" if "_store" in op else ("D" if re.search(r"\blds\b", ins) else "L")
            cur.append((kind, 1))
    return {k: [t for t in v if t[0] != "_"] for k, v in kernels.items()}


def signature(tokens):
    """run-length form of the stream between the first and the last COUNTED wait (n > 0); '' when the kernel has none"""
    idx = [i for i, (k, n) in enumerate(tokens) if k == "W" and n > 0]
    if not idx:
        return ""
    # a little context on both sides: the operations a first wait counts were issued before it
    lo = idx[0]
    while lo > 0 and tokens[lo - 1][0] not in ("W",):
        lo -= 1
    out, prev, run = [], None, 0
    for k, n in tokens[lo:idx[-1] + 1]:
        if k in ("L", "D", "S"):
            if prev == k:
                run += 1
                continue
            if prev in ("L", "D", "S"):
                out.append(f"{prev}{run}")
            prev, run = k, 1
            continue
        if prev in ("L", "D", "S"):
            out.append(f"{prev}{run}")
        prev = k
        if k == "W":
            out.append(f"W{n}")
        elif k == "B":
            out.append("B")
        elif k == "|" and (not out or out[-1] != "|"):
            out.append("|")
    return " ".join(out)


def load_sigs():
    sigs = {}
    if os.path.exists(SIG):
        for line in open(SIG):
            if line.strip() and not line.startswith("#"):
                name, sig = line.rstrip("\n").split("\t")
                sigs[name] = sig
    return sigs


def main():
    args = sys.argv[1:]
    update = "--update" in args
    objs = [a for a in args if not a.startswith("--")]
    sigs = load_sigs()
    bad = 0
    for obj in objs:
        streams = kernel_streams(device_disassembly(obj))
        names = list(streams)
        pretty = dict(zip(names, [re.sub(r"fid::\(anonymous namespace\)::|void ", "", d).split("(fid::")[0] for d in demangle(names)]))
        unit = os.path.basename(obj).replace(".o", "")
        have = {pretty[n]: signature(streams[n]) for n in names}
        have = {f"{unit}:{k}": v for k, v in have.items() if v}
        if update:
            sigs = {k: v for k, v in sigs.items() if not k.startswith(unit + ":")}
            sigs.update(have)
            continue
        for k, v in sorted(have.items()):
            if k not in sigs:
                print(f"check_waitcnt: {k}: a kernel with counted waits has no recorded signature (review it, then tools/check_waitcnt.py --update {obj})")
                bad += 1
            elif sigs[k] != v:
                print(f"check_waitcnt: {k}: the vector-memory operation stream around the counted waits CHANGED\n  recorded: {sigs[k]}\n  built:    {v}\n"
                      f"  -> re-derive the kernel's vmcnt constants against the new stream, then tools/check_waitcnt.py --update {obj}")
                bad += 1
        for k in sorted(k for k in sigs if k.startswith(unit + ":") and k not in have):
            print(f"check_waitcnt: {k}: recorded kernel is no longer in {obj} (tools/check_waitcnt.py --update {obj})")
            bad += 1
    if update:
        with open(SIG, "w") as f:
            f.write("# kernel <TAB> vector-memory event stream around its hand-counted s_waitcnt vmcnt(N) (tools/check_waitcnt.py; W wait, L loads,\n"
                    "# D loads to LDS, S stores, B barrier, | branch).  Regenerate ONLY after re-deriving the counts in the kernel's source.\n")
            for k in sorted(sigs):
                f.write(f"{k}\t{sigs[k]}\n")
        print(f"check_waitcnt: {len(sigs)} signatures written to {os.path.relpath(SIG)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
