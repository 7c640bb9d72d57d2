#!/usr/bin/env python
"""Per-kernel hashes of the gfx950 instruction streams the product build is made of.

    tools/isa_stamp.py                      -> JSON on stdout: {"head": ..., "files": {"attn.hip": {"<mangled kernel>": "<sha1[:12]>", ...}}}
    tools/isa_stamp.py --out profiles/r06_ISA.json
    tools/isa_stamp.py --against <git-ref>  -> lists every kernel whose instruction stream differs from the one <git-ref>'s sources give
    tools/isa_stamp.py --check profiles/r06_ISA.json   -> exit 1 if the working tree's kernels differ from the stamped ones
    tools/isa_stamp.py --stamp-ref dce9137 --out profiles/r05_ISA.json   -> the stamp of a commit's sources (what its profiles measured)

Every .hip under csrc/ is compiled with the Makefile's flags (`--cuda-device-only -S`), the listing is cut into kernels and the
instruction text of each (labels renumbered, comments dropped) is hashed -- the same normalisation as tools/kernel_code_diff.py.
"The profiles describe HEAD" is then a mechanical check: the stamp written when the profiles were collected must equal the stamp of
the tree under judgement (a refactor changed a default-path kernel's code after its last GPU run, unnoticed)."""
import sys, os, re, json, hashlib, subprocess, tempfile, shutil, argparse
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video-description-with-spatial-temporal-attention_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"]
EXTRA = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def kernels_of_listing(text):
    res = {}; cur = None; buf = []
    for l in text.splitlines():
        m = re.match(r'^(_Z\S+):\s*', l)
        if m:
            cur = m.group(1); buf = []; continue
        if cur is None: continue
        t = l.split(';')[0].strip()
        if t.startswith('s_endpgm'):
            buf.append(t); res[cur] = hashlib.sha1('\n'.join(buf).encode()).hexdigest()[:12]; cur = None; continue
        if not t or (t.startswith('.') and not t.startswith('.LBB')): continue
        buf.append(re.sub(r'\.LBB\d+_', '.LBB_', t))
    return res


def listing(csrc, f, extra_flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc"] + FLAGS + EXTRA.get(f, []) + list(extra_flags) + ["--cuda-device-only", "-S", os.path.join(csrc, f), "-o", out],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


def stamp(csrc, extra_flags=()):
    files = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    with ThreadPoolExecutor(8) as ex:
        ks = list(ex.map(lambda f: kernels_of_listing(listing(csrc, f, extra_flags)), files))
    return dict(zip(files, ks))


def head():
    try:
        h = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"]).decode().strip()
        dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", CSRC]).strip())
        return h + ("+dirty-csrc" if dirty else "")
    except Exception:
        return "unknown"


def compare(a, b, na, nb):
    bad = 0
    for f in sorted(set(a) | set(b)):
        ka, kb = a.get(f, {}), b.get(f, {})
        changed = [k for k in ka if k in kb and ka[k] != kb[k]]
        new = sorted(set(kb) - set(ka)); gone = sorted(set(ka) - set(kb))
        print("%-20s %s %3d kernels -> %s %3d; identical %3d; changed %d; new %d; gone %d" % (f, na, len(ka), nb, len(kb),
              len([k for k in ka if k in kb]) - len(changed), len(changed), len(new), len(gone)))
        for k in changed: print("    CHANGED", k[:150])
        for k in new: print("    new    ", k[:150])
        for k in gone: print("    gone   ", k[:150])
        bad += len(changed) + len(new) + len(gone)
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out"); ap.add_argument("--against"); ap.add_argument("--check")
    ap.add_argument("--stamp-ref", help="write the stamp of a git ref's sources (with --out) instead of the working tree's")
    a = ap.parse_args()
    if a.stamp_ref:
        d = tempfile.mkdtemp()
        try:
            rel = os.path.relpath(CSRC, ROOT)
            subprocess.check_call("git -C %s archive %s %s | tar -x -C %s" % (ROOT, a.stamp_ref, rel, d), shell=True)
            ref = subprocess.check_output(["git", "-C", ROOT, "rev-parse", a.stamp_ref]).decode().strip()
            doc = {"head": ref, "flags": FLAGS, "extra": EXTRA, "files": stamp(os.path.join(d, rel))}
        finally:
            shutil.rmtree(d, ignore_errors=True)
        s = json.dumps(doc, indent=1, sort_keys=True)
        if a.out: open(a.out, "w").write(s + "\n")
        else: print(s)
        sys.exit(0)
    cur = stamp(CSRC)
    if a.against:
        d = tempfile.mkdtemp()
        try:
            rel = os.path.relpath(CSRC, ROOT)
            subprocess.check_call("git -C %s archive %s %s | tar -x -C %s" % (ROOT, a.against, rel, d), shell=True)
            old = stamp(os.path.join(d, rel))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        compare(old, cur, a.against[:8], "tree")
        sys.exit(0)
    if a.check:
        old = json.load(open(a.check))
        bad = compare(old["files"], cur, "stamp", "tree")
        print("stamp %s (%s): %s" % (a.check, old.get("head"), "MATCHES the tree" if not bad else "%d kernels DIFFER" % bad))
        sys.exit(1 if bad else 0)
    doc = {"head": head(), "flags": FLAGS, "extra": EXTRA, "files": cur}
    s = json.dumps(doc, indent=1, sort_keys=True)
    if a.out: open(a.out, "w").write(s + "\n")
    else: print(s)
